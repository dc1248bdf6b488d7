"""Compile the C oracle (and, when /root/reference is present, the reference's own Cython GAE
into oracle/_ref/).  Building the checker is not using it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def build_oracle(force=False):
    src = os.path.join(HERE, 'puffer_oracle.c')
    out_dir = os.path.join(HERE, '_build')
    out = os.path.join(out_dir, 'libpuffer_oracle.so')
    os.makedirs(out_dir, exist_ok=True)
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) >= max(os.path.getmtime(src),
                                             os.path.getmtime(os.path.join(HERE, 'puffer_oracle.h')))):
        return out
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-builtin-pow', '-Wall',
                           src, '-o', out, '-lm'])
    return out


def build_ref(force=False):
    """oracle/_ref/c_gae*.so: the reference's c_gae.pyx compiled from where it lies.

    Only possible where /root/reference exists (the build container); the GPU box uses the
    prebuilt file that travels with the snapshot.  Returns the path or None.
    """
    import sysconfig
    out_dir = os.path.join(HERE, '_ref')
    suffix = sysconfig.get_config_var('EXT_SUFFIX')
    out = os.path.join(out_dir, 'c_gae' + suffix)
    pyx = os.path.join(REF, 'c_gae.pyx')
    if not os.path.exists(pyx):
        return out if os.path.exists(out) else None
    if os.path.exists(out) and not force:
        return out
    os.makedirs(out_dir, exist_ok=True)
    import numpy as np
    c_file = os.path.join(out_dir, 'c_gae.c')
    subprocess.check_call([sys.executable, '-m', 'cython', '-3', pyx, '-o', c_file])
    inc = sysconfig.get_paths()['include']
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-fwrapv', '-fno-strict-aliasing',
                           '-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION',
                           '-I', inc, '-I', np.get_include(), c_file, '-o', out])
    return out


if __name__ == '__main__':
    print(build_oracle(force=True))
    print(build_ref(force='--force-ref' in sys.argv))
