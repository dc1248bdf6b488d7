"""Stage the files of the reference that the hot path is made of into a GITIGNORED scratch directory (`_refstage/`) so that ONE
gpurun job can (a) run the reference's unmodified demo.py end to end on the device engine and (b) time the reference's own CPU
path on the GPU box's host cores.  The GPU box has no /root/reference; gpurun ships git-ignored files of the tree.

    python tools/stage_reference.py           # copy
    python tools/stage_reference.py --clean   # remove (tools/gpu_jobs/with_reference.sh does this when the job ends)

The staging directory is scratch: it is never committed (.gitignore), nothing under pufferlib_amd/ imports it, and it must not
exist at round end.  What is copied is SURVEY.md section 8c's list — the files the CPU oracle *is* — plus demo.py / config.yaml:

    demo.py config.yaml clean_pufferl.py c_gae.pyx
    pufferlib/*.py pufferlib/extensions.pyx
    pufferlib/frameworks/{__init__,cleanrl}.py
    pufferlib/environments/__init__.py pufferlib/environments/ocean/*.py
"""
import argparse
import glob
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(REPO, '_refstage')
REF = '/root/reference'


def files():
    out = ['demo.py', 'config.yaml', 'clean_pufferl.py', 'c_gae.pyx', 'pufferlib/extensions.pyx',
           'pufferlib/frameworks/__init__.py', 'pufferlib/frameworks/cleanrl.py', 'pufferlib/environments/__init__.py']
    out += [os.path.relpath(p, REF) for p in sorted(glob.glob(os.path.join(REF, 'pufferlib', '*.py')))]
    out += [os.path.relpath(p, REF) for p in sorted(glob.glob(os.path.join(REF, 'pufferlib', 'environments', 'ocean', '*.py')))]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clean', action='store_true')
    args = ap.parse_args()
    if args.clean:
        shutil.rmtree(STAGE, ignore_errors=True)
        print('[stage_reference] removed', STAGE)
        return 0
    if not os.path.exists(os.path.join(REF, 'demo.py')):
        print('[stage_reference] no reference checkout at', REF, file=sys.stderr)
        return 1
    ignored = os.popen(f'cd {REPO} && git check-ignore _refstage/demo.py').read().strip()
    if ignored != '_refstage/demo.py':
        print('[stage_reference] refusing: _refstage/ is not git-ignored', file=sys.stderr)
        return 1
    shutil.rmtree(STAGE, ignore_errors=True)
    n = 0
    for rel in files():
        dst = os.path.join(STAGE, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, rel), dst)
        n += 1
    print(f'[stage_reference] {n} files -> {STAGE}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
