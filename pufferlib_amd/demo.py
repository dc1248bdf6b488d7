"""Run the reference's UNMODIFIED demo.py against this engine.

    python -m pufferlib_amd.demo [--reference DIR] [--host-vec] -- <demo.py arguments, e.g. --env squared --mode train>

demo.py (demo.py:1-20) does ``import clean_pufferl`` and picks its vecenv backend with ``--vec serial|multiprocessing|ray``
mapped to ``pufferlib.vector.Serial / Multiprocessing / Ray`` (demo.py:159-167).  The launcher therefore

  1. makes ``clean_pufferl`` resolve to ``pufferlib_amd.clean_pufferl`` (same function surface: create / evaluate / train / close /
     try_load_checkpoint / rollout, the same ``data`` namespace);
  2. rebinds ``pufferlib.vector.Serial`` to ``DeviceOrHost``: a backend factory that builds the device-resident vecenv of this
     package when the env creator is one of the ocean envs it hosts (squared, stochastic, memory, bandit, multiagent, spaces)
     and the reference's own ``Serial`` otherwise (the trainer then takes its host-vecenv path, pufferlib_amd/hostpath.py);
     ``--host-vec`` keeps the reference backends untouched;
  3. supplies a stand-in for ``rich_argparse`` if it is not installed (demo.py only uses its help formatter);
  4. runs demo.py with ``runpy`` as ``__main__`` from a working directory that holds the reference's config.yaml
     (demo.py:22 opens it relative to the cwd) — the current directory if it has one, else a scratch copy.

Nothing of the reference is modified or copied except that one yaml file into a scratch directory.
"""
import argparse
import os
import runpy
import shutil
import sys
import tempfile
import types


def find_reference(explicit=None):
    """Directory that holds the reference's demo.py + config.yaml: --reference, $PUFFERLIB_DIR, or next to the importable
    ``pufferlib`` package."""
    cands = [explicit, os.environ.get('PUFFERLIB_DIR')]
    try:
        import importlib.util
        spec = importlib.util.find_spec('pufferlib')
        if spec is not None and spec.submodule_search_locations:
            cands.append(os.path.dirname(list(spec.submodule_search_locations)[0]))
    except Exception:
        pass
    for c in cands:
        if c and os.path.exists(os.path.join(c, 'demo.py')) and os.path.exists(os.path.join(c, 'config.yaml')):
            return os.path.abspath(c)
    raise FileNotFoundError('cannot find the PufferLib checkout (demo.py + config.yaml): pass --reference DIR or set PUFFERLIB_DIR')


def _creator_name(creator):
    import functools
    while isinstance(creator, functools.partial):
        creator = creator.func
    return getattr(creator, '__name__', '').lower()


def device_backend_for(creator):
    """The device-resident backend class that hosts this env creator, or None (-> a host backend)."""
    from . import vector
    name = _creator_name(creator)
    for key, cls in (('squared', 'Squared'), ('stochastic', 'Stochastic'), ('memory', 'Memory'), ('bandit', 'Bandit'),
                     ('multiagent', 'Multiagent'), ('spaces', 'Spaces'), ('synthetic', 'Synthetic')):
        if key in name and hasattr(vector, cls):
            return getattr(vector, cls)
    return None


def make_device_or_host(host_backend):
    """The object bound to ``pufferlib.vector.Serial``: called like a backend class (vector.py:637)."""
    def DeviceOrHost(env_creators, env_args, env_kwargs, num_envs, **kwargs):
        cls = device_backend_for(env_creators[0]) if len(env_creators) else None
        if cls is not None and all(device_backend_for(c) is cls for c in env_creators):
            if num_envs is None:        # reference quirk: `backend is Serial and 'batch_size' in kwargs` with batch_size None
                num_envs = len(env_creators)
            return cls(env_creators, env_args, env_kwargs, num_envs, **kwargs)
        return host_backend(env_creators, env_args, env_kwargs, num_envs, **kwargs)
    DeviceOrHost.host_backend = host_backend
    DeviceOrHost.__doc__ = 'device-resident pufferlib_amd backend where the env is hosted on device, else ' + repr(host_backend)
    return DeviceOrHost


def install(host_vec=False):
    """Steps 1-3 of the module docstring.  Returns the names that were (re)bound, for the tests."""
    done = []
    try:
        import rich_argparse  # noqa: F401
    except Exception:
        mod = types.ModuleType('rich_argparse')
        mod.RichHelpFormatter = argparse.HelpFormatter
        sys.modules['rich_argparse'] = mod
        done.append('rich_argparse (stand-in)')
    import pufferlib          # the reference package: this launcher is for processes that have it
    import pufferlib.vector
    from . import clean_pufferl
    sys.modules['clean_pufferl'] = clean_pufferl
    done.append('clean_pufferl')
    if not host_vec and not hasattr(pufferlib.vector.Serial, 'host_backend'):
        pufferlib.vector.Serial = make_device_or_host(pufferlib.vector.Serial)
        done.append('pufferlib.vector.Serial')
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(prog='python -m pufferlib_amd.demo', add_help=True,
                                 description='run the reference demo.py on the MI355X-native engine')
    ap.add_argument('--reference', default=None, help='PufferLib checkout (demo.py, config.yaml); default: next to the importable pufferlib')
    ap.add_argument('--host-vec', action='store_true', help='keep the reference vecenv backends (host path of the trainer)')
    args, rest = ap.parse_known_args(argv)
    if rest and rest[0] == '--':
        rest = rest[1:]
    ref = find_reference(args.reference)
    if ref not in sys.path:
        sys.path.insert(0, ref)          # what `python demo.py` gives the script: its own directory first
    sys.dont_write_bytecode = True       # never leave __pycache__ in the reference tree
    install(host_vec=args.host_vec)
    cwd = os.getcwd()
    if not os.path.exists(os.path.join(cwd, 'config.yaml')):
        scratch = tempfile.mkdtemp(prefix='pfa_demo_')
        shutil.copy(os.path.join(ref, 'config.yaml'), os.path.join(scratch, 'config.yaml'))
        os.chdir(scratch)
        print(f'[pufferlib_amd.demo] running in {scratch} (config.yaml copied from {ref})', file=sys.stderr)
    sys.argv = [os.path.join(ref, 'demo.py')] + list(rest)
    runpy.run_path(os.path.join(ref, 'demo.py'), run_name='__main__')


if __name__ == '__main__':
    main()
