"""`python -m pufferlib_amd.demo` runs the reference's UNMODIFIED demo.py (demo.py:153-201) up to the construction of the vecenv
backend: clean_pufferl resolves to pufferlib_amd.clean_pufferl, `--vec serial` reaches pufferlib_amd.vector.Squared with the
arguments pufferlib.vector.make hands a backend.  Build-container test (needs /root/reference; gym/gymnasium/pettingzoo come
from tests/shims); the GPU half — constructing the backend and training — is tests/test_gpu_demo.py."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

DRIVER = r'''
import sys
sys.dont_write_bytecode = True
from pufferlib_amd import demo, vector

class Reached(Exception):
    pass

def fake_init(self, env_creators, env_args, env_kwargs, num_envs, **kwargs):
    import clean_pufferl
    print('BACKEND', type(self).__name__, len(env_creators), num_envs, sorted(kwargs), dict(env_kwargs[0]), flush=True)
    print('TRAINER', clean_pufferl.__name__, callable(clean_pufferl.rollout), flush=True)
    raise Reached()

vector.Squared.__init__ = fake_init
try:
    demo.main(['--reference', sys.argv[1], '--', '--env', 'squared', '--vec', 'serial', '--train.num-envs', '8'])
except Reached:
    print('REACHED', flush=True)
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'demo.py')), reason='needs the reference checkout (build container)')
def test_unmodified_demo_py_reaches_the_device_backend(tmp_path):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, 'tests', 'shims'), REF]),
               PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', DRIVER, REF], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout
    assert 'REACHED' in out, (out[-2000:], r.stderr[-3000:])
    line = [l for l in out.splitlines() if l.startswith('BACKEND')][0]
    assert 'Squared 8 8' in line and "'distance_to_target': 3" in line, line          # config.yaml's ocean/squared section reached us
    assert 'TRAINER pufferlib_amd.clean_pufferl True' in out
    assert not os.path.exists(os.path.join(REF, '__pycache__'))                        # the reference tree stays untouched


def test_backend_selection_by_env_creator():
    import functools
    from pufferlib_amd import demo, vector

    def make_squared(distance_to_target=3, num_targets=1):
        pass

    def make_nethack():
        pass
    assert demo.device_backend_for(make_squared) is vector.Squared
    assert demo.device_backend_for(functools.partial(make_squared, distance_to_target=2)) is vector.Squared
    assert demo.device_backend_for(vector.make_memory) is vector.Memory
    assert demo.device_backend_for(make_nethack) is None
    calls = []
    host = lambda *a, **k: calls.append((a, k)) or 'host'  # noqa: E731
    backend = demo.make_device_or_host(host)
    assert backend([make_nethack] * 2, [[]] * 2, [{}] * 2, 2, num_workers=1) == 'host' and len(calls) == 1
