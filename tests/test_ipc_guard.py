"""pufferlib_amd picks dmabuf IPC (HSA_ENABLE_IPC_MODE_LEGACY=0) at IMPORT time — it must be in the environment when HIP
initialises — and records a process in which it is too late, so that dist.init_p2p leaves the peer path closed with a reason
instead of discovering it in the self-test."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import json, os, sys, types
if sys.argv[1] == 'hip_up':           # a process whose torch has already initialised the device
    t = types.ModuleType('torch'); t.cuda = types.SimpleNamespace(is_initialized=lambda: True); sys.modules['torch'] = t
import pufferlib_amd
print(json.dumps(dict(mode=pufferlib_amd.IPC_MODE, env=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))))
'''


def probe(case, value):
    env = {k: v for k, v in os.environ.items() if k != 'HSA_ENABLE_IPC_MODE_LEGACY'}
    if value is not None:
        env['HSA_ENABLE_IPC_MODE_LEGACY'] = value
    env['PYTHONPATH'] = REPO
    r = subprocess.run([sys.executable, '-c', PROBE, case], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_import_sets_the_variable_before_hip_initialises():
    d = probe('fresh', None)
    assert d['env'] == '0' and d['mode']['ok'] and d['mode']['value_at_import'] is None and d['mode']['reason'] is None


def test_import_keeps_a_value_that_is_already_right():
    d = probe('fresh', '0')
    assert d['env'] == '0' and d['mode']['ok'] and d['mode']['value_at_import'] == '0'


def test_a_wrong_value_is_reported_not_overridden():
    d = probe('fresh', '1')
    assert d['env'] == '1' and not d['mode']['ok'] and 'HSA_ENABLE_IPC_MODE_LEGACY=1' in d['mode']['reason']


def test_hip_already_initialised_without_the_variable_is_too_late():
    d = probe('hip_up', None)
    assert not d['mode']['ok'] and d['mode']['hip_initialised_at_import'] and 'before pufferlib_amd was imported' in d['mode']['reason']
    d = probe('hip_up', '0')             # ... but fine when the launcher had exported it
    assert d['mode']['ok']
