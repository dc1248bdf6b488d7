"""Every surviving PFA_* switch under its NON-DEFAULT value (<= 10 runtime switches, none untested): each runs, in
a fresh process (several are read once per process), the reference golden replay and two full-size iterations of the bench workload,
whose results must equal the default-settings run — bit for bit where the switch only moves WHERE or WHEN something runs, within the
contract's 1e-5 where it changes arithmetic (the bf16x6 product form).  The data-parallel switches (PFA_ALLREDUCE, PFA_FUSED_DP,
PFA_RANK_AFFINITY) are the transports of tests/test_gpu_dp.py.

  PFA_LAZY_READBACK=1      evaluate() / train() hand back lazy containers, the host waits when they are first read
  PFA_EARLY_GAE=0          the update's GAE + statistics pass runs in train() instead of behind evaluate()'s readback
  PFA_GAE_SELF=0           GAE's f64-seeded two-launch window instead of the self-starting one
  PFA_FUSED_ADAM=0         partial reduce and clip + Adam as two kernels instead of one (the recovery path of a grid hand-off timeout)
  PFA_WAIT_TIMEOUT_MS      bound of the in-kernel waits (grid hand-off, peer exchanges)
  PFA_MATRIX_PRODUCTS      bf16x6: six bf16 partial products per fp32 product in the gradient step
  PFA_LSTM_OBS_CACHE_MB=0  the recurrent update gathers every minibatch's observation rows every epoch instead of keeping them"""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _probe(tmp, name, policy, env_over):
    out = os.path.join(str(tmp), f'{name}.npz')
    env = {k: v for k, v in os.environ.items() if not k.startswith('PFA_')}
    env.update(env_over)
    r = subprocess.run([sys.executable, os.path.join(HERE, 'switch_probe.py'), '--out', out, '--policy', policy], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return np.load(out)


@pytest.fixture(scope='module')
def defaults(tmp_path_factory):
    tmp = tmp_path_factory.mktemp('switches')
    return {p: _probe(tmp, f'default_{p}', p, {}) for p in ('mlp', 'lstm')}


@pytest.mark.parametrize('policy,switch,value,exact', [
    ('mlp', 'PFA_LAZY_READBACK', '1', True), ('mlp', 'PFA_EARLY_GAE', '0', True), ('mlp', 'PFA_GAE_SELF', '0', True),
    ('mlp', 'PFA_FUSED_ADAM', '0', True), ('mlp', 'PFA_WAIT_TIMEOUT_MS', '5000', True), ('mlp', 'PFA_MATRIX_PRODUCTS', 'bf16x6', False),
    ('lstm', 'PFA_LSTM_OBS_CACHE_MB', '0', True), ('lstm', 'PFA_EARLY_GAE', '0', True), ('lstm', 'PFA_LAZY_READBACK', '1', True)])
def test_non_default_switch_value_gives_the_default_runs_results(tmp_path, defaults, policy, switch, value, exact):
    got, want = _probe(tmp_path, switch, policy, {switch: value}), defaults[policy]
    for it in range(2 if exact else 1):     # (another product form rounds differently: after its first update the rollouts part ways)
        assert np.array_equal(got[f'{it}.actions'], want[f'{it}.actions'])
        assert np.array_equal(got[f'{it}.stats'], want[f'{it}.stats'], equal_nan=True)
        if exact:
            assert np.array_equal(got[f'{it}.advantages'].view(np.uint32), want[f'{it}.advantages'].view(np.uint32)), (switch, it)
            assert np.array_equal(got[f'{it}.losses'], want[f'{it}.losses'], equal_nan=True), (switch, it, got[f'{it}.losses'], want[f'{it}.losses'])
            assert np.array_equal(got[f'{it}.flat'].view(np.uint32), want[f'{it}.flat'].view(np.uint32)), (switch, it)
        else:
            np.testing.assert_allclose(got[f'{it}.advantages'], want[f'{it}.advantages'], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(got[f'{it}.losses'], want[f'{it}.losses'], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(got[f'{it}.flat'], want[f'{it}.flat'], rtol=1e-5, atol=1e-5)
