"""ocean.Spaces (ocean.py:356-404) as a device-resident vecenv (csrc/spaces.hip): Dict observation emulated to 108-byte rows,
Dict action emulated to MultiDiscrete([2, 2]), observations from numpy's process-global legacy generator whose
data-dependent stream positions the tape kernel resolves in parallel.  Against (a) the C oracle (sequential restatement,
pinned against numpy and the reference) over many reset rounds and window boundaries, bit for bit; (b) the unmodified
reference's create/evaluate/train run on Spaces (tests/golden/ppo_spaces.npz) — this time with the observations GENERATED on
device from the seed instead of played back."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


def _make(n, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_spaces, num_envs=n, backend=vector.Spaces, **kw)


@pytest.mark.parametrize('n,seed,sends', [(1, 1, 61), (37, 7, 41), (700, 42, 24), (4096, 3, 9)])
def test_protocol_path_equals_the_oracle_bit_for_bit(n, seed, sends):
    from oracle import c_oracle
    dev = _make(n)
    ref = c_oracle.SpacesSerial(n, global_seed=seed)
    dev.async_reset(seed)
    ref.async_reset(seed)
    rng = np.random.default_rng(n)
    for t in range(sends):
        o, r, te, tr, infos, ids, mask = dev.recv()
        o2, r2, te2, _, infos2, _, _ = ref.recv()
        assert o.dtype == torch.uint8 and tuple(o.shape) == (n, 108)
        assert np.array_equal(o.cpu().numpy(), o2), f'observation bytes differ at send {t}'
        assert np.array_equal(r.cpu().numpy(), r2) and np.array_equal(te.cpu().numpy(), te2), t
        assert len(infos) == len(infos2)
        for a, b in zip(infos, infos2):
            assert a['score'] == b['score'] and a['episode_return'] == b['episode_return'] and a['episode_length'] == 1
        a = rng.integers(0, 2, (n, 2)).astype(np.int64)
        dev.send(a)
        ref.send(a)
    st = dev.stats_with_flag(reset=False).cpu().numpy()
    assert st[4] == 0                                   # no tape underrun


def test_protocol_misuse_and_spaces():
    from pufferlib_amd.exceptions import APIUsageError
    vec = _make(8)
    assert vec.single_action_space.nvec.tolist() == [2, 2] and vec.single_observation_space.shape == (108,)
    assert vec.emulated.emulated_observation_dtype.itemsize == 108
    with pytest.raises(APIUsageError):
        vec.recv()                                       # before reset
    vec.async_reset(1)
    vec.recv()
    with pytest.raises(APIUsageError, match='Actions do not match action space'):
        vec.send(np.zeros(8, np.int64))                  # wrong shape
    vec2 = _make(8)
    vec2.async_reset(1)
    vec2.recv()
    with pytest.raises(APIUsageError, match='Actions do not match action space'):
        vec2.send(np.full((8, 2), 2, np.int64))          # out of range


def _spaces_noise(g, it, n, horizon):
    q = g[f'it{it}.noise']
    return q.reshape(horizon, 2, n, 2).transpose(0, 2, 1, 3).reshape(horizon, n, 4)


def test_golden_run_of_the_reference_reproduced_from_the_seed(golden_dir):
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    g = np.load(os.path.join(golden_dir, 'ppo_spaces.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    hp = [float(x) for x in g['hparams']]
    vec = _make(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    pol.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith('w0.')})
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, hp, seed=1), vec, pol)
    assert data.host_bridge is None and data.flat_params.obs_stride == 128 and data.flat_params.nvec == [2, 2]
    step_major = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(n * horizon, *x.shape[1:]).cpu().numpy()  # noqa: E731
    for it in range(iters):
        data.noise = torch.as_tensor(_spaces_noise(g, it, n, horizon))
        stats, _ = clean_pufferl.evaluate(data)
        e = data.experience
        assert np.array_equal(step_major(e.obs)[:, :108], g[f'it{it}.obs'].astype(np.float32)), 'observation stream differs'
        acts = data.flat_params.unpack_actions(e.actions.long())
        assert np.array_equal(step_major(acts), g[f'it{it}.actions'].astype(np.int64)), 'actions differ'
        assert np.array_equal(step_major(e.rewards), g[f'it{it}.rewards']) and np.array_equal(step_major(e.dones), g[f'it{it}.dones'])
        np.testing.assert_allclose(step_major(e.logprobs), g[f'it{it}.logprobs'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-12)
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                                   g[f'it{it}.losses'][:6], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.' + k], rtol=1e-5, atol=1e-5, err_msg=k)


def test_train_loop_runs_on_device_spaces():
    """create -> evaluate -> train on the device env with the Philox noise (no host sync per step): statistics are those of
    one-step episodes, the weights move and stay finite.  (Whether PPO learns the task from raw float32 BYTES is a property of
    the reference's Default policy on emulated rows, not of this backend, and is not asserted.)"""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    n, horizon = 512, 32
    hp = [2.5e-4, 0.99, 0.95, 0.2, 0.5, 0.2, 0.5, 0.001]
    vec = _make(n)
    torch.manual_seed(0)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 8, 2, n * horizon * 40, hp, seed=5), vec, pol)
    w0 = data.flat_params.flat.clone()
    for it in range(4):
        stats, _ = clean_pufferl.evaluate(data)
        assert stats['episode_length'] == 1 and stats['score'] == stats['episode_return'] and 0.0 <= stats['score'] <= 1.0
        clean_pufferl.train(data)
    assert data.global_step == 4 * n * horizon
    assert torch.isfinite(data.flat_params.flat).all() and not torch.equal(w0, data.flat_params.flat)


def test_long_horizon_on_few_envs_draws_the_tape_in_chunks():
    """batch_size / num_envs >= 512 on Spaces needs more reset rounds than the tape ring holds at once; the stepwise
    rollout draws them in chunks (clean_pufferl._rollout_stepwise) instead of asking for the whole rollout up front."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    n, horizon = 2, 1024
    hp = [2.5e-4, 0.99, 0.95, 0.2, 0.5, 0.2, 0.5, 0.001]
    vec = _make(n)
    assert horizon // 2 >= vec.tape_rounds                      # the up-front request this used to make would be refused
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 8, 1, n * horizon * 4, hp, seed=5), vec, pol)
    stats, _ = clean_pufferl.evaluate(data)
    assert stats['episode_length'] == 1 and data.global_step == n * horizon
    clean_pufferl.train(data)
    assert torch.isfinite(data.flat_params.flat).all()
