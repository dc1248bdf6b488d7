"""HIP Squared vecenv vs the golden reference trajectories and vs the C oracle (bit-exact)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vec(n, d, nt, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=d, num_targets=nt), num_envs=n,
                       backend=vector.Squared, **kw)


@pytest.mark.parametrize('tag', ['d3t1', 'd1t4', 'd2t2', 'd4t3', 'd3t1_big'])
def test_golden_trajectory_bit_exact(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f'squared_{tag}.npz'))
    n, d, nt, seed, steps = (int(x) for x in g['config'])
    v = _vec(n, d, nt)
    v.async_reset(seed)
    infos = []
    gsz = 2 * d + 1
    for k in range(steps + 1):
        o, r, te, tr, info, ids, masks = v.recv()
        assert tuple(o.shape) == (n, gsz, gsz)
        assert np.array_equal(o.cpu().numpy(), g['obs'][k].astype(np.float32)), (tag, k)
        assert np.array_equal(r.cpu().numpy().view(np.uint32), g['rewards'][k].view(np.uint32)), (tag, k)
        assert np.array_equal(te.cpu().numpy(), g['terminals'][k])
        assert np.array_equal(tr.cpu().numpy(), g['truncations'][k])
        assert masks.cpu().numpy().all()
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        got_t, want_t = v.debug_targets(), g['targets'][k]
        for e in range(n):   # remaining targets, in draw order (the golden list is compacted by list.remove)
            assert [c for c in got_t[e] if c >= 0] == [c for c in want_t[e] if c >= 0], (tag, k, e)
        if k < steps:
            v.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])
    # padding columns of the live buffer stay zero
    assert float(v.obs_buf[:, v.obs_dim:].abs().sum()) == 0.0


@pytest.mark.parametrize('n,d,nt,seed', [(4096, 3, 1, 1), (1000, 2, 3, 12345), (257, 5, 7, 2 ** 33 + 9), (64, 1, 1, 0)])
def test_matches_oracle_at_size(n, d, nt, seed):
    from oracle import c_oracle
    v = _vec(n, d, nt)
    ref = c_oracle.SquaredSerial(n, d, nt)
    v.async_reset(seed)
    ref.async_reset(seed)
    rng = np.random.RandomState(7)
    steps = 3 * (nt * d + 1) + 2
    for k in range(steps):
        o, r, te, tr, info, _, _ = v.recv()
        ro, rr, rte, rtr, rinfo, _, _ = ref.recv()
        assert np.array_equal(o.cpu().numpy(), ro), k
        assert np.array_equal(r.cpu().numpy().view(np.uint32), rr.view(np.uint32)), k
        assert np.array_equal(te.cpu().numpy(), rte), k
        assert len(info) == len(rinfo)
        for a, b in zip(info, rinfo):
            assert a['episode_return'] == b['episode_return'] and a['episode_length'] == b['episode_length']
            assert a['score'] == b['score']
        a = rng.randint(0, 8, size=n)
        v.send(a)
        ref.send(a)
    assert v.debug_stream_pos() == ref.stream_pos()


def test_protocol_misuse():
    from pufferlib_amd import vector
    from pufferlib_amd.exceptions import APIUsageError
    v = _vec(4, 3, 1)
    with pytest.raises(APIUsageError):
        v.recv()                      # step before reset
    with pytest.raises(APIUsageError):
        v.send(np.zeros(4, np.int64))
    v.async_reset(1)
    v.recv()
    with pytest.raises(APIUsageError):
        v.recv()                      # recv twice
    with pytest.raises(APIUsageError):
        v.send(np.array([0, 1, 2, 9]))  # action out of space
    with pytest.raises(APIUsageError):
        vector.make(vector.make_squared, num_envs=0)
    with pytest.raises(APIUsageError):
        vector.make(vector.make_squared, num_envs=4, bogus=1)


def test_reset_step_helpers_and_lazy_infos():
    v = _vec(8, 3, 1, info_mode='lazy')
    obs, infos = v.reset(seed=3)
    assert infos == [] and tuple(obs.shape) == (8, 7, 7)
    for _ in range(8):
        obs, rew, term, trunc, infos = v.step(np.zeros(8, np.int64))
        assert infos == []
    st = v.episode_stats().cpu().numpy()
    assert st[0] == 16 and st[2] == 16 * 3   # two finished 3-step episodes per env


def test_reset_target_tape_keeps_ahead_of_back_to_back_rollouts():
    """with update_epochs = num_minibatches = 1 (the only setting in which north_star's "single
    all-reduce per update" is literal) train() shrinks to one optimizer step, rollouts run nearly back to back, and the side-stream
    draw of the NEXT rollout's reset targets (the sequential MT19937 recurrence: csrc/squared.hip squared_tape_words_kernel) has one
    rollout's time to finish.  20 iterations at the BASELINE shape: never a tape underrun (evaluate() raises on one), and the whole
    tape draw (words + the two parallel passes) takes at most 0.6 of the rollout it hides under."""
    import ctypes as C
    import torch
    from pufferlib_amd import _lib, clean_pufferl, cleanrl, models, vector
    from test_gpu_ppo import _config
    n, horizon = 4096, 128
    B = n * horizon
    vec = vector.make(vector.make_squared, num_envs=n, backend=vector.Squared, obs_stride=64)
    torch.manual_seed(3)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, B, 16, 1, B * 64, [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01], seed=2), vec, pol)
    L = _lib.lib()
    for _ in range(4):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    L.pfa_timing_reset()
    L.pfa_timing_enable(2)
    for _ in range(20):
        stats, _ = clean_pufferl.evaluate(data)      # (a tape underrun raises here, from the readback's flag)
        clean_pufferl.train(data)
    torch.cuda.synchronize()
    L.pfa_timing_enable(0)

    def ms(name):
        k, t = C.c_int64(0), C.c_double(0.0)
        _lib.check(L.pfa_timing_read(name.encode(), C.byref(k), C.byref(t)), 'timing_read')
        return k.value, t.value
    (n_tape, t_tape), (n_roll, t_roll) = ms('squared_tape'), ms('rollout_mlp_squared')
    assert n_tape == 20 and n_roll == 20, (n_tape, n_roll)
    print(f'[tape] {t_tape / n_tape * 1e3:.1f} us per draw vs {t_roll / n_roll * 1e3:.1f} us per rollout')
    assert t_tape / n_tape <= 0.6 * t_roll / n_roll, (t_tape / n_tape, t_roll / n_roll)
    assert np.isfinite(stats['episode_return']) and torch.isfinite(data.flat_params.flat).all()
