"""Host-vecenv rollout path (pufferlib_amd/hostpath.py): a CPU vecenv speaking the reference's
recv/send protocol feeds the device trainer.  With the C oracle's SquaredSerial as that CPU vecenv the whole
create -> evaluate -> train loop must reproduce the device-resident Squared path bit for bit (same envs, same policy kernel,
same Philox rows), whatever order the rows of a batch arrive in."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu

HP = [2.5e-3, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]


def _run(kind, recurrent, n=48, horizon=16, iters=2, order='natural'):
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from host_vecenv import HostSquared
    from test_gpu_ppo import _config
    torch.manual_seed(3)
    if kind == 'device':
        vec = vector.make(vector.make_squared, num_envs=n, backend=vector.Squared)
    else:
        vec = HostSquared(n, order=order)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent else cleanrl.Policy(base)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 8, 2, n * horizon * 8, HP, seed=11), vec, pol)
    out = []
    for _ in range(iters):
        stats, _ = clean_pufferl.evaluate(data)
        e = data.experience
        snap = [x.clone() for x in (e.obs, e.actions, e.logprobs, e.values, e.rewards, e.dones)]
        clean_pufferl.train(data)
        out.append((snap, dict(stats), data.flat_params.flat.clone(), dict(data.losses), data.global_step))
    return out


@pytest.mark.parametrize('recurrent', [False, True])
@pytest.mark.parametrize('order', ['natural', 'reversed', 'shuffled'])
def test_host_vecenv_rollout_equals_device_resident_path(recurrent, order):
    dev = _run('device', recurrent)
    host = _run('host', recurrent, order=order)
    for (sd, std, wd, ld, gd), (sh, sth, wh, lh, gh) in zip(dev, host):
        for x, y in zip(sd, sh):
            assert torch.equal(x, y)
        assert gd == gh
        assert set(std) == set(sth)
        for k in std:
            assert abs(std[k] - sth[k]) < 1e-12, k
        assert torch.equal(wd, wh)
        for k in ld:
            assert ld[k] == lh[k] or (ld[k] != ld[k] and lh[k] != lh[k]), k


def test_store_rows_places_rows_by_env_and_counts_drops():
    from pufferlib_amd import _lib, clean_pufferl
    L = _lib.lib()
    N, T, DP = 5, 3, 16
    exp = clean_pufferl.Experience(N * T, T, None, DP, N, 'cuda')
    counters = torch.zeros(N, dtype=torch.int32, device='cuda')
    sd = torch.zeros(2, dtype=torch.int32, device='cuda')
    rng = np.random.RandomState(0)
    want_obs = np.zeros((N, T, DP), np.float32)
    held = [0] * N
    for step in range(T + 1):                                        # one step too many: every row of it must be dropped
        ids = rng.permutation(N).astype(np.int32)
        mask = np.ones(N, np.uint8)
        if step == 0:
            mask[2] = 0                                               # masked row: not stored, its env lags one step behind
        obs = rng.randn(N, DP).astype(np.float32)
        t = lambda a, dt=None: torch.as_tensor(a if dt is None else a.astype(dt)).cuda()
        rew, done = rng.randn(N).astype(np.float32), (rng.rand(N) < 0.5).astype(np.uint8)
        act, lp, val = rng.randint(0, 8, N).astype(np.int64), rng.randn(N).astype(np.float32), rng.randn(N).astype(np.float32)
        args = [t(obs), t(rew), t(done), t(act), t(lp), t(val), t(ids), t(mask)]
        _lib.check(L.pfa_store_rows(C.byref(exp.c), N, N, DP, *[_lib.ptr(a) for a in args], _lib.ptr(counters), _lib.ptr(sd), None),
                   'store_rows')
        torch.cuda.synchronize()
        for i in range(N):
            e = int(ids[i])
            if mask[i] and held[e] < T:
                want_obs[e, held[e]] = obs[i]
                held[e] += 1
    assert np.array_equal(exp.obs.view(N, T, DP).cpu().numpy(), want_obs)
    assert counters.cpu().tolist() == held == [T] * N
    stored, dropped = sd.cpu().tolist()
    assert stored == N * T and dropped == N - 1                       # the lagging env used the extra step, the others were dropped


def test_wide_observations_take_the_gemm_path():
    """Rows beyond the fused kernels' 128 floats are not rejected any more: they get a stride of the next multiple of 16 and the
    policy adopts a general.GeneralParams buffer (tests/test_gpu_general.py runs them end to end)."""
    from pufferlib_amd import hostpath
    assert hostpath.obs_stride_for(49) == 64 and hostpath.obs_stride_for(128) == 128
    assert hostpath.obs_stride_for(129) == 144 and hostpath.obs_stride_for(1000) == 1008


def test_host_path_throughput_report(capsys):
    """Not a pass/fail bound: prints the end-to-end rate of the host path (C oracle envs on one host core + PCIe both ways
    per step + device policy/update) so DESIGN.md can quote it next to the device-resident rate."""
    import time
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from host_vecenv import HostSquared
    from test_gpu_ppo import _config
    n, horizon = 4096, 32
    vec = HostSquared(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, 2, n * horizon * 64, HP, seed=1), vec, pol)
    for _ in range(2):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = data.profile
    with capsys.disabled():
        print(f'\n[host path] {n} envs x {horizon} steps: {iters * n * horizon / dt / 1e6:.2f} M env steps/s '
              f'(env {p.env.elapsed:.2f}s, forward {p.eval_forward.elapsed:.2f}s, misc {p.eval_misc.elapsed:.2f}s of {dt:.2f}s)')
    assert data.global_step == (iters + 2) * n * horizon


@pytest.mark.parametrize('recurrent', [False, True])
def test_envpool_batches_smaller_than_the_env_count(recurrent):
    """agents_per_batch < num_agents: groups of envs take turns (async EnvPool), so a rollout needs workers*T recvs and every
    recv carries a different env_id range.  Each env's rows must land in its own env-major slot: replaying the stored
    actions of a group on a fresh oracle group reproduces the stored observations / rewards / dones, and (recurrent) the
    LSTM state rows advanced only with their own group."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from host_vecenv import HostSquaredPool
    from oracle import c_oracle
    from test_gpu_ppo import _config
    n, workers, horizon = 48, 3, 8
    per = n // workers
    torch.manual_seed(1)
    vec = HostSquaredPool(n, workers)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent else cleanrl.Policy(base)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 8, HP, seed=9), vec, pol)
    assert data.host_bridge.max_rows == per
    for it in range(2):
        clean_pufferl.evaluate(data)
        e = data.experience
        assert data.global_step == (it + 1) * n * horizon
        obs = e.obs.view(n, horizon, -1)[:, :, :49].cpu().numpy()
        acts, rew, done = (x.view(n, horizon).cpu().numpy() for x in (e.actions, e.rewards, e.dones))
        if it == 0:                                   # replay from reset: group w = envs [w*per, (w+1)*per), seed 9 + w*per
            for w in range(workers):
                ref = c_oracle.SquaredSerial(per, 3, 1)
                ref.async_reset(9 + w * per)
                rows = slice(w * per, (w + 1) * per)
                for t in range(horizon):
                    o, r, d, _, _, _, _ = ref.recv()
                    assert np.array_equal(o.reshape(per, -1), obs[rows, t]), (w, t)
                    assert np.array_equal(r, rew[rows, t]) and np.array_equal(d.astype(np.float32), done[rows, t]), (w, t)
                    ref.send(acts[rows, t].astype(np.int64))
        clean_pufferl.train(data)
        assert torch.isfinite(data.flat_params.flat).all()
    if recurrent:
        h = data.lstm_engine.lstm_h[0]
        assert torch.isfinite(h).all() and float(h.abs().sum()) > 0


def test_uneven_async_pool_fast_workers_return_more_often():
    """A genuinely async pool (vector.py:382-390: first-ready workers): group 0 answers twice per cycle, group 1 once.  The
    rollout keeps stepping until the slow group has its T rows; the fast group's surplus rows are acted on, not stored, and
    nothing raises.  Stored rows of every group replay on a fresh oracle group (the first T steps of each)."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from host_vecenv import HostSquaredPool
    from oracle import c_oracle
    from test_gpu_ppo import _config
    n, workers, horizon = 32, 2, 8
    per = n // workers
    vec = HostSquaredPool(n, workers, schedule=[0, 0, 1])
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 8, HP, seed=4), vec, pol)
    clean_pufferl.evaluate(data)
    e = data.experience
    assert data.host_rows_dropped > 0 and e.ptr == e.batch_size
    assert vec.turn == 3 * horizon                                     # the slow group needed T of its turns
    obs = e.obs.view(n, horizon, -1)[:, :, :49].cpu().numpy()
    acts, rew = (x.view(n, horizon).cpu().numpy() for x in (e.actions, e.rewards))
    for w in range(workers):
        ref = c_oracle.SquaredSerial(per, 3, 1)
        ref.async_reset(4 + w * per)
        rows = slice(w * per, (w + 1) * per)
        for t in range(horizon):
            o, r, d, _, _, _, _ = ref.recv()
            assert np.array_equal(o.reshape(per, -1), obs[rows, t]), (w, t)
            assert np.array_equal(r, rew[rows, t]), (w, t)
            ref.send(acts[rows, t].astype(np.int64))
    clean_pufferl.train(data)
    assert torch.isfinite(data.flat_params.flat).all()


def test_reference_exact_store_on_an_uneven_pool():
    """config.async_store = 'reference' (VERDICT r2 item 10): Experience.store + sort_training_data as the reference runs them when
    agents report unevenly (clean_pufferl.py:436-464) — the FIRST batch_size masked rows in arrival order, one stable sort by
    (env_id, step), runs of different lengths per agent — restated here in numpy from the recorded recv()/send() stream."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from host_vecenv import HostSquaredPool
    from oracle import c_oracle
    from test_gpu_ppo import _config
    n, workers, horizon = 32, 2, 8
    B = n * horizon
    inner = HostSquaredPool(n, workers, schedule=[0, 0, 1])
    log = dict(recv=[], send=[])

    class Recording:
        def __getattr__(self, k):
            return getattr(inner, k)

        def recv(self):
            out = inner.recv()
            log['recv'].append(tuple(np.array(x).copy() if not isinstance(x, list) else x for x in out))
            return out

        def send(self, actions):
            log['send'].append(np.array(actions).copy())
            inner.send(actions)
    vec = Recording()
    pol = cleanrl.Policy(models.Default(inner.driver_env))
    data = clean_pufferl.create(_config(n, horizon, B // 2, 4, 1, B * 8, HP, seed=4, async_store='reference'), vec, pol)
    clean_pufferl.evaluate(data)
    e = data.experience
    # the reference's store + sort on the recorded stream
    rows, ptr = [], 0
    for step, ((o, r, d, t, info, env_id, mask), a) in enumerate(zip(log['recv'], log['send'])):
        idx = np.nonzero(mask)[0][:B - ptr]
        for i in idx:
            rows.append((int(env_id[i]), step, o[i].reshape(-1), float(r[i]), float(d[i]), int(np.asarray(a).reshape(-1)[i])))
        ptr += len(idx)
    assert ptr == B and len(log['recv']) == len(log['send'])
    order = sorted(range(B), key=lambda j: (rows[j][0], rows[j][1]))
    per_agent = np.bincount([rows[j][0] for j in range(B)], minlength=n)
    assert per_agent.min() < horizon < per_agent.max()                    # genuinely uneven: the fast group holds more rows than the slow one
    want_obs = np.stack([rows[j][2] for j in order])
    assert np.array_equal(e.obs[:, :49].cpu().numpy(), want_obs)
    assert np.array_equal(e.rewards.cpu().numpy(), np.array([rows[j][3] for j in order], np.float32))
    assert np.array_equal(e.dones.cpu().numpy(), np.array([rows[j][4] for j in order], np.float32))
    assert np.array_equal(e.actions.cpu().numpy(), np.array([rows[j][5] for j in order], np.int32))
    assert data.host_rows_dropped == sum(int(m.sum()) for (_, _, _, _, _, _, m) in log['recv']) - B
    clean_pufferl.train(data)
    # GAE is the reference's single scan over the sorted flat batch (clean_pufferl.py:163-169)
    want = c_oracle.compute_gae(e.dones.cpu().numpy(), e.values.cpu().numpy(), e.rewards.cpu().numpy(), HP[1], HP[2])
    np.testing.assert_allclose(e.advantages.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert torch.isfinite(data.flat_params.flat).all() and np.isfinite(data.losses.explained_variance)


class _ReplayPool:
    """A host vecenv that hands out, recv by recv, exactly what the REFERENCE's pufferlib.vector.Multiprocessing backend handed
    the reference trainer in tests/golden/ppo_mp.npz (EnvPool mode: 8 of 16 envs per recv, shared-memory worker processes), and
    checks that every send() carries the actions the reference sent."""

    def __init__(self, g):
        import json
        from pufferlib_amd import vector
        self.g = g
        n, _, _, _, _, _, _, per, _ = (int(x) for x in g['config'])
        self.driver_env = vector.SquaredSpec(3, 1)
        self.single_observation_space = self.driver_env.single_observation_space
        self.single_action_space = self.driver_env.single_action_space
        self.num_envs = self.num_agents = n
        self.agents_per_batch = per
        self.agent_ids = np.arange(n)
        self.emulated = True
        self.infos = json.loads(str(g['recv.infos']))
        self.k = 0

    def async_reset(self, seed=42):
        pass

    def recv(self):
        g, k = self.g, self.k
        assert k < len(g['recv.obs']), 'the trainer asked for more batches than the reference run consumed'
        return (g['recv.obs'][k].astype(np.float32).reshape(-1, 7, 7), g['recv.rewards'][k], g['recv.terminals'][k].astype(bool),
                g['recv.truncations'][k].astype(bool), self.infos[k], g['recv.env_id'][k].astype(np.int64), g['recv.mask'][k].astype(bool))

    def send(self, actions):
        assert np.array_equal(np.asarray(actions).reshape(-1), self.g['send.actions'][self.k].astype(np.int64)), self.k
        self.k += 1

    def close(self):
        pass


def test_host_path_replays_the_reference_run_over_its_multiprocessing_backend(golden_dir):
    """create -> evaluate -> train on the batches the unmodified reference received from its own Multiprocessing backend
    (tests/golden/ppo_mp.npz, generated by tests/golden/make_golden.py::gen_ppo_mp): same actions back to every recv (bit-exact,
    under the recorded multinomial noise), the env-major experience equals what the reference's store + sort_training_data
    left (exact for observations / rewards / dones / actions, 1e-5 for log-probabilities and values), and advantages, returns,
    losses and weights after each update match to 1e-5.  (explained_variance is left out: the reference's formula pairs rows
    by ARRIVAL position, which for a pool depends on worker timing.)"""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import TOL, _config, _load_weights
    g = np.load(os.path.join(golden_dir, 'ppo_mp.npz'))
    n, horizon, mbs, bptt, epochs, total, iters, per, _ = (int(x) for x in g['config'])
    vec = _ReplayPool(g)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    _load_weights(pol, g, 'w0.')
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, [float(x) for x in g['hparams']]), vec, pol)
    exp = data.experience
    A = 8
    for it in range(iters):
        k0, k1 = (int(x) for x in g[f'it{it}.recvs'])
        assert vec.k == k0
        assert abs(data.optimizer.param_groups[0]['lr'] - float(g[f'it{it}.lr_used'])) < 1e-12
        noise = torch.ones(k1 - k0, n, A)
        for k in range(k0, k1):                                   # rows of recv k belong to the agents it reported
            noise[k - k0, torch.as_tensor(g['recv.env_id'][k].astype(np.int64))] = torch.as_tensor(g['recv.noise'][k])
        data.noise = noise
        stats, _ = clean_pufferl.evaluate(data)
        assert vec.k == k1 and data.host_rows_dropped == 0
        assert np.array_equal(exp.actions.cpu().numpy(), g[f'it{it}.actions'].astype(np.int32))
        assert np.array_equal(exp.obs.cpu().numpy()[:, :49], g[f'it{it}.obs'].astype(np.float32))
        assert np.array_equal(exp.rewards.cpu().numpy(), g[f'it{it}.rewards'])
        assert np.array_equal(exp.dones.cpu().numpy(), g[f'it{it}.dones'])
        np.testing.assert_allclose(exp.logprobs.cpu().numpy(), g[f'it{it}.logprobs'], **TOL)
        np.testing.assert_allclose(exp.values.cpu().numpy(), g[f'it{it}.values'], **TOL)
        assert data.global_step == int(g[f'it{it}.global_step'])
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-9)
        clean_pufferl.train(data)
        for m in range(exp.num_minibatches):
            idx = exp.minibatch_rows_index(m)
            np.testing.assert_allclose(exp.advantages[idx].cpu().numpy(), g[f'it{it}.advantages'][m], **TOL)
            np.testing.assert_allclose(exp.returns[idx].cpu().numpy(), g[f'it{it}.returns'][m], **TOL)
        L = data.losses
        got = [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.{k}'], err_msg=k, **TOL)
