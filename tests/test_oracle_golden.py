"""Pin the CPU oracle (oracle/) against outputs of the unmodified reference (tests/golden/*.npz,
produced by tests/golden/make_golden.py) and the known answers in SURVEY.md Appendix B."""
import os
import sys
import random

import numpy as np
import pytest
import torch

from oracle import c_oracle, ppo_torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cnn_golden  # noqa: E402


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ---------------------------------------------------------------- MT19937 / random.sample
@pytest.mark.parametrize('seed', [0, 1, 42, 4097, 2 ** 32 + 5, 2 ** 45 + 77])
def test_mt_stream_matches_cpython(seed):
    random.seed(seed)
    m = c_oracle.MT().seed(seed)
    assert [random.getrandbits(32) for _ in range(1500)] == [m.u32() for _ in range(1500)]


@pytest.mark.parametrize('n,k', [(24, 1), (8, 4), (16, 2), (32, 3), (21, 5), (22, 5), (85, 6), (86, 6), (1, 1)])
def test_sample_matches_cpython(n, k):
    random.seed(11)
    m = c_oracle.MT().seed(11)
    for _ in range(300):
        assert random.sample(range(n), k) == m.sample(n, k)


def test_appendix_b_first_targets():
    # first-episode targets after reset(seed=s), d=3, nt=1
    want = {1: (0, 4), 2: (0, 1), 3: (1, 0), 4: (1, 0), 42: (6, 3)}
    for s, t in want.items():
        v = c_oracle.SquaredSerial(1, 3, 1)
        v.async_reset(s)
        assert v.targets(0) == [t]


# ---------------------------------------------------------------- Squared Serial vecenv
@pytest.mark.parametrize('tag', ['d3t1', 'd1t4', 'd2t2', 'd4t3', 'd3t1_big'])
def test_squared_trajectory_bit_exact(golden_dir, tag):
    g = _load(golden_dir, f'squared_{tag}.npz')
    n, d, nt, seed, steps = (int(x) for x in g['config'])
    v = c_oracle.SquaredSerial(n, d, nt)
    v.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, masks = v.recv()
        assert np.array_equal(o.astype(np.int8), g['obs'][k]), (tag, k)
        assert np.array_equal(o, g['obs'][k].astype(np.float32))
        assert np.array_equal(r.view(np.uint32), g['rewards'][k].view(np.uint32)), (tag, k)
        assert np.array_equal(te, g['terminals'][k]) and np.array_equal(tr, g['truncations'][k])
        assert masks.all()
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        tg = g['targets'][k]
        for e in range(n):
            want = [(c // v.g, c % v.g) for c in tg[e] if c >= 0]
            assert v.targets(e) == want, (tag, k, e)
        if k < steps:
            v.send(g['actions'][k].astype(np.int64))
    got = np.array(infos, np.float64).reshape(-1, 5)
    assert np.array_equal(got, g['infos'])  # python-float sums reproduced exactly


# ---------------------------------------------------------------- GAE
def test_gae_matches_reference(golden_dir):
    g = _load(golden_dir, 'gae.npz')
    cases = sorted({k.split('_')[0] for k in g.files})
    assert len(cases) == 8
    for c in cases:
        gamma, lam = g[c + '_gl']
        adv = c_oracle.compute_gae(g[c + '_dones'], g[c + '_values'], g[c + '_rewards'], gamma, lam)
        assert np.array_equal(adv.view(np.uint32), g[c + '_adv'].view(np.uint32)), c


def test_gae_appendix_b():
    adv = c_oracle.compute_gae(np.zeros(8), np.arange(8), np.ones(8), .99, .95)
    want = [11.514251, 10.126795, 8.662194, 7.11557, 5.481733, 3.7551653, 1.9300003, 0.]
    assert np.allclose(adv, want, rtol=1e-6, atol=0)


def test_ref_c_gae_agrees_when_built(golden_dir):
    """oracle/_ref/c_gae*.so is the reference's own Cython kernel compiled from /root/reference."""
    import importlib.util
    import glob
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = glob.glob(os.path.join(here, 'oracle', '_ref', 'c_gae*.so'))
    if not so:
        pytest.skip('oracle/_ref not built (no /root/reference here)')
    spec = importlib.util.spec_from_file_location('c_gae', so[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.RandomState(5)
    d = (rng.rand(5000) < .05).astype(np.float32)
    v, r = rng.randn(5000).astype(np.float32), rng.randn(5000).astype(np.float32)
    a = mod.compute_gae(d, v, r, .99, .95)
    b = c_oracle.compute_gae(d, v, r, .99, .95)
    assert np.array_equal(np.asarray(a).view(np.uint32), b.view(np.uint32))


# ---------------------------------------------------------------- policy / PPO update
def _replay(golden_dir, tag):
    g = _load(golden_dir, f'ppo_{tag}.npz')
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = (float(x) for x in g['hparams'])
    sd = {k[3:]: g[k] for k in g.files if k.startswith('w0.')}
    pol = ppo_torch.Policy.from_reference_state_dict(sd)
    vec = c_oracle.SquaredSerial(n, 3, 1)
    tr = ppo_torch.Trainer(pol, vec, batch_size=n * horizon, minibatch_size=mbs, bptt_horizon=bptt,
                           update_epochs=epochs, learning_rate=lr, gamma=gamma, gae_lambda=lam, clip_coef=clip,
                           vf_coef=vf_coef, vf_clip_coef=vf_clip, max_grad_norm=mgn, ent_coef=ent,
                           total_timesteps=total, seed=1)
    return g, pol, tr, iters


@pytest.mark.parametrize('tag', ['mlp', 'lstm', 'mlp_h256'])     # mlp_h256: models.Default(hidden_size=256)
def test_ppo_replay_matches_reference(golden_dir, tag):
    torch.set_num_threads(1)
    g, pol, tr, iters = _replay(golden_dir, tag)
    recurrent = tag == 'lstm'
    for it in range(iters):
        assert abs(tr.opt.param_groups[0]['lr'] - float(g[f'it{it}.lr_used'])) < 1e-15
        stats = tr.evaluate(g[f'it{it}.noise'])
        assert np.array_equal(tr.actions, g[f'it{it}.actions'].astype(np.int64)), 'actions differ'
        assert np.array_equal(tr.obs.numpy().astype(np.int8), g[f'it{it}.obs'])
        assert np.array_equal(tr.rewards, g[f'it{it}.rewards'])
        assert np.array_equal(tr.dones, g[f'it{it}.dones'])
        tol = dict(rtol=1e-5, atol=1e-6) if recurrent else dict(rtol=0, atol=0)
        np.testing.assert_allclose(tr.logprobs, g[f'it{it}.logprobs'], **tol)
        np.testing.assert_allclose(tr.values, g[f'it{it}.values'], **tol)
        assert tr.global_step == int(g[f'it{it}.global_step'])
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']],
                                   g[f'it{it}.stats'], rtol=1e-12)
        L = tr.train()
        np.testing.assert_allclose(tr.b_advantages.numpy(), g[f'it{it}.advantages'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(tr.b_returns.numpy(), g[f'it{it}.returns'], rtol=1e-5, atol=1e-6)
        assert np.array_equal(tr.b_idxs.numpy(), g[f'it{it}.b_idxs'])
        got = [L['policy_loss'], L['value_loss'], L['entropy'], L['old_approx_kl'], L['approx_kl'], L['clipfrac'],
               L['explained_variance']]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=2e-5, atol=1e-7)
        prefix = 'policy.policy.' if recurrent else 'policy.'
        m, v = tr.adam_moments()
        for name, arr in pol.state_arrays().items():
            key = ('policy.recurrent.' + name) if name.endswith('_l0') else (prefix + name)
            np.testing.assert_allclose(arr, g[f'it{it}.w.' + key], rtol=1e-5, atol=1e-6, err_msg=name)
            np.testing.assert_allclose(m[name], g[f'it{it}.m.' + key], rtol=1e-4, atol=1e-7, err_msg=name)
            np.testing.assert_allclose(v[name], g[f'it{it}.v.' + key], rtol=1e-4, atol=1e-9, err_msg=name)
        assert abs(tr.opt.param_groups[0]['lr'] - float(g[f'it{it}.lr_next'])) < 1e-15


def _close_digest(got, want, what, atol=1e-6, rtol=1e-5):
    got = np.asarray(got)
    d = cnn_golden.digest(got)
    np.testing.assert_allclose(d[2:], want[2:], rtol=rtol, atol=atol, err_msg=what)
    tol = 0.1 * atol * got.size + rtol * abs(want[1])
    assert abs(d[0] - want[0]) <= tol and abs(d[1] - want[1]) <= tol, (what, d[:2], want[:2], tol)


@pytest.mark.parametrize('tag', ['c1_mlp', 'c1_lstm', 'demo_lstm', 'c2_mlp'])
def test_ppo_replay_matches_reference_at_baseline_sizes(golden_dir, tag):
    """The oracle against the reference at BASELINE's own sizes (digest-form fixtures of `make_golden.py big`): configs[0] with both
    policies, the shape demo.py --env squared trains (bptt 4, 8 minibatches, lr 0.017), one iteration of configs[1] (4096 x 128).
    c2's multinomial noise is not stored (16.8 MB): regenerated as the reference drew it when this box's torch draws the same
    numbers (digest check), else noise that forces the recorded actions."""
    import hashlib
    torch.set_num_threads(8 if tag == 'c2_mlp' else 1)
    g, pol, tr, iters = _replay(golden_dir, f'{tag}')
    n, horizon = int(g['config'][0]), int(g['config'][1])
    recurrent = tag.endswith('lstm')
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for it in range(iters):
        assert abs(tr.opt.param_groups[0]['lr'] - float(g[f'it{it}.lr_used'])) < 1e-15
        if f'it{it}.noise' in g.files:
            noise = g[f'it{it}.noise']
        else:
            torch.manual_seed(1)
            for _ in range(it + 1):
                noise = np.stack([torch.empty(n, 8).exponential_(1).numpy() for _ in range(horizon)])
            if not np.array_equal(cnn_golden.digest(noise), g[f'it{it}.noise_digest']):
                noise = np.ones((horizon, n, 8), np.float32)
                np.put_along_axis(noise, g[f'it{it}.actions'].reshape(horizon, n, 1).astype(np.int64), np.float32(1e-30), axis=2)
        stats = tr.evaluate(noise)
        assert np.array_equal(tr.actions, g[f'it{it}.actions'].astype(np.int64)), 'actions differ'
        assert sha(tr.obs.numpy().astype(np.int8)) == str(g[f'it{it}.obs_sha'])
        assert sha(tr.rewards.astype(np.float32)) == str(g[f'it{it}.rewards_sha']) and sha(tr.dones.astype(np.float32)) == str(g[f'it{it}.dones_sha'])
        _close_digest(tr.logprobs, g[f'it{it}.logprobs'], 'logprobs')
        _close_digest(tr.values, g[f'it{it}.values'], 'values')
        assert tr.global_step == int(g[f'it{it}.global_step'])
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-12)
        L = tr.train()
        _close_digest(tr.b_advantages.numpy(), g[f'it{it}.advantages'], 'advantages')
        _close_digest(tr.b_returns.numpy(), g[f'it{it}.returns'], 'returns')
        got = [L['policy_loss'], L['value_loss'], L['entropy'], L['old_approx_kl'], L['approx_kl'], L['clipfrac'], L['explained_variance']]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=2e-5, atol=1e-6)
        prefix = 'policy.policy.' if recurrent else 'policy.'
        m, v = tr.adam_moments()
        for name, arr in pol.state_arrays().items():
            key = ('policy.recurrent.' + name) if name.endswith('_l0') else (prefix + name)
            _close_digest(arr, g[f'it{it}.w.' + key], name, atol=2e-6)
            _close_digest(m[name], g[f'it{it}.m.' + key], name, atol=1e-7, rtol=2e-4)
            _close_digest(v[name], g[f'it{it}.v.' + key], name, atol=1e-9, rtol=2e-4)
        assert abs(tr.opt.param_groups[0]['lr'] - float(g[f'it{it}.lr_next'])) < 1e-15


def test_demo_shape_update_sits_on_a_knife_edge(golden_dir):
    """Why tests/test_gpu_big_goldens.py tolerates a few isolated weight entries on the demo shape (config.yaml:498-509: lr 0.017,
    4 epochs x 8 minibatches of 128 rows): the reference's OWN update is not reproducible to 1e-5 in every entry there.  The oracle
    (bit-compatible with the reference on this fixture: test above) re-run with every gradient entry multiplied by 1 + 1e-6 N(0, 1) —
    the size of any re-ordered fp32 summation — lands either on the recorded weights (all entries within 2e-6) or on ANOTHER set:
    a few dozen entries of the encoder (one hidden unit, the few observation cells occupied in one row) off by up to ~7e-3, the rest
    unchanged.  One row sits within rounding distance of a branch point (a ReLU kink / a clipping branch), and Adam's 32 steps at
    lr 0.017 turn the flipped branch into a visible offset of exactly those entries."""
    torch.set_num_threads(1)
    g = _load(golden_dir, 'ppo_demo_lstm.npz')

    def run(eps, seed):
        _, pol, tr, _ = _replay(golden_dir, 'demo_lstm')
        gen = torch.Generator().manual_seed(seed)
        plain = tr.opt.step

        def step(*a, **k):
            if eps:
                with torch.no_grad():
                    for p in pol.params:
                        if p.grad is not None:
                            p.grad.mul_(1 + eps * torch.randn(p.grad.shape, generator=gen))
            return plain(*a, **k)
        tr.opt.step = step
        tr.evaluate(g['it0.noise'])
        tr.train()
        return pol.state_arrays()
    base = run(0.0, 0)
    outcomes = []
    for seed in range(4):
        w = run(1e-6, seed)
        worst = max(float(np.abs(w[k] - base[k]).max()) for k in base)
        n_off = sum(int((np.abs(w[k] - base[k]) > 1e-5).sum()) for k in base)
        total = sum(base[k].size for k in base)
        outcomes.append((worst, n_off))
        assert worst < 2e-6 or (1e-4 < worst < 0.5 * 0.017 and n_off < 0.01 * total), (seed, worst, n_off)
    assert any(w < 2e-6 for w, _ in outcomes) and any(w > 1e-4 for w, _ in outcomes), outcomes      # both branches are reachable


def test_conv_ppo_replay_matches_reference(golden_dir):
    """oracle ConvPolicy + Trainer against the unmodified reference's run with models.Convolutional (ppo_cnn.npz): same initial
    weights (digests), identical actions given the recorded multinomial noise, log-probabilities / values / losses / updated weights."""
    import cnn_golden
    torch.set_num_threads(1)
    g = _load(golden_dir, 'ppo_cnn.npz')
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = (float(x) for x in g['hparams'])
    net = cnn_golden.container()
    for k, v in net.state_dict().items():      # layer_init of the container == the reference's, up to the QR's rounding
        np.testing.assert_allclose(cnn_golden.digest(v.numpy()), g['init.policy.' + k], rtol=1e-5, atol=1e-6, err_msg=k)
    w0 = cnn_golden.start_weights(net)
    for k, v in w0.items():
        assert np.array_equal(cnn_golden.digest(v), g['w0.policy.' + k]), k
    pol = ppo_torch.ConvPolicy(w0)
    tr = ppo_torch.Trainer(pol, cnn_golden.ReplayVec(g), batch_size=n * horizon, minibatch_size=mbs, bptt_horizon=bptt, update_epochs=epochs,
                           learning_rate=lr, gamma=gamma, gae_lambda=lam, clip_coef=clip, vf_coef=vf_coef, vf_clip_coef=vf_clip,
                           max_grad_norm=mgn, ent_coef=ent, total_timesteps=total, seed=1)
    tr.evaluate(g['it0.noise'])
    assert np.array_equal(tr.actions, g['it0.actions'].astype(np.int64)), 'actions differ'
    np.testing.assert_allclose(tr.logprobs, g['it0.logprobs'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tr.values, g['it0.values'], rtol=1e-5, atol=1e-6)
    L = tr.train()
    np.testing.assert_allclose(tr.b_advantages.numpy(), g['it0.advantages'], rtol=1e-5, atol=1e-6)
    got = [L['policy_loss'], L['value_loss'], L['entropy'], L['old_approx_kl'], L['approx_kl'], L['clipfrac'], L['explained_variance']]
    np.testing.assert_allclose(got, g['it0.losses'], rtol=2e-5, atol=1e-7)
    for name, arr in pol.state_arrays().items():
        np.testing.assert_allclose(cnn_golden.digest(arr), g['it0.w.policy.' + name], rtol=1e-5, atol=1e-6, err_msg=name)


def test_recurrent_conv_ppo_replay_matches_reference(golden_dir):
    """oracle RecurrentConvPolicy + Trainer against the unmodified reference's run with LSTMWrapper(512, 512) over
    models.Convolutional behind cleanrl.RecurrentPolicy (ppo_cnn_lstm.npz; environments/atari/torch.py:4-6): identical actions
    given the recorded multinomial noise, log-probabilities / values / final LSTM state / losses / updated weights."""
    import cnn_golden
    torch.set_num_threads(1)
    g = _load(golden_dir, 'ppo_cnn_lstm.npz')
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = (float(x) for x in g['hparams'])
    w0 = cnn_golden.recurrent_start_weights(cnn_golden.container())
    for k, v in w0.items():
        assert np.array_equal(cnn_golden.digest(v), g['w0.' + cnn_golden.golden_key(k)]), k
    pol = ppo_torch.RecurrentConvPolicy(w0)
    tr = ppo_torch.Trainer(pol, cnn_golden.ReplayVec(g), batch_size=n * horizon, minibatch_size=mbs, bptt_horizon=bptt, update_epochs=epochs,
                           learning_rate=lr, gamma=gamma, gae_lambda=lam, clip_coef=clip, vf_coef=vf_coef, vf_clip_coef=vf_clip,
                           max_grad_norm=mgn, ent_coef=ent, total_timesteps=total, seed=1)
    tr.evaluate(g['it0.noise'])
    assert np.array_equal(tr.actions, g['it0.actions'].astype(np.int64)), 'actions differ'
    np.testing.assert_allclose(tr.logprobs, g['it0.logprobs'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tr.values, g['it0.values'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tr.lstm_h.numpy(), g['it0.lstm_h'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tr.lstm_c.numpy(), g['it0.lstm_c'], rtol=1e-5, atol=1e-6)
    L = tr.train()
    np.testing.assert_allclose(tr.b_advantages.numpy(), g['it0.advantages'], rtol=1e-5, atol=1e-6)
    got = [L['policy_loss'], L['value_loss'], L['entropy'], L['old_approx_kl'], L['approx_kl'], L['clipfrac'], L['explained_variance']]
    np.testing.assert_allclose(got, g['it0.losses'], rtol=2e-5, atol=1e-7)
    for name, arr in pol.state_arrays().items():
        np.testing.assert_allclose(cnn_golden.digest(arr), g['it0.w.' + cnn_golden.golden_key(name)], rtol=1e-5, atol=1e-6, err_msg=name)


def test_stochastic_oracle_replays_reference_trajectory(golden_dir):
    """ocean.Stochastic under Serial + GymnasiumPufferEnv + EpisodeStats (tests/golden/stochastic.npz from the unmodified
    reference): observations, f32 rewards, terminals, auto-reset rows and the episode infos, bit for bit."""
    from oracle import c_oracle
    g = np.load(os.path.join(golden_dir, 'stochastic.npz'))
    n, seed, steps = (int(x) for x in g['config'])
    vec = c_oracle.StochasticSerial(n, float(g['p'][0]), 100)
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o, g['obs'][k]) and np.array_equal(r, g['rewards'][k]), k
        assert np.array_equal(te, g['terminals'][k]) and np.array_equal(tr, g['truncations'][k]) and m.all(), k
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])      # f64 sums in the same order
    assert len(infos) == 12


def test_stochastic_reward_formula_every_state():
    """The C restatement (libm pow, like CPython's float ** 2) vs the python expression of ocean.py:571-576 for every
    reachable (tick, count, action) of a 100-step episode and a few p."""
    from oracle import c_oracle
    for p in (0.7, 0.75, 0.5, 1 / 3):
        for tick in range(1, 101):
            for count in range(0, tick + 1):
                frac = count / tick
                prox = 1 - (p - frac) ** 2
                for a in (0, 1):
                    if (a == 0 and count == 0) or (a == 1 and count == tick):
                        continue                                   # unreachable: the last action is counted
                    want = prox if ((a == 0 and frac < p) or (a == 1 and frac >= p)) else 0
                    got, gp = c_oracle.stochastic_reward(p, tick, count, a)
                    assert got == want and gp == prox, (p, tick, count, a)


@pytest.mark.parametrize('tag', ['l2d2', 'l3d1'])
def test_memory_oracle_replays_reference_trajectory(golden_dir, tag):
    """ocean.Memory under Serial (tests/golden/memory_<tag>.npz from the unmodified reference): the solutions drawn from
    numpy's global legacy stream (per-env seeding at async_reset, shared stream afterwards, MT19937 block crossings),
    observations, rewards, terminals, auto-reset rows and episode infos, bit for bit."""
    from oracle import c_oracle
    g = np.load(os.path.join(golden_dir, f'memory_{tag}.npz'))
    n, L, D, seed, steps = (int(x) for x in g['config'])
    vec = c_oracle.MemorySerial(n, L, D)
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o, g['obs'][k]) and np.array_equal(r, g['rewards'][k]), k
        assert np.array_equal(te, g['terminals'][k]) and not tr.any() and m.all(), k
        assert np.array_equal(np.stack([vec.solution(e) for e in range(n)]).astype(np.int8), g['solutions'][k]), k
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])


def test_numpy_legacy_stream_restatement_matches_numpy():
    """po_mt_seed_numpy + genrand == np.random.RandomState(seed).randint(0, 2, size) (masked rejection, one word per draw)."""
    from oracle import c_oracle
    import ctypes as C
    L = c_oracle.lib()
    for seed in (0, 1, 42, 4219, 2 ** 32 - 1):
        st = (C.c_uint8 * 4096)()          # po_mt_t is ~2.5 KB
        L.po_mt_seed_numpy(st, seed)
        want = np.random.RandomState(seed).randint(0, 2, size=2000)
        got = np.array([L.po_mt_u32(st) & 1 for _ in range(2000)])
        assert np.array_equal(got, want), seed


def test_bandit_oracle_replays_reference_trajectory(golden_dir):
    """ocean.Bandit under Serial (tests/golden/bandit.npz): reseed-to-42 at every reset, solution from numpy's legacy randint,
    reward noise from its legacy gauss (cached second value across envs) — rewards (f32) and infos (f64) bit for bit."""
    from oracle import c_oracle
    g = np.load(os.path.join(golden_dir, 'bandit.npz'))
    n, na, seed, steps = (int(x) for x in g['config'])
    vec = c_oracle.BanditSerial(n, na, *g['scale_noise'])
    vec.async_reset(seed)
    assert vec.solution == int(g['solution'][0])
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o, g['obs'][k]) and np.array_equal(r, g['rewards'][k]) and np.array_equal(te, g['terminals'][k]), k
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])


def test_numpy_legacy_gauss_and_randint_restatement_matches_numpy():
    from oracle import c_oracle
    import ctypes as C
    L = c_oracle.lib()
    for seed in (0, 42, 99991):
        st = (C.c_uint8 * 4096)()
        L.po_np_seed(st, seed)
        rs = np.random.RandomState(seed)
        for n in (10, 2, 7, 1000, 2 ** 31):
            assert L.po_np_randint(st, n) == rs.randint(0, n), (seed, n)
        want = [rs.randn() for _ in range(1001)]                     # odd count: leaves a cached value behind
        got = [L.po_np_randn(st) for _ in range(1001)]
        assert got == want, seed
        L.po_np_seed(st, seed + 1)                                    # reseeding must drop the cached value
        rs.seed(seed + 1)
        assert L.po_np_randn(st) == rs.randn()


def test_multiagent_oracle_replays_reference_trajectory(golden_dir):
    """ocean.Multiagent under Serial + PettingZooPufferEnv (tests/golden/multiagent.npz): env-major agent rows, per-slot scoring
    rule, terminal-every-step / reset-row alternation and the per-env info dicts."""
    from oracle import c_oracle
    g = np.load(os.path.join(golden_dir, 'multiagent.npz'))
    n, seed, steps = (int(x) for x in g['config'])
    vec = c_oracle.MultiagentSerial(n)
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o, g['obs'][k]) and np.array_equal(r, g['rewards'][k]) and np.array_equal(te, g['terminals'][k]), k
        assert m.all() and not tr.any()
        infos += [(k, j, i[1]['score'], i[2]['score']) for j, i in enumerate(info)]
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.int64).reshape(-1, 4), g['infos'])


def spaces_noise(g, it, n, horizon):
    """The recording holds one [N, 2] slab per torch.multinomial call, two calls (heads) per step: -> [T, N, 4]."""
    q = g[f'it{it}.noise']
    return q.reshape(horizon, 2, n, 2).transpose(0, 2, 1, 3).reshape(horizon, n, 4)


def test_multidiscrete_ppo_replay_matches_reference(golden_dir):
    """models.Default's per-head decoders + sample_logits' list branch + [batch, heads] actions through create/evaluate/train
    (tests/golden/ppo_spaces.npz: the unmodified reference on ocean Spaces, Dict obs -> 108-byte rows, Dict action ->
    MultiDiscrete([2, 2])).  Observations are played back (host_vecenv.SpacesReplay), everything else is recomputed."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from host_vecenv import SpacesReplay
    torch.set_num_threads(1)
    g = _load(golden_dir, 'ppo_spaces.npz')
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = (float(x) for x in g['hparams'])
    pol = ppo_torch.Policy.from_reference_state_dict({k[3:]: g[k] for k in g.files if k.startswith('w0.')})
    assert pol.heads == [2, 2]
    rounds = np.concatenate([g[f'it{it}.obs'].reshape(horizon, n, 108) for it in range(iters)])
    rounds = np.concatenate([rounds, rounds[-1:]])                    # the recv after the last send is never looked at
    tr = ppo_torch.Trainer(pol, SpacesReplay(rounds), batch_size=n * horizon, minibatch_size=mbs, bptt_horizon=bptt,
                           update_epochs=epochs, learning_rate=lr, gamma=gamma, gae_lambda=lam, clip_coef=clip, vf_coef=vf_coef,
                           vf_clip_coef=vf_clip, max_grad_norm=mgn, ent_coef=ent, total_timesteps=total, seed=1)
    for it in range(iters):
        stats = tr.evaluate(spaces_noise(g, it, n, horizon))
        assert np.array_equal(tr.actions, g[f'it{it}.actions'].astype(np.int64)), 'actions differ'
        assert np.array_equal(tr.rewards, g[f'it{it}.rewards']) and np.array_equal(tr.dones, g[f'it{it}.dones'])
        np.testing.assert_allclose(tr.logprobs, g[f'it{it}.logprobs'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tr.values, g[f'it{it}.values'], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-12)
        L = tr.train()
        got = [L['policy_loss'], L['value_loss'], L['entropy'], L['old_approx_kl'], L['approx_kl'], L['clipfrac'], L['explained_variance']]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=2e-5, atol=1e-7)
        m, v = tr.adam_moments()
        for name, arr in pol.state_arrays().items():
            np.testing.assert_allclose(arr, g[f'it{it}.w.policy.' + name], rtol=1e-5, atol=1e-6, err_msg=name)
            np.testing.assert_allclose(m[name], g[f'it{it}.m.policy.' + name], rtol=1e-4, atol=1e-7, err_msg=name)


def test_numpy_int8_randint_and_f32_sum_restatements_match_numpy():
    """RandomState.randint(low, high, size, dtype=int8) (bytes of buffered words, fresh buffer per call) interleaved with randn, and
    np.sum over contiguous float32 arrays (pairwise blocks) — the two pieces of numpy arithmetic ocean.Spaces leans on."""
    import ctypes as C
    from oracle import c_oracle
    L = c_oracle.lib()
    for seed in (0, 1, 12345):
        st = (C.c_uint8 * 4096)()
        L.po_np_seed(st, seed)
        rs = np.random.RandomState(seed)
        for low, high, cnt in ((-1, 2, 5), (0, 2, 7), (-3, 4, 33), (0, 1, 4), (-128, 128, 9), (5, 6, 3), (-1, 2, 1)):
            got = np.zeros(cnt, np.int8)
            L.po_np_randint_i8(st, low, high, cnt, got.ctypes.data)
            assert np.array_equal(got, rs.randint(low, high, (cnt,), dtype=np.int8)), (seed, low, high)
            assert L.po_np_randn(st) == rs.randn()
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 24, 25, 31, 64, 127):
        for _ in range(50):
            a = rng.standard_normal(n).astype(np.float32)
            assert L.po_np_sum_f32(a.ctypes.data, n) == np.sum(a.reshape(-1)), n
    for _ in range(200):
        a = rng.standard_normal((5, 5)).astype(np.float32)
        assert L.po_np_sum_f32(a.ctypes.data, 25) == np.sum(a)


def test_spaces_oracle_replays_reference_observation_stream_and_rewards(golden_dir):
    """ocean.Spaces under Serial as the reference's create/evaluate drove it (tests/golden/ppo_spaces.npz: np.random.seed(1) by
    seed_everything, envs never reseed): the emulated 108-byte rows of 2 x 32 steps x 16 envs, rewards and terminals under the
    recorded actions, bit for bit."""
    from oracle import c_oracle
    g = _load(golden_dir, 'ppo_spaces.npz')
    n, horizon, _, _, _, _, iters = (int(x) for x in g['config'])
    vec = c_oracle.SpacesSerial(n, global_seed=1)
    vec.async_reset(1)
    for it in range(iters):
        obs = g[f'it{it}.obs'].reshape(horizon, n, 108)
        acts = g[f'it{it}.actions'].reshape(horizon, n, 2).astype(np.int64)
        rew, done = g[f'it{it}.rewards'].reshape(horizon, n), g[f'it{it}.dones'].reshape(horizon, n)
        scores = []                                   # infos of THIS evaluate's T recv() calls (clean_pufferl.py:110-113)
        for t in range(horizon):
            o, r, d, _, info, _, _ = vec.recv()
            assert np.array_equal(o, obs[t]), (it, t)
            assert np.array_equal(r, rew[t]) and np.array_equal(d.astype(np.float32), done[t]), (it, t)
            scores += [i['score'] for i in info]
            vec.send(acts[t])
        assert abs(np.mean(scores) - g[f'it{it}.stats'][2]) < 1e-12, it


def test_parallel_form_of_the_spaces_stream_equals_the_sequential_one():
    """oracle/spaces_stream.py (acceptance marks -> per-position reset lengths -> pointer doubling -> independent row fills) against
    the sequential C restatement over 3 rounds of 40 envs: the algorithm the device tape kernel for ocean.Spaces will implement."""
    from oracle import c_oracle, spaces_stream
    n, rounds = 40, 3
    for seed in (1, 7):
        vec = c_oracle.SpacesSerial(n, global_seed=seed)
        vec.async_reset()
        want = [vec.observations.copy()]
        for _ in range(rounds - 1):
            vec.send(np.zeros((n, 2), np.int64))          # terminal step
            vec.send(np.zeros((n, 2), np.int64))          # reset row: fresh observations
            want.append(vec.observations.copy())
        got = spaces_stream.parallel_rows(seed, n * rounds, window=n * rounds * 90)
        assert np.array_equal(got, np.concatenate(want)), seed


def test_vectorised_philox_equals_the_c_restatement():
    """oracle/c_oracle.philox_exp_noise (numpy) feeds the full-size GPU parity test: pin it word for word against
    po_philox4x32_10 (the restatement of pufferlib_amd/csrc/philox.hpp's stream), including 64-bit seeds / steps and row offsets."""
    from oracle import c_oracle
    for seed, step, off, rows, cols in [(1, 0, 0, 33, 8), (1, 127, 4096, 5, 8), ((9 << 32) | 7, (3 << 32) + 11, 100, 4, 15), (42, 5, 0, 3, 3)]:
        got = c_oracle.philox_exp_noise(seed, step, rows, cols, row_offset=off)
        assert got.shape == (rows, cols) and got.dtype == np.float32
        for e in range(rows):
            for j in range((cols + 3) // 4):
                w = c_oracle.philox4x32_10([e + off, j, step & 0xFFFFFFFF, step >> 32], [seed & 0xFFFFFFFF, seed >> 32])
                u = ((w >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
                n = min(4, cols - 4 * j)
                assert np.array_equal(got[e, 4 * j:4 * j + n], (-np.log(u))[:n]), (seed, step, e, j)


@pytest.mark.parametrize('seed,pre,need,n_pop', [(1, 0, 1000, 24), (2, 7, 5000, 24), (3, 623, 4096 * 33, 24), (4, 100, 3000, 16),
                                                  (5, 0, 777, 40), (6, 5, 624 * 3, 8), (7, 311, 1, 24)])
def test_parallel_form_of_the_squared_tape_equals_cpython_random(seed, pre, need, n_pop):
    """oracle/squared_tape.py (the algorithm of csrc/squared.hip's single-target tape kernels: sliding 227-word MT19937 window,
    parallel accept / prefix / select, generator state handed back) against CPython itself: the draws of
    random.sample(range(n_pop), 1) in order, and random.getstate() afterwards — block AND index, incl. the convention that an
    index of 624 stays on the old block.  Sizes: the headline fill (4096 envs x 33 rounds), power-of-two and non-power-of-two
    populations (acceptance 1/2 .. 3/4 .. 5/8), a start on the last word of a block, a single draw."""
    from oracle import squared_tape
    random.seed(seed)
    for _ in range(pre):
        random.getrandbits(32)
    st = random.getstate()[1]
    want = [random.sample(range(n_pop), 1)[0] for _ in range(need)]
    after = random.getstate()[1]
    draws, block, idx, consumed = squared_tape.fill(st[:624], st[624], need, n_pop)
    assert draws.tolist() == want
    assert idx == after[624] and block.tolist() == list(after[:624])
    assert consumed >= need


def test_squared_tape_raw_words_are_successive_mt19937_blocks():
    from oracle import squared_tape
    random.seed(11)
    st = random.getstate()[1]
    raw = squared_tape.raw_words(np.array(st[:624], dtype=np.uint32), 120)      # crosses the linear window's wrap several times
    for j in range(1, 121):
        for _ in range(624):
            random.getrandbits(32)
        assert raw[624 * j:624 * (j + 1)].tolist() == list(random.getstate()[1][:624]), j


def test_multiprocessing_backend_run_is_consistent_and_oracle_policy_reproduces_it(golden_dir):
    """tests/golden/ppo_mp.npz — the unmodified reference trained over its OWN pufferlib.vector.Multiprocessing backend
    (vector.py:218-447, EnvPool mode: 8 of 16 envs per recv) — pinned two ways: (1) the experience the reference sorted
    (sort_training_data, clean_pufferl.py:452-464) is exactly its recv() batches re-ordered by (env id, arrival), which is the
    env-major layout the device trainer writes directly; (2) the oracle's restatement of models.Default + sample_logits
    reproduces every action the reference sent back (bit-exact) and its log-probabilities / values (1e-5) from the recorded
    observations and multinomial noise."""
    g = np.load(os.path.join(golden_dir, 'ppo_mp.npz'))
    n, horizon, _, _, _, _, iters, per, workers = (int(x) for x in g['config'])
    ids, mask = g['recv.env_id'], g['recv.mask'].astype(bool)
    assert mask.all() and ids.shape[1] == per and n // per == 2
    assert [int(r[0]) for r in ids] == [per * (k % 2) for k in range(len(ids))]          # the two worker blocks answer in turn
    for it in range(iters):
        k0, k1 = (int(x) for x in g[f'it{it}.recvs'])
        env = ids[k0:k1].reshape(-1)
        order = np.lexsort((np.arange(env.size), env))                                   # stable (env id, arrival) order
        assert np.array_equal(np.bincount(env), np.full(n, horizon))
        for key, src in (('obs', 'recv.obs'), ('rewards', 'recv.rewards'), ('dones', 'recv.terminals')):
            flat = g[src][k0:k1].reshape(env.size, -1)
            assert np.array_equal(flat[order].reshape(g[f'it{it}.{key}'].shape).astype(np.float32),
                                  g[f'it{it}.{key}'].astype(np.float32)), key
        assert np.array_equal(g['send.actions'][k0:k1].reshape(-1)[order], g[f'it{it}.actions'])
        # the oracle policy with the weights this iteration started from
        sd = {k[len('w0.'):] if it == 0 else k[len(f'it{it - 1}.w.'):]: g[k] for k in g.files
              if k.startswith('w0.' if it == 0 else f'it{it - 1}.w.')}
        pol = ppo_torch.Policy.from_reference_state_dict(sd)
        with torch.no_grad():
            for k in range(k0, k1):
                logits, value, _ = pol.forward(torch.as_tensor(g['recv.obs'][k].astype(np.float32)))
                action, logprob, _ = ppo_torch.sample_logits(logits, noise=torch.as_tensor(g['recv.noise'][k]))
                assert np.array_equal(action.numpy(), g['send.actions'][k].astype(np.int64)), k
                rows = np.nonzero(np.isin(order, np.arange((k - k0) * per, (k - k0 + 1) * per)))[0]   # where this recv's rows landed
                np.testing.assert_allclose(logprob.numpy(), g[f'it{it}.logprobs'][rows][np.argsort(order[rows])], rtol=1e-5, atol=1e-5)
                np.testing.assert_allclose(value.flatten().numpy(), g[f'it{it}.values'][rows][np.argsort(order[rows])], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('d', [1, 2, 3, 4, 5])
def test_single_target_form_of_the_squared_step_equals_the_general_form(d):
    """oracle/squared_nt1.py: the single-target env step of the fused rollout (csrc/squared_env.hpp squared_step_nt1 /
    squared_reset_nt1: reward table, kept target coordinates, two-cell clear at a reset) against the general restatement of
    ocean.py:448-513, 300 episodes of random actions per grid size: identical grids after every step and reset, rewards equal
    as float32 bit patterns, same dones and scores."""
    from oracle import squared_nt1
    rs = np.random.RandomState(d)
    g = 2 * d + 1
    perimeter = [(x, y) for x in range(g) for y in range(g) if x in (0, g - 1) or y in (0, g - 1)]   # ocean.py:444-446
    a, b = squared_nt1.General(d), squared_nt1.SingleTarget(d)
    # the device env enters the rollout from async_reset's freshly drawn grid: same starting state for both forms
    first = perimeter[rs.randint(len(perimeter))]
    a.reset(first)
    b.grid[:] = a.grid
    b.pos, b.target, b.tick, b.rem = a.pos, first, 0, 1
    for episode in range(300):
        done = False
        while not done:
            act = int(rs.randint(8))
            ra, done, sa = a.step(act)
            rb, db, sb = b.step(act)
            assert ra.tobytes() == np.float32(rb).tobytes() and done == db and sa == sb, (episode, act)
            assert np.array_equal(a.grid, b.grid), (episode, act)
        t = perimeter[rs.randint(len(perimeter))]
        a.reset(t)
        b.reset(t)
        assert np.array_equal(a.grid, b.grid), episode
