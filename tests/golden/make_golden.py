"""Generate the golden fixtures in tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported read-only from /root/reference with the tests-only shims in
tests/shims/ standing in for gymnasium / gym / pettingzoo (not installed, no network).  Nothing
from the reference is copied: this script calls its public functions and records inputs/outputs.

Fixtures written (all small, committed):
  gae.npz            c_gae.compute_gae (c_gae.pyx:11-32) on KAT / random / boundary inputs
  squared_<tag>.npz  pufferlib.vector.Serial over ocean make_squared (vector.py:70-166,
                     ocean.py:406-513, emulation.py:124-234, postprocess.py:8-54): lock-step trajectories
  bandit.npz         pufferlib.vector.Serial over ocean make_bandit (ocean.py:8-63): numpy legacy randint + gauss
  memory_<tag>.npz   pufferlib.vector.Serial over ocean make_memory (ocean.py:65-123): numpy's global legacy stream
  multiagent.npz     pufferlib.vector.Serial over ocean make_multiagent (ocean.py:148-224) under PettingZooPufferEnv
  nativize.npz       pufferlib.pytorch.nativize_dtype / nativize_tensor (pytorch.py:48-145) on emulated Dict/Tuple spaces
  stochastic.npz     pufferlib.vector.Serial over ocean make_stochastic (ocean.py:529-582): deterministic trajectories
  ppo_mlp.npz        clean_pufferl.create/evaluate/train (clean_pufferl.py:30-292) with models.Default
  ppo_lstm.npz       same with models.LSTMWrapper (models.py:64-111)
  ppo_mlp_h256.npz   same as ppo_mlp with models.Default(hidden_size=256) (models.py:24): a width outside the fused kernels (`make_golden.py wide`)
  ppo_mp.npz         same as ppo_mlp over the reference's own pufferlib.vector.Multiprocessing backend (vector.py:218-447) in
                     EnvPool mode (8 of 16 envs per recv): every recv() batch, the noise, the actions sent back, the sorted experience
  ppo_cnn.npz        same with models.Convolutional (models.py:113-157, NatureCNN) on a stub env with uint8 (4, 84, 84) frames;
                     frames are re-derivable from recorded frame numbers, big tensors recorded as digests (sum, |sum|, 64 samples)
  ppo_cnn_lstm.npz   same as ppo_cnn with pufferlib.models.LSTMWrapper(input_size=512, hidden_size=512) on top and frameworks.cleanrl.RecurrentPolicy —
                     the `Recurrent` policy of environments/atari/torch.py:4-6 (`make_golden.py cnn_lstm`)
  ppo_c1_mlp / ppo_c1_lstm / ppo_demo_lstm / ppo_c2_mlp.npz   the same call sequence at BASELINE's own sizes, digest form (`make_golden.py big`,
                     gen_big): configs[0] (64 x 128), the shape `demo.py --env squared` trains (config.yaml:498-509), one iteration of
                     configs[1] (4096 x 128)
  ppo_spaces.npz     same as ppo_mlp on ocean make_spaces: Dict observation emulated to 108-byte rows, Dict action emulated to
                     MultiDiscrete([2, 2]) -> models.Default's per-head decoders and sample_logits' list branch (cleanrl.py:25-47)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(REPO, 'tests', 'shims'), '/root/reference']

import warnings  # noqa: E402
warnings.filterwarnings('ignore')

import numpy as np  # noqa: E402
import torch  # noqa: E402


def gen_gae(compute_gae):
    out = {}
    rng = np.random.RandomState(0)
    cases = {
        'kat8': (np.zeros(8, np.float32), np.arange(8, dtype=np.float32), np.ones(8, np.float32), .99, .95),
        'len1': (np.zeros(1, np.float32), np.ones(1, np.float32), np.ones(1, np.float32), .99, .95),
        'len2': (np.array([0, 1], np.float32), np.array([.5, -1], np.float32), np.array([1, 2], np.float32), .99, .95),
        'alldone': (np.ones(257, np.float32), rng.randn(257).astype(np.float32), rng.randn(257).astype(np.float32), .99, .95),
        'nodone': (np.zeros(1000, np.float32), rng.randn(1000).astype(np.float32), rng.randn(1000).astype(np.float32), .99, .95),
        'p01': ((rng.rand(4097) < .01).astype(np.float32), rng.randn(4097).astype(np.float32), rng.randn(4097).astype(np.float32), .99, .95),
        'p25': ((rng.rand(65537) < .25).astype(np.float32), rng.randn(65537).astype(np.float32), rng.randn(65537).astype(np.float32), .9, .8),
        'gamma1': (np.zeros(300, np.float32), rng.randn(300).astype(np.float32), rng.randn(300).astype(np.float32), 1.0, 1.0),
    }
    for k, (d, v, r, g, l) in cases.items():
        adv = compute_gae(d, v, r, g, l)
        out[k + '_dones'], out[k + '_values'], out[k + '_rewards'] = d, v, r
        out[k + '_gl'] = np.array([g, l], np.float64)
        out[k + '_adv'] = np.asarray(adv)
    np.savez_compressed(os.path.join(HERE, 'gae.npz'), **out)
    print('gae.npz', len(cases), 'cases')


def gen_squared(tag, num_envs, d, nt, seed, steps):
    import pufferlib.vector
    import pufferlib.environments.ocean as ocean
    vec = pufferlib.vector.make(ocean.env_creator('squared'), env_kwargs=dict(distance_to_target=d, num_targets=nt),
                                num_envs=num_envs, backend=pufferlib.vector.Serial)
    rng = np.random.RandomState(1000 + seed)
    vec.async_reset(seed)
    obs, rew, term, trunc, acts = [], [], [], [], []
    info_rows = []  # (recv index, ordinal in list, episode_return, episode_length, score)
    targets = []    # remaining target cells per env after every recv, -1 padded
    g = 2 * d + 1
    nt_eff = 4 * d if nt == -1 else nt

    def snap(k):
        o, r, te, tr, infos, ids, masks = vec.recv()
        assert masks.all() and (ids == np.arange(num_envs)).all()
        obs.append(o.copy()); rew.append(r.copy()); term.append(te.copy()); trunc.append(tr.copy())
        for j, i in enumerate(infos):
            info_rows.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        tg = -np.ones((num_envs, nt_eff), np.int16)
        for e, env in enumerate(vec.envs):
            for j, (x, y) in enumerate(env.env.env.targets):
                tg[e, j] = x * g + y
        targets.append(tg)

    snap(0)
    for k in range(1, steps + 1):
        a = rng.randint(0, 8, size=num_envs)
        acts.append(a)
        vec.send(a)
        snap(k)
    obs = np.stack(obs)
    assert set(np.unique(obs)) <= {-1.0, 0.0, 1.0}
    np.savez_compressed(
        os.path.join(HERE, f'squared_{tag}.npz'),
        config=np.array([num_envs, d, nt, seed, steps], np.int64),
        obs=obs.astype(np.int8), rewards=np.stack(rew), terminals=np.stack(term), truncations=np.stack(trunc),
        actions=np.stack(acts).astype(np.int8), infos=np.array(info_rows, np.float64).reshape(-1, 5),
        targets=np.stack(targets))
    print(f'squared_{tag}.npz', obs.shape, 'infos', len(info_rows))


def gen_stochastic(num_envs=6, p=0.7, seed=3, steps=230):
    """pufferlib.vector.Serial over ocean make_stochastic (ocean.py:529-582; horizon is fixed to 100 by
    ocean/environment.py:61-64): deterministic env, no RNG — rewards are python-float arithmetic cast to f32."""
    import pufferlib.vector
    import pufferlib.environments.ocean as ocean
    vec = pufferlib.vector.make(ocean.env_creator('stochastic'), env_kwargs=dict(p=p), num_envs=num_envs,
                                backend=pufferlib.vector.Serial)
    rng = np.random.RandomState(77)
    vec.async_reset(seed)
    obs, rew, term, trunc, acts, info_rows = [], [], [], [], [], []

    def snap(k):
        o, r, te, tr, infos, ids, masks = vec.recv()
        assert masks.all() and (ids == np.arange(num_envs)).all()
        obs.append(o.copy()); rew.append(r.copy()); term.append(te.copy()); trunc.append(tr.copy())
        for j, i in enumerate(infos):
            info_rows.append((k, j, i['episode_return'], i['episode_length'], i['score']))

    snap(0)
    bias = rng.rand(num_envs)            # every env plays action 0 with its own probability
    for k in range(1, steps + 1):
        a = (rng.rand(num_envs) >= bias).astype(np.int64)
        acts.append(a)
        vec.send(a)
        snap(k)
    np.savez_compressed(os.path.join(HERE, 'stochastic.npz'), config=np.array([num_envs, seed, steps], np.int64),
                        p=np.array([p]), obs=np.stack(obs), rewards=np.stack(rew), terminals=np.stack(term),
                        truncations=np.stack(trunc), actions=np.stack(acts).astype(np.int8),
                        infos=np.array(info_rows, np.float64).reshape(-1, 5))
    print('stochastic.npz', np.stack(obs).shape, 'infos', len(info_rows))


def gen_memory(tag, num_envs, mem_length, mem_delay, seed, steps):
    """pufferlib.vector.Serial over ocean make_memory (ocean.py:65-123): every reset draws np.random.randint(0, 2, horizon) from
    numpy's process-global legacy stream (seeded per env at async_reset, shared afterwards)."""
    import pufferlib.vector
    import pufferlib.environments.ocean as ocean
    vec = pufferlib.vector.make(ocean.env_creator('memory'), env_kwargs=dict(mem_length=mem_length, mem_delay=mem_delay),
                                num_envs=num_envs, backend=pufferlib.vector.Serial)
    rng = np.random.RandomState(500 + seed)       # a private generator: must not touch the global stream under test
    vec.async_reset(seed)
    obs, rew, term, trunc, acts, info_rows, sols = [], [], [], [], [], [], []

    def snap(k):
        o, r, te, tr, infos, ids, masks = vec.recv()
        assert masks.all() and (ids == np.arange(num_envs)).all()
        obs.append(o.copy()); rew.append(r.copy()); term.append(te.copy()); trunc.append(tr.copy())
        for j, i in enumerate(infos):
            info_rows.append((k, j, i['episode_return'], i['episode_length'], float(i['score'])))
        sols.append(np.stack([env.env.env.solution.copy() for env in vec.envs]))

    snap(0)
    for k in range(1, steps + 1):
        # half of the envs echo the remembered digits at the right time, so scores of both kinds appear
        a = rng.randint(0, 2, size=num_envs)
        for e, env in enumerate(vec.envs[::2]):
            m = env.env.env
            if not env.done and m.tick >= m.mem_length + m.mem_delay:
                a[2 * e] = int(m.solution[m.tick - m.mem_length - m.mem_delay])
        acts.append(a)
        vec.send(a)
        snap(k)
    np.savez_compressed(os.path.join(HERE, f'memory_{tag}.npz'),
                        config=np.array([num_envs, mem_length, mem_delay, seed, steps], np.int64), obs=np.stack(obs),
                        rewards=np.stack(rew), terminals=np.stack(term), truncations=np.stack(trunc),
                        actions=np.stack(acts).astype(np.int8), infos=np.array(info_rows, np.float64).reshape(-1, 5),
                        solutions=np.stack(sols).astype(np.int8))
    print(f'memory_{tag}.npz', np.stack(obs).shape, 'infos', len(info_rows))


def gen_bandit(num_envs=37, num_actions=10, reward_scale=1, reward_noise=1, seed=5, steps=9):
    """pufferlib.vector.Serial over ocean make_bandit (ocean.py:8-63): every reset reseeds numpy's global generator with the
    hard fixed seed 42, draws the solution with randint, and each step adds np.random.randn() * reward_scale."""
    import pufferlib.vector
    import pufferlib.environments.ocean as ocean
    vec = pufferlib.vector.make(ocean.env_creator('bandit'), env_kwargs=dict(num_actions=num_actions, reward_scale=reward_scale,
                                                                              reward_noise=reward_noise),
                                num_envs=num_envs, backend=pufferlib.vector.Serial)
    rng = np.random.RandomState(900)
    vec.async_reset(seed)
    obs, rew, term, acts, info_rows = [], [], [], [], []

    def snap(k):
        o, r, te, tr, infos, ids, masks = vec.recv()
        assert masks.all() and not tr.any()
        obs.append(o.copy()); rew.append(r.copy()); term.append(te.copy())
        for j, i in enumerate(infos):
            info_rows.append((k, j, i['episode_return'], i['episode_length'], float(i['score'])))

    snap(0)
    sol = vec.envs[0].env.env.solution_idx
    for k in range(1, steps + 1):
        a = rng.randint(0, num_actions, size=num_envs)
        a[::3] = sol
        acts.append(a)
        vec.send(a)
        snap(k)
    np.savez_compressed(os.path.join(HERE, 'bandit.npz'), config=np.array([num_envs, num_actions, seed, steps], np.int64),
                        scale_noise=np.array([reward_scale, reward_noise], np.float64), solution=np.array([sol]),
                        obs=np.stack(obs), rewards=np.stack(rew), terminals=np.stack(term), actions=np.stack(acts).astype(np.int8),
                        infos=np.array(info_rows, np.float64).reshape(-1, 5))
    print('bandit.npz', np.stack(obs).shape, 'infos', len(info_rows), 'solution', sol)


def gen_multiagent(num_envs=19, seed=3, steps=8):
    """pufferlib.vector.Serial over ocean make_multiagent (ocean.py:148-224): two agent rows per env through
    PettingZooPufferEnv (emulation.py:236-420); infos are the env's own {agent: {'score': reward}} per env."""
    import pufferlib.vector
    import pufferlib.environments.ocean as ocean
    vec = pufferlib.vector.make(ocean.env_creator('multiagent'), num_envs=num_envs, backend=pufferlib.vector.Serial)
    assert vec.num_agents == 2 * num_envs
    rng = np.random.RandomState(901)
    vec.async_reset(seed)
    obs, rew, term, acts, info_rows = [], [], [], [], []

    def snap(k):
        o, r, te, tr, infos, ids, masks = vec.recv()
        assert masks.all() and not tr.any() and np.array_equal(ids, np.arange(2 * num_envs))
        obs.append(o.copy()); rew.append(r.copy()); term.append(te.copy())
        for j, i in enumerate(infos):
            assert sorted(i) == [1, 2] and all(list(v) == ['score'] for v in i.values())
            info_rows.append((k, j, i[1]['score'], i[2]['score']))

    snap(0)
    for k in range(1, steps + 1):
        a = rng.randint(0, 2, size=2 * num_envs)
        acts.append(a)
        vec.send(a)
        snap(k)
    np.savez_compressed(os.path.join(HERE, 'multiagent.npz'), config=np.array([num_envs, seed, steps], np.int64),
                        obs=np.stack(obs), rewards=np.stack(rew), terminals=np.stack(term), actions=np.stack(acts).astype(np.int8),
                        infos=np.array(info_rows, np.int64).reshape(-1, 4))
    print('multiagent.npz', np.stack(obs).shape, 'infos', len(info_rows))


def nativize_cases():
    """name -> observation space (tests/shims gymnasium).  Shared with the tests through the fixture only."""
    import gymnasium.spaces as S
    box = lambda shape, dt: S.Box(low=0, high=1, shape=shape, dtype=dt)   # noqa: E731
    return {
        'spaces_env': S.Dict({'image': box((5, 5), np.float32), 'flat': box((5,), np.int8)}),               # ocean.Spaces
        'all_f32': S.Dict({'a': box((3,), np.float32), 'b': box((2, 2), np.float32)}),                      # non-byte sample rows
        'mixed': S.Dict({'u8': box((3,), np.uint8), 'f64': box((2,), np.float64), 'i16': box((5,), np.int16),
                         'f32': box((4, 3), np.float32), 'i64': box((1,), np.int64), 'f16': box((7,), np.float16)}),
        'nested': S.Dict({'x': box((3,), np.uint8), 'inner': S.Dict({'p': box((2,), np.int32), 'q': box((3,), np.uint8)}),
                          'y': S.Tuple([box((2,), np.uint16), box((1,), np.float32)])}),
        'tuple': S.Tuple([box((4,), np.int8), box((2, 3), np.int32), S.Discrete(5)]),
        'wide': S.Dict({'map': box((40, 40), np.uint8), 'vec': box((700,), np.float32), 'id': box((3,), np.int32)}),
    }


def dtype_spec(dt):
    """Nested literal of an aligned structured dtype: [(name, spec), ...] for structs, (base, shape) for leaves."""
    if dt.fields is not None:
        return [(name, dtype_spec(sub)) for name, (sub, _) in dt.fields.items()]
    base, shape = dt.subdtype if dt.subdtype is not None else (dt, ())
    return (str(base), tuple(shape))


def dtype_from_spec(spec):
    """Inverse of dtype_spec, built the way the reference builds it (emulation.py:68-80: align=True at every level)."""
    if isinstance(spec, list):
        return np.dtype([(name, dtype_from_spec(sub)) for name, sub in spec], align=True)
    return np.dtype((spec[0], spec[1]), align=True)


def gen_nativize(rows_per_case=21):
    """pufferlib.pytorch.nativize_dtype + nativize_tensor (pytorch.py:48-145) on the dtypes the reference's own emulation builds
    (emulation.emulate_observation_space, emulation.py:96-110) for several observation spaces; random row bytes."""
    import pufferlib
    import pufferlib.emulation
    import pufferlib.pytorch
    out = {}
    rng = np.random.RandomState(77)
    for name, space in nativize_cases().items():
        emulated_space, structured = pufferlib.emulation.emulate_observation_space(space)
        emulated = pufferlib.namespace(observation_dtype=emulated_space.dtype, emulated_observation_dtype=structured)
        native = pufferlib.pytorch.nativize_dtype(emulated)
        sample = np.dtype(emulated_space.dtype)
        D = int(emulated_space.shape[0])
        if sample.kind == 'f':
            rows = rng.randn(rows_per_case, D).astype(sample)
        else:
            rows = rng.randint(0, 256, size=(rows_per_case, D * sample.itemsize)).astype(np.uint8).view(sample)
            if sample.itemsize == 1:     # keep f16/f32/f64 leaves free of NaN payloads (NaN != NaN in comparisons)
                rows = rows.copy()
        leaves = pufferlib.pytorch.nativize_tensor(torch.from_numpy(rows), native)
        table = []

        def walk(nd, lv, path):
            if isinstance(nd, tuple):
                dt, shape, off, delta = nd
                table.append(('/'.join(path), str(dt).replace('torch.', ''), list(shape), int(off), int(delta)))
                out[f'{name}:leaf:' + '/'.join(path)] = lv.contiguous().view(torch.uint8).numpy().copy()   # bytes: NaN-safe
            else:
                for k in nd:
                    walk(nd[k], lv[k], path + (str(k),))
        walk(native, leaves, ())
        out[f'{name}:rows'] = rows.view(np.uint8).reshape(rows_per_case, -1)
        out[f'{name}:sample'] = np.array(str(sample))
        spec = dtype_spec(structured)
        assert dtype_from_spec(spec) == structured and dtype_from_spec(spec).itemsize == structured.itemsize
        out[f'{name}:descr'] = np.array(repr(spec))
        out[f'{name}:table'] = np.array(repr(table))
        print('nativize', name, 'row bytes', rows.view(np.uint8).reshape(rows_per_case, -1).shape[1], 'leaves', len(table))
    out['cases'] = np.array(list(nativize_cases()))
    np.savez_compressed(os.path.join(HERE, 'nativize.npz'), **out)


def gen_ppo(tag, use_rnn, num_envs=16, horizon=32, iters=2, env='squared', hidden=128):
    import pufferlib
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import pufferlib.environments.ocean as ocean
    import clean_pufferl

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None

    batch = num_envs * horizon
    config = pufferlib.namespace(
        env=env, seed=1, torch_deterministic=True, cpu_offload=False, device='cpu',
        total_timesteps=batch * 8, learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95,
        update_epochs=2, norm_adv=True, clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1,
        max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=batch, minibatch_size=batch // 4,
        bptt_horizon=8, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
        data_dir='/tmp/golden_experiments', exp_id='golden')
    vec = pufferlib.vector.make(ocean.env_creator(env), num_envs=num_envs, backend=pufferlib.vector.Serial)

    torch.manual_seed(1)
    policy = pufferlib.models.Default(vec.driver_env, hidden_size=hidden)
    if use_rnn:
        policy = pufferlib.models.LSTMWrapper(vec.driver_env, policy, input_size=hidden, hidden_size=hidden)
        policy = pufferlib.frameworks.cleanrl.RecurrentPolicy(policy)
    else:
        policy = pufferlib.frameworks.cleanrl.Policy(policy)

    out = {}
    for k, v in policy.state_dict().items():
        out['w0.' + k] = v.detach().numpy().copy()

    # record the exponential noise behind every torch.multinomial call (rollout sampling)
    noise = []
    orig_multinomial = torch.multinomial

    def recording_multinomial(p, n, *a, **kw):
        st = torch.get_rng_state()
        res = orig_multinomial(p, n, *a, **kw)
        st2 = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty_like(p).exponential_(1)
        assert torch.equal((p / q).argmax(-1, keepdim=True), res), 'multinomial != argmax(p/q)'
        torch.set_rng_state(st2)
        noise.append(q.numpy().copy())
        return res

    torch.multinomial = recording_multinomial
    try:
        data = clean_pufferl.create(config, vec, policy)
        exp = data.experience
        for it in range(iters):
            noise.clear()
            clean_pufferl.evaluate(data)
            out[f'it{it}.noise'] = np.stack(noise)                       # (T, N, A)
            # storage (step-major) order; Squared cells are -1/0/1, Spaces rows are the emulated bytes
            out[f'it{it}.obs'] = exp.obs.numpy().reshape(batch, -1).astype(np.int8 if env == 'squared' else np.uint8)
            out[f'it{it}.actions'] = exp.actions_np.copy().astype(np.int8)      # [batch] or [batch, heads] (MultiDiscrete)
            out[f'it{it}.logprobs'] = exp.logprobs_np.copy()
            out[f'it{it}.rewards'] = exp.rewards_np.copy()
            out[f'it{it}.dones'] = exp.dones_np.copy()
            out[f'it{it}.values'] = exp.values_np.copy()
            out[f'it{it}.global_step'] = np.array(data.global_step, np.int64)
            out[f'it{it}.stats'] = np.array([data.stats.get('episode_return', np.nan),
                                             data.stats.get('episode_length', np.nan),
                                             data.stats.get('score', np.nan)], np.float64)
            if use_rnn:
                out[f'it{it}.lstm_h'] = exp.lstm_h.numpy().copy()
                out[f'it{it}.lstm_c'] = exp.lstm_c.numpy().copy()
            lr_used = data.optimizer.param_groups[0]['lr']
            clean_pufferl.train(data)
            out[f'it{it}.lr_used'] = np.array(lr_used, np.float64)
            out[f'it{it}.lr_next'] = np.array(data.optimizer.param_groups[0]['lr'], np.float64)
            out[f'it{it}.advantages'] = exp.b_advantages.numpy().copy()   # (nmb, minibatch)
            out[f'it{it}.returns'] = exp.b_returns.numpy().copy()
            out[f'it{it}.b_idxs'] = exp.b_idxs.numpy().copy()
            L = data.losses
            out[f'it{it}.losses'] = np.array([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl,
                                              L.approx_kl, L.clipfrac, L.explained_variance], np.float64)
            for k, v in policy.state_dict().items():
                out[f'it{it}.w.' + k] = v.detach().numpy().copy()
            st = data.optimizer.state_dict()['state']
            names = [k for k, _ in policy.named_parameters()]
            for i, nme in enumerate(names):
                out[f'it{it}.m.' + nme] = st[i]['exp_avg'].numpy().copy()
                out[f'it{it}.v.' + nme] = st[i]['exp_avg_sq'].numpy().copy()
    finally:
        torch.multinomial = orig_multinomial
    out['config'] = np.array([num_envs, horizon, config.minibatch_size, config.bptt_horizon,
                              config.update_epochs, config.total_timesteps, iters], np.int64)
    out['hparams'] = np.array([config.learning_rate, config.gamma, config.gae_lambda, config.clip_coef,
                               config.vf_coef, config.vf_clip_coef, config.max_grad_norm, config.ent_coef],
                              np.float64)
    np.savez_compressed(os.path.join(HERE, f'ppo_{tag}.npz'), **out)
    print(f'ppo_{tag}.npz', len(out), 'arrays; losses it0', out['it0.losses'])


def gen_ppo_mp(num_envs=16, num_workers=4, batch_envs=8, horizon=32, iters=2, attempts=20):
    """clean_pufferl.create/evaluate/train over the reference's OWN pufferlib.vector.Multiprocessing backend (vector.py:218-447:
    worker processes, shared-memory buffers, `batch_envs` of the `num_envs` envs per recv = EnvPool mode) on ocean squared.
    Recorded per recv(): what the backend handed out (observations, rewards, terminals, env ids, masks, infos), the multinomial
    noise and the actions sent back; per iteration: the experience in the order sort_training_data leaves it, advantages, losses,
    weights.  Which workers answer a recv() depends on timing; the run is repeated until every env contributes exactly `horizon`
    rows to every batch (the schedule the device trainer's env-major experience assumes, hostpath.py) — the CONTENT does not
    depend on timing (each env's trajectory follows from its own recorded actions)."""
    import json
    import pufferlib
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import pufferlib.environments.ocean as ocean
    import clean_pufferl

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None
    batch = num_envs * horizon
    for attempt in range(attempts):
        config = pufferlib.namespace(
            env='squared', seed=1, torch_deterministic=True, cpu_offload=False, device='cpu',
            total_timesteps=batch * 8, learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95,
            update_epochs=2, norm_adv=True, clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1,
            max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=batch, minibatch_size=batch // 4,
            bptt_horizon=8, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
            data_dir='/tmp/golden_experiments', exp_id='golden')
        vec = pufferlib.vector.make(ocean.env_creator('squared'), num_envs=num_envs, num_workers=num_workers,
                                    batch_size=batch_envs, backend=pufferlib.vector.Multiprocessing)
        torch.manual_seed(1)
        policy = pufferlib.frameworks.cleanrl.Policy(pufferlib.models.Default(vec.driver_env, hidden_size=128))
        out = {}
        for k, v in policy.state_dict().items():
            out['w0.' + k] = v.detach().numpy().copy()
        noise, recvs, sends = [], [], []
        orig_multinomial, orig_recv, orig_send = torch.multinomial, vec.recv, vec.send

        def recording_multinomial(p, n, *a, **kw):
            st = torch.get_rng_state()
            res = orig_multinomial(p, n, *a, **kw)
            st2 = torch.get_rng_state()
            torch.set_rng_state(st)
            q = torch.empty_like(p).exponential_(1)
            assert torch.equal((p / q).argmax(-1, keepdim=True), res), 'multinomial != argmax(p/q)'
            torch.set_rng_state(st2)
            noise.append(q.numpy().copy())
            return res

        def recording_recv():
            o, r, d, t, info, env_id, mask = orig_recv()
            recvs.append((np.array(o).copy(), np.array(r).copy(), np.array(d).copy(), np.array(t).copy(),
                          [dict(pufferlib.utils.unroll_nested_dict(i)) for i in info], np.array(env_id).copy(), np.array(mask).copy()))
            return o, r, d, t, info, env_id, mask

        def recording_send(actions):
            sends.append(np.array(actions).copy())
            return orig_send(actions)

        torch.multinomial, vec.recv, vec.send = recording_multinomial, recording_recv, recording_send
        even = True
        try:
            data = clean_pufferl.create(config, vec, policy)
            exp = data.experience
            first = 0
            for it in range(iters):
                clean_pufferl.evaluate(data)
                k1 = len(recvs)
                ids = np.concatenate([r[5][r[6].astype(bool)] for r in recvs[first:k1]])[:batch]
                even = even and np.array_equal(np.bincount(ids, minlength=num_envs), np.full(num_envs, horizon))
                even = even and [int(r[5][0]) for r in recvs[:k1]] == [batch_envs * (j % (num_envs // batch_envs)) for j in range(k1)]   # canonical turn order: fixtures must regenerate bit for bit
                out[f'it{it}.recvs'] = np.array([first, k1], np.int64)
                first = k1
                # the experience as sort_training_data orders it (train() sorts; same stable (env, step) order here)
                order = np.asarray(sorted(range(len(exp.sort_keys)), key=exp.sort_keys.__getitem__))
                out[f'it{it}.obs'] = exp.obs.numpy().reshape(batch, -1)[order].astype(np.int8)
                out[f'it{it}.actions'] = exp.actions_np[order].astype(np.int8)
                out[f'it{it}.logprobs'] = exp.logprobs_np[order].copy()
                out[f'it{it}.rewards'] = exp.rewards_np[order].copy()
                out[f'it{it}.dones'] = exp.dones_np[order].copy()
                out[f'it{it}.values'] = exp.values_np[order].copy()
                out[f'it{it}.global_step'] = np.array(data.global_step, np.int64)
                out[f'it{it}.stats'] = np.array([data.stats.get('episode_return', np.nan), data.stats.get('episode_length', np.nan),
                                                 data.stats.get('score', np.nan)], np.float64)
                lr_used = data.optimizer.param_groups[0]['lr']
                clean_pufferl.train(data)
                out[f'it{it}.lr_used'] = np.array(lr_used, np.float64)
                out[f'it{it}.advantages'] = exp.b_advantages.numpy().copy()
                out[f'it{it}.returns'] = exp.b_returns.numpy().copy()
                L = data.losses
                out[f'it{it}.losses'] = np.array([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                                                 np.float64)      # (explained_variance depends on the arrival order: not recorded)
                for k, v in policy.state_dict().items():
                    out[f'it{it}.w.' + k] = v.detach().numpy().copy()
        finally:
            torch.multinomial = orig_multinomial
            vec.close()
        if not even:
            print(f'ppo_mp attempt {attempt}: uneven arrival schedule, retrying')
            continue
        K = len(recvs)
        assert len(noise) == K == len(sends)
        out['recv.obs'] = np.stack([r[0].reshape(batch_envs, -1) for r in recvs]).astype(np.int8)
        out['recv.rewards'] = np.stack([r[1] for r in recvs]).astype(np.float32)
        out['recv.terminals'] = np.stack([r[2] for r in recvs]).astype(np.uint8)
        out['recv.truncations'] = np.stack([r[3] for r in recvs]).astype(np.uint8)
        out['recv.env_id'] = np.stack([r[5] for r in recvs]).astype(np.int32)
        out['recv.mask'] = np.stack([r[6] for r in recvs]).astype(np.uint8)
        out['recv.infos'] = np.array(json.dumps([r[4] for r in recvs]))
        out['recv.noise'] = np.stack(noise).astype(np.float32)            # [recv][batch_envs][A]
        out['send.actions'] = np.stack(sends).astype(np.int8)
        out['config'] = np.array([num_envs, horizon, config.minibatch_size, config.bptt_horizon, config.update_epochs,
                                  config.total_timesteps, iters, batch_envs, num_workers], np.int64)
        out['hparams'] = np.array([config.learning_rate, config.gamma, config.gae_lambda, config.clip_coef, config.vf_coef,
                                   config.vf_clip_coef, config.max_grad_norm, config.ent_coef], np.float64)
        np.savez_compressed(os.path.join(HERE, 'ppo_mp.npz'), **out)
        print('ppo_mp.npz', len(out), 'arrays;', K, 'recvs; env-id blocks of the first recvs:',
              [int(r[5][0]) for r in recvs[:8]], '; losses it0', out['it0.losses'])
        return
    raise RuntimeError('ppo_mp: no even arrival schedule in %d attempts' % attempts)


def cnn_frame(counter, base_seed=777):
    """Frame number `counter` of the stub Atari-shaped env: uint8 (4, 84, 84) from numpy's legacy generator seeded per frame
    (the GPU test regenerates the same frames from the recorded counters instead of storing 28 KB per observation)."""
    return np.random.RandomState(base_seed + int(counter)).randint(0, 256, (4, 84, 84)).astype(np.uint8)


def cnn_start_weight(name, shape, seed=4242):
    """Start value of parameter `name` of the golden NatureCNN run: N(0, gain^2 / fan_in) weights (gain as layer_init's std:
    sqrt(2), actor 0.01, value_fn 1), N(0, 0.01^2) biases, from numpy's legacy generator keyed by the parameter name."""
    import zlib
    rs = np.random.RandomState(seed + zlib.crc32(name.encode()) % 100000)
    if name.endswith('bias') or name.startswith('bias_'):          # (bias_ih_l0 / bias_hh_l0 of the recurrent variant)
        return (0.01 * rs.standard_normal(shape)).astype(np.float32)
    gain = 0.01 if 'actor' in name else 1.0 if ('value_fn' in name or name.startswith('weight_')) else np.sqrt(2)
    return (gain / np.sqrt(np.prod(shape[1:])) * rs.standard_normal(shape)).astype(np.float32)


def digest(a, samples=64):
    """Compact fingerprint of a big tensor: sum, sum of |.|, and `samples` evenly spaced elements."""
    f = np.asarray(a, np.float64).reshape(-1)
    idx = np.linspace(0, f.size - 1, min(samples, f.size)).astype(np.int64)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])


def gen_ppo_cnn(num_envs=4, horizon=16, iters=1, use_rnn=False):
    """clean_pufferl.create/evaluate/train (clean_pufferl.py:30-292) with pufferlib.models.Convolutional (models.py:113-157, the
    NatureCNN of BASELINE configs[3]) behind frameworks.cleanrl.Policy, on a stub env with Atari-shaped observations."""
    import gymnasium
    import pufferlib
    import pufferlib.emulation
    import pufferlib.postprocess
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import clean_pufferl

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None
    counters = {'next': 0}

    class FrameEnv(gymnasium.Env):
        def __init__(self):
            self.observation_space = gymnasium.spaces.Box(low=0, high=255, shape=(4, 84, 84), dtype=np.uint8)
            self.action_space = gymnasium.spaces.Discrete(4)
            self.render_mode = 'ansi'
            self.tick = 0
            self.frame = None
            self.counter = -1

        def _draw(self):
            self.counter = counters['next']
            counters['next'] += 1
            self.frame = cnn_frame(self.counter)
            return self.frame

        def reset(self, seed=None):
            self.tick = 0
            return self._draw(), {}

        def step(self, action):
            reward = float(int(action) == int(self.frame[0, 0, 0]) % 4)
            self.tick += 1
            done = self.tick >= 5
            return self._draw(), reward, done, False, {'score': reward} if done else {}

    def make_env():
        return pufferlib.emulation.GymnasiumPufferEnv(env=pufferlib.postprocess.EpisodeStats(FrameEnv()))

    batch = num_envs * horizon
    config = pufferlib.namespace(
        env='frames', seed=1, torch_deterministic=True, cpu_offload=False, device='cpu',
        total_timesteps=batch * 8, learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95,
        update_epochs=2, norm_adv=True, clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1,
        max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=batch, minibatch_size=batch // 2,
        bptt_horizon=8, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
        data_dir='/tmp/golden_experiments', exp_id='golden')
    vec = pufferlib.vector.make(make_env, num_envs=num_envs, backend=pufferlib.vector.Serial)
    torch.manual_seed(1)
    net = pufferlib.models.Convolutional(vec.driver_env, framestack=4, flat_size=64 * 7 * 7)
    if use_rnn:    # environments/atari/torch.py:4-6: Recurrent = LSTMWrapper(input_size=512, hidden_size=512) over the NatureCNN Policy
        policy = pufferlib.frameworks.cleanrl.RecurrentPolicy(pufferlib.models.LSTMWrapper(vec.driver_env, net, input_size=512, hidden_size=512))
    else:
        policy = pufferlib.frameworks.cleanrl.Policy(net)
    bare = lambda k: k.split('.', 2)[2] if use_rnn and k.startswith('policy.policy.') else k.split('.', 2)[2] if use_rnn else k[len('policy.'):]  # noqa: E731
    out = {}
    for k, v in policy.state_dict().items():
        out['init.' + k] = digest(v.detach().numpy())      # layer_init under torch.manual_seed(1) (QR: equal up to LAPACK rounding)
    # the run itself starts from weights any platform can rebuild bit for bit: numpy's legacy normal stream, layer_init's scales
    with torch.no_grad():
        for k, v in policy.state_dict().items():
            v.copy_(torch.from_numpy(cnn_start_weight(bare(k), tuple(v.shape))))   # keyed by the bare parameter name
    for k, v in policy.state_dict().items():
        out['w0.' + k] = digest(v.detach().numpy())
    noise = []
    orig_multinomial = torch.multinomial

    def recording_multinomial(p, n, *a, **kw):
        st = torch.get_rng_state()
        res = orig_multinomial(p, n, *a, **kw)
        st2 = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty_like(p).exponential_(1)
        assert torch.equal((p / q).argmax(-1, keepdim=True), res), 'multinomial != argmax(p/q)'
        torch.set_rng_state(st2)
        noise.append(q.numpy().copy())
        return res

    torch.multinomial = recording_multinomial
    try:
        data = clean_pufferl.create(config, vec, policy)
        exp = data.experience
        for it in range(iters):
            noise.clear()
            # which frame every stored observation is: envs step in index order, so observation (t, e) is the frame env e holds at recv t
            frame_ids = []
            orig_recv = vec.recv

            def recv():
                frame_ids.append([env.env.env.counter for env in vec.envs])
                return orig_recv()
            vec.recv = recv
            clean_pufferl.evaluate(data)
            vec.recv = orig_recv
            out[f'it{it}.frame_ids'] = np.array(frame_ids[:horizon], np.int64)            # (T, N)
            obs = exp.obs.numpy().reshape(batch, 4, 84, 84)
            for t in range(horizon):
                for e in range(num_envs):
                    assert np.array_equal(obs[t * num_envs + e], cnn_frame(frame_ids[t][e])), (t, e)
            out[f'it{it}.noise'] = np.stack(noise)                                        # (T, N, A)
            out[f'it{it}.actions'] = exp.actions_np.copy().astype(np.int8)
            out[f'it{it}.logprobs'] = exp.logprobs_np.copy()
            out[f'it{it}.rewards'] = exp.rewards_np.copy()
            out[f'it{it}.dones'] = exp.dones_np.copy()
            out[f'it{it}.values'] = exp.values_np.copy()
            out[f'it{it}.global_step'] = np.array(data.global_step, np.int64)
            if use_rnn:
                out[f'it{it}.lstm_h'] = exp.lstm_h.numpy().copy()
                out[f'it{it}.lstm_c'] = exp.lstm_c.numpy().copy()
            lr_used = data.optimizer.param_groups[0]['lr']
            clean_pufferl.train(data)
            out[f'it{it}.lr_used'] = np.array(lr_used, np.float64)
            out[f'it{it}.advantages'] = exp.b_advantages.numpy().copy()
            out[f'it{it}.returns'] = exp.b_returns.numpy().copy()
            L = data.losses
            out[f'it{it}.losses'] = np.array([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac,
                                              L.explained_variance], np.float64)
            for k, v in policy.state_dict().items():
                out[f'it{it}.w.' + k] = digest(v.detach().numpy())
    finally:
        torch.multinomial = orig_multinomial
    out['config'] = np.array([num_envs, horizon, config.minibatch_size, config.bptt_horizon, config.update_epochs,
                              config.total_timesteps, iters], np.int64)
    out['hparams'] = np.array([config.learning_rate, config.gamma, config.gae_lambda, config.clip_coef, config.vf_coef,
                               config.vf_clip_coef, config.max_grad_norm, config.ent_coef], np.float64)
    fname = 'ppo_cnn_lstm.npz' if use_rnn else 'ppo_cnn.npz'
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, len(out), 'arrays; losses it0', out['it0.losses'])


def sha(a):
    """sha256 of the array's bytes (C order) as a numpy string: the fingerprint of a tensor that must match bit for bit."""
    import hashlib
    return np.array(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())


def gen_ppo_big(tag, use_rnn, num_envs, horizon, minibatch, bptt, epochs, lr, iters, total_timesteps=None, store_noise=True, full_state=False):
    """clean_pufferl.create/evaluate/train (clean_pufferl.py:30-292) by the UNMODIFIED reference at BASELINE's own sizes, recorded in
    digest form: what must match bit for bit as sha256 (observations, rewards, dones in storage order) or in
    full where the replay needs it (actions, initial weights, the multinomial noise unless `store_noise` is off), what must match
    within 1e-5 as digest(): sum, |sum|, 64 evenly spaced elements (values, log-probabilities, advantages, returns, LSTM state, updated
    weights and Adam moments), and the scalars (losses, learning rates, statistics, step counts).  store_noise=False (C2: 16.8 MB per
    rollout): only digest() of the noise and the recipe — the noise IS torch.manual_seed(seed) followed by one
    torch.empty(N, A).exponential_(1) per step (torch.multinomial's own draw, asserted below), which the test box regenerates; the
    replay falls back to the recorded actions where the box's torch draws other numbers.
    full_state: weights and Adam moments after every iteration but the last also in full (`it{k}.full.w.* / m.* / v.*`), so that a
    replay can continue from the reference's exact state (the demo shape's 32 Adam steps at lr 0.017 sit on a knife edge: see
    tests/test_oracle_golden.py::test_demo_shape_update_sits_on_a_knife_edge)."""
    import pufferlib
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import pufferlib.environments.ocean as ocean
    import clean_pufferl

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None

    batch = num_envs * horizon
    config = pufferlib.namespace(
        env='squared', seed=1, torch_deterministic=True, cpu_offload=False, device='cpu',
        total_timesteps=total_timesteps or batch * 8, learning_rate=lr, anneal_lr=True, gamma=0.99, gae_lambda=0.95,
        update_epochs=epochs, norm_adv=True, clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1,
        max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=batch, minibatch_size=minibatch,
        bptt_horizon=bptt, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
        data_dir='/tmp/golden_experiments', exp_id='golden')
    vec = pufferlib.vector.make(ocean.env_creator('squared'), num_envs=num_envs, backend=pufferlib.vector.Serial)
    torch.manual_seed(1)
    policy = pufferlib.models.Default(vec.driver_env, hidden_size=128)
    if use_rnn:
        policy = pufferlib.models.LSTMWrapper(vec.driver_env, policy, input_size=128, hidden_size=128)
        policy = pufferlib.frameworks.cleanrl.RecurrentPolicy(policy)
    else:
        policy = pufferlib.frameworks.cleanrl.Policy(policy)
    out = {}
    for k, v in policy.state_dict().items():
        out['w0.' + k] = v.detach().numpy().copy()
    noise = []
    orig_multinomial = torch.multinomial

    def recording_multinomial(p, n, *a, **kw):
        st = torch.get_rng_state()
        res = orig_multinomial(p, n, *a, **kw)
        st2 = torch.get_rng_state()
        torch.set_rng_state(st)
        q = torch.empty_like(p).exponential_(1)
        assert torch.equal((p / q).argmax(-1, keepdim=True), res), 'multinomial != argmax(p/q)'
        torch.set_rng_state(st2)
        noise.append(q.numpy().copy())
        return res

    torch.multinomial = recording_multinomial
    import time
    t_start = time.time()
    try:
        data = clean_pufferl.create(config, vec, policy)
        exp = data.experience
        for it in range(iters):
            noise.clear()
            clean_pufferl.evaluate(data)
            nz = np.stack(noise)                                               # (T, N, A)
            assert nz.shape == (horizon, num_envs, 8) and nz.dtype == np.float32
            if store_noise:
                out[f'it{it}.noise'] = nz
            out[f'it{it}.noise_digest'] = digest(nz)
            out[f'it{it}.obs_sha'] = sha(exp.obs.numpy().reshape(batch, -1).astype(np.int8))        # storage (step-major) order
            out[f'it{it}.rewards_sha'] = sha(exp.rewards_np.astype(np.float32))
            out[f'it{it}.dones_sha'] = sha(exp.dones_np.astype(np.float32))
            out[f'it{it}.rewards_sum'] = np.array(float(exp.rewards_np.astype(np.float64).sum()))
            out[f'it{it}.dones_sum'] = np.array(float(exp.dones_np.astype(np.float64).sum()))
            out[f'it{it}.actions'] = exp.actions_np.copy().astype(np.int8)
            out[f'it{it}.logprobs'] = digest(exp.logprobs_np)
            out[f'it{it}.values'] = digest(exp.values_np)
            out[f'it{it}.global_step'] = np.array(data.global_step, np.int64)
            out[f'it{it}.stats'] = np.array([data.stats.get('episode_return', np.nan), data.stats.get('episode_length', np.nan),
                                             data.stats.get('score', np.nan)], np.float64)
            if use_rnn:
                out[f'it{it}.lstm_h'] = digest(exp.lstm_h.numpy())
                out[f'it{it}.lstm_c'] = digest(exp.lstm_c.numpy())
            lr_used = data.optimizer.param_groups[0]['lr']
            clean_pufferl.train(data)
            out[f'it{it}.lr_used'] = np.array(lr_used, np.float64)
            out[f'it{it}.lr_next'] = np.array(data.optimizer.param_groups[0]['lr'], np.float64)
            out[f'it{it}.advantages'] = digest(exp.b_advantages.numpy())        # (nmb, minibatch) order
            out[f'it{it}.returns'] = digest(exp.b_returns.numpy())
            L = data.losses
            out[f'it{it}.losses'] = np.array([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac,
                                              L.explained_variance], np.float64)
            st = data.optimizer.state_dict()['state']
            names = [k for k, _ in policy.named_parameters()]
            for k, v in policy.state_dict().items():
                out[f'it{it}.w.' + k] = digest(v.detach().numpy())
            for i, nme in enumerate(names):
                out[f'it{it}.m.' + nme] = digest(st[i]['exp_avg'].numpy())
                out[f'it{it}.v.' + nme] = digest(st[i]['exp_avg_sq'].numpy())
            if full_state and it < iters - 1:
                for k, v in policy.state_dict().items():
                    out[f'it{it}.full.w.' + k] = v.detach().numpy().copy()
                for i, nme in enumerate(names):
                    out[f'it{it}.full.m.' + nme] = st[i]['exp_avg'].numpy().copy()
                    out[f'it{it}.full.v.' + nme] = st[i]['exp_avg_sq'].numpy().copy()
                out[f'it{it}.full.step'] = np.array(int(st[0]['step']), np.int64)
            print(f'  ppo_{tag} it{it}: {time.time() - t_start:.1f} s, losses', out[f'it{it}.losses'][:3])
    finally:
        torch.multinomial = orig_multinomial
    out['config'] = np.array([num_envs, horizon, config.minibatch_size, config.bptt_horizon, config.update_epochs,
                              config.total_timesteps, iters], np.int64)
    out['hparams'] = np.array([config.learning_rate, config.gamma, config.gae_lambda, config.clip_coef, config.vf_coef,
                               config.vf_clip_coef, config.max_grad_norm, config.ent_coef], np.float64)
    out['torch_version'] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, f'ppo_{tag}.npz'), **out)
    print(f'ppo_{tag}.npz', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, f'ppo_{tag}.npz')) // 1024, 'KiB')


def gen_big():
    """The reference at BASELINE's own sizes (`make_golden.py big`; ~1 minute of reference CPU time):
      ppo_c1_mlp / ppo_c1_lstm  configs[0]: 64 envs x 128 steps, minibatch 2048, bptt 16, 4 epochs (config.yaml:12-41 defaults)
      ppo_demo_lstm             what `demo.py --env squared` really trains (config.yaml:498-509): 8 envs, batch 1024, minibatch 128,
                                bptt 4, lr 0.017, LSTM — a partition the one-pass GAE sums refuse (bptt < 8): the un-fused path
      ppo_c2_mlp                ONE iteration of configs[1]: 4096 envs x 128 steps, 4 minibatches x 4 epochs"""
    gen_ppo_big('c1_mlp', False, 64, 128, 2048, 16, 4, 2.5e-4, iters=2)
    gen_ppo_big('c1_lstm', True, 64, 128, 2048, 16, 4, 2.5e-4, iters=2)
    gen_ppo_big('demo_lstm', True, 8, 128, 128, 4, 4, 0.017, iters=2, total_timesteps=30_000, full_state=True)
    gen_ppo_big('c2_mlp', False, 4096, 128, 131072, 16, 4, 2.5e-4, iters=1, store_noise=False)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'cnn':
        gen_ppo_cnn()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'cnn_lstm':   # the recurrent NatureCNN of environments/atari/torch.py:4-6
        gen_ppo_cnn(use_rnn=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'mp':
        gen_ppo_mp()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'big':      # the reference at BASELINE's own sizes, digest form
        gen_big()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'wide':     # pufferlib.models.Default(hidden_size=256): outside the fused kernels' width
        gen_ppo('mlp_h256', use_rnn=False, hidden=256)
        sys.exit(0)
    import clean_pufferl  # builds c_gae through pyximport exactly as the reference does (clean_pufferl.py:24-27)
    gen_gae(clean_pufferl.compute_gae)
    gen_squared('d3t1', 64, 3, 1, 1, 60)
    gen_squared('d1t4', 8, 1, -1, 42, 40)      # n=8 <= 21: pool-swap sampling, variable bit widths
    gen_squared('d2t2', 16, 2, 2, 7, 40)       # n=16: pool path with 2 draws
    gen_squared('d4t3', 16, 4, 3, 3, 60)       # n=32 > 21: set-rejection sampling
    gen_squared('d3t1_big', 700, 3, 1, 4090, 12)  # crosses several MT19937 regenerations per reset round
    gen_stochastic()
    gen_bandit()
    gen_nativize()
    gen_multiagent()
    gen_memory('l2d2', 5, 2, 2, 11, 40)
    gen_memory('l3d1', 130, 3, 1, 4090, 30)    # 130 x 7 words per reset round: crosses MT19937 blocks
    gen_ppo('mlp', use_rnn=False)
    gen_ppo('lstm', use_rnn=True)
    gen_ppo('spaces', use_rnn=False, env='spaces')
    gen_ppo('mlp_h256', use_rnn=False, hidden=256)
    gen_ppo_mp()
    gen_ppo_cnn()
    gen_ppo_cnn(use_rnn=True)
    gen_big()
