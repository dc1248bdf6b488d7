"""Device-resident ocean Stochastic vecenv (csrc/stochastic.hip) vs the golden trajectory of the
unmodified reference and the C oracle: protocol path bit-exact (observations, f32 rewards, terminals, auto-reset rows, episode
infos), every reachable reward state, fused rollout == stepwise protocol path, and a full create -> evaluate -> train against
the torch-fp32 oracle trainer."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(n, p=0.7, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_stochastic, env_kwargs=dict(p=p), num_envs=n, backend=vector.Stochastic, **kw)


def test_protocol_replays_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'stochastic.npz'))
    n, seed, steps = (int(x) for x in g['config'])
    vec = _make(n, float(g['p'][0]))
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o.cpu().numpy(), g['obs'][k]) and np.array_equal(r.cpu().numpy(), g['rewards'][k]), k
        assert np.array_equal(te.cpu().numpy(), g['terminals'][k]) and not tr.any() and m.all(), k
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])
    st = vec.episode_stats().cpu().numpy()
    assert st[0] == 12 and abs(st[1] - g['infos'][:, 2].sum()) < 1e-9 and st[2] == 1200


def test_every_reachable_reward_state_matches_the_oracle():
    """N = 101 envs, env c plays one action on its first c steps and the other after, then the mirrored schedule — together
    they visit every (tick, count, last action) of a 100-step episode (incl. the states where float ** 2 != x * x)."""
    from oracle import c_oracle
    for p in (0.7, 0.5, 1 / 3):
        for first in (0, 1):
            n = 101
            dev = _make(n, p)
            ref = c_oracle.StochasticSerial(n, p, 100)
            dev.async_reset(0)
            ref.async_reset(0)
            dev.recv()
            for t in range(101):
                a = np.where(np.arange(n) > t, first, 1 - first).astype(np.int64)
                dev.send(a)
                ref.send(a)
                o, r, te, _, info_d, _, _ = dev.recv()
                o2, r2, te2, _, info_r, _, _ = ref.recv()
                assert np.array_equal(r.cpu().numpy(), r2) and np.array_equal(te.cpu().numpy(), te2), (p, first, t)
                assert len(info_d) == len(info_r)
                for x, y in zip(info_d, info_r):
                    assert x['episode_return'] == y['episode_return'] and x['score'] == y['score'], (p, first, t)


def _config(n, horizon, mbs, bptt, hp, **over):
    from test_gpu_ppo import _config as base
    return base(n, horizon, mbs, bptt, 2, n * horizon * 10, hp, **over)


def test_fused_rollout_equals_stepwise_protocol_and_oracle_update():
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from pufferlib_amd import clean_pufferl, cleanrl, models, _lib
    from oracle import c_oracle, ppo_torch
    n, horizon = 40, 128                      # 128 steps cross the 100-step episode end and its auto-reset row
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    torch.manual_seed(4)
    vec = _make(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    with torch.no_grad():
        for p_ in pol.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    cfg = _config(n, horizon, n * horizon // 2, 16, hp, seed=5)
    data = clean_pufferl.create(cfg, vec, pol)
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    stats, _ = clean_pufferl.evaluate(data)
    e = data.experience
    acts = e.actions.view(n, horizon).cpu().numpy()
    # stepwise protocol path on a second vecenv with the same policy and noise stream
    vec2 = _make(n)
    pol2 = cleanrl.Policy(models.Default(vec2.driver_env))
    pol2.load_state_dict(pol.state_dict())
    pol2.noise_seed = 5
    vec2.async_reset(5)
    rew2, done2, act2, lp2 = [], [], [], []
    for t in range(horizon):
        o, r, d, _, _, _, _ = vec2.recv()
        a, lp, _, val = pol2(o)
        rew2.append(r.clone()); done2.append(d.clone()); act2.append(a.clone()); lp2.append(lp.clone())
        vec2.send(a)
    assert torch.equal(torch.stack(act2, 1).int(), e.actions.view(n, horizon))
    assert torch.equal(torch.stack(rew2, 1), e.rewards.view(n, horizon))
    assert torch.equal(torch.stack(done2, 1).float(), e.dones.view(n, horizon))
    assert torch.equal(torch.stack(lp2, 1), e.logprobs.view(n, horizon))
    # env side vs the C oracle on the same actions
    ref = c_oracle.StochasticSerial(n, 0.7, 100)
    ref.async_reset(5)
    rets = []
    for t in range(horizon):
        o, r, d, _, info, _, _ = ref.recv()
        assert np.array_equal(r, e.rewards.view(n, horizon)[:, t].cpu().numpy()), t
        assert np.array_equal(d.astype(np.float32), e.dones.view(n, horizon)[:, t].cpu().numpy()), t
        rets += [i['episode_return'] for i in info]
        ref.send(acts[:, t].astype(np.int64))
    rets += [i['episode_return'] for i in ref.recv()[4]]
    assert len(rets) == n and abs(stats['episode_return'] - np.mean(rets)) < 1e-12 and stats['episode_length'] == 100
    # update vs the torch-fp32 oracle trainer on the same rollout
    from test_gpu_ppo import _step_major
    B = n * horizon
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, c_oracle.StochasticSerial(n, 0.7, 100), batch_size=B, minibatch_size=B // 2, bptt_horizon=16,
                           update_epochs=2, learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4],
                           vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=5)
    tr.obs = torch.as_tensor(_step_major(e.obs, n, horizon)[:, :1].copy())
    tr.actions = _step_major(e.actions, n, horizon).astype(np.int64)
    tr.logprobs = _step_major(e.logprobs, n, horizon).copy()
    tr.rewards = _step_major(e.rewards, n, horizon).copy()
    tr.dones = _step_major(e.dones, n, horizon).copy()
    tr.values = _step_major(e.values, n, horizon).copy()
    tr.global_step = data.global_step
    Lo = tr.train()
    clean_pufferl.train(data)
    L = data.losses
    np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.approx_kl], [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'approx_kl')],
                               rtol=1e-5, atol=1e-5)
    sd = pol.state_dict()
    for k, arr in opol.state_arrays().items():
        np.testing.assert_allclose(sd['policy.' + k].cpu().numpy(), arr, rtol=1e-5, atol=1e-5, err_msg=k)


def test_recurrent_policy_is_refused_and_wrong_creator_raises():
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from pufferlib_amd.exceptions import APIUsageError
    vec = _make(16)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    with pytest.raises(NotImplementedError):
        clean_pufferl.create(_config(16, 32, 256, 16, [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]), vec, pol)
    with pytest.raises(APIUsageError):
        vector.make(vector.make_squared, num_envs=4, backend=vector.Stochastic)
