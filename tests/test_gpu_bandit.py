"""Device-resident ocean Bandit vecenv (csrc/bandit.hip) vs the golden trajectory of the unmodified
reference and the C oracle: protocol path bit-exact (f32 rewards incl. the legacy-gauss noise table, terminals, reset rows,
f64 episode infos), other configurations against the oracle, and create -> evaluate -> train end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(n, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_bandit, env_kwargs=kw, num_envs=n, backend=vector.Bandit)


def test_protocol_replays_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'bandit.npz'))
    n, na, seed, steps = (int(x) for x in g['config'])
    scale, noise = g['scale_noise']
    vec = _make(n, num_actions=na, reward_scale=scale, reward_noise=noise)
    assert vec.solution == int(g['solution'][0])
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


@pytest.mark.parametrize('cfg', [dict(num_actions=10, reward_scale=1, reward_noise=1), dict(num_actions=4, reward_scale=0.5, reward_noise=1),
                                 dict(num_actions=15, reward_scale=3, reward_noise=0)])
def test_other_configurations_match_the_oracle(cfg):
    from oracle import c_oracle
    n = 300
    dev = _make(n, **cfg)
    ref = c_oracle.BanditSerial(n, cfg['num_actions'], cfg['reward_scale'], cfg['reward_noise'])
    dev.async_reset(3)
    ref.async_reset(3)
    assert dev.solution == ref.solution
    rng = np.random.default_rng(1)
    for t in range(9):
        o, r, te, _, info_d, _, _ = dev.recv()
        o2, r2, te2, _, info_r, _, _ = ref.recv()
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(r.cpu().numpy(), r2) and np.array_equal(te.cpu().numpy(), te2), t
        assert len(info_d) == len(info_r)
        for x, y in zip(info_d, info_r):
            assert x['episode_return'] == y['episode_return'] and x['score'] == y['score'] and x['episode_length'] == 1, t
        a = rng.integers(0, cfg['num_actions'], n).astype(np.int64)
        dev.send(a)
        ref.send(a)
    st = dev.episode_stats().cpu().numpy()
    assert st[0] == n * 5 and st[2] == n * 5


def test_rollout_rows_match_the_oracle_and_the_update_runs():
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from oracle import c_oracle
    from test_gpu_ppo import _config
    n, horizon = 64, 16
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    torch.manual_seed(2)
    vec = _make(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 8, 2, n * horizon * 10, hp, seed=9), vec, pol)
    stats, _ = clean_pufferl.evaluate(data)
    e = data.experience
    acts = e.actions.view(n, horizon).cpu().numpy()
    ref = c_oracle.BanditSerial(n)
    ref.async_reset(9)
    scores = []
    for t in range(horizon):
        o, r, d, _, info, _, _ = ref.recv()
        assert np.array_equal(r, e.rewards.view(n, horizon)[:, t].cpu().numpy()), t
        assert np.array_equal(d.astype(np.float32), e.dones.view(n, horizon)[:, t].cpu().numpy()), t
        scores += [i['score'] for i in info]
        ref.send(acts[:, t].astype(np.int64))
    scores += [i['score'] for i in ref.recv()[4]]
    assert len(scores) == n * horizon // 2 and abs(stats['score'] - np.mean(scores)) < 1e-12 and stats['episode_length'] == 1
    before = pol.state_dict()['policy.decoder.bias'].clone()
    clean_pufferl.train(data)
    assert np.isfinite(data.losses.value_loss) and not torch.equal(before, pol.state_dict()['policy.decoder.bias'])


def test_wrong_creator_and_too_many_actions_raise():
    from pufferlib_amd import vector
    from pufferlib_amd.exceptions import APIUsageError
    with pytest.raises(APIUsageError):
        vector.make(vector.make_squared, num_envs=4, backend=vector.Bandit)
    with pytest.raises(APIUsageError):
        _make(4, num_actions=16)
