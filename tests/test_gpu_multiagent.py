"""Device-resident ocean Multiagent vecenv (csrc/multiagent.hip): two agent rows per env behind the
backend protocol, against the golden trajectory of the unmodified reference (PettingZooPufferEnv + Serial) and the oracle;
then create -> evaluate -> train with num_agents = 2 x envs, whose statistics are the reference's `1/score`, `2/score`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(n, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_multiagent, num_envs=n, backend=vector.Multiagent, **kw)


def test_protocol_replays_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'multiagent.npz'))
    n, seed, steps = (int(x) for x in g['config'])
    vec = _make(n)
    assert vec.num_agents == 2 * n and vec.agents_per_env == [2] * n and vec.action_space.shape == (2 * n,)
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o.cpu().numpy(), g['obs'][k]) and np.array_equal(r.cpu().numpy(), g['rewards'][k]), k
        assert np.array_equal(te.cpu().numpy(), g['terminals'][k]) and not tr.any() and m.all(), k
        assert np.array_equal(ids, np.arange(2 * n))
        infos += [(k, j, i[1]['score'], i[2]['score']) for j, i in enumerate(info)]
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.int64).reshape(-1, 4), g['infos'])
    st = vec.episode_stats().cpu().numpy()
    fin = g['infos']
    assert st[0] == st[2] == len(fin) and st[1] == fin[:, 2].sum() and st[3] == fin[:, 3].sum()


@pytest.mark.parametrize('n', [1, 127, 128, 5000])
def test_sizes_across_workgroup_boundaries_match_the_oracle(n):
    from oracle import c_oracle
    dev, ref = _make(n, info_mode='lazy'), c_oracle.MultiagentSerial(n)
    dev.async_reset(0)
    ref.async_reset(0)
    rng = np.random.default_rng(n)
    tot = np.zeros(4)
    for t in range(7):
        o, r, te, _, _, _, _ = dev.recv()
        o2, r2, te2, _, info, _, _ = ref.recv()
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(r.cpu().numpy(), r2) and np.array_equal(te.cpu().numpy(), te2), t
        for i in info:
            tot += (1, i[1]['score'], 1, i[2]['score'])
        a = rng.integers(0, 2, 2 * n).astype(np.int64)
        dev.send(a)
        ref.send(a)
    for i in ref.recv()[4]:
        tot += (1, i[1]['score'], 1, i[2]['score'])
    assert np.array_equal(dev.episode_stats().cpu().numpy(), tot)
    assert not dev.episode_stats().cpu().numpy().any()           # accumulators were reset by the read above


def test_ppo_trains_both_agent_slots_through_create_evaluate_train():
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from oracle import c_oracle
    from test_gpu_ppo import _config
    n, horizon = 512, 32
    hp = [2.5e-3, 0.95, 0.9, 0.1, 0.5, 0.1, 0.5, 0.01]
    torch.manual_seed(3)
    vec = _make(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    A = vec.num_agents
    cfg = _config(A, horizon, A * horizon // 4, 16, 4, A * horizon * 30, hp, seed=2)
    data = clean_pufferl.create(cfg, vec, pol)
    stats, _ = clean_pufferl.evaluate(data)
    assert data.global_step == A * horizon and set(stats) == {'1/score', '2/score'}
    # rollout rows vs the oracle under the policy's own actions
    e = data.experience
    acts = e.actions.view(A, horizon).cpu().numpy()
    ref = c_oracle.MultiagentSerial(n)
    ref.async_reset(2)
    s1 = []
    for t in range(horizon):
        o, r, d, _, info, _, _ = ref.recv()
        assert np.array_equal(o[:, 0], e.obs.view(A, horizon, -1)[:, t, 0].cpu().numpy()), t
        assert np.array_equal(r, e.rewards.view(A, horizon)[:, t].cpu().numpy()), t
        assert np.array_equal(d.astype(np.float32), e.dones.view(A, horizon)[:, t].cpu().numpy()), t
        s1 += [i[1]['score'] for i in info]
        ref.send(acts[:, t].astype(np.int64))
    s1 += [i[1]['score'] for i in ref.recv()[4]]
    assert abs(stats['1/score'] - np.mean(s1)) < 1e-12
    first = dict(stats)
    clean_pufferl.train(data)
    for _ in range(29):
        stats, _ = clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    assert first['1/score'] < 0.65 and first['2/score'] < 0.65 and stats['1/score'] > 0.95 and stats['2/score'] > 0.95, (first, stats)
