"""Device-resident ocean Memory vecenv (csrc/memory.hip) vs golden trajectories of the unmodified
reference and the C oracle: solutions drawn from numpy's global legacy stream (per-env seeding, shared stream afterwards, tape
ahead of the sends), observations, rewards, terminals, auto-reset rows, episode infos — all bit for bit; and the task itself:
the recurrent policy learns it, the memory-less one cannot."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(n, L=2, D=2, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_memory, env_kwargs=dict(mem_length=L, mem_delay=D), num_envs=n, backend=vector.Memory, **kw)


def _bits_to_solutions(bits, L, H):
    sol = -np.ones((len(bits), H), np.int8)
    for j in range(L):
        sol[:, j] = (bits >> j) & 1
    return sol


@pytest.mark.parametrize('tag', ['l2d2', 'l3d1'])
def test_protocol_replays_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f'memory_{tag}.npz'))
    n, L, D, seed, steps = (int(x) for x in g['config'])
    vec = _make(n, L, D)
    vec.async_reset(seed)
    infos = []
    for k in range(steps + 1):
        o, r, te, tr, info, ids, m = vec.recv()
        assert np.array_equal(o.cpu().numpy(), g['obs'][k]) and np.array_equal(r.cpu().numpy(), g['rewards'][k]), k
        assert np.array_equal(te.cpu().numpy(), g['terminals'][k]) and not tr.any() and m.all(), k
        bits, under = vec.debug_solutions()
        assert under == 0
        assert np.array_equal(_bits_to_solutions(bits, L, 2 * L + D), g['solutions'][k]), k
        for j, i in enumerate(info):
            infos.append((k, j, i['episode_return'], i['episode_length'], i['score']))
        if k < steps:
            vec.send(g['actions'][k].astype(np.int64))
    assert np.array_equal(np.array(infos, np.float64).reshape(-1, 5), g['infos'])


@pytest.mark.parametrize('n,L,D,sends', [(700, 2, 2, 80), (4096, 1, 0, 40), (33, 16, 3, 120)])
def test_device_equals_oracle_over_many_reset_rounds(n, L, D, sends):
    """Sizes that cross many MT19937 blocks per reset round, the shortest possible episode (L=1, D=0: 1 step + reset row) and
    the longest digit string; random actions; tape filled one send at a time (protocol path)."""
    from oracle import c_oracle
    dev = _make(n, L, D)
    ref = c_oracle.MemorySerial(n, L, D)
    dev.async_reset(123)
    ref.async_reset(123)
    rng = np.random.RandomState(n + L)
    for k in range(sends):
        o, r, te, _, info_d, _, _ = dev.recv()
        o2, r2, te2, _, info_r, _, _ = ref.recv()
        assert np.array_equal(o.cpu().numpy(), o2) and np.array_equal(r.cpu().numpy(), r2) and np.array_equal(te.cpu().numpy(), te2), k
        assert [(i['episode_return'], i['episode_length'], i['score']) for i in info_d] == \
               [(i['episode_return'], i['episode_length'], i['score']) for i in info_r], k
        a = rng.randint(0, 2, n).astype(np.int64)
        dev.send(a)
        ref.send(a)
    assert dev.debug_solutions()[1] == 0


def _train(recurrent, updates, L=2, D=2, n=1024, horizon=96):
    from pufferlib_amd import clean_pufferl, cleanrl, models, namespace
    torch.manual_seed(0)
    vec = _make(n, L, D)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent else cleanrl.Policy(base)
    B = n * horizon
    cfg = namespace(env='memory', seed=1, torch_deterministic=True, cpu_offload=False, device='cuda', total_timesteps=B * updates,
                    learning_rate=5e-3, anneal_lr=True, gamma=0.95, gae_lambda=0.9, update_epochs=4, norm_adv=True,
                    clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01,
                    target_kl=None, batch_size=B, minibatch_size=B // 4, bptt_horizon=8, compile=False,
                    checkpoint_interval=0, data_dir='/tmp/pfa_experiments', exp_id='mem')
    data = clean_pufferl.create(cfg, vec, pol)
    scores = []
    for _ in range(updates):
        stats, _ = clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        if 'score' in stats:
            scores.append(stats['score'])
    return scores


def test_only_the_recurrent_policy_learns_memory():
    lstm = _train(True, 60)
    mlp = _train(False, 60)
    print('memory scores: lstm', [round(x, 3) for x in lstm[::10]], 'mlp', [round(x, 3) for x in mlp[::10]])
    assert lstm[0] < 0.4 and lstm[-1] > 0.9, lstm[-5:]
    assert mlp[-1] < 0.6, mlp[-5:]          # two hidden digits: a memory-less policy cannot beat chance by much


@pytest.mark.parametrize('n,L,D,T', [(50, 2, 2, 37), (4096, 3, 1, 16)])
def test_fused_recurrent_rollout_equals_the_stepwise_protocol_pieces(n, L, D, T):
    """evaluate() for Memory + the LSTM policy is ONE persistent kernel (pfa_rollout_lstm_memory); it must equal the
    protocol-level loop (policy step, Experience.store, device_send: three launches per step) bit for bit — experience rows,
    live buffers, recurrent state, episode statistics — over two rollouts (state and env phase carry over), ragged tile."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    runs = []
    for fused in (True, False):
        torch.manual_seed(4)
        vec = _make(n, L, D)
        pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
        bptt = T if T < 16 else 16
        data = clean_pufferl.create(_config(n, T, n * T, 1, 1, n * T * 8, hp, seed=11), vec, pol)
        out = []
        for it in range(2):
            if fused:
                stats, _ = clean_pufferl.evaluate(data)
            else:
                clean_pufferl._rollout_stepwise(data, None, T, n)
                stats, _ = clean_pufferl._finish_evaluate(data, n, T)
            e = data.experience
            out.append([x.clone() for x in (e.obs, e.actions, e.logprobs, e.values, e.rewards, e.dones, vec.obs_buf, vec.rewards,
                                            vec.terminals_u8, data.lstm_engine.lstm_h, data.lstm_engine.lstm_c)] + [stats])
        runs.append(out)
        assert vec.debug_solutions()[1] == 0
    for it in range(2):
        for k, (a, b) in enumerate(zip(runs[0][it][:-1], runs[1][it][:-1])):
            assert torch.equal(a, b), (it, k)
        assert runs[0][it][-1] == runs[1][it][-1]
    assert runs[0][1][-1]['episode_length'] == 2 * L + D - 1
