"""The synthetic byte-row vecenv (csrc/synth_env.hpp, vector.Synthetic) — BASELINE configs[2]'s workload shape (MiniGrid-shaped
160-byte rows, 7 actions, 100-step episodes; the simulator itself is third-party, env parity unpinned) — and the recurrent
policy on it: generator values against an independent Philox restatement, protocol state machine, fused rollout == stepwise
pieces bit for bit, and the BPTT update against the torch-fp32 oracle trainer on the device rollout's experience."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


def _make(n, **kw):
    from pufferlib_amd import vector
    return vector.make(vector.make_synthetic, env_kwargs=kw, num_envs=n, backend=vector.Synthetic)


def _row(seed, env, episode, tick, values=160, high=10):
    from oracle import c_oracle
    out = np.zeros(values, np.float32)
    for chunk in range((values + 15) // 16):
        w = c_oracle.philox4x32_10([env, chunk, episode, tick], [seed & 0xFFFFFFFF, 0x5359 ^ (seed >> 32)])
        b = np.frombuffer(w.astype('<u4').tobytes(), np.uint8)
        k = min(16, values - 16 * chunk)
        out[16 * chunk:16 * chunk + k] = (b % (high + 1))[:k]
    return out


def test_generator_values_and_protocol_state_machine():
    n, ep = 37, 5
    vec = _make(n, episode_length=ep)
    assert vec.obs_stride == 160 and vec.single_observation_space.shape == (160,) and vec.single_action_space.n == 7
    vec.async_reset(9)
    tick = np.zeros(n, int)          # steps taken in the current episode; == ep: the env is done, the next send is its reset row
    episode = np.zeros(n, int)
    want_r, want_t = np.zeros(n, np.float32), np.zeros(n, bool)
    rng = np.random.default_rng(0)
    finished = 0
    for t in range(3 * (ep + 1) + 2):
        o, r, te, tr, infos, ids, mask = vec.recv()
        o = o.cpu().numpy()
        assert o.shape == (n, 160) and o.min() >= 0 and o.max() <= 10
        for e in (0, 5, n - 1):
            assert np.array_equal(o[e], _row(9, e, episode[e], tick[e])), (t, e)
        assert np.array_equal(r.cpu().numpy(), want_r) and np.array_equal(te.cpu().numpy(), want_t), t
        finished += len(infos)
        for i in infos:
            assert i['episode_length'] == ep and 0 <= i['score'] <= 1 and i['episode_return'] == i['score'] * ep
        a = rng.integers(0, 7, n)
        done = tick == ep
        want_r = np.where(done, 0.0, (a == o[:, 0].astype(int) % 7)).astype(np.float32)     # a reset row ignores the action
        tick = np.where(done, 0, tick + 1)
        episode = episode + done
        want_t = (tick == ep) & ~done
        vec.send(a)
    assert finished == 3 * n


@pytest.mark.parametrize('n,T', [(50, 23), (4096, 16)])
def test_fused_recurrent_rollout_equals_the_stepwise_pieces(n, T):
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    runs = []
    for fused in (True, False):
        torch.manual_seed(4)
        vec = _make(n, episode_length=7)
        pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
        with torch.no_grad():
            pol.policy.policy.encoder.weight.mul_(0.3)
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
    for it in range(2):
        for k, (a, b) in enumerate(zip(runs[0][it][:-1], runs[1][it][:-1])):
            assert torch.equal(a, b), (it, k)
        assert runs[0][it][-1] == runs[1][it][-1]
    assert runs[0][1][-1]['episode_length'] == 7


def test_c3_shaped_update_vs_oracle_trainer_on_device_rollout():
    """160-float rows, 7 actions, LSTM(128), bptt 16: the device rollout's experience into the HIP BPTT update and into the
    torch-fp32 restatement of clean_pufferl.train (two updates, lr schedule and Adam state carried)."""
    from host_vecenv import HostByteRows
    from oracle import ppo_torch
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    n, horizon, nmb, bptt = 64, 32, 2, 16
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    vec = _make(n, episode_length=9)
    torch.manual_seed(8)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    with torch.no_grad():
        pol.policy.policy.encoder.weight.mul_(0.3)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=3), vec, pol)
    opol = ppo_torch.Policy.from_reference_state_dict({k: v.detach().cpu().clone() for k, v in pol.state_dict().items()})
    sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
    tr = None
    for it in range(2):
        clean_pufferl.evaluate(data)
        e = data.experience
        new = ppo_torch.Trainer(opol, HostByteRows(n), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                                learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4], vf_clip_coef=hp[5],
                                max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=3)
        if tr is not None:
            new.opt = tr.opt
        tr = new
        tr.obs = torch.as_tensor(sm(e.obs)[:, :160].copy())
        tr.actions = sm(e.actions).astype(np.int64)
        tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (e.logprobs, e.rewards, e.dones, e.values))
        tr.global_step = data.global_step
        Lo = tr.train()
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.approx_kl], [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'approx_kl')],
                                   rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k, arr in opol.state_arrays().items():
            key = ('policy.recurrent.' + k) if k.endswith('_l0') else ('policy.policy.' + k)
            np.testing.assert_allclose(sd[key].cpu().numpy(), arr, rtol=1e-5, atol=1e-5, err_msg=k)
