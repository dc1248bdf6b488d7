"""MultiDiscrete action heads on the MLP policy (models.py:29-35,55-58; cleanrl.py:25-47 list branch): per-head sampling in
the forward kernel, per-head log-softmax / entropy in the fused PPO gradient kernel, [rows, heads] actions at the protocol
boundary.  Against (a) the unmodified reference's create/evaluate/train on ocean Spaces (tests/golden/ppo_spaces.npz: Dict
observation -> 108-byte rows -> obs stride 128, Dict action -> MultiDiscrete([2, 2])), observations played back, and (b) the
torch-fp32 oracle trainer on a three-head host env."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu


def _policy(vec, weights=None):
    from pufferlib_amd import cleanrl, models
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    if weights is not None:
        pol.load_state_dict({k: torch.as_tensor(v) for k, v in weights.items()})
    return pol


def test_forward_sample_matches_the_oracle_for_four_heads():
    from host_vecenv import HostMultiHead
    from oracle import ppo_torch
    nvec = [3, 2, 5, 4]
    vec = HostMultiHead(300, nvec, obs_dim=33)
    torch.manual_seed(0)
    pol = _policy(vec)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.3 * torch.randn_like(p))
    w = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    assert list(w)[2:6] == ['decoder.0.weight', 'decoder.0.bias', 'decoder.1.weight', 'decoder.1.bias']
    obs = torch.randn(300, 33)
    noise = torch.empty(300, sum(nvec)).exponential_(1)
    a, lp, ent, val = pol(obs.cuda(), noise=noise)
    opol = ppo_torch.Policy(w)
    with torch.no_grad():
        logits, oval, _ = opol.forward(obs)
        oa, olp, oent = ppo_torch.sample_logits(logits, noise=noise)
    assert a.shape == (300, 4) and torch.equal(a.cpu(), oa)
    np.testing.assert_allclose(lp.cpu().numpy(), olp.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), oent.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(val.cpu().numpy(), oval.numpy(), rtol=1e-5, atol=1e-5)


def _spaces_noise(g, it, n, horizon):
    q = g[f'it{it}.noise']
    return q.reshape(horizon, 2, n, 2).transpose(0, 2, 1, 3).reshape(horizon, n, 4)


def test_spaces_golden_replay_through_create_evaluate_train(golden_dir):
    from host_vecenv import SpacesReplay
    from pufferlib_amd import clean_pufferl
    from test_gpu_ppo import _config
    g = np.load(os.path.join(golden_dir, 'ppo_spaces.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    hp = [float(x) for x in g['hparams']]
    rounds = np.concatenate([g[f'it{it}.obs'].reshape(horizon, n, 108) for it in range(iters)])
    vec = SpacesReplay(np.concatenate([rounds, rounds[-1:]]))
    pol = _policy(vec, {k[3:]: g[k] for k in g.files if k.startswith('w0.')})
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, hp, seed=1), vec, pol)
    assert data.flat_params.obs_stride == 128 and data.flat_params.nvec == [2, 2]
    step_major = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(n * horizon, *x.shape[1:]).cpu().numpy()  # noqa: E731
    for it in range(iters):
        data.noise = torch.as_tensor(_spaces_noise(g, it, n, horizon))
        stats, _ = clean_pufferl.evaluate(data)
        e = data.experience
        acts = data.flat_params.unpack_actions(e.actions.long())
        assert np.array_equal(step_major(acts), g[f'it{it}.actions'].astype(np.int64)), 'actions differ'
        assert np.array_equal(step_major(e.obs)[:, :108], g[f'it{it}.obs'].astype(np.float32))
        assert np.array_equal(step_major(e.rewards), g[f'it{it}.rewards']) and np.array_equal(step_major(e.dones), g[f'it{it}.dones'])
        np.testing.assert_allclose(step_major(e.logprobs), g[f'it{it}.logprobs'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(step_major(e.values), g[f'it{it}.values'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-12)
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                                   g[f'it{it}.losses'][:6], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.' + k], rtol=1e-5, atol=1e-5, err_msg=k)


def test_three_heads_rollout_and_update_vs_oracle_trainer():
    from host_vecenv import HostMultiHead
    from oracle import ppo_torch
    from pufferlib_amd import clean_pufferl
    from test_gpu_ppo import _config
    nvec, n, horizon, nmb, bptt = [3, 4, 2], 40, 32, 2, 8
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    vec = HostMultiHead(n, nvec)
    torch.manual_seed(5)
    pol = _policy(vec)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.1 * torch.randn_like(p))
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=3), vec, pol)
    assert data.flat_params.obs_stride == 32
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, HostMultiHead(n, nvec), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4], vf_clip_coef=hp[5],
                           max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=3)
    for it in range(2):
        noise = torch.empty(horizon, n, sum(nvec)).exponential_(1)
        data.noise = noise.clone()
        stats, _ = clean_pufferl.evaluate(data)
        ostats = tr.evaluate(noise.numpy())
        e = data.experience
        sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
        assert np.array_equal(sm(data.flat_params.unpack_actions(e.actions.long())), tr.actions), it
        assert np.array_equal(sm(e.rewards), tr.rewards) and np.array_equal(sm(e.dones), tr.dones)
        np.testing.assert_allclose(sm(e.logprobs), tr.logprobs, rtol=1e-5, atol=1e-5)
        assert abs(stats['score'] - ostats['score']) < 1e-9
        Lo = tr.train()
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                                   [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')],
                                   rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k, arr in opol.state_arrays().items():
            np.testing.assert_allclose(sd['policy.' + k].cpu().numpy(), arr, rtol=1e-5, atol=1e-5, err_msg=k)


def test_three_heads_with_the_recurrent_policy_vs_oracle_trainer():
    """LSTMWrapper over the multi-head Default (models.py:84-111 -> decode_actions list branch): recurrent policy step with
    per-head sampling on the host path, then the BPTT update whose heads/loss kernel takes the per-head softmax."""
    from host_vecenv import HostMultiHead
    from oracle import ppo_torch
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    nvec, n, horizon, nmb, bptt = [3, 4, 2], 32, 32, 2, 8
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    vec = HostMultiHead(n, nvec)
    torch.manual_seed(6)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=3), vec, pol)
    opol = ppo_torch.Policy.from_reference_state_dict(sd0)
    assert opol.recurrent and opol.heads == nvec
    tr = ppo_torch.Trainer(opol, HostMultiHead(n, nvec), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4], vf_clip_coef=hp[5],
                           max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=3)
    for it in range(2):
        noise = torch.empty(horizon, n, sum(nvec)).exponential_(1)
        data.noise = noise.clone()
        clean_pufferl.evaluate(data)
        tr.evaluate(noise.numpy())
        e = data.experience
        sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
        assert np.array_equal(sm(data.flat_params.unpack_actions(e.actions.long())), tr.actions), it
        assert np.array_equal(sm(e.rewards), tr.rewards)
        np.testing.assert_allclose(sm(e.logprobs), tr.logprobs, rtol=1e-5, atol=1e-5)
        Lo = tr.train()
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.approx_kl], [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'approx_kl')],
                                   rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k, arr in opol.state_arrays().items():
            key = ('policy.recurrent.' + k) if k.endswith('_l0') else ('policy.policy.' + k)
            np.testing.assert_allclose(sd[key].cpu().numpy(), arr, rtol=1e-5, atol=1e-5, err_msg=k)


def test_unsupported_combinations_fail_loudly():
    from host_vecenv import HostMultiHead
    from pufferlib_amd import clean_pufferl
    from test_gpu_ppo import _config
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    with pytest.raises(NotImplementedError):
        # (two heads of 9 = 18 logits run in the GEMM path now; a head of more than 15 choices does not fit the 4-bit action packing)
        clean_pufferl.create(_config(16, 16, 128, 8, 1, 10 ** 5, hp), HostMultiHead(16, [16, 2]), _policy(HostMultiHead(16, [16, 2])))
