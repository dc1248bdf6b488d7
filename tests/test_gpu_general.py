"""The width-general policy path (pufferlib_amd/general.py, csrc/general.hip) against the torch-fp32 oracle (oracle/ppo_torch.py),
north_star's 1e-5:

  * the row-wise kernels on their own — sample_logits / given-action scoring / PPO loss with its gradient, up to 63 logits and
    MultiDiscrete heads; the LSTM cell and its back-propagation;
  * policy(obs, action=...) — frameworks.cleanrl.Policy / RecurrentPolicy in training mode (cleanrl.py:60-66,87-93) — for the
    128-wide policies whose training runs in the fused kernels and for the wide ones;
  * pufferlib.models.Default(hidden_size=64/256/512), wide observation rows, 40 logits; LSTMWrapper(256, 256); and the recurrent
    NatureCNN (environments/atari/torch.py:4-6) through create / evaluate / train: rollout values and log-probabilities, then
    losses and post-update weights of a whole update (every epoch and minibatch) against the oracle trainer on the same experience.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)
HP = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]


def _heads_word(nvec):
    return sum(n << (4 * h) for h, n in enumerate(nvec))


@pytest.mark.parametrize('nvec', [[4], [17], [63], [3, 2, 5, 4], [15, 15, 15, 15]])
def test_row_kernels_sample_and_score_like_the_oracle(nvec):
    from pufferlib_amd import _lib
    from oracle import ppo_torch
    L = _lib.lib()
    A, rows = sum(nvec), 777
    NO = (A + 1 + 15) // 16 * 16
    multi = len(nvec) > 1
    heads = _heads_word(nvec) if multi else 0
    g = torch.Generator().manual_seed(A)
    out = torch.zeros(rows, NO)
    out[:, :A + 1] = torch.randn(rows, A + 1, generator=g) * 2
    noise = torch.empty(rows, A).exponential_(1, generator=g)
    cols = np.cumsum([0] + nvec)
    logits = [out[:, cols[h]:cols[h + 1]] for h in range(len(nvec))]
    oa, olp, oent = ppo_torch.sample_logits(logits if multi else logits[0], noise=noise)
    d = out.cuda()
    actions = torch.empty(rows, dtype=torch.int64, device='cuda')
    lp, ent, val = (torch.empty(rows, device='cuda') for _ in range(3))
    _lib.check(L.pfa_heads_rows_sample(_lib.ptr(d), NO, rows, A, heads, _lib.ptr(noise.cuda()), None, 0, _lib.ptr(actions), _lib.ptr(lp),
                                       _lib.ptr(ent), _lib.ptr(val), None), 'sample')
    got = actions.cpu()
    if multi:
        got = torch.stack([(got >> (4 * h)) & 15 for h in range(len(nvec))], dim=1)
    mism = int((got != oa).sum())
    assert mism <= 1, mism              # argmax(p / q): a near-tie may flip under a different summation order
    ok = (got == oa).all(dim=1) if multi else got == oa
    np.testing.assert_allclose(lp.cpu().numpy()[ok], olp.numpy()[ok], **TOL)
    np.testing.assert_allclose(ent.cpu().numpy(), oent.numpy(), **TOL)
    assert torch.equal(val.cpu(), out[:, A])
    # the same rows scored with GIVEN actions
    packed = actions.clone()
    lp2, ent2, val2 = (torch.empty(rows, device='cuda') for _ in range(3))
    _lib.check(L.pfa_heads_rows_eval(_lib.ptr(d), NO, rows, A, heads, _lib.ptr(packed), _lib.ptr(lp2), _lib.ptr(ent2), _lib.ptr(val2), None), 'eval')
    _, olp2, oent2 = ppo_torch.sample_logits(logits if multi else logits[0], action=got)
    np.testing.assert_allclose(lp2.cpu().numpy(), olp2.numpy(), **TOL)
    np.testing.assert_allclose(ent2.cpu().numpy(), oent2.numpy(), **TOL)


@pytest.mark.parametrize('nvec,time_major', [([5], False), ([40], True), ([3, 2, 5], True)])
def test_row_loss_kernel_matches_autograd(nvec, time_major):
    """PPO loss sums and d loss / d (head outputs) of one minibatch against torch autograd on the reference's loss (clean_pufferl.py:202-238)."""
    from pufferlib_amd import _lib
    from oracle import ppo_torch
    L = _lib.lib()
    A = sum(nvec)
    multi = len(nvec) > 1
    NO = (A + 1 + 15) // 16 * 16
    heads = _heads_word(nvec) if multi else 0
    N, T, Th, nmb = 24, 8, 4, 2
    B = N * T
    M = B // nmb
    R = M // Th
    g = torch.Generator().manual_seed(7 + A)
    hp = _lib.PpoHparams(HP[3], HP[5], HP[4], HP[7], 1, 1, nmb, Th)
    if multi:
        acts = torch.stack([torch.randint(0, n, (B,), generator=g) for n in nvec], dim=1)
        packed = sum(acts[:, h] << (4 * h) for h in range(len(nvec))).to(torch.int32)
    else:
        acts = torch.randint(0, A, (B,), generator=g)
        packed = acts.to(torch.int32)
    dev = 'cuda'
    bufs = dict(actions=packed.to(dev), logprobs=(-np.log(A) + 0.1 * torch.randn(B, generator=g)).to(dev), values=torch.randn(B, generator=g).to(dev),
                rewards=torch.zeros(B, device=dev), dones=torch.zeros(B, device=dev), advantages=torch.randn(B, generator=g).to(dev),
                returns=torch.randn(B, generator=g).to(dev))
    obs = torch.zeros(B, 16, device=dev)
    exp = _lib.Experience(obs.data_ptr(), bufs['actions'].data_ptr(), bufs['logprobs'].data_ptr(), bufs['values'].data_ptr(), bufs['rewards'].data_ptr(),
                          bufs['dones'].data_ptr(), bufs['advantages'].data_ptr(), bufs['returns'].data_ptr(), T)
    mb = 1
    # flat env-major rows of minibatch mb: segments {mb + k*nmb}
    k = torch.arange(R)
    idx = ((mb + k[:, None] * nmb) * Th + torch.arange(Th)[None, :]).reshape(-1)           # minibatch order (row k*Th + t)
    order = idx.view(R, Th).t().reshape(-1) if time_major else idx                          # row order of the chunk handed to the kernel
    out = torch.zeros(M, NO)
    out[:, :A + 1] = torch.randn(M, A + 1, generator=g)
    out_t = out.clone().requires_grad_(True)
    cols = np.cumsum([0] + nvec)
    logits = [out_t[:, cols[h]:cols[h + 1]] for h in range(len(nvec))]
    a_rows = acts[order]
    _, newlogprob, entropy = ppo_torch.sample_logits(logits if multi else logits[0], action=a_rows)
    adv_all = bufs['advantages'].cpu()[idx]
    adv = bufs['advantages'].cpu()[order]
    a_n = (adv - adv_all.mean()) / (adv_all.std() + 1e-8)
    logratio = newlogprob - bufs['logprobs'].cpu()[order]
    ratio = logratio.exp()
    pg = torch.max(-a_n * ratio, -a_n * torch.clamp(ratio, 1 - HP[3], 1 + HP[3])).mean()
    nv = out_t[:, A]
    ret, val = bufs['returns'].cpu()[order], bufs['values'].cpu()[order]
    vl = 0.5 * torch.max((nv - ret) ** 2, (val + torch.clamp(nv - val, -HP[5], HP[5]) - ret) ** 2).mean()
    loss = pg - HP[7] * entropy.mean() + HP[4] * vl
    loss.backward()
    s1, s2 = float(adv_all.double().sum()), float((adv_all.double() ** 2).sum())
    stats = torch.zeros(nmb, 2, dtype=torch.float64, device=dev)
    stats[mb, 0], stats[mb, 1] = s1, s2
    dout = torch.full((M, NO), 7.0, device=dev)
    pairs = torch.zeros(16, device=dev)
    ws = torch.zeros(L.pfa_heads_rows_loss_workspace_bytes(M), dtype=torch.uint8, device=dev)
    _lib.check(L.pfa_heads_rows_loss(_lib.ptr(out.to(dev)), NO, C.byref(exp), B, mb, 0, M, R if time_major else 0, A, heads, C.byref(hp), _lib.ptr(stats),
                                     M, _lib.ptr(dout), NO, NO, _lib.ptr(pairs), 0, _lib.ptr(ws), None), 'loss')
    np.testing.assert_allclose(dout.cpu().numpy()[:, :A + 1], out_t.grad.numpy()[:, :A + 1], rtol=1e-4, atol=1e-8)
    assert float(dout[:, A + 1:].abs().sum()) == 0.0
    sums = (pairs[0::2].double() + pairs[1::2].double()).cpu().numpy() / M
    np.testing.assert_allclose(sums[:3], [pg.item(), vl.item(), entropy.mean().item()], **TOL)


@pytest.mark.parametrize('H', [64, 512])
def test_lstm_cell_forward_backward_match_autograd(H):
    from pufferlib_amd import _lib
    L = _lib.lib()
    R = 37
    g = torch.Generator().manual_seed(H)
    G = torch.randn(R, 4 * H, generator=g)
    c0 = torch.randn(R, H, generator=g)
    Gt, c0t = G.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    i, f, gg, o = Gt.chunk(4, dim=1)
    c1 = torch.sigmoid(f) * c0t + torch.sigmoid(i) * torch.tanh(gg)
    h1 = torch.sigmoid(o) * torch.tanh(c1)
    dh, dc1 = torch.randn(R, H, generator=g), torch.randn(R, H, generator=g)
    (h1 * dh).sum().add((c1 * dc1).sum()).backward()
    d = lambda x: x.cuda().contiguous()   # noqa: E731
    Gd, c0d = d(G), d(c0)
    c1d, h1d = torch.empty(R, H, device='cuda'), torch.empty(R, H + 16, device='cuda')
    _lib.check(L.pfa_lstm_cell_forward(_lib.ptr(Gd), _lib.ptr(c0d), _lib.ptr(c1d), _lib.ptr(h1d), H + 16, None, 0, R, H, None), 'fwd')
    np.testing.assert_allclose(c1d.cpu().numpy(), c1.detach().numpy(), **TOL)
    np.testing.assert_allclose(h1d[:, :H].cpu().numpy(), h1.detach().numpy(), **TOL)
    dcd = d(dc1)
    dG = torch.empty(R, 4 * H, device='cuda')
    half = d(dh * 0.25)
    _lib.check(L.pfa_lstm_cell_backward(_lib.ptr(d(dh * 0.75)), H, _lib.ptr(half), H, _lib.ptr(dcd), _lib.ptr(Gd), _lib.ptr(c0d), _lib.ptr(c1d),
                                        _lib.ptr(dG), R, H, None), 'bwd')
    np.testing.assert_allclose(dG.cpu().numpy(), Gt.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dcd.cpu().numpy(), c0t.grad.numpy(), rtol=1e-4, atol=1e-6)


def _squared(n, d=3):
    from pufferlib_amd import vector
    return vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=d, num_targets=1), num_envs=n, backend=vector.Squared)


def _perturb(pol, seed=3, scale=0.05):
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(scale * torch.randn_like(p))


def _weights(pol, strip='policy.'):
    return {k[len(strip):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}


def test_training_mode_call_on_the_fused_mlp_policy():
    """policy(obs, action=atn) -> (action, logprob, entropy, value) for Default(128): cleanrl.py:60-66."""
    from pufferlib_amd import cleanrl, models
    from oracle import ppo_torch
    vec = _squared(8)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    _perturb(pol, scale=0.2)
    w = _weights(pol)
    obs = torch.randn(300, 7, 7)
    atn = torch.randint(0, 8, (300,))
    a, lp, ent, val = pol(obs.cuda(), action=atn.cuda())
    opol = ppo_torch.Policy(w)
    with torch.no_grad():
        logits, oval, _ = opol.forward(obs.reshape(300, -1))
        _, olp, oent = ppo_torch.sample_logits(logits, action=atn)
    assert torch.equal(a.cpu(), atn)
    np.testing.assert_allclose(lp.cpu().numpy(), olp.numpy(), **TOL)
    np.testing.assert_allclose(ent.cpu().numpy(), oent.numpy(), **TOL)
    np.testing.assert_allclose(val.cpu().numpy(), oval.numpy(), **TOL)
    # rollout mode still runs the fused kernel and the same parameters
    a2, lp2, _, val2 = pol(obs.cuda(), noise=torch.ones(300, 8))
    np.testing.assert_allclose(val2.cpu().numpy(), oval.numpy(), **TOL)


def test_training_mode_call_on_the_fused_recurrent_policy_with_a_time_axis():
    """RecurrentPolicy(obs[B, TT, ...], state, action=atn): encoder -> LSTM over TT steps from `state` -> heads (cleanrl.py:87-93,
    models.py:84-111); returns the final state."""
    from pufferlib_amd import cleanrl, models
    from oracle import ppo_torch
    vec = _squared(8)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    _perturb(pol, scale=0.2)
    sd = pol.state_dict()
    w = {k[len('policy.policy.'):]: v.cpu().numpy() for k, v in sd.items() if k.startswith('policy.policy.')}
    w.update({k[len('policy.recurrent.'):]: v.cpu().numpy() for k, v in sd.items() if k.startswith('policy.recurrent.')})
    B, TT = 20, 6
    obs = torch.randn(B, TT, 7, 7)
    atn = torch.randint(0, 8, (B, TT))
    h0, c0 = torch.randn(1, B, 128) * 0.3, torch.randn(1, B, 128) * 0.3
    a, lp, ent, val, (h1, c1) = pol(obs.cuda(), state=(h0.cuda(), c0.cuda()), action=atn.cuda())
    opol = ppo_torch.Policy(w, recurrent=True)
    with torch.no_grad():
        logits, oval, (oh, oc) = opol.forward(obs.reshape(B, TT, -1), (h0, c0))
        _, olp, oent = ppo_torch.sample_logits(logits, action=atn.reshape(-1))
    np.testing.assert_allclose(lp.cpu().numpy(), olp.numpy(), **TOL)
    np.testing.assert_allclose(ent.cpu().numpy(), oent.numpy(), **TOL)
    np.testing.assert_allclose(val.cpu().numpy(), oval.numpy(), **TOL)
    np.testing.assert_allclose(h1.cpu().numpy(), oh.numpy(), **TOL)
    np.testing.assert_allclose(c1.cpu().numpy(), oc.numpy(), **TOL)


def _inject(tr, exp, n, horizon, obs_dim, frames=False):
    sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(n * horizon, *x.shape[1:]).cpu().numpy()  # noqa: E731
    tr.obs = torch.as_tensor(sm(exp.obs)[:, :obs_dim].copy()).float() if not frames else torch.as_tensor(sm(exp.obs).copy())
    tr.actions = sm(exp.actions).astype(np.int64)
    tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (exp.logprobs, exp.rewards, exp.dones, exp.values))


def _check_update(data, pol, opol, tr, strip):
    from pufferlib_amd import clean_pufferl
    torch.set_num_threads(8)
    Lo = tr.train()
    clean_pufferl.train(data)
    L = data.losses
    np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                               [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')], **TOL)
    sd = pol.state_dict()
    for k, arr in opol.state_arrays().items():
        np.testing.assert_allclose(sd[strip(k)].cpu().numpy(), arr, err_msg=k, **TOL)


@pytest.mark.parametrize('hidden,d', [(64, 3), (256, 3), (512, 3), (512, 5), (256, 1), (64, 2)])
def test_wide_default_policy_rollout_and_update_vs_oracle(hidden, d, matrix_products):
    """Default(hidden_size != 128) on the device Squared vecenv behind create / evaluate / train: rows of up to 64 floats (d <= 3: 9 / 25 / 49
    columns) run the tile-kernel rollout and the fused gradient kernel of csrc/ppo_wide.hip (hidden split over the four waves), d = 5 (121
    columns, 128-float rows) the GEMM-path engine."""
    from pufferlib_amd import clean_pufferl, cleanrl, general, models
    from test_gpu_ppo import _config
    from oracle import c_oracle, ppo_torch
    n, horizon, nmb, bptt = 64, 32, 2, 8
    B = n * horizon
    vec = _squared(n, d)
    pol = cleanrl.Policy(models.Default(vec.driver_env, hidden_size=hidden))
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.gen_engine is not None
    w0 = _weights(pol)
    D = vec.obs_dim
    noise = torch.empty(horizon, n, 8).exponential_(1)
    data.noise = noise
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, d, 1), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=HP[0], gamma=HP[1], gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5],
                           max_grad_norm=HP[6], ent_coef=HP[7], total_timesteps=B * 10, seed=1)
    tr.evaluate(noise.numpy())                      # the oracle's own rollout under the same noise: the env is bit-exact given equal actions
    sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(n * horizon, *x.shape[1:]).cpu().numpy()  # noqa: E731
    assert np.array_equal(sm(exp.actions).astype(np.int64), tr.actions)
    assert np.array_equal(sm(exp.obs)[:, :D], tr.obs.numpy())
    np.testing.assert_allclose(sm(exp.logprobs), tr.logprobs, **TOL)
    np.testing.assert_allclose(sm(exp.values), tr.values, **TOL)
    _check_update(data, pol, opol, tr, lambda k: 'policy.' + k)


@pytest.mark.parametrize('hidden', [64, 256])
def test_wide_default_policy_with_a_minibatch_that_is_no_multiple_of_16(hidden, matrix_products):
    """24 envs x 5 steps, 2 minibatches of 60 rows: pfa_ppo_wide_grad tiles minibatches in 16-row blocks, so the engine must leave the
    fused wide kernel off and train through the GEMM path (which takes any minibatch size, like the 128-wide fused kernel)."""
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    from oracle import c_oracle, ppo_torch
    n, horizon, nmb, bptt, d = 24, 5, 2, 5, 3
    B = n * horizon
    vec = _squared(n, d)
    pol = cleanrl.Policy(models.Default(vec.driver_env, hidden_size=hidden))
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP), vec, pol)
    assert data.gen_engine is not None and data.gen_engine.wide_ws is None
    w0 = _weights(pol)
    noise = torch.empty(horizon, n, 8).exponential_(1)
    data.noise = noise
    clean_pufferl.evaluate(data)
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, d, 1), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=HP[0], gamma=HP[1], gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5],
                           max_grad_norm=HP[6], ent_coef=HP[7], total_timesteps=B * 10, seed=1)
    tr.evaluate(noise.numpy())
    _check_update(data, pol, opol, tr, lambda k: 'policy.' + k)


def test_wide_observations_and_many_logits_on_a_host_vecenv():
    """Rows of 300 floats (beyond the fused kernels' 128) and 40 logits (beyond their 15) through the host-vecenv path."""
    from host_vecenv import HostMultiHead
    from pufferlib_amd import clean_pufferl, cleanrl, general, models
    from test_gpu_ppo import _config
    from oracle import ppo_torch
    n, horizon, nmb, bptt = 16, 16, 2, 4
    B = n * horizon
    from pufferlib_amd import spaces

    class HostDiscrete(HostMultiHead):          # the same env behind a Discrete(40) action space
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.single_action_space = spaces.Discrete(self.nvec[0])
    vec = HostDiscrete(n, [40], obs_dim=300)
    pol = cleanrl.Policy(models.Default(vec.driver_env, hidden_size=128))
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP, env='host'), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.flat_params.obs_stride == 304
    w0 = _weights(pol)
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy(w0)

    class _V:
        num_envs = n
        observations = np.zeros((n, 300), np.float32)

        def async_reset(self, seed):
            pass
    tr = ppo_torch.Trainer(opol, _V(), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2, learning_rate=HP[0], gamma=HP[1],
                           gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5], max_grad_norm=HP[6], ent_coef=HP[7],
                           total_timesteps=B * 10, seed=1)
    _inject(tr, exp, n, horizon, 300)
    with torch.no_grad():                            # the rollout's own numbers on the stored observations
        logits, oval, _ = opol.forward(tr.obs)
        _, olp, _ = ppo_torch.sample_logits(logits, action=torch.as_tensor(tr.actions))
    np.testing.assert_allclose(tr.logprobs, olp.numpy(), **TOL)
    np.testing.assert_allclose(tr.values, oval.flatten().numpy(), **TOL)
    tr.global_step = data.global_step
    _check_update(data, pol, opol, tr, lambda k: 'policy.' + k)


def test_wide_lstm_policy_rollout_and_update_vs_oracle(matrix_products):
    """LSTMWrapper(256, 256) over Default(256): state carried through the rollout and across the minibatches of an epoch."""
    from pufferlib_amd import clean_pufferl, cleanrl, general, models
    from test_gpu_ppo import _config
    from oracle import c_oracle, ppo_torch
    n, horizon, nmb, bptt, H = 32, 16, 2, 8, 256
    B = n * horizon
    vec = _squared(n)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env, hidden_size=H), input_size=H, hidden_size=H))
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.gen_engine.net.lstm == (H, H)
    sd = pol.state_dict()
    w0 = {k[len('policy.policy.'):]: v.cpu().numpy().copy() for k, v in sd.items() if k.startswith('policy.policy.')}
    w0.update({k[len('policy.recurrent.'):]: v.cpu().numpy().copy() for k, v in sd.items() if k.startswith('policy.recurrent.')})
    noise = torch.empty(horizon, n, 8).exponential_(1)
    data.noise = noise
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy(w0, recurrent=True)
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, 3, 1), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=HP[0], gamma=HP[1], gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5],
                           max_grad_norm=HP[6], ent_coef=HP[7], total_timesteps=B * 10, seed=1)
    tr.evaluate(noise.numpy())
    sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(n * horizon, *x.shape[1:]).cpu().numpy()  # noqa: E731
    assert np.array_equal(sm(exp.actions).astype(np.int64), tr.actions)
    np.testing.assert_allclose(sm(exp.logprobs), tr.logprobs, **TOL)
    np.testing.assert_allclose(sm(exp.values), tr.values, **TOL)
    np.testing.assert_allclose(data.gen_engine.lstm_h.cpu().numpy(), tr.lstm_h.numpy(), **TOL)

    def name(k):
        return 'policy.recurrent.' + k if k in ppo_torch.Policy.LSTM_NAMES else 'policy.policy.' + k
    _check_update(data, pol, opol, tr, name)


def test_recurrent_nature_cnn_rollout_and_update_vs_oracle(matrix_products):
    """environments/atari/torch.py:4-6: LSTMWrapper(512, 512) over the NatureCNN on the device frame vecenv."""
    from pufferlib_amd import clean_pufferl, cleanrl, general, models, vector
    from test_gpu_ppo import _config
    from oracle import ppo_torch
    n, horizon, nmb, bptt = 4, 8, 2, 4
    B = n * horizon
    vec = vector.make(vector.make_frames, env_kwargs=dict(episode_length=5), num_envs=n, backend=vector.Frames)
    torch.manual_seed(0)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Convolutional(vec.driver_env, framestack=4), input_size=512, hidden_size=512))
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP, env='frames'), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.gen_engine.net.kind == 'cnn' and data.gen_engine.net.lstm == (512, 512)
    sd = pol.state_dict()
    w0 = {k[len('policy.policy.'):]: v.cpu().numpy().copy() for k, v in sd.items() if k.startswith('policy.policy.')}
    w0.update({k[len('policy.recurrent.'):]: v.cpu().numpy().copy() for k, v in sd.items() if k.startswith('policy.recurrent.')})
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.RecurrentConvPolicy(w0)

    class _V:
        num_envs = n
        observations = np.zeros((n, 4 * 84 * 84), np.uint8)

        def async_reset(self, seed):
            pass
    tr = ppo_torch.Trainer(opol, _V(), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2, learning_rate=HP[0], gamma=HP[1],
                           gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5], max_grad_norm=HP[6], ent_coef=HP[7],
                           total_timesteps=B * 10, seed=1)
    _inject(tr, exp, n, horizon, 4 * 84 * 84, frames=True)
    tr.obs = tr.obs.float()
    # the rollout: every env's state carried step to step from zeros
    with torch.no_grad():
        frames = tr.obs.reshape(horizon, n, -1)
        state = None
        vals, lps = [], []
        for t in range(horizon):
            logits, v, state = opol.forward(frames[t], state)
            _, lp, _ = ppo_torch.sample_logits(logits, action=torch.as_tensor(tr.actions[t * n:(t + 1) * n]))
            vals.append(v.flatten())
            lps.append(lp)
    np.testing.assert_allclose(tr.values, torch.cat(vals).numpy(), **TOL)
    np.testing.assert_allclose(tr.logprobs, torch.cat(lps).numpy(), **TOL)
    tr.global_step = data.global_step

    def name(k):
        return 'policy.recurrent.' + k if k in ppo_torch.Policy.LSTM_NAMES else 'policy.policy.' + k
    _check_update(data, pol, opol, tr, name)


def test_recurrent_nature_cnn_replays_the_reference_golden(golden_dir, matrix_products):
    """tests/golden/ppo_cnn_lstm.npz: the unmodified reference's cleanrl.RecurrentPolicy(LSTMWrapper(Convolutional, 512, 512))
    (environments/atari/torch.py:4-6) through its create / evaluate / train on the frame stub. Rollout mode: policy(obs, state) on
    its frames with its multinomial's exponential draws -> its actions bit for bit, log-probabilities, values, and the LSTM state it
    ends the rollout with. Training mode: its experience through the update -> losses and every updated tensor."""
    import cnn_golden
    from pufferlib_amd import clean_pufferl, cleanrl, general, models, vector
    from test_gpu_ppo import _config
    g = np.load(os.path.join(golden_dir, 'ppo_cnn_lstm.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    hp = [float(x) for x in g['hparams']]
    B = n * horizon
    vec = vector.make(vector.make_frames, num_envs=n, backend=vector.Frames)
    conv = cnn_golden.container()
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, conv, input_size=512, hidden_size=512))
    start = cnn_golden.recurrent_start_weights(conv)
    with torch.no_grad():
        for k, v in pol.state_dict().items():
            bare = k.split('.', 2)[2]
            assert cnn_golden.golden_key(bare) == k
            v.copy_(torch.from_numpy(start[bare]))
            assert np.array_equal(cnn_golden.digest(v.numpy()), g['w0.' + k]), k
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, hp, seed=1, env='frames'), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.gen_engine.net.kind == 'cnn' and data.gen_engine.net.lstm == (512, 512)
    frame_ids, noise = g['it0.frame_ids'], g['it0.noise']
    frames = np.stack([[cnn_golden.cnn_frame(frame_ids[t, e]) for e in range(n)] for t in range(horizon)])   # (T, N, 4, 84, 84)
    dev = vec.device
    state = None
    for t in range(horizon):
        a, lp, ent, val, state = pol(torch.as_tensor(frames[t]).to(dev), state, noise=torch.as_tensor(noise[t]))
        assert np.array_equal(a.cpu().numpy().reshape(-1), g['it0.actions'][t * n:(t + 1) * n]), t
        np.testing.assert_allclose(lp.cpu().numpy(), g['it0.logprobs'][t * n:(t + 1) * n], **TOL)
        np.testing.assert_allclose(val.cpu().numpy().reshape(-1), g['it0.values'][t * n:(t + 1) * n], **TOL)
    np.testing.assert_allclose(state[0].cpu().numpy(), g['it0.lstm_h'], **TOL)
    np.testing.assert_allclose(state[1].cpu().numpy(), g['it0.lstm_c'], **TOL)
    e = data.experience
    em = lambda x: torch.as_tensor(np.ascontiguousarray(np.asarray(x).reshape(horizon, n, *np.asarray(x).shape[1:]).swapaxes(0, 1))  # noqa: E731
                                   .reshape(B, *np.asarray(x).shape[1:])).to(dev)
    e.obs.copy_(em(frames.reshape(B, -1)))
    e.actions.copy_(em(g['it0.actions'].astype(np.int32)).view_as(e.actions))
    for dst, key in ((e.logprobs, 'logprobs'), (e.values, 'values'), (e.rewards, 'rewards'), (e.dones, 'dones')):
        dst.copy_(em(g['it0.' + key].astype(np.float32)))
    e.ptr = B
    data.global_step = int(g['it0.global_step'])
    clean_pufferl.train(data)
    L = data.losses
    got = [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac, L.explained_variance]
    np.testing.assert_allclose(got, g['it0.losses'], **TOL)
    for k, v in pol.state_dict().items():
        got, want = cnn_golden.digest(v.cpu().numpy()), g['it0.w.' + k]
        np.testing.assert_allclose(got[2:], want[2:], err_msg=k, **TOL)                 # the sampled elements
        np.testing.assert_allclose(got[:2], want[:2], rtol=0, atol=1e-5 * max(1.0, want[1]), err_msg=k + ' (sums)')


def test_hidden_256_replays_the_reference_golden(golden_dir, matrix_products):
    """tests/golden/ppo_mlp_h256.npz: the unmodified reference's create / evaluate / train with models.Default(hidden_size=256) on
    Serial(Squared) — its multinomial noise in, its actions bit for bit, experience, advantages, losses, weights and Adam moments out."""
    from pufferlib_amd import clean_pufferl, cleanrl, general, models
    from test_gpu_ppo import _config, _load_weights, _step_major
    g = np.load(os.path.join(golden_dir, 'ppo_mlp_h256.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    vec = _squared(n)
    pol = cleanrl.Policy(models.Default(vec.driver_env, hidden_size=256))
    _load_weights(pol, g, 'w0.')
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, [float(x) for x in g['hparams']]), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams)
    exp = data.experience
    for it in range(iters):
        assert abs(data.optimizer.param_groups[0]['lr'] - float(g[f'it{it}.lr_used'])) < 1e-12
        data.noise = torch.as_tensor(g[f'it{it}.noise'])
        stats, _ = clean_pufferl.evaluate(data)
        assert np.array_equal(_step_major(exp.actions, n, horizon), g[f'it{it}.actions'].astype(np.int32))
        assert np.array_equal(_step_major(exp.obs, n, horizon)[:, :49], g[f'it{it}.obs'].astype(np.float32))
        assert np.array_equal(_step_major(exp.rewards, n, horizon), g[f'it{it}.rewards'])
        assert np.array_equal(_step_major(exp.dones, n, horizon), g[f'it{it}.dones'])
        np.testing.assert_allclose(_step_major(exp.logprobs, n, horizon), g[f'it{it}.logprobs'], **TOL)
        np.testing.assert_allclose(_step_major(exp.values, n, horizon), g[f'it{it}.values'], **TOL)
        assert data.global_step == int(g[f'it{it}.global_step'])
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']], g[f'it{it}.stats'], rtol=1e-9)
        clean_pufferl.train(data)
        for m in range(exp.num_minibatches):
            idx = exp.minibatch_rows_index(m)
            np.testing.assert_allclose(exp.advantages[idx].cpu().numpy(), g[f'it{it}.advantages'][m], **TOL)
        L = data.losses
        got = [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac, L.explained_variance]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        m_, v_ = data.flat_params.split(data.optimizer.exp_avg), data.flat_params.split(data.optimizer.exp_avg_sq)
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.{k}'], err_msg=k, **TOL)
            short = k[len('policy.'):]
            np.testing.assert_allclose(m_[short].cpu().numpy(), g[f'it{it}.m.{k}'], rtol=1e-4, atol=1e-6, err_msg=k)
            np.testing.assert_allclose(v_[short].cpu().numpy(), g[f'it{it}.v.{k}'], rtol=1e-4, atol=1e-8, err_msg=k)
        assert abs(data.optimizer.param_groups[0]['lr'] - float(g[f'it{it}.lr_next'])) < 1e-12


def _general_run(kind, recurrent, order='natural', n=32, horizon=16, iters=2, H=256):
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from host_vecenv import HostSquared
    from test_gpu_ppo import _config
    torch.manual_seed(3)
    vec = _squared(n) if kind == 'device' else HostSquared(n, order=order)
    base = models.Default(vec.driver_env, hidden_size=H)
    pol = (cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base, input_size=H, hidden_size=H)) if recurrent else cleanrl.Policy(base))
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 8, 2, n * horizon * 8, [2.5e-3] + HP[1:], seed=11), vec, pol)
    out = []
    for _ in range(iters):
        clean_pufferl.evaluate(data)
        e = data.experience
        snap = [x.clone() for x in (e.obs, e.actions, e.logprobs, e.values, e.rewards, e.dones)]
        clean_pufferl.train(data)
        out.append((snap, data.flat_params.flat.clone(), data.global_step))
    return out, data, pol


@pytest.mark.parametrize('recurrent', [False, True])
def test_general_path_host_vecenv_equals_device_vecenv(recurrent):
    """The same wide policy behind the device Squared vecenv and behind a host vecenv handing out shuffled batches (the LSTM state
    rows follow the env ids): identical experience and identical parameters after two iterations."""
    dev, _, _ = _general_run('device', recurrent)
    host, _, _ = _general_run('host', recurrent, order='shuffled')
    for (sd, wd, gd), (sh, wh, gh) in zip(dev, host):
        for x, y in zip(sd, sh):
            assert torch.equal(x, y)
        assert gd == gh and torch.equal(wd, wh)


def test_general_policy_checkpoint_round_trip(tmp_path):
    from pufferlib_amd import clean_pufferl
    out, data, pol = _general_run('device', True, iters=1)
    data.config.data_dir, data.config.exp_id = str(tmp_path), 'wide'
    path = clean_pufferl.save_checkpoint(data)
    want = {k: v.clone() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)
    clean_pufferl.train(data)
    after = data.flat_params.flat.clone()
    assert any(not torch.equal(want[k], v) for k, v in pol.state_dict().items())
    clean_pufferl.try_load_checkpoint(data)
    for k, v in pol.state_dict().items():
        assert torch.equal(want[k], v), k
    # the packed operand copies follow the loaded weights: the next rollout is not the one the overwritten weights would give
    loaded = torch.load(path, weights_only=False)
    obs = torch.randn(5, 7, 7)
    a0 = loaded(obs.cuda(), noise=torch.ones(5, 8))
    a1 = pol(obs.cuda(), noise=torch.ones(5, 8))
    for x, y in zip(a0[:4], a1[:4]):
        assert torch.equal(x, y)
    assert not torch.equal(after, data.flat_params.flat)


def test_two_multidiscrete_heads_of_nine_run_end_to_end():
    """18 logits in two heads: more than the fused kernels' 15, inside the 4-bit action packing — the GEMM path with per-head
    log-softmax / sampling / entropy, against the oracle's list branch of sample_logits (cleanrl.py:31-44)."""
    from host_vecenv import HostMultiHead
    from pufferlib_amd import clean_pufferl, cleanrl, general, models
    from test_gpu_ppo import _config
    from oracle import ppo_torch
    n, horizon, nmb, bptt = 16, 16, 2, 4
    B = n * horizon
    vec = HostMultiHead(n, [9, 9], obs_dim=20)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, HP, env='host'), vec, pol)
    assert isinstance(data.flat_params, general.GeneralParams) and data.flat_params.nvec == [9, 9]
    w0 = _weights(pol)
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy(w0)

    class _V:
        num_envs = n
        observations = np.zeros((n, 20), np.float32)

        def async_reset(self, seed):
            pass
    tr = ppo_torch.Trainer(opol, _V(), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2, learning_rate=HP[0], gamma=HP[1],
                           gae_lambda=HP[2], clip_coef=HP[3], vf_coef=HP[4], vf_clip_coef=HP[5], max_grad_norm=HP[6], ent_coef=HP[7],
                           total_timesteps=B * 10, seed=1)
    _inject(tr, exp, n, horizon, 20)
    packed = tr.actions
    tr.actions = np.stack([packed & 15, (packed >> 4) & 15], axis=1)       # the kernels' nibble packing -> [rows, heads]
    assert tr.actions.max() < 9
    with torch.no_grad():
        logits, oval, _ = opol.forward(tr.obs)
        _, olp, _ = ppo_torch.sample_logits(logits, action=torch.as_tensor(tr.actions))
    np.testing.assert_allclose(tr.logprobs, olp.numpy(), **TOL)
    tr.global_step = data.global_step
    _check_update(data, pol, opol, tr, lambda k: 'policy.' + k)


@pytest.mark.parametrize('hidden,nt', [(64, 1), (256, 1), (512, 1), (256, 2)])
def test_wide_fused_rollout_equals_stepwise_protocol_bit_for_bit(hidden, nt, monkeypatch):
    """Default(hidden 64 / 256 / 512) on vector.Squared: evaluate()'s ONE persistent kernel (csrc/rollout.hip templated on the hidden
    tiles per wave, W1 fragments of that width in registers) == T x {recv, policy(obs), store, send} through the public protocol,
    bit for bit — policy(obs) runs the same tile code as a launch of its own (pfa_mlp_view_forward_sample).  And the GEMM path the
    same policy took before (general.USE_TILE_VIEW = False: igemm rows + head kernels per step) agrees within the fp32 tolerance."""
    from pufferlib_amd import clean_pufferl, cleanrl, general, models, vector
    from test_gpu_ppo import _config, _t
    n, horizon = 48, 20          # not a multiple of 16 envs: the masked tail tile
    def make():
        vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=3, num_targets=nt), num_envs=n, backend=vector.Squared)
        torch.manual_seed(7)
        return vec, cleanrl.Policy(models.Default(vec.driver_env, hidden_size=hidden))
    vec, pol = make()
    _perturb(pol)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 4, HP, seed=5), vec, pol)
    assert data.gen_engine is not None and data.gen_engine.mlp_view is not None
    clean_pufferl.evaluate(data)
    exp = data.experience

    vec2, pol2 = make()
    pol2.load_state_dict(pol.state_dict())
    pol2.noise_seed = 5
    vec2.async_reset(5)
    obs_l, act_l, lp_l, val_l, rew_l, done_l = [], [], [], [], [], []
    for t in range(horizon):
        o, r, d, tr_, info, ids, mask = vec2.recv()
        a, lp, ent, val = pol2(o)
        obs_l.append(o.reshape(n, -1).clone()); rew_l.append(r.clone()); done_l.append(d.clone().float())
        act_l.append(a.clone()); lp_l.append(lp.clone()); val_l.append(val.flatten().clone())
        vec2.send(a)
    assert general.tile_view(pol2.flat_params) is not None
    assert torch.equal(_t(exp.obs, n, horizon)[:, :49], torch.cat(obs_l))
    assert torch.equal(_t(exp.actions, n, horizon).long(), torch.cat(act_l))
    assert torch.equal(_t(exp.logprobs, n, horizon), torch.cat(lp_l))
    assert torch.equal(_t(exp.values, n, horizon), torch.cat(val_l))
    assert torch.equal(_t(exp.rewards, n, horizon), torch.cat(rew_l))
    assert torch.equal(_t(exp.dones, n, horizon), torch.cat(done_l))
    assert torch.equal(vec.observations, vec2.observations)
    assert torch.equal(vec.rewards, vec2.rewards) and torch.equal(vec.terminals, vec2.terminals)

    # the GEMM path on the recorded observations, under the noise the fused rollout drew
    monkeypatch.setattr(general, 'USE_TILE_VIEW', False)
    vec3, pol3 = make()
    pol3.load_state_dict(pol.state_dict())
    obs_all = torch.cat(obs_l).cuda()
    import ctypes as C
    from pufferlib_amd import _lib
    noise = torch.empty(n * horizon, 8, device='cuda')
    key, L = _lib.NoiseKey(5, 0), _lib.lib()
    _lib.check(L.pfa_philox_exp_noise(_lib.ptr(noise), horizon, n, 8, C.byref(key), 0, _lib.stream_handle()), 'noise')
    a3, lp3, _, val3 = pol3(obs_all.view(n * horizon, 7, 7), noise=noise)
    assert general.tile_view(pol3.flat_params) is None
    np.testing.assert_allclose(lp3.cpu().numpy(), torch.cat(lp_l).cpu().numpy(), **TOL)
    np.testing.assert_allclose(val3.flatten().cpu().numpy(), torch.cat(val_l).cpu().numpy(), **TOL)
    assert float((a3 != torch.cat(act_l)).float().mean()) < 0.01          # (near-ties of p/q may flip between the two summation orders)
