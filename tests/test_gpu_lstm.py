"""Recurrent policy path (LSTMWrapper) vs the golden outputs of the unmodified reference (tests/golden/ppo_lstm.npz):
rollout with the recorded multinomial noise, LSTM state carry, BPTT update, two full iterations.  Plus cell kernels vs
torch autograd.  Tolerances: 1e-5 on activations/advantages, a little looser on weights after 2 x 8 optimizer steps of
an LSTM (different but fp32-exact summation orders in the BLAS)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-5, atol=1e-5)    # north_star: within 1e-5 fp32


def _config(n, horizon, mbs, bptt, epochs, total, hp, **over):
    from pufferlib_amd import namespace
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = hp
    cfg = dict(env='squared', seed=1, torch_deterministic=True, cpu_offload=False, device='cuda', total_timesteps=total,
               learning_rate=lr, anneal_lr=True, gamma=gamma, gae_lambda=lam, update_epochs=epochs, norm_adv=True,
               clip_coef=clip, clip_vloss=True, vf_coef=vf_coef, vf_clip_coef=vf_clip, max_grad_norm=mgn, ent_coef=ent,
               target_kl=None, batch_size=n * horizon, minibatch_size=mbs, bptt_horizon=bptt, compile=False,
               checkpoint_interval=0, data_dir='/tmp/pfa_experiments', exp_id='test')
    cfg.update(over)
    return namespace(**cfg)


def _make(n):
    from pufferlib_amd import vector, models, cleanrl
    vec = vector.make(vector.make_squared, num_envs=n, backend=vector.Squared)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    return vec, pol


def _step_major(x, n, t):
    return x.view(n, t, *x.shape[1:]).transpose(0, 1).reshape(n * t, *x.shape[1:]).cpu().numpy()


def test_create_evaluate_train_replays_golden_lstm(golden_dir):
    from pufferlib_amd import clean_pufferl
    g = np.load(os.path.join(golden_dir, 'ppo_lstm.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    vec, pol = _make(n)
    pol.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith('w0.')})
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, [float(x) for x in g['hparams']]), vec, pol)
    exp = data.experience
    assert exp.lstm_h.shape == (1, n, 128)
    for it in range(iters):
        data.noise = torch.as_tensor(g[f'it{it}.noise'])
        stats, _ = clean_pufferl.evaluate(data)
        assert np.array_equal(_step_major(exp.actions, n, horizon), g[f'it{it}.actions'].astype(np.int32)), 'actions'
        assert np.array_equal(_step_major(exp.obs, n, horizon)[:, :49], g[f'it{it}.obs'].astype(np.float32))
        assert np.array_equal(_step_major(exp.rewards, n, horizon), g[f'it{it}.rewards'])
        np.testing.assert_allclose(_step_major(exp.logprobs, n, horizon), g[f'it{it}.logprobs'], **TOL)
        np.testing.assert_allclose(_step_major(exp.values, n, horizon), g[f'it{it}.values'], **TOL)
        np.testing.assert_allclose(exp.lstm_h.cpu().numpy(), g[f'it{it}.lstm_h'], **TOL)      # state after the rollout,
        np.testing.assert_allclose(exp.lstm_c.cpu().numpy(), g[f'it{it}.lstm_c'], **TOL)      # never reset on done
        assert data.global_step == int(g[f'it{it}.global_step'])
        clean_pufferl.train(data)
        for m in range(exp.num_minibatches):
            idx = exp.minibatch_rows_index(m)
            np.testing.assert_allclose(exp.advantages[idx].cpu().numpy(), g[f'it{it}.advantages'][m], rtol=1e-5, atol=1e-5)
        L = data.losses
        got = [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac, L.explained_variance]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.{k}'], rtol=1e-5, atol=1e-5, err_msg=k)
        m_ = data.flat_params.split(data.optimizer.exp_avg)
        for k, v in m_.items():
            key = 'policy.' + k if k.startswith('recurrent.') else 'policy.policy.' + k
            np.testing.assert_allclose(v.cpu().numpy(), g[f'it{it}.m.{key}'], rtol=2e-4, atol=1e-6, err_msg=k)


def test_recurrent_policy_protocol_step_matches_fused_engine():
    """policy(obs, state) through the public wrapper == the engine's rollout step (same kernels, same noise stream)."""
    from pufferlib_amd import clean_pufferl
    n, horizon = 32, 8
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    vec, pol = _make(n)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 4, hp, seed=3), vec, pol)
    clean_pufferl.evaluate(data)
    exp = data.experience
    vec2, pol2 = _make(n)
    pol2.load_state_dict(pol.state_dict())
    pol2.noise_seed = 3
    vec2.async_reset(3)
    state = None
    acts = []
    for t in range(horizon):
        o = vec2.recv()[0]
        a, lp, ent, val, state = pol2(o, state)
        acts.append(a.clone())
        vec2.send(a)
    assert torch.equal(exp.actions.view(n, horizon).t().reshape(-1).long(), torch.cat(acts))
    assert torch.equal(state[0], exp.lstm_h) and torch.equal(state[1], exp.lstm_c)


@pytest.mark.parametrize('mo,no,k,pad', [(512, 128, 131072, 0), (128, 64, 8192, 0), (16, 128, 4099, 0), (128, 128, 37, 0),
                                          (256, 256, 1000, 32), (16, 128, 131072, 0), (128, 32, 5000, 0), (128, 16, 777, 16),
                                          (128, 96, 2048, 0)])
def test_gemm_tn_matches_f64_contraction(mo, no, k, pad):
    """C = A^T B over the k rows (csrc/gemm.hip) vs the same contraction in f64; ragged k and padded row strides."""
    import ctypes as C
    from pufferlib_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(mo + no + k)
    a_full = torch.randn(k, mo + pad, device='cuda', generator=g)
    b_full = torch.randn(k, no + pad, device='cuda', generator=g)
    a, b = a_full[:, :mo], b_full[:, :no]
    out_full = torch.full((mo, no + 8), float('nan'), device='cuda')
    out = out_full[:, :no]
    nbytes = L.pfa_gemm_tn_workspace_bytes(mo, no, k)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    _lib.check(L.pfa_gemm_tn_f32(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), _lib.ptr(out), out.stride(0), mo, no, k,
                                 _lib.ptr(ws), _lib.stream_handle()), 'gemm_tn')
    want = (a.double().t() @ b.double())
    err = (out.double() - want).abs().max().item()
    assert err <= 2e-6 * (k ** 0.5) * 4 + 1e-5, err          # fp32 partial sums of ~N(0, k) terms
    assert torch.isnan(out_full[:, no:]).all()                 # nothing written outside the tile
    # deterministic: bit-identical on a second run
    out2 = torch.empty(mo, no, device='cuda')
    _lib.check(L.pfa_gemm_tn_f32(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), _lib.ptr(out2), no, mo, no, k, _lib.ptr(ws),
                                 _lib.stream_handle()), 'gemm_tn')
    assert torch.equal(out2, out.contiguous())


@pytest.mark.parametrize('mo,k,pad', [(512, 131072, 0), (512, 4099, 0), (128, 37, 16), (256, 10000, 0)])
def test_gemm_tn2_is_the_two_products_in_one_pass_over_the_shared_operand(mo, k, pad):
    """pfa_gemm_tn2_f32: c0 = a^T b0 and c1 = a^T b1 (the recurrent layer's dW_ih = dG^T xe and dW_hh = dG^T h_prev) from one pass
    over a — vs the f64 contractions; operands in separate buffers with their own strides, outputs into separate (strided) tensors,
    ragged k; deterministic."""
    from pufferlib_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(mo + k)
    a = torch.randn(k, mo + pad, device='cuda', generator=g)[:, :mo]
    b0 = torch.randn(k, 128, device='cuda', generator=g)
    b1 = torch.randn(k, 128 + 2 * pad, device='cuda', generator=g)[:, :128]
    o0_full, o1 = torch.full((mo, 136), float('nan'), device='cuda'), torch.full((mo, 128), float('nan'), device='cuda')
    o0 = o0_full[:, :128]
    nbytes = L.pfa_gemm_tn2_workspace_bytes(mo, k)
    assert nbytes > 0 and L.pfa_gemm_tn2_workspace_bytes(96, k) == 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')

    def run(c0, c1):
        _lib.check(L.pfa_gemm_tn2_f32(_lib.ptr(a), a.stride(0), _lib.ptr(b0), b0.stride(0), _lib.ptr(b1), b1.stride(0), _lib.ptr(c0), c0.stride(0),
                                      _lib.ptr(c1), c1.stride(0), mo, k, _lib.ptr(ws), _lib.stream_handle()), 'gemm_tn2')
    run(o0, o1)
    tol = 2e-6 * (k ** 0.5) * 4 + 1e-5
    assert (o0.double() - a.double().t() @ b0.double()).abs().max().item() <= tol
    assert (o1.double() - a.double().t() @ b1.double()).abs().max().item() <= tol
    assert torch.isnan(o0_full[:, 128:]).all()
    p0, p1 = torch.empty(mo, 128, device='cuda'), torch.empty(mo, 128, device='cuda')
    run(p0, p1)
    assert torch.equal(p0, o0.contiguous()) and torch.equal(p1, o1)


def test_gemm_tn_rejects_unsupported_shapes():
    from pufferlib_amd import _lib
    from pufferlib_amd.exceptions import ExtensionError
    L = _lib.lib()
    assert L.pfa_gemm_tn_workspace_bytes(24, 100, 64) == 0 and L.pfa_gemm_tn_workspace_bytes(64, 64, 64) == 0
    x = torch.zeros(64, 128, device='cuda')
    with pytest.raises(ExtensionError):
        _lib.check(L.pfa_gemm_tn_f32(_lib.ptr(x), 128, _lib.ptr(x), 128, _lib.ptr(x), 128, 24, 100, 64, _lib.ptr(x),
                                     _lib.stream_handle()), 'gemm_tn')


def test_fused_lstm_rollout_equals_stepwise_protocol_pieces():
    """Engine.rollout (one persistent kernel) vs Engine.rollout_stepwise (policy_step / store / send launches): identical
    experience, env state and LSTM state, bit for bit; 40 envs = a ragged last tile, two rollouts = state carry."""
    from pufferlib_amd import clean_pufferl
    n, horizon = 40, 16
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]

    def run(stepwise):
        torch.manual_seed(7)
        vec, pol = _make(n)
        data = clean_pufferl.create(_config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 4, hp, seed=5), vec, pol)
        eng = data.lstm_engine
        if stepwise:
            eng.rollout = eng.rollout_stepwise
        out = []
        for _ in range(2):
            clean_pufferl.evaluate(data)
            e = data.experience
            out.append([x.clone() for x in (e.obs, e.actions, e.logprobs, e.values, e.rewards, e.dones, e.lstm_h, e.lstm_c,
                                            vec.obs_buf, vec.rewards)])
        return out

    fused, step = run(False), run(True)
    for a, b in zip(fused, step):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_lstm_policy_step_matches_torch_modules():
    """pfa_lstm_policy_step vs nn.Linear / nn.LSTM / Categorical on the same weights (fp32, 1e-5)."""
    from pufferlib_amd import _lib, lstm as plstm
    torch.manual_seed(11)
    n = 50
    vec, pol = _make(n)
    with torch.no_grad():                      # the default init zeroes most biases: make every one of them count
        for name, p in pol.named_parameters():
            if name.endswith('bias') or 'bias_' in name:
                p.copy_(torch.randn_like(p) * 0.2)
    vec.async_reset(1)
    obs = vec.recv()[0]
    state = (torch.randn(1, n, 128, device='cuda') * 0.3, torch.randn(1, n, 128, device='cuda') * 0.3)
    noise = torch.empty(n, 8, device='cuda').exponential_()
    a, lp, ent, val, (h1, c1) = pol(obs, state, noise=noise)
    sd = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    enc_w, enc_b = sd['policy.policy.encoder.weight'], sd['policy.policy.encoder.bias']
    x = torch.relu(obs.reshape(n, -1)[:, :enc_w.shape[1]].float() @ enc_w.t() + enc_b)
    lstm = torch.nn.LSTM(128, 128).cuda()
    lstm.load_state_dict({k.split('recurrent.')[1]: v for k, v in sd.items() if 'recurrent.' in k})
    with torch.no_grad():
        y, (h_ref, c_ref) = lstm(x.unsqueeze(0), state)
    y = y[0]
    logits = y @ sd['policy.policy.decoder.weight'].t() + sd['policy.policy.decoder.bias']
    value = y @ sd['policy.policy.value_head.weight'].t() + sd['policy.policy.value_head.bias']
    np.testing.assert_allclose(h1.cpu().numpy(), h_ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c1.cpu().numpy(), c_ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(val.cpu().numpy(), value.cpu().numpy(), rtol=1e-5, atol=1e-5)
    logp = torch.log_softmax(logits, dim=1)
    want_a = torch.argmax(torch.softmax(logits, 1) / noise, dim=1)
    assert torch.equal(a, want_a)
    np.testing.assert_allclose(lp.cpu().numpy(), logp.gather(1, want_a[:, None])[:, 0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), -(logp * logp.exp()).sum(1).cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('R,Th', [(40, 5), (64, 16), (7, 3), (8192, 16)])    # last: one full-size minibatch (BASELINE configs[1])
def test_lstm_seq_forward_backward_match_torch_autograd(R, Th):
    """pfa_lstm_seq_forward / pfa_lstm_seq_backward vs nn.Linear + nn.LSTM under torch autograd: every kept activation, the
    state after Th steps, d loss/d gate pre-activations, d loss/d encoder pre-activations and the bias gradients.  Ragged
    row counts (not a multiple of the 32-row workgroup), nonzero biases, nonzero carried-in state."""
    from pufferlib_amd import _lib, lstm as plstm
    L = _lib.lib()
    torch.manual_seed(R * 100 + Th)
    vec, pol = _make(16)
    with torch.no_grad():
        for name, p in pol.named_parameters():
            if name.endswith('bias') or 'bias_' in name:
                p.copy_(torch.randn_like(p) * 0.2)
    fp = pol.adopt(64, 'cuda')
    dev = 'cuda'
    obs = torch.zeros(Th, R, 64, device=dev)
    obs[:, :, :49] = (torch.rand(Th, R, 49, device=dev) < 0.1).float() + 0.1 * torch.randn(Th, R, 49, device=dev)
    h0, c0 = 0.3 * torch.randn(R, 128, device=dev), 0.3 * torch.randn(R, 128, device=dev)
    dh = 0.1 * torch.randn(Th, R, 128, device=dev)                    # d loss / d h_t through the heads

    xe = torch.full((Th, R, 128), float('nan'), device=dev)
    gates = torch.full((Th, R, 512), float('nan'), device=dev)
    Hs = torch.full((Th + 1, R, 128), float('nan'), device=dev)
    Cs = torch.full((Th + 1, R, 128), float('nan'), device=dev)
    Hs[0], Cs[0] = h0, c0
    wpack = plstm.pack_gates(fp)
    st = _lib.stream_handle()
    _lib.check(L.pfa_lstm_seq_forward(_lib.ptr(obs), R, Th, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(wpack), _lib.ptr(xe),
                                      _lib.ptr(gates), _lib.ptr(Hs), _lib.ptr(Cs), 0, st), 'fwd')
    wb = torch.empty_like(wpack)
    _lib.check(L.pfa_lstm_pack_bwd(_lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(wb), st), 'pack_bwd')
    dG = torch.full((Th, R, 512), float('nan'), device=dev)
    dxe = torch.full((Th, R, 128), float('nan'), device=dev)
    gb, eb = torch.empty(512, device=dev), torch.empty(128, device=dev)
    ws = torch.empty(L.pfa_lstm_seq_backward_workspace_bytes(R), dtype=torch.uint8, device=dev)
    _lib.check(L.pfa_lstm_seq_backward(_lib.ptr(gates), _lib.ptr(Cs), _lib.ptr(xe), _lib.ptr(dh), R, Th, _lib.ptr(wb), _lib.ptr(dG),
                                       _lib.ptr(dxe), _lib.ptr(gb), _lib.ptr(eb), _lib.ptr(ws), st), 'bwd')

    # torch reference in f64
    sd = {k: v.detach().double() for k, v in pol.state_dict().items()}
    W1 = sd['policy.policy.encoder.weight']
    b1 = sd['policy.policy.encoder.bias'].clone().requires_grad_(True)
    Wih, Whh = sd['policy.recurrent.weight_ih_l0'], sd['policy.recurrent.weight_hh_l0']
    bih = sd['policy.recurrent.bias_ih_l0'].clone().requires_grad_(True)
    bhh = sd['policy.recurrent.bias_hh_l0']
    pre = obs[:, :, :49].double() @ W1.t() + b1
    pre.retain_grad()
    x = torch.relu(pre)
    h, c = h0.double(), c0.double()
    acts, hs, cs, pres = [], [], [], []
    for t in range(Th):
        p = x[t] @ Wih.t() + h @ Whh.t() + bih + bhh
        p.retain_grad()
        i, f, g, o = p.chunk(4, 1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * c + i * g
        h = o * torch.tanh(c)
        acts.append(torch.cat([i, f, g, o], 1)); hs.append(h); cs.append(c); pres.append(p)
    loss = sum((hs[t] * dh[t].double()).sum() for t in range(Th))
    loss.backward()
    tol = dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(xe.cpu().numpy(), x.detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(gates.cpu().numpy(), torch.stack(acts).detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(Hs[1:].cpu().numpy(), torch.stack(hs).detach().cpu().numpy(), **tol)
    np.testing.assert_allclose(Cs[1:].cpu().numpy(), torch.stack(cs).detach().cpu().numpy(), **tol)
    assert torch.equal(Hs[0], h0) and torch.equal(Cs[0], c0)
    np.testing.assert_allclose(dG.cpu().numpy(), torch.stack([p.grad for p in pres]).cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dxe.cpu().numpy(), pre.grad.cpu().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(gb.cpu().numpy(), bih.grad.cpu().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(eb.cpu().numpy(), b1.grad.cpu().numpy(), rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('d,nt', [(1, 1), (2, 2), (4, 3)])
def test_recurrent_policy_trains_on_other_grid_sizes(d, nt):
    """obs 3x3 / 5x5 / 9x9 -> row strides 16 / 32 / 96: every fused recurrent kernel and the weight-gradient contraction
    have those instantiations; the update is compared with the torch-fp32 oracle trainer on the same rollout."""
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from oracle import c_oracle, ppo_torch
    n, horizon, nmb, bptt = 32, 16, 2, 8
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    torch.manual_seed(d)
    vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=d, num_targets=nt), num_envs=n, backend=vector.Squared)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cfg = _config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=7)
    data = clean_pufferl.create(cfg, vec, pol)
    D = (2 * d + 1) ** 2
    assert vec.obs_stride == max(16, (D + 15) // 16 * 16)
    opol = ppo_torch.Policy.from_reference_state_dict({k: v.detach().cpu().clone() for k, v in pol.state_dict().items()})
    assert opol.recurrent
    clean_pufferl.evaluate(data)
    exp = data.experience
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, d, nt), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt,
                           update_epochs=2, learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3],
                           vf_coef=hp[4], vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7],
                           total_timesteps=B * 10, seed=7)
    tr.obs = torch.as_tensor(_step_major(exp.obs, n, horizon)[:, :D].copy())
    tr.actions = _step_major(exp.actions, n, horizon).astype(np.int64)
    tr.logprobs = _step_major(exp.logprobs, n, horizon).copy()
    tr.rewards = _step_major(exp.rewards, n, horizon).copy()
    tr.dones = _step_major(exp.dones, n, horizon).copy()
    tr.values = _step_major(exp.values, n, horizon).copy()
    tr.global_step = data.global_step
    tr.train()
    clean_pufferl.train(data)
    want = opol.state_arrays()                      # short names: 'encoder.weight', ..., 'weight_ih_l0', ...
    sd = pol.state_dict()
    for k, w in want.items():
        full = ('policy.recurrent.' if k.endswith('_l0') else 'policy.policy.') + k
        np.testing.assert_allclose(sd[full].cpu().numpy(), w, rtol=1e-5, atol=1e-5, err_msg=k)


def test_minigrid_shaped_160_byte_rows_vs_oracle_trainer():
    """BASELINE configs[2] / SURVEY config C3 shape: LSTM policy on 160-byte observation rows (read as 160 floats), 7 actions,
    through the host path: rollout (recurrent policy step, state carried per agent) and two BPTT updates against the
    torch-fp32 oracle trainer."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from host_vecenv import HostByteRows
    from oracle import ppo_torch
    from pufferlib_amd import clean_pufferl, cleanrl, models
    from test_gpu_ppo import _config
    n, horizon, nmb, bptt = 48, 32, 2, 16
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    vec = HostByteRows(n)
    torch.manual_seed(8)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    with torch.no_grad():
        pol.policy.policy.encoder.weight.mul_(0.3)            # byte-valued inputs: keep the hidden layer in a sane range
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=3), vec, pol)
    assert data.flat_params.obs_stride == 160 and data.flat_params.obs_dim == 160
    opol = ppo_torch.Policy.from_reference_state_dict(sd0)
    tr = ppo_torch.Trainer(opol, HostByteRows(n), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4], vf_clip_coef=hp[5],
                           max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=3)
    sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
    for it in range(2):
        noise = torch.empty(horizon, n, 7).exponential_(1)
        data.noise = noise.clone()
        clean_pufferl.evaluate(data)
        tr.evaluate(noise.numpy())
        e = data.experience
        assert np.array_equal(sm(e.obs), tr.obs.numpy()), it
        assert np.array_equal(sm(e.actions.long()), tr.actions), it
        assert np.array_equal(sm(e.rewards), tr.rewards)
        np.testing.assert_allclose(sm(e.logprobs), tr.logprobs, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(sm(e.values), tr.values, rtol=1e-5, atol=1e-5)
        Lo = tr.train()
        clean_pufferl.train(data)
        L = data.losses
        np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.approx_kl], [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'approx_kl')],
                                   rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        for k, arr in opol.state_arrays().items():
            key = ('policy.recurrent.' + k) if k.endswith('_l0') else ('policy.policy.' + k)
            np.testing.assert_allclose(sd[key].cpu().numpy(), arr, rtol=1e-5, atol=1e-5, err_msg=k)
    # the fused MLP kernels stop at 128 floats per row: the same rows behind a non-recurrent Default take the GEMM path (general.py)
    from pufferlib_amd import general
    d2 = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=3), HostByteRows(n),
                              cleanrl.Policy(models.Default(vec.driver_env)))
    assert isinstance(d2.flat_params, general.GeneralParams) and d2.gen_engine is not None and d2.flat_params.obs_stride == 160
