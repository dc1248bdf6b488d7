"""HIP policy forward / fused rollout / PPO update vs the golden outputs of the unmodified reference
(tests/golden/ppo_mlp.npz) and vs the torch-fp32 oracle at larger sizes.  Tolerance: north_star's 1e-5 fp32,
stated as allclose(rtol=1e-5, atol=1e-5); actions (integers) bit-exact given the same multinomial noise."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-5, atol=1e-5)


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


def _make(n, d=3, nt=1):
    from pufferlib_amd import vector, models, cleanrl
    vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=d, num_targets=nt), num_envs=n,
                      backend=vector.Squared)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    return vec, pol


def _load_weights(pol, g, prefix):
    sd = {k[len(prefix):]: torch.as_tensor(g[k]) for k in g.files if k.startswith(prefix)}
    pol.load_state_dict(sd)


def _step_major(x, n, t):
    """env-major device tensor -> the reference's storage (step-major) order as numpy."""
    return x.view(n, t, *x.shape[1:]).transpose(0, 1).reshape(n * t, *x.shape[1:]).cpu().numpy()


def test_forward_sample_matches_reference_rollout(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ppo_mlp.npz'))
    n, horizon = int(g['config'][0]), int(g['config'][1])
    vec, pol = _make(n)
    _load_weights(pol, g, 'w0.')
    obs = torch.as_tensor(g['it0.obs'].astype(np.float32)).cuda()
    for t in (0, 1, 5, horizon - 1):
        rows = slice(t * n, (t + 1) * n)
        a, lp, ent, val = pol(obs[rows].view(n, 7, 7), noise=torch.as_tensor(g['it0.noise'][t]))
        assert np.array_equal(a.cpu().numpy(), g['it0.actions'][rows].astype(np.int64)), t
        np.testing.assert_allclose(lp.cpu().numpy(), g['it0.logprobs'][rows], **TOL)
        np.testing.assert_allclose(val.flatten().cpu().numpy(), g['it0.values'][rows], **TOL)
        np.testing.assert_allclose(ent.cpu().numpy(), np.full(n, np.log(8)), rtol=1e-3)


def test_create_evaluate_train_replays_golden(golden_dir):
    from pufferlib_amd import clean_pufferl
    g = np.load(os.path.join(golden_dir, 'ppo_mlp.npz'))
    n, horizon, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    vec, pol = _make(n)
    _load_weights(pol, g, 'w0.')
    data = clean_pufferl.create(_config(n, horizon, mbs, bptt, epochs, total, [float(x) for x in g['hparams']]), vec, pol)
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
        np.testing.assert_allclose([stats['episode_return'], stats['episode_length'], stats['score']],
                                   g[f'it{it}.stats'], rtol=1e-9)
        clean_pufferl.train(data)
        # golden b_advantages[m, q] is minibatch m's q-th row; same partition on our env-major buffer
        for m in range(exp.num_minibatches):
            idx = exp.minibatch_rows_index(m)
            np.testing.assert_allclose(exp.advantages[idx].cpu().numpy(), g[f'it{it}.advantages'][m], **TOL)
            np.testing.assert_allclose(exp.returns[idx].cpu().numpy(), g[f'it{it}.returns'][m], **TOL)
        L = data.losses
        got = [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac, L.explained_variance]
        np.testing.assert_allclose(got, g[f'it{it}.losses'], rtol=1e-5, atol=1e-5)
        sd = pol.state_dict()
        m_, v_ = data.flat_params.split(data.optimizer.exp_avg), data.flat_params.split(data.optimizer.exp_avg_sq)
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'it{it}.w.{k}'], err_msg=k, **TOL)
            short = k[len('policy.'):]
            np.testing.assert_allclose(m_[short].cpu().numpy(), g[f'it{it}.m.{k}'], rtol=1e-5, atol=1e-6, err_msg=k)
            np.testing.assert_allclose(v_[short].cpu().numpy(), g[f'it{it}.v.{k}'], rtol=1e-5, atol=1e-8, err_msg=k)
        assert abs(data.optimizer.param_groups[0]['lr'] - float(g[f'it{it}.lr_next'])) < 1e-12
    # pad columns of the encoder never move
    H, DP = 128, vec.obs_stride
    assert float(data.flat_params.flat[:H * DP].view(H, DP)[:, 49:].abs().sum()) == 0.0


@pytest.mark.parametrize('n,horizon,nmb,bptt', [(256, 64, 4, 16), (64, 32, 2, 4), (1024, 128, 4, 16), (704, 64, 2, 16)])
def test_update_vs_torch_oracle(n, horizon, nmb, bptt, matrix_products):
    """Same experience into the HIP update and the torch-fp32 restatement of clean_pufferl.train.  Both product forms of the fused
    gradient step: exact fp32 MFMA chains (csrc/ppo_update.hip) and the opt-in six bf16 partial products per fp32 product
    (csrc/ppo_bf16.hpp: minibatches of whole 32-row tiles — 128, 32, 1024 and 704 tiles here, the last one a ragged second round of
    the 512-workgroup grid; bptt 4 = tiles that straddle segments)."""
    from pufferlib_amd import clean_pufferl
    from oracle import c_oracle, ppo_torch
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    B = n * horizon
    vec, pol = _make(n)
    torch.manual_seed(3)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))     # move off the near-uniform init so ratios/clipping engage
    cfg = _config(n, horizon, B // nmb, bptt, 2, B * 10, hp)
    data = clean_pufferl.create(cfg, vec, pol)
    import ctypes as C
    from pufferlib_amd import _lib
    assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(data.flat_params.dims), B // nmb) == (1 if matrix_products == 'bf16x6' else 0)
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)          # Philox noise; fills the experience on device
    exp = data.experience

    ovec = c_oracle.SquaredSerial(n, 3, 1)
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, ovec, batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2,
                           learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4],
                           vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=1)
    # hand the oracle the device rollout's experience in its storage (step-major) order
    tr.obs = torch.as_tensor(_step_major(exp.obs, n, horizon)[:, :49].copy())
    tr.actions = _step_major(exp.actions, n, horizon).astype(np.int64)
    tr.logprobs = _step_major(exp.logprobs, n, horizon).copy()
    tr.rewards = _step_major(exp.rewards, n, horizon).copy()
    tr.dones = _step_major(exp.dones, n, horizon).copy()
    tr.values = _step_major(exp.values, n, horizon).copy()
    tr.global_step = data.global_step
    torch.set_num_threads(8)
    Lo = tr.train()
    clean_pufferl.train(data)
    L = data.losses
    np.testing.assert_allclose(
        [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
        [Lo['policy_loss'], Lo['value_loss'], Lo['entropy'], Lo['old_approx_kl'], Lo['approx_kl'], Lo['clipfrac']],
        rtol=1e-5, atol=1e-5)
    sd = pol.state_dict()
    for k, arr in opol.state_arrays().items():
        np.testing.assert_allclose(sd['policy.' + k].cpu().numpy(), arr, err_msg=k, **TOL)


def test_fused_rollout_equals_stepwise_protocol():
    """evaluate()'s persistent kernel == T x {recv, policy(obs), store, send} through the public protocol,
    bit for bit (same Philox stream)."""
    from pufferlib_amd import clean_pufferl
    n, horizon = 48, 20          # not a multiple of 16 envs: exercises the masked tail tile
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    vec, pol = _make(n)
    cfg = _config(n, horizon, n * horizon // 2, 4, 1, n * horizon * 4, hp, seed=5)
    data = clean_pufferl.create(cfg, vec, pol)
    clean_pufferl.evaluate(data)
    exp = data.experience

    vec2, pol2 = _make(n)
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
    assert torch.equal(_t(exp.obs, n, horizon)[:, :49], torch.cat(obs_l))
    assert torch.equal(_t(exp.actions, n, horizon).long(), torch.cat(act_l))
    assert torch.equal(_t(exp.logprobs, n, horizon), torch.cat(lp_l))
    assert torch.equal(_t(exp.values, n, horizon), torch.cat(val_l))
    assert torch.equal(_t(exp.rewards, n, horizon), torch.cat(rew_l))
    assert torch.equal(_t(exp.dones, n, horizon), torch.cat(done_l))
    # and the live buffers continue identically
    assert torch.equal(vec.observations, vec2.observations)
    assert torch.equal(vec.rewards, vec2.rewards) and torch.equal(vec.terminals, vec2.terminals)


def _t(x, n, t):
    return x.view(n, t, *x.shape[1:]).transpose(0, 1).reshape(n * t, *x.shape[1:])


def test_philox_actions_match_oracle_at_full_size():
    """4096 envs x 128 steps with the in-kernel Philox noise: replay the same stream on the CPU oracle
    (C env + torch MLP + C Philox) and require identical integer trajectories, modulo near-ties."""
    from pufferlib_amd import clean_pufferl
    from oracle import c_oracle, ppo_torch
    n, horizon = 4096, 128
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    vec, pol = _make(n)
    cfg = _config(n, horizon, n * horizon // 4, 16, 1, n * horizon * 4, hp, seed=1)
    data = clean_pufferl.create(cfg, vec, pol)
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)
    exp = data.experience
    # CPU side: Philox -> Exp(1) noise for the first 8 steps (the oracle is slow; 8 x 4096 rows suffices)
    steps = 8
    rows = np.arange(n)
    noise = np.empty((steps, n, 8), np.float32)
    for t in range(steps):
        for e in rows:
            for j in range(2):
                w = c_oracle.philox4x32_10([e, j, t, 0], [1, 0])
                u = ((w >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
                noise[t, e, 4 * j:4 * j + 4] = -np.log(u)
    ovec = c_oracle.SquaredSerial(n, 3, 1)
    opol = ppo_torch.Policy(w0)
    ovec.async_reset(1)
    acts = _step_major(exp.actions, n, horizon)
    obs = _step_major(exp.obs, n, horizon)[:, :49]
    mism = 0
    for t in range(steps):
        o = ovec.recv()[0]
        assert np.array_equal(o.reshape(n, -1), obs[t * n:(t + 1) * n]), t     # envs in lock-step so far
        with torch.no_grad():
            logits, value, _ = opol.forward(torch.as_tensor(o.reshape(n, -1).copy()))
            a, lp, _ = ppo_torch.sample_logits(logits, noise=torch.as_tensor(noise[t]))
        dev = acts[t * n:(t + 1) * n]
        bad = np.nonzero(a.numpy() != dev)[0]
        mism += len(bad)
        ovec.send(dev.astype(np.int64))       # follow the device's actions so later steps stay comparable
    assert mism <= 2, f'{mism} action mismatches in {steps * n} samples (only near-ties in p/q may differ)'


def test_native_data_parallel_path_single_rank_equals_plain():
    """The RCCL-native DP loop (csrc/dist.cpp + pfa_ppo_mlp_train data_parallel=1) with a 1-rank communicator must
    reproduce the plain single-GPU update bit for bit (all-reduce over one rank is the identity; the clip norm is
    recomputed from the reduced gradient instead of the reduce kernel's partial sums)."""
    from pufferlib_amd import clean_pufferl, dist as pdist
    n, horizon = 256, 64
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    results = []
    for force in (False, True):
        vec, pol = _make(n)
        cfg = _config(n, horizon, n * horizon // 4, 16, 2, n * horizon * 4, hp, force_native_dp=force)
        data = clean_pufferl.create(cfg, vec, pol)
        assert data.native_dp == force
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        results.append((data.flat_params.flat.clone(), dict(data.losses)))
        pdist.finalize_native()
    (w0, l0), (w1, l1) = results
    np.testing.assert_allclose(w1.cpu().numpy(), w0.cpu().numpy(), rtol=1e-6, atol=1e-7)
    for k in l0:
        np.testing.assert_allclose(l1[k], l0[k], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('n,horizon,nmb,nt', [(256, 64, 4, 1), (64, 32, 2, 2), (4096, 32, 4, 1)])
def test_one_launch_reduce_and_adam_equals_the_two_kernel_form_bit_for_bit(monkeypatch, n, horizon, nmb, nt):
    """pfa_ppo_mlp_train sums the workgroup partials, takes the clip norm and applies Adam in ONE launch (grid barrier inside,
    csrc/ppo_update.hip ppo_reduce_adam_kernel); PFA_FUSED_ADAM=0 runs the two kernels pfa_ppo_mlp_grad + pfa_adam_clip_step are
    made of.  Same arithmetic in the same order: parameters, moments, gradient bucket and losses must be identical bits, over
    several updates (the barrier's generation word carries over between launches)."""
    from pufferlib_amd import clean_pufferl
    hp = [2.5e-3, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    runs = []
    for fused in ('1', '0'):
        monkeypatch.setenv('PFA_FUSED_ADAM', fused)
        torch.manual_seed(11)
        vec, pol = _make(n, nt=nt)
        data = clean_pufferl.create(_config(n, horizon, n * horizon // nmb, 16, 3, n * horizon * 8, hp, seed=9), vec, pol)
        for _ in range(3):
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        runs.append((data.flat_params.flat.clone(), data.optimizer.exp_avg.clone(), data.optimizer.exp_avg_sq.clone(), data.grads.clone(),
                     dict(data.losses)))
    for name, a, b in zip(('parameters', 'exp_avg', 'exp_avg_sq', 'gradient bucket'), runs[0][:4], runs[1][:4]):
        assert torch.equal(a, b), (name, float((a - b).abs().max()), int((a != b).sum()))
    assert repr(runs[0][4]) == repr(runs[1][4])
    assert torch.isfinite(runs[0][0]).all()


@pytest.mark.parametrize('d,nt,n,horizon,nmb,bptt', [(1, 4, 40, 16, 2, 8), (2, 2, 96, 32, 4, 16), (3, 2, 24, 48, 1, 4), (4, 3, 72, 32, 2, 16),
                                                     (5, 2, 200, 32, 4, 8)])
def test_other_grid_sizes_rollout_and_update_vs_oracle(d, nt, n, horizon, nmb, bptt):
    """obs_stride 16 / 32 / 64 / 96 / 128 (d = 1..5; the two wide strides run the 2-pair form of the gradient kernel), several targets, env counts that are not multiples of 16, one
    minibatch, short bptt: replay the device rollout's actions on the C oracle (bit-exact env side) and compare the
    update with the torch-fp32 restatement."""
    from pufferlib_amd import clean_pufferl
    from oracle import c_oracle, ppo_torch
    hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
    B = n * horizon
    vec, pol = _make(n, d, nt)
    torch.manual_seed(d)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cfg = _config(n, horizon, B // nmb, bptt, 2, B * 10, hp, seed=7)
    data = clean_pufferl.create(cfg, vec, pol)
    D = (2 * d + 1) ** 2
    assert vec.obs_stride == max(16, (D + 15) // 16 * 16)
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)
    exp = data.experience
    acts = _step_major(exp.actions, n, horizon)
    obs = _step_major(exp.obs, n, horizon)
    assert float(np.abs(obs[:, D:]).sum()) == 0.0            # pad columns
    ovec = c_oracle.SquaredSerial(n, d, nt)
    ovec.async_reset(7)
    for t in range(horizon):
        o, r, dn, _, _, _, _ = ovec.recv()
        rows = slice(t * n, (t + 1) * n)
        assert np.array_equal(o.reshape(n, -1), obs[rows, :D]), t
        assert np.array_equal(r, _step_major(exp.rewards, n, horizon)[rows]), t
        assert np.array_equal(dn.astype(np.float32), _step_major(exp.dones, n, horizon)[rows]), t
        ovec.send(acts[rows].astype(np.int64))
    opol = ppo_torch.Policy(w0)
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, d, nt), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt,
                           update_epochs=2, learning_rate=hp[0], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3],
                           vf_coef=hp[4], vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7],
                           total_timesteps=B * 10, seed=7)
    tr.obs = torch.as_tensor(obs[:, :D].copy())
    tr.actions = acts.astype(np.int64)
    tr.logprobs = _step_major(exp.logprobs, n, horizon).copy()
    tr.rewards = _step_major(exp.rewards, n, horizon).copy()
    tr.dones = _step_major(exp.dones, n, horizon).copy()
    tr.values = _step_major(exp.values, n, horizon).copy()
    tr.global_step = data.global_step
    Lo = tr.train()
    clean_pufferl.train(data)
    L = data.losses
    np.testing.assert_allclose(
        [L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
        [Lo['policy_loss'], Lo['value_loss'], Lo['entropy'], Lo['old_approx_kl'], Lo['approx_kl'], Lo['clipfrac']],
        rtol=1e-5, atol=1e-5)
    sd = pol.state_dict()
    for k, arr in opol.state_arrays().items():
        np.testing.assert_allclose(sd['policy.' + k].cpu().numpy(), arr, err_msg=k, **TOL)
    np.testing.assert_allclose(L.explained_variance, Lo['explained_variance'], rtol=1e-3, atol=1e-4)


def test_argument_errors_match_reference_messages():
    from pufferlib_amd import clean_pufferl
    vec, pol = _make(16)
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    with pytest.raises(ValueError, match='batch_size must be divisible by minibatch_size'):
        clean_pufferl.create(_config(16, 32, 100, 4, 1, 10 ** 6, hp), vec, pol)
    vec, pol = _make(16)
    with pytest.raises(ValueError, match='minibatch_size must be divisible by bptt_horizon'):
        clean_pufferl.create(_config(16, 32, 128, 3, 1, 10 ** 6, hp), vec, pol)
    vec, pol = _make(16)
    data = clean_pufferl.create(_config(16, 3, 24, 3, 1, 10 ** 6, hp), vec, pol)   # minibatch of 24 rows: not 16-aligned
    clean_pufferl.evaluate(data)
    from pufferlib_amd.exceptions import ExtensionError
    with pytest.raises(ExtensionError, match='multiple of 16'):
        clean_pufferl.train(data)


def test_checkpoint_round_trip_resumes_bit_identically(tmp_path):
    """save_checkpoint / try_load_checkpoint (clean_pufferl.py:509-546): a resumed trainer must continue exactly like
    the one that never stopped — parameters, Adam moments, step count and lr all live in the flat device buffers."""
    from pufferlib_amd import clean_pufferl
    n, horizon = 64, 32
    hp = [2.5e-3, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]

    def fresh(seed):
        torch.manual_seed(seed)
        vec, pol = _make(n)
        cfg = _config(n, horizon, 512, 16, 2, n * horizon * 8, hp, data_dir=str(tmp_path), exp_id='ckpt')
        return clean_pufferl.create(cfg, vec, pol)

    a = fresh(0)
    for _ in range(2):
        clean_pufferl.evaluate(a)
        clean_pufferl.train(a)
    path = clean_pufferl.save_checkpoint(a)
    assert os.path.exists(path) and os.path.exists(tmp_path / 'ckpt' / 'trainer_state.pt')
    saved = torch.load(path, weights_only=False)                      # whole module, as the reference saves it
    assert isinstance(saved, torch.nn.Module)

    b = fresh(123)                                                     # different initial weights
    assert not torch.equal(a.flat_params.flat, b.flat_params.flat)
    clean_pufferl.try_load_checkpoint(b)
    assert torch.equal(a.flat_params.flat, b.flat_params.flat)
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg) and torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)
    assert (b.global_step, b.epoch, b.optimizer.step_count) == (a.global_step, a.epoch, a.optimizer.step_count)
    assert b.optimizer.param_groups[0]['lr'] == a.optimizer.param_groups[0]['lr']

    # same experience + same state -> the next update is bit-identical
    clean_pufferl.evaluate(a)
    for name in ('obs', 'actions', 'logprobs', 'advantages', 'returns'):
        getattr(b.experience, name).copy_(getattr(a.experience, name))
    b.experience._rdv.copy_(a.experience._rdv)
    b.experience.ptr, b.experience.step = a.experience.ptr, a.experience.step
    b.global_step = a.global_step
    clean_pufferl.train(a)
    clean_pufferl.train(b)
    assert torch.equal(a.flat_params.flat, b.flat_params.flat)
    for k in a.losses:
        assert a.losses[k] == b.losses[k], k


def test_gpu_utilisation_counter_for_the_dashboard():
    """clean_pufferl.Utilization's gpu_util row (clean_pufferl.py:501: torch.cuda.utilization()) read from amdgpu's sysfs counter of
    the current device; 0 is also what it reports where the container hides the counter."""
    from pufferlib_amd import clean_pufferl
    x = torch.randn(4096, 4096, device='cuda')
    for _ in range(20):
        x = x @ x * 1e-4
    v = clean_pufferl.gpu_busy_percent()
    torch.cuda.synchronize()
    assert isinstance(v, int) and 0 <= v <= 100
    u = clean_pufferl.Utilization(delay=0.01)
    import time
    time.sleep(0.1)
    u.stop()
    assert len(u.gpu_util) >= 1 and len(u.gpu_mem) == len(u.gpu_util) and 0 < u.gpu_mem[-1] <= 1


def test_deferred_readback_mode_gives_the_same_numbers(monkeypatch):
    """PFA_LAZY_READBACK=1 (readback.py): stats and losses arrive through lazy containers one event later; three
    evaluate()+train() iterations give bit-identical statistics, losses and weights to the immediate mode, nothing is waited
    for until the numbers are read, and the readbacks of an iteration are resolved by the end of the next."""
    from pufferlib_amd import clean_pufferl, readback
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    n, horizon = 256, 32
    runs = []
    for lazy in (False, True):
        monkeypatch.setenv('PFA_LAZY_READBACK', '1' if lazy else '0')
        torch.manual_seed(3)
        vec, pol = _make(n)
        data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, 2, n * horizon * 8, hp, seed=5), vec, pol)
        out = []
        for it in range(3):
            stats, infos = clean_pufferl.evaluate(data)
            if lazy:
                assert isinstance(stats, readback.LazyDict) and data._rb_eval.outstanding      # nothing waited for yet
            clean_pufferl.train(data)
            if lazy:
                assert data._rb_train.outstanding
                if it:
                    assert not prev_stats._pending and len(prev_stats) == 3                       # resolved by this evaluate()
                prev_stats = stats
            out.append((stats, infos, data.losses))
        final = [(dict(s), dict(i), dict(l)) for s, i, l in out]
        assert not data._rb_eval.outstanding and not data._rb_train.outstanding
        runs.append((final, data.flat_params.flat.clone()))
    assert repr(runs[0][0]) == repr(runs[1][0])          # (repr: an explained_variance of nan still compares equal)
    assert torch.equal(runs[0][1], runs[1][1])
    assert set(runs[0][0][-1][0]) == {'episode_return', 'episode_length', 'score'} and runs[0][0][-1][2]['value_loss'] > 0


@pytest.mark.parametrize('num_actions', [3, 11, 12, 15])
def test_49_float_rows_with_other_action_counts_vs_oracle(num_actions, matrix_products):
    """The 7x7-grid instantiation of the gradient kernel (3 dW1 k-tiles + the column-48 accumulators) keeps the head outputs in
    PERMUTED fragment rows for up to 11 actions (three outputs per lane group) and in natural rows above that: both against the
    oracle trainer on a host vecenv with 49-float observations and 3 / 11 (permuted) and 12 / 15 (natural) actions; in both product
    forms (the bf16-path kernel of csrc/ppo_bf16.hpp has the same two head layouts)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from host_vecenv import HostMultiHead
    from pufferlib_amd import clean_pufferl, cleanrl, models, spaces
    from pufferlib_amd.models import FlatParams
    from oracle import ppo_torch

    class HostDiscrete(HostMultiHead):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.single_action_space = spaces.Discrete(self.nvec[0])
    n, horizon, nmb, bptt = 32, 16, 2, 8
    B = n * horizon
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    vec = HostDiscrete(n, [num_actions], obs_dim=49)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    torch.manual_seed(3)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.05 * torch.randn_like(p))
    data = clean_pufferl.create(_config(n, horizon, B // nmb, bptt, 2, B * 10, hp, env='host'), vec, pol)
    assert isinstance(data.flat_params, FlatParams) and data.flat_params.obs_stride == 64       # the fused kernels, not the GEMM path
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy(w0)

    class _V:
        num_envs = n
        observations = np.zeros((n, 49), np.float32)

        def async_reset(self, seed):
            pass
    tr = ppo_torch.Trainer(opol, _V(), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=2, learning_rate=hp[0], gamma=hp[1],
                           gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4], vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7],
                           total_timesteps=B * 10, seed=1)
    tr.obs = torch.as_tensor(_step_major(exp.obs, n, horizon)[:, :49].copy())
    tr.actions = _step_major(exp.actions, n, horizon).astype(np.int64)
    tr.logprobs, tr.rewards, tr.dones, tr.values = (_step_major(x, n, horizon).copy() for x in (exp.logprobs, exp.rewards, exp.dones, exp.values))
    tr.global_step = data.global_step
    Lo = tr.train()
    clean_pufferl.train(data)
    L = data.losses
    np.testing.assert_allclose([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac],
                               [Lo[k] for k in ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')], **TOL)
    sd = pol.state_dict()
    for k, arr in opol.state_arrays().items():
        np.testing.assert_allclose(sd['policy.' + k].cpu().numpy(), arr, err_msg=k, **TOL)


def test_early_gae_pass_is_the_same_update_and_yields_to_in_place_edits(monkeypatch):
    """evaluate() enqueues the update's GAE + advantage-statistics pass behind its statistics readback (the device runs it while
    the host walks from evaluate() into train()); train() reuses it only while its key holds (clean_pufferl._gae_key).
    (a) three iterations with the early pass == three iterations with PFA_EARLY_GAE=0, bit for bit (statistics, losses, weights,
    advantages); (b) rewards scaled IN PLACE between evaluate() and train() (reward shaping): train() runs its own pass over the
    edited rows — same numbers as the run that never had an early pass; (c) a second train() on the same rows runs its own pass."""
    from pufferlib_amd import clean_pufferl
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    n, horizon = 256, 32
    runs = []
    for early in ('1', '0'):
        monkeypatch.setenv('PFA_EARLY_GAE', early)
        torch.manual_seed(3)
        vec, pol = _make(n)
        data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, 2, n * horizon * 16, hp, seed=5), vec, pol)
        out = []
        for it in range(5):
            stats, infos = clean_pufferl.evaluate(data)
            assert (data._gae_done is not None) == (early == '1')
            if it == 2:                                   # reward shaping between the two calls
                data.experience.rewards.mul_(0.5)
                assert early == '0' or data._gae_done != clean_pufferl._gae_key(data)
            clean_pufferl.train(data)
            assert data._gae_done is None
            if it == 3:                                   # the same rows again: no early pass to reuse
                clean_pufferl.train(data)
            out.append((dict(stats), dict(data.losses), data.experience.advantages.clone(), data.experience.returns.clone(),
                        data.adv_stats.clone()))
        runs.append((out, data.flat_params.flat.clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert repr(a[:2]) == repr(b[:2])
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert torch.equal(runs[0][1], runs[1][1])
    shaped, plain = runs[0][0][2][2], runs[0][0][1][2]
    assert not torch.equal(shaped, plain) and float(shaped.abs().max()) > 0


@pytest.mark.parametrize('n,horizon,epochs', [(256, 32, 2), (64, 32, 1)])
def test_report_rides_the_updates_last_launch(n, horizon, epochs):
    """pfa_ppo_mlp_train_logged: the last reduce + Adam launch of an update writes train()'s ten report numbers itself (the lanes
    that own the loss sums) — losses and explained variance == the oracle's on the same rows are covered by the golden replays; here:
    the report is there and finite, and the entry point says what it did."""
    import ctypes as C
    from pufferlib_amd import _lib, clean_pufferl
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    torch.manual_seed(3)
    vec, pol = _make(n)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, epochs, n * horizon * 8, hp, seed=5), vec, pol)
    out = []
    for it in range(3):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        out.append(dict(data.losses))
    assert all(np.isfinite(v) for o in out for k, v in o.items() if k != 'explained_variance')
    assert out[-1]['value_loss'] > 0 and out[-1]['entropy'] > 0
    # the entry point itself: report written by the update (log_packed = 1) == losses + ev4 packed by hand
    L = _lib.lib()
    ex, fp, opt = data.experience, data.flat_params, data.optimizer
    nmb = ex.num_minibatches
    hpar = clean_pufferl._make_hparams(data.config, ex)
    ev4 = torch.arange(1, 5, dtype=torch.float64, device='cuda') * 0.25
    out10 = torch.full((10,), -1.0, dtype=torch.float64, device='cuda')
    acc = torch.zeros(8, dtype=torch.float64, device='cuda')
    packed = C.c_int32(-1)
    g = opt.param_groups[0]
    _lib.check(L.pfa_ppo_mlp_train_logged(
        C.byref(ex.c), ex.batch_size, _lib.ptr(fp.flat), C.byref(fp.dims), C.byref(hpar), _lib.ptr(data.adv_stats), _lib.ptr(data.grads),
        _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), opt.step_count, float(g['lr']), 0.9, 0.999, 1e-5, 0.5, epochs, _lib.ptr(acc),
        _lib.ptr(data.workspace), 0, _lib.ptr(ev4), _lib.ptr(out10), C.byref(packed), _lib.stream_handle()), 'train_logged')
    torch.cuda.synchronize()
    assert packed.value == 1
    assert torch.equal(out10[:6], acc[:6]) and torch.equal(out10[6:], ev4) and float(acc[:3].abs().sum()) > 0


def test_update_does_not_depend_on_what_the_caller_left_in_the_workspace():
    """Advisor (round 5): the grid-wide hand-off words of the one-launch reduce + Adam used to live at the end of the caller's workspace,
    whose contract silently was "zeroed once, written by nobody else" — a recycled buffer whose bits happened to look like the
    launch generation would have been read as a norm piece.  They are library-owned now (csrc/ppo_update.hip grid_words_of): an update
    over a workspace full of 0xFF bytes, and one over a workspace that held another trainer's data, give the bits of the clean run."""
    from pufferlib_amd import _lib, clean_pufferl
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    n, horizon = 256, 32
    outs = []
    for fill in (None, 0xFF, 'recycled'):
        torch.manual_seed(3)
        vec, pol = _make(n)
        data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, 2, n * horizon * 8, hp, seed=5), vec, pol)
        for it in range(2):
            clean_pufferl.evaluate(data)
            if fill == 0xFF:
                torch.cuda.synchronize()
                data._gae_done = None                     # (the early GAE pass keeps its sums in the workspace: let train() redo it)
                data.workspace.fill_(0xFF)
            elif fill == 'recycled':
                torch.cuda.synchronize()
                data._gae_done = None
                data.workspace.copy_(torch.randint(0, 256, data.workspace.shape, dtype=torch.uint8, device='cuda',
                                                   generator=torch.Generator(device='cuda').manual_seed(it)))
            clean_pufferl.train(data)
        assert _lib.lib().pfa_ppo_grid_status() == 0
        outs.append((data.flat_params.flat.clone(), dict(data.losses)))
    for flat, losses in outs[1:]:
        assert torch.equal(flat, outs[0][0]) and repr(losses) == repr(outs[0][1])
    assert _lib.lib().pfa_ppo_grid_reset() == 0           # nothing was raised; the reset entry point reports the value it cleared
