"""Host orchestration of the recurrent-policy path (pufferlib.models.LSTMWrapper, models.py:64-111).  Every product is a
hand-written fp32 MFMA kernel behind the C ABI; this module only sequences them:

  pack_gates       [W_ih | W_hh] -> MFMA fragment order (csrc/lstm_fused.hip), after every weight change
  policy_step      encode_observations -> one nn.LSTM step -> decode_actions + sample_logits, one kernel   (protocol path)
  Engine.rollout   clean_pufferl.evaluate for a Squared vecenv: ONE persistent kernel for all T steps (csrc/lstm_fused.hip)
  Engine.update    one minibatch of clean_pufferl.train: gather -> fused forward over the bptt_horizon steps with the carried,
                   detached state (clean_pufferl.py:186-191; csrc/lstm_seq.hip) -> heads + PPO loss (csrc/lstm.hip) -> fused
                   BPTT (csrc/lstm_seq.hip) -> weight-gradient contractions over the rows (csrc/gemm.hip) -> flat gradient

Row order inside a minibatch is TIME-MAJOR (row t*R + k = segment mb + k*nmb at step t) so that every per-step slice is
a contiguous [R][...] block.
"""
import ctypes as C

import torch

from . import _lib

H = 128


def pack_gates(fp, out=None):
    """[W_ih | W_hh] in the fragment order the fused kernels stream (csrc/lstm_fused.hip); call after the weights changed."""
    L = _lib.lib()
    if out is None:
        out = torch.empty(L.pfa_lstm_pack_bytes() // 4, device=fp.flat.device)
    _lib.check(L.pfa_lstm_pack(_lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(out), _lib.stream_handle()), 'lstm_pack')
    return out


def policy_step(fp, obs, h, c, noise, key, row_offset, wpack=None):
    """One rollout step on ``rows`` observation rows [rows][obs_stride]; h, c [rows][128] are updated in place.
    Returns (actions int64, logprob, entropy, value[:, None])."""
    L = _lib.lib()
    stream = _lib.stream_handle()
    rows = obs.shape[0]
    dev = obs.device
    if wpack is None:
        wpack = pack_gates(fp)
    actions = torch.empty(rows, dtype=torch.int64, device=dev)
    logprob = torch.empty(rows, device=dev)
    entropy = torch.empty(rows, device=dev)
    value = torch.empty(rows, device=dev)
    _lib.check(L.pfa_lstm_policy_step(_lib.ptr(obs), rows, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(wpack), _lib.ptr(h),
                                      _lib.ptr(c), _lib.ptr(noise), C.byref(key), row_offset, _lib.ptr(actions),
                                      _lib.ptr(logprob), _lib.ptr(entropy), _lib.ptr(value), stream), 'lstm_policy_step')
    return fp.unpack_actions(actions), logprob, entropy, value.unsqueeze(1)


class Engine:
    """Buffers and step functions of the recurrent path for one (vecenv, experience) pair."""

    def __init__(self, fp, experience, vecenv):
        self.fp, self.exp, self.vec = fp, experience, vecenv
        dev = fp.flat.device
        N = vecenv.num_agents
        self.lstm_h = torch.zeros(1, N, H, device=dev)      # Experience.lstm_h / lstm_c (clean_pufferl.py:407-412)
        self.lstm_c = torch.zeros(1, N, H, device=dev)
        self.wpack = torch.empty(_lib.lib().pfa_lstm_pack_bytes() // 4, device=dev)
        self.wpack_bwd = torch.empty_like(self.wpack)
        M, Th = experience.minibatch_size, experience.bptt_horizon
        self.M, self.Th, self.R = M, Th, M // Th
        R = self.R
        DP = fp.obs_stride
        # the minibatches' observation rows in time-major order: the rows do not change between the epochs of an update, so each
        # minibatch is gathered once per update (epoch 0) and kept — one buffer per minibatch, up to PFA_LSTM_OBS_CACHE_MB (default
        # 1024: a second copy of the observation batch) in total, else one buffer re-gathered every time as before.  The cache is
        # dropped whenever the experience is rewritten (clean_pufferl._finish_evaluate / hostpath.evaluate call invalidate_obs_cache)
        import os
        nmb = experience.num_minibatches
        keep = nmb * M * DP * 4 <= int(os.environ.get('PFA_LSTM_OBS_CACHE_MB', '1024')) << 20
        self.obs_tm_all = [torch.empty(M, DP, device=dev) for _ in range(nmb if keep else 1)]
        self.obs_tm = self.obs_tm_all[0]
        self._gathered = set()              # (update id, minibatch) pairs present in obs_tm_all
        self.update_id = 0                  # set by clean_pufferl.train(): one id per train() call
        self.xe = torch.empty(M, H, device=dev)
        self.gates = torch.empty(Th, R, 4 * H, device=dev)
        self.Cs = torch.zeros(Th + 1, R, H, device=dev)
        self.Hs = torch.zeros(Th + 1, R, H, device=dev)
        self.dout = torch.empty(M, 16, device=dev)
        self.dh_heads = torch.empty(M, H, device=dev)
        self.dG = torch.empty(Th, R, 4 * H, device=dev)
        self.dxe = torch.empty(M, H, device=dev)
        self.bsum16 = torch.empty(16, device=dev)
        L = _lib.lib()
        self.ws = torch.empty(L.pfa_lstm_heads_loss_workspace_bytes(),
                              dtype=torch.uint8, device=dev)
        shapes = [(4 * H, H), (H, DP), (16, H)]
        # one split-partial workspace per weight-gradient product: the three products' partials are summed by ONE launch (pfa_reduce_multi)
        self.gemm_ws = [torch.empty(nb, dtype=torch.uint8, device=dev) for nb in
                        (L.pfa_gemm_tn_workspace_bytes(H, DP, M), L.pfa_gemm_tn2_workspace_bytes(4 * H, M), L.pfa_gemm_tn_workspace_bytes(16, H, M))]
        self.g16 = torch.empty(16, H, device=dev)
        self.bwd_ws = torch.empty(L.pfa_lstm_seq_backward_workspace_bytes(R), dtype=torch.uint8, device=dev)
        self.norm_partials = torch.empty(256, dtype=torch.float64, device=dev)
        self.state = None      # truthy once slot Th of Hs/Cs holds the previous minibatch's final state (this epoch)

    # -------------------------------------------------------------------------------------------- rollout
    def rollout(self, T, noise, key_seed, step0, env_offset):
        """clean_pufferl.evaluate's loop for a Squared or Memory vecenv: one persistent kernel (csrc/lstm_fused.hip).  The caller has
        made sure the reset-target tape holds the rounds of these T sends and accounts for them afterwards."""
        L = _lib.lib()
        vec, exp, fp = self.vec, self.exp, self.fp
        assert T == exp.horizon
        pack_gates(fp, self.wpack)
        key = _lib.NoiseKey(key_seed, step0)
        fn = {'Memory': L.pfa_rollout_lstm_memory, 'Synthetic': L.pfa_rollout_lstm_synth}.get(type(vec).__name__, L.pfa_rollout_lstm_squared)   # same signature
        _lib.check(fn(_lib.ptr(vec.state), C.byref(vec.cfg), _lib.ptr(fp.flat), C.byref(fp.dims),
                      _lib.ptr(self.wpack), _lib.ptr(self.lstm_h), _lib.ptr(self.lstm_c), C.byref(exp.c),
                      _lib.ptr(noise), C.byref(key), env_offset, _lib.ptr(vec.obs_buf),
                      _lib.ptr(vec.rewards), _lib.ptr(vec.terminals_u8), _lib.ptr(vec.truncations_u8),
                      _lib.ptr(vec.masks_u8), _lib.stream_handle()), 'rollout_lstm')

    def rollout_stepwise(self, T, noise, key_seed, step0, env_offset):
        """The same rollout through the protocol-level pieces (policy_step / store / send), one launch sequence per step —
        what a generic vecenv would run; bit-identical to rollout() (tests)."""
        L = _lib.lib()
        vec, exp, fp = self.vec, self.exp, self.fp
        stream = _lib.stream_handle()
        h, c = self.lstm_h[0], self.lstm_c[0]
        N = vec.num_agents
        pack_gates(fp, self.wpack)
        for t in range(T):
            key = _lib.NoiseKey(key_seed, step0 + t)
            nz = None if noise is None else noise[t]
            actions, logprob, _, value = policy_step(fp, vec.obs_buf, h, c, nz, key, env_offset, wpack=self.wpack)
            _lib.check(L.pfa_store_step(C.byref(exp.c), t, N, fp.obs_stride, _lib.ptr(vec.obs_buf), _lib.ptr(vec.rewards),
                                        _lib.ptr(vec.terminals_u8), _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(value),
                                        stream), 'store_step')
            vec.ensure_tape(1)
            _lib.check(L.pfa_squared_send(_lib.ptr(vec.state), C.byref(vec.cfg), _lib.ptr(actions), _lib.ptr(vec.obs_buf),
                                          _lib.ptr(vec.rewards), _lib.ptr(vec.terminals_u8), _lib.ptr(vec.truncations_u8),
                                          _lib.ptr(vec.masks_u8), stream), 'send')
            vec.sends += 1
        vec.sends -= T      # the caller (clean_pufferl.evaluate) accounts for the T sends of a rollout

    def invalidate_obs_cache(self):
        """The experience rows were rewritten (a new rollout): the gathered time-major copies are stale whoever drives update()."""
        self._gathered = set()

    # -------------------------------------------------------------------------------------------- update
    def update(self, mb, hp, adv_stats, global_mb_rows, grads, B):
        """Forward + loss + BPTT for minibatch `mb`; writes the flat gradient (+8 loss sums) into `grads`."""
        L = _lib.lib()
        stream = _lib.stream_handle()
        fp, exp = self.fp, self.exp
        M, Th, R = self.M, self.Th, self.R
        # ---- forward -----------------------------------------------------------------------------------------------
        keep = len(self.obs_tm_all) > 1
        self.obs_tm = self.obs_tm_all[mb if keep else 0]
        if not keep or self.update_id == 0 or (self.update_id, mb) not in self._gathered:     # (id 0: update() driven without train())
            _lib.check(L.pfa_gather_obs_time_major(C.byref(exp.c), B, mb, C.byref(hp), fp.obs_stride, _lib.ptr(self.obs_tm), stream),
                       'gather_obs')
            if keep:
                self._gathered = {k for k in self._gathered if k[0] == self.update_id} | {(self.update_id, mb)}
        # lstm_state = None at the start of every epoch (clean_pufferl.py:176), else carried across minibatches, detached (:188-191):
        # the previous minibatch's final state is still in slot Th, and the forward kernel itself moves it (or zeros) into slot 0
        init_slot = -1 if (mb == 0 or self.state is None) else Th
        # both re-tilings of the gate matrix in one launch (the weights changed in the previous optimizer step)
        _lib.check(L.pfa_lstm_pack_both(_lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(self.wpack), _lib.ptr(self.wpack_bwd), stream), 'lstm_pack_both')
        _lib.check(L.pfa_lstm_seq_forward(_lib.ptr(self.obs_tm), R, Th, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(self.wpack),
                                          _lib.ptr(self.xe), _lib.ptr(self.gates), _lib.ptr(self.Hs), _lib.ptr(self.Cs), init_slot, stream),
                   'lstm_seq_forward')
        self.state = True
        h_all = self.Hs[1:].view(M, H)
        # ---- heads + loss ----------------------------------------------------------------------------------------
        loss_sums = grads[fp.count:fp.count + 16]         # 8 f64 sums as (hi, lo) float pairs
        _lib.check(L.pfa_lstm_heads_loss(_lib.ptr(h_all), C.byref(exp.c), B, mb, _lib.ptr(fp.flat), C.byref(fp.dims), C.byref(hp),
                                         _lib.ptr(adv_stats), global_mb_rows, _lib.ptr(self.dout), _lib.ptr(self.dh_heads),
                                         _lib.ptr(loss_sums), _lib.ptr(self.bsum16), _lib.ptr(self.ws), stream), 'lstm_heads_loss')
        # ---- back-propagation through time (csrc/lstm_seq.hip) + weight gradients (csrc/gemm.hip) ----------------------
        gv = fp.split(grads[:fp.count])
        # (bias gradients: the kernel leaves its per-workgroup column sums in bwd_ws; they are summed with the products' partials below)
        _lib.check(L.pfa_lstm_seq_backward(_lib.ptr(self.gates), _lib.ptr(self.Cs), _lib.ptr(self.xe), _lib.ptr(self.dh_heads), R, Th,
                                           _lib.ptr(self.wpack_bwd), _lib.ptr(self.dG), _lib.ptr(self.dxe), None, None, _lib.ptr(self.bwd_ws),
                                           stream), 'lstm_seq_backward')
        dG = self.dG.view(M, 4 * H)
        gW1p = fp.encoder_weight_padded(grads[:fp.count])
        hp_, gih, ghh = self.Hs[:Th].view(M, H), gv['recurrent.weight_ih_l0'], gv['recurrent.weight_hh_l0']
        jobs = (_lib.ReduceJob * 4)()
        # dW1 = dxe^T obs;  dW_ih = dG^T xe and dW_hh = dG^T h_prev in one pass over dG;  [16][128] heads: rows < A decoder, row A value head
        _lib.check(L.pfa_gemm_tn_partial_f32(_lib.ptr(self.dxe), self.dxe.stride(0), _lib.ptr(self.obs_tm), self.obs_tm.stride(0), H, self.obs_tm.shape[1], M,
                                             _lib.ptr(self.gemm_ws[0]), C.byref(jobs[0]), stream), 'gemm_tn dW1')
        _lib.check(L.pfa_gemm_tn2_partial_f32(_lib.ptr(dG), dG.stride(0), _lib.ptr(self.xe), self.xe.stride(0), _lib.ptr(hp_), hp_.stride(0), 4 * H, M,
                                              _lib.ptr(self.gemm_ws[1]), C.byref(jobs[1]), stream), 'gemm_tn2 dW_ih dW_hh')
        _lib.check(L.pfa_gemm_tn_partial_f32(_lib.ptr(self.dout), self.dout.stride(0), _lib.ptr(h_all), h_all.stride(0), 16, H, M,
                                             _lib.ptr(self.gemm_ws[2]), C.byref(jobs[2]), stream), 'gemm_tn heads')
        jobs[0].c, jobs[0].ldc = gW1p.data_ptr(), gW1p.stride(0)
        jobs[1].c, jobs[1].ldc, jobs[1].c2, jobs[1].ldc2 = gih.data_ptr(), gih.stride(0), ghh.data_ptr(), ghh.stride(0)
        jobs[2].c, jobs[2].ldc = self.g16.data_ptr(), self.g16.stride(0)
        gb, eb = gv['recurrent.bias_ih_l0'], gv['encoder.bias']
        jobs[3] = _lib.ReduceJob(1, (R + 31) // 32, 1, 4 * H + H, self.bwd_ws.data_ptr(), gb.data_ptr(), 0, eb.data_ptr(), 0, 4 * H, 0)
        _lib.check(L.pfa_reduce_multi(jobs, 4, stream), 'reduce_multi')
        _lib.check(L.pfa_lstm_finish_grads(_lib.ptr(grads), C.byref(fp.dims), _lib.ptr(self.g16), _lib.ptr(self.bsum16), stream),
                   'lstm_finish_grads')

    def _gemm_tn(self, a, b, out):
        """out[mo][no] = a[k][mo]^T b[k][no] — the weight-gradient contraction over the minibatch rows (csrc/gemm.hip)."""
        L = _lib.lib()
        _lib.check(L.pfa_gemm_tn_f32(_lib.ptr(a), a.stride(0), _lib.ptr(b), b.stride(0), _lib.ptr(out), out.stride(0),
                                     a.shape[1], b.shape[1], a.shape[0], _lib.ptr(self.gemm_ws[1]), _lib.stream_handle()), 'gemm_tn')
        return out

    def clip_adam(self, grads, opt, max_grad_norm, loss_acc, loss_scale):
        L = _lib.lib()
        stream = _lib.stream_handle()
        fp = self.fp
        n = self.norm_partials.numel()
        _lib.check(L.pfa_sumsq_partials(_lib.ptr(grads), fp.count, _lib.ptr(self.norm_partials), n, stream), 'sumsq')
        opt.step_count += 1
        g = opt.param_groups[0]
        _lib.check(L.pfa_adam_clip_step(_lib.ptr(fp.flat), _lib.ptr(grads), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq),
                                        fp.count, float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                        opt.step_count, float(max_grad_norm), 1.0, C.c_void_p(grads.data_ptr() + 4 * fp.count),
                                        _lib.ptr(loss_acc), loss_scale, _lib.ptr(self.norm_partials), n, stream), 'adam')
