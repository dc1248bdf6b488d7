"""NatureCNN policy engine (BASELINE configs[3]): pufferlib.models.Convolutional (models.py:113-157) behind
frameworks.cleanrl.Policy, every product on the fp32-MFMA implicit-GEMM kernels of csrc/igemm.hip.

  Layer.forward / backward_dx / backward_dw      one conv or linear layer = one kernel launch each
  Engine.forward(frames)                         the rollout / training forward of a batch of uint8 frames -> hidden [n][512]
  Engine.policy_step(frames, ...)                policy(obs) in rollout mode: + heads + sample_logits (csrc/cnn_heads.hip)
  Engine.update(mb, ...)                         forward + PPO loss + backward of one minibatch (in chunks of `chunk` rows) ->
                                                 flat gradient in named_parameters() order + the 16-float loss tail

Activations are NHWC f32 (conv outputs [n*OH*OW][OC]); the Linear behind nn.Flatten sees NHWC rows, its weight columns are
re-ordered when packed.  Algorithmic work per frame: forward 2 (256*32*400 + 512*64*81 + 576*64*49 + 3136*512 +
512*(A+1)) = 18.7 MFLOP, forward + backward ~ 3x that minus conv1's dX."""
import ctypes as C

import torch

from . import _lib

F32 = torch.float32
MODE_DENSE, MODE_IM2COL_F32, MODE_IM2COL_U8, MODE_COL2IM = 0, 1, 2, 3
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK = 0, 1, 2, 3


def _operand(mode, tensor, lda=0, geom=(0,) * 9):
    return _lib.IgemmOperand(mode, 0, tensor.data_ptr(), lda, *geom)


class ConvLayer:
    """Conv2d(IC, OC, K, stride S), valid padding, + ReLU.  Input NHWC f32 [n][IH][IW][IC], or uint8 NCHW frames (first layer)."""

    def __init__(self, weight, bias, ih, iw, stride, u8_input, device):
        oc, ic, kh, kw = weight.shape
        self.w, self.b = weight, bias
        self.IC, self.IH, self.IW, self.OC, self.KH, self.KW, self.S = ic, ih, iw, oc, kh, kw, stride
        self.OH, self.OW = (ih - kh) // stride + 1, (iw - kw) // stride + 1
        self.K = ic * kh * kw
        self.u8 = u8_input
        self.geom = (ic, ih, iw, oc, self.OH, self.OW, kh, kw, stride)
        # forward B [oc][k] in the loader's patch order (the first layer: torch's own order, weights / 255 — its loader hands out bytes)
        self.w_fwd = torch.empty(oc, self.K, device=device)
        # dX B [phase][ic][(jy, jx, oc)] (no dX for the first layer)
        self.phases = stride * stride
        self.KP = (kh // stride) * (kw // stride) * oc if not u8_input else 0
        self.w_dx = None if u8_input else torch.empty(self.phases, ic, self.KP, device=device)

    def pack(self):
        L = _lib.lib()
        _lib.check(L.pfa_cnn_pack_conv(_lib.ptr(self.w), C.byref(_operand(0, self.w, 0, self.geom)), 1 if self.u8 else 0, _lib.ptr(self.w_fwd),
                                       _lib.ptr(self.w_dx), _lib.stream_handle()), 'pack_conv')

    def out_rows(self, n):
        return n * self.OH * self.OW

    def forward(self, x, n, out):
        """out [n*OH*OW][OC] = relu(conv(x) + bias)."""
        a = _operand(MODE_IM2COL_U8 if self.u8 else MODE_IM2COL_F32, x, 0, self.geom)
        _lib.check(_lib.lib().pfa_igemm_rows(C.byref(a), self.out_rows(n), self.K, _lib.ptr(self.w_fwd), self.K, self.OC, _lib.ptr(out), self.OC,
                                             EPI_BIAS_RELU, _lib.ptr(self.b), None, 0, _lib.stream_handle()), 'conv_forward')

    def backward_dx(self, dout, n, act_in, dx):
        """dx [n*IH*IW][IC] = conv_transpose(dout) masked by relu'(act_in) — act_in is the (post-ReLU) activation this layer read."""
        a = _operand(MODE_COL2IM, dout, 0, self.geom)
        _lib.check(_lib.lib().pfa_igemm_rows(C.byref(a), n * self.IH * self.IW, self.KH * self.KW * self.OC, _lib.ptr(self.w_dx), self.KP, self.IC,
                                             _lib.ptr(dx), self.IC, EPI_MASK, None, _lib.ptr(act_in), self.IC, _lib.stream_handle()), 'conv_dx')

    def backward_dw(self, x, n, dout, gw, gb, accumulate, ws):
        """gw (torch layout [OC][IC][KH][KW]) (+)= dout^T im2col(x); gb (+)= column sums of dout."""
        a = _operand(MODE_IM2COL_U8 if self.u8 else MODE_IM2COL_F32, x, 0, self.geom)
        _lib.check(_lib.lib().pfa_igemm_weights(C.byref(a), self.out_rows(n), self.K, _lib.ptr(dout), self.OC, self.OC, _lib.ptr(gw),
                                                3 if self.u8 else 2, 1 if accumulate else 0, _lib.ptr(gb), _lib.ptr(ws), _lib.stream_handle()), 'conv_dw')

    def dw_workspace(self, n):
        return _lib.lib().pfa_igemm_weights_workspace_bytes(self.out_rows(n), self.K, self.OC)


class LinearLayer:
    """Linear(K, N) (+ ReLU) on dense rows.  `flatten` = (C, H, W): the input is the NHWC form of an NCHW tensor nn.Flatten'ed."""

    def __init__(self, weight, bias, relu, flatten, device):
        self.w, self.b, self.relu, self.flatten = weight, bias, relu, flatten
        self.N, self.K = weight.shape
        self.w_t = torch.empty(self.K, self.N, device=device)          # dX B [k][n]
        self.w_p = torch.empty(self.N, self.K, device=device) if flatten else None   # forward B [n][k'] (columns in NHWC order)

    def pack(self):
        L = _lib.lib()
        if self.flatten:
            c, h, w = self.flatten
            _lib.check(L.pfa_cnn_pack_fc(_lib.ptr(self.w), self.N, c, h * w, _lib.ptr(self.w_p), _lib.ptr(self.w_t), _lib.stream_handle()), 'pack_fc')
        else:
            _lib.check(L.pfa_cnn_transpose(_lib.ptr(self.w), self.N, self.K, _lib.ptr(self.w_t), _lib.stream_handle()), 'transpose')

    def forward(self, x, rows, out):
        a = _operand(MODE_DENSE, x, self.K)
        wb = self.w_p if self.flatten else self.w
        _lib.check(_lib.lib().pfa_igemm_rows(C.byref(a), rows, self.K, _lib.ptr(wb), self.K, self.N, _lib.ptr(out), self.N,
                                             EPI_BIAS_RELU if self.relu else EPI_BIAS, _lib.ptr(self.b), None, 0, _lib.stream_handle()), 'linear_forward')

    def backward_dx(self, dout, rows, act_in, dx):
        a = _operand(MODE_DENSE, dout, self.N)
        _lib.check(_lib.lib().pfa_igemm_rows(C.byref(a), rows, self.N, _lib.ptr(self.w_t), self.N, self.K, _lib.ptr(dx), self.K, EPI_MASK, None,
                                             _lib.ptr(act_in), self.K, _lib.stream_handle()), 'linear_dx')

    def backward_dw(self, x, rows, dout, gw, gb, accumulate, ws):
        geom = (self.flatten[0], self.flatten[1], self.flatten[2], 0, 0, 0, 0, 0, 0) if self.flatten else (0,) * 9
        a = _operand(MODE_DENSE, x, self.K, geom)
        _lib.check(_lib.lib().pfa_igemm_weights(C.byref(a), rows, self.K, _lib.ptr(dout), self.N, self.N, _lib.ptr(gw), 4 if self.flatten else 1,
                                                1 if accumulate else 0, _lib.ptr(gb), _lib.ptr(ws), _lib.stream_handle()), 'linear_dw')

    def dw_workspace(self, rows):
        return _lib.lib().pfa_igemm_weights_workspace_bytes(rows, self.K, self.N)


class Engine:
    """Forward / update of the NatureCNN policy over a ConvParams buffer.  `chunk` = frames per kernel batch (bounds the
    activation memory: ~170 KB per frame with the gradients)."""

    def __init__(self, cp, experience=None, chunk=8192):
        self.cp, self.dev = cp, cp.flat.device
        self.fp = cp
        self.experience = experience      # the trainer's buffers (uint8 obs [B][frame_bytes]) for update()
        self.state = None                 # (no recurrent state; the trainer's epoch loop resets it for either engine)
        self.norm_partials = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        v = cp.views
        dev = self.dev
        self.conv1 = ConvLayer(v['network.0.weight'], v['network.0.bias'], 84, 84, 4, True, dev)
        self.conv2 = ConvLayer(v['network.2.weight'], v['network.2.bias'], 20, 20, 2, False, dev)
        self.conv3 = ConvLayer(v['network.4.weight'], v['network.4.bias'], 9, 9, 1, False, dev)
        self.fc = LinearLayer(v['network.7.weight'], v['network.7.bias'], True, (64, 7, 7), dev)
        self.layers = [self.conv1, self.conv2, self.conv3, self.fc]
        self.frame_bytes = cp.framestack * 84 * 84
        self.chunk = 0
        self.packed_version = -1
        self.version = 0          # bumped by whoever changes the weights (optimizer step, checkpoint load)
        self._alloc(chunk)

    def _alloc(self, chunk):
        if chunk <= self.chunk:
            return
        dev, n = self.dev, chunk
        self.chunk = n
        self.a1 = torch.empty(n * 400, 32, device=dev)
        self.a2 = torch.empty(n * 81, 64, device=dev)
        self.a3 = torch.empty(n * 49, 64, device=dev)
        self.h = torch.empty(n, 512, device=dev)
        self.d1 = torch.empty_like(self.a1)
        self.d2 = torch.empty_like(self.a2)
        self.d3 = torch.empty_like(self.a3)
        self.dh = torch.empty_like(self.h)
        self.dout = torch.empty(n, 16, device=dev)
        self.frames = torch.empty(n, self.frame_bytes, dtype=torch.uint8, device=dev)
        L = _lib.lib()
        ws = max([self.conv1.dw_workspace(n), self.conv2.dw_workspace(n), self.conv3.dw_workspace(n), self.fc.dw_workspace(n),
                  L.pfa_igemm_weights_workspace_bytes(n, 512, 16)])
        self.ws = torch.empty(ws, dtype=torch.uint8, device=dev)
        self.ws_loss = torch.empty(L.pfa_cnn_heads_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.g16 = torch.empty(16, 512, device=dev)
        self.gb16 = torch.empty(16, device=dev)

    def pack(self):
        if self.packed_version != self.version:
            for layer in self.layers:
                layer.pack()
            self.packed_version = self.version

    # ------------------------------------------------------------------------------------------------------------ forward
    def forward(self, frames, n):
        """frames uint8 [n][F*84*84] (NCHW per frame) -> self.h[:n] (hidden, post-ReLU); keeps a1/a2/a3 for a backward."""
        assert n <= self.chunk
        self.pack()
        self.conv1.forward(frames, n, self.a1)
        self.conv2.forward(self.a1, n, self.a2)
        self.conv3.forward(self.a2, n, self.a3)
        self.fc.forward(self.a3, n, self.h)
        return self.h[:n]

    def policy_step(self, frames, n, noise, key, row_offset, actions, logprob, entropy, value):
        """policy(obs) in rollout mode for n frames (any n: processed in chunks)."""
        L = _lib.lib()
        v = self.cp.views
        for lo in range(0, n, self.chunk):
            m = min(self.chunk, n - lo)
            h = self.forward(frames[lo:lo + m], m)
            nz = None if noise is None else noise[lo:lo + m]
            _lib.check(L.pfa_cnn_heads_sample(_lib.ptr(h), m, _lib.ptr(v['actor.weight']), _lib.ptr(v['actor.bias']), _lib.ptr(v['value_fn.weight']),
                                              _lib.ptr(v['value_fn.bias']), self.cp.num_actions, _lib.ptr(nz), C.byref(key), row_offset + lo,
                                              _lib.ptr(actions[lo:lo + m]), _lib.ptr(logprob[lo:lo + m]),
                                              None if entropy is None else _lib.ptr(entropy[lo:lo + m]), _lib.ptr(value[lo:lo + m]),
                                              _lib.stream_handle()), 'cnn_heads_sample')

    # ------------------------------------------------------------------------------------------------------------- update
    def backward(self, frames, m, dh_pre, gv, acc):
        """Back-propagate d loss / d (pre-ReLU hidden) [m][512] through Linear(3136,512) and the three conv layers of the chunk
        whose forward just ran (activations a1/a2/a3 live); weight / bias gradients into the views `gv` (accumulate = acc)."""
        self.fc.backward_dw(self.a3, m, dh_pre, gv['network.7.weight'], gv['network.7.bias'], acc, self.ws)
        self.fc.backward_dx(dh_pre, m, self.a3, self.d3)            # d3 masked by relu'(a3)
        self.conv3.backward_dw(self.a2, m, self.d3, gv['network.4.weight'], gv['network.4.bias'], acc, self.ws)
        self.conv3.backward_dx(self.d3, m, self.a2, self.d2)
        self.conv2.backward_dw(self.a1, m, self.d2, gv['network.2.weight'], gv['network.2.bias'], acc, self.ws)
        self.conv2.backward_dx(self.d2, m, self.a1, self.d1)
        self.conv1.backward_dw(frames, m, self.d1, gv['network.0.weight'], gv['network.0.bias'], acc, self.ws)

    def update(self, mb, hp, adv_stats, global_mb_rows, grads, B):
        """Forward + PPO loss + backward for minibatch `mb` of the trainer's experience (clean_pufferl.py:179-244 up to
        loss.backward()); writes the flat gradient (named_parameters order) + the loss tail."""
        return self.update_from(self.experience.c, self.experience.obs, B, mb, hp, adv_stats, global_mb_rows, grads)

    def clip_adam(self, grads, opt, max_grad_norm, loss_acc, loss_scale):
        """clip_grad_norm_ + optimizer.step() (clean_pufferl.py:240-244) on the flat buffer; the packed weight forms go stale."""
        L = _lib.lib()
        stream = _lib.stream_handle()
        cp = self.cp
        n = self.norm_partials.numel()
        _lib.check(L.pfa_sumsq_partials(_lib.ptr(grads), cp.count, _lib.ptr(self.norm_partials), n, stream), 'sumsq')
        opt.step_count += 1
        g = opt.param_groups[0]
        _lib.check(L.pfa_adam_clip_step(_lib.ptr(cp.flat), _lib.ptr(grads), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), cp.count,
                                        float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), opt.step_count,
                                        float(max_grad_norm), 1.0, C.c_void_p(grads.data_ptr() + 4 * cp.count), _lib.ptr(loss_acc), loss_scale,
                                        _lib.ptr(self.norm_partials), n, stream), 'adam')
        self.version += 1

    def update_from(self, exp_c, obs_u8, B, mb, hp, adv_stats, global_mb_rows, grads):
        L = _lib.lib()
        cp = self.cp
        v = cp.views
        gv = cp.split(grads[:cp.count])
        mbs = B // hp.num_minibatches
        tail = grads[cp.count:cp.count + 16]
        stream = _lib.stream_handle()
        for ci, q0 in enumerate(range(0, mbs, self.chunk)):
            m = min(self.chunk, mbs - q0)
            acc = ci > 0
            _lib.check(L.pfa_cnn_gather_frames(_lib.ptr(obs_u8), self.frame_bytes, B, mb, C.byref(hp), q0, m, _lib.ptr(self.frames), stream), 'gather')
            h = self.forward(self.frames, m)
            _lib.check(L.pfa_cnn_heads_loss(_lib.ptr(h), C.byref(exp_c), B, mb, q0, m, _lib.ptr(v['actor.weight']), _lib.ptr(v['actor.bias']),
                                            _lib.ptr(v['value_fn.weight']), _lib.ptr(v['value_fn.bias']), cp.num_actions, C.byref(hp),
                                            _lib.ptr(adv_stats), global_mb_rows, _lib.ptr(self.dout), _lib.ptr(self.dh), _lib.ptr(tail),
                                            1 if acc else 0, _lib.ptr(self.ws_loss), stream), 'cnn_heads_loss')
            # heads: dW = dout^T h ([16][512]: rows < A actor, row A value_fn), db = column sums of dout
            a = _operand(MODE_DENSE, h, 512)
            _lib.check(L.pfa_igemm_weights(C.byref(a), m, 512, _lib.ptr(self.dout), 16, 16, _lib.ptr(self.g16), 1, 1 if acc else 0,
                                           _lib.ptr(self.gb16), _lib.ptr(self.ws), stream), 'heads_dw')
            dh_pre = self.dh                       # already w.r.t. the pre-ReLU hidden (masked in the heads kernel)
            self.backward(self.frames, m, dh_pre, gv, acc)
        A = cp.num_actions
        gv['actor.weight'].copy_(self.g16[:A])
        gv['actor.bias'].copy_(self.gb16[:A])
        gv['value_fn.weight'].copy_(self.g16[A:A + 1])
        gv['value_fn.bias'].copy_(self.gb16[A:A + 1])
