"""Width-general policy engine: every policy shape of the reference that the fused 128-wide kernels do not cover.

  pufferlib.models.Default(env, hidden_size=H)                         models.py:24-39    any H that is a multiple of 16, any flat
                                                                                          observation width, up to 63 logits
  pufferlib.models.LSTMWrapper(env, policy, input_size=I, hidden_size=Hl)  models.py:64-82  any I (= the encoder width), Hl
  environments/atari/torch.py:4-6  Recurrent = LSTMWrapper(512, 512) over the NatureCNN   (what config.yaml's atari section trains)

and the training-mode call ``policy(obs, action=...)`` of frameworks.cleanrl.Policy / RecurrentPolicy (cleanrl.py:60-66,87-93)
for EVERY policy, the 128-wide ones included (`Evaluator`).

Every product is a launch of the fp32-MFMA implicit-GEMM kernels of csrc/igemm.hip (pfa_igemm_rows: C = A B^T with bias / ReLU /
relu'-mask epilogues; pfa_igemm_weights: the weight-gradient contraction over the rows, scattered to torch's layout, with the bias
gradient from the same pass); the row-wise pieces (sampling, PPO loss, the LSTM cell and its back-propagation, the row-order
changes) are csrc/general.hip.  The host side only sequences launches and keeps the packed operand forms current:

  encoder   Linear(obs, H) + ReLU  on zero-padded rows [rows][Kp]           (or cnn.Engine's conv stack + Linear(3136, 512))
  LSTM      gates = [x | h] Wcat^T + (b_ih + b_hh), Wcat = [W_ih | W_hh]     one GEMM per step, then pfa_lstm_cell_forward
  heads     out = feature W2v^T + b2v, W2v = decoder rows, value row, zero rows up to a multiple of 16

Row order of a chunk is TIME-MAJOR (row t * R + k = segment k at step t) so that every LSTM step is a contiguous block.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .models import decoder_heads, find_cnn, find_lstm, find_mlp

MODE_DENSE = 0
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK = 0, 1, 2, 3
MAX_LOGITS = 63


def _round_up(x, a):
    return (x + a - 1) // a * a


def _operand(t, lda, geom=(0,) * 9):
    return _lib.IgemmOperand(MODE_DENSE, 0, t.data_ptr(), lda, *geom)


def gemm_rows(a, lda, rows, K, B, ldb, N, out, ldc, epi=EPI_NONE, bias=None, mask=None, ldmask=0):
    """out[rows][:N] (row stride ldc) = epilogue(a[rows][:K] B[N][:K]^T)."""
    if rows == 0:
        return
    _lib.check(_lib.lib().pfa_igemm_rows(C.byref(_operand(a, lda)), rows, K, _lib.ptr(B), ldb, N, _lib.ptr(out), ldc, epi, _lib.ptr(bias),
                                         _lib.ptr(mask), ldmask, _lib.stream_handle()), 'gemm_rows')


def gemm_weights(a, lda, rows, K, D, ldd, N, out, accumulate, bias_out, ws):
    """out [N][K] (torch Linear layout) (+)= D[rows][:N]^T a[rows][:K]; bias_out [N] (+)= column sums of D."""
    _lib.check(_lib.lib().pfa_igemm_weights(C.byref(_operand(a, lda)), rows, K, _lib.ptr(D), ldd, N, _lib.ptr(out), 1, 1 if accumulate else 0,
                                            _lib.ptr(bias_out), _lib.ptr(ws), _lib.stream_handle()), 'gemm_weights')


def rows_perm(src, lds, dst, ldd, rows, cols, segments=0, steps=0, to_time_major=False, act=None, lda=0):
    _lib.check(_lib.lib().pfa_rows_perm(_lib.ptr(src), lds, _lib.ptr(dst), ldd, _lib.ptr(act), lda, rows, cols, segments, steps,
                                        1 if to_time_major else 0, _lib.stream_handle()), 'rows_perm')


def pack_heads(nvec, multidiscrete):
    """pfa_mlp_dims.heads: nibble-packed head sizes, 0 for one Discrete head."""
    if not multidiscrete:
        return 0
    if len(nvec) > 8 or max(nvec) > 15:
        raise NotImplementedError(f'MultiDiscrete heads {nvec}: up to 8 heads of up to 15 choices (one Discrete head may have {MAX_LOGITS})')
    return sum(n << (4 * h) for h, n in enumerate(nvec))


class GeneralParams:
    """One flat fp32 device buffer with ALL parameters of the policy module in named_parameters() order and torch shapes (what
    the reference's optimizer iterates over); every module parameter is re-pointed at its view.  Interface of models.FlatParams."""

    def __init__(self, policy_module, device, obs_stride=None):
        self.module = policy_module
        self.cnn = find_cnn(policy_module)
        self.lstm = find_lstm(policy_module)
        self.mlp = None if self.cnn is not None else find_mlp(policy_module)
        if self.lstm is not None and (self.lstm.num_layers != 1 or self.lstm.bidirectional or self.lstm.batch_first):
            raise NotImplementedError('nn.LSTM(input, hidden, num_layers=1), time-first, unidirectional (models.py:76)')
        if self.cnn is not None:
            self.kind = 'cnn'
            self.nvec = [int(self.cnn.actor.weight.shape[0])]
            self.multidiscrete = False
            self.features = int(self.cnn.actor.weight.shape[1])
            convs = [m for m in self.cnn.network if isinstance(m, nn.Conv2d)]
            self.framestack = int(convs[0].weight.shape[1])
            self.obs_dim = self.obs_stride = self.framestack * 84 * 84
        else:
            self.kind = 'mlp'
            self.nvec = decoder_heads(self.mlp)
            self.multidiscrete = not isinstance(self.mlp.decoder, nn.Linear)
            self.features, self.obs_dim = (int(x) for x in self.mlp.encoder.weight.shape)
            self.obs_stride = int(obs_stride) if obs_stride else _round_up(self.obs_dim, 16)     # the vecenv's row stride
            if self.obs_stride % 16 != 0 or self.obs_stride < self.obs_dim:
                raise ValueError(f'observation row stride {self.obs_stride}: a multiple of 16 that holds {self.obs_dim} floats')
            if self.features % 16 != 0:
                raise NotImplementedError(f'hidden_size {self.features}: a multiple of 16')
        self.num_actions = sum(self.nvec)
        if self.num_actions > MAX_LOGITS:
            raise NotImplementedError(f'{self.num_actions} logits: the head kernels take up to {MAX_LOGITS}')
        self.heads = pack_heads(self.nvec, self.multidiscrete)
        if self.lstm is not None:
            if self.lstm.input_size != self.features:
                raise ValueError(f'nn.LSTM input_size {self.lstm.input_size} != encoder width {self.features}')
            if self.lstm.hidden_size % 16 != 0:
                raise NotImplementedError('LSTM hidden_size: a multiple of 16')
        self.head_in = self.lstm.hidden_size if self.lstm is not None else self.features
        self.names = [n for n, _ in policy_module.named_parameters()]
        self.count = int(sum(p.numel() for _, p in policy_module.named_parameters()))
        self.flat = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.views = self.split(self.flat)
        with torch.no_grad():
            for name, p in policy_module.named_parameters():
                v = self.views[name]
                v.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = v
            if self.lstm is not None:
                self.lstm._flat_weights = [getattr(self.lstm, n) for n in self.lstm._flat_weights_names]
        # dims of the kernels' MLP descriptor, for the host code that only reads num_actions / heads from it
        self.dims = _lib.MlpDims(int(min(self.obs_dim, 2 ** 31 - 1)), int(self.obs_stride), int(self.features), int(self.num_actions), self.heads)

    def split(self, flat):
        out, o = {}, 0
        for name, p in self.module.named_parameters():
            n = p.numel()
            out[name] = flat[o:o + n].view(p.shape)
            o += n
        return out

    def flat_like(self):
        return torch.zeros_like(self.flat)

    def unpack_actions(self, packed):
        if not self.multidiscrete:
            return packed
        shifts = torch.arange(0, 4 * len(self.nvec), 4, device=packed.device, dtype=packed.dtype)
        return (packed.unsqueeze(-1) >> shifts) & 15

    def name_of(self, param):
        """Full dotted name of a module parameter (views are keyed by it)."""
        for n, p in self.module.named_parameters():
            if p is param:
                return n
        raise KeyError('parameter not in the policy module')


class _ConvAdapter:
    """What cnn.Engine reads from a ConvParams, served from a GeneralParams (the conv stack's parameters by their short names)."""
    multidiscrete = False

    def __init__(self, gp):
        self.gp = gp
        self.flat = gp.flat
        self.framestack = gp.framestack
        self.num_actions = gp.num_actions
        self.count = gp.count
        self.prefix = {}
        for short, p in gp.cnn.named_parameters():
            self.prefix[short] = gp.name_of(p)
        self.views = {short: gp.views[full] for short, full in self.prefix.items()}

    def split(self, flat):
        full = self.gp.split(flat)
        return {short: full[name] for short, name in self.prefix.items()}


class Net:
    """The packed operand forms and the forward / backward launch sequences of one policy over a dict of parameter VIEWS (so the
    same code serves a GeneralParams buffer and the kernel-layout buffer of models.FlatParams)."""

    def __init__(self, kind, views, names, obs_dim, obs_stride, features, num_actions, heads, lstm_sizes, device, conv_engine=None):
        """names: dict role -> parameter name in `views`: enc_w, enc_b (mlp), actors [(w, b), ...], value_w, value_b,
        w_ih, w_hh, b_ih, b_hh (when lstm_sizes = (I, Hl))."""
        self.kind, self.views, self.names, self.dev = kind, views, names, device
        self.obs_dim, self.Kp, self.F, self.A, self.heads = obs_dim, obs_stride, features, num_actions, heads
        self.NO = _round_up(num_actions + 1, 16)
        self.lstm = lstm_sizes
        self.FH = lstm_sizes[1] if lstm_sizes else features
        self.conv = conv_engine
        dev = device
        if kind == 'mlp':
            self.w1p = torch.zeros(features, self.Kp, device=dev)
        self.w2v = torch.zeros(self.NO, self.FH, device=dev)
        self.b2v = torch.zeros(self.NO, device=dev)
        self.w2vT = torch.zeros(self.FH, self.NO, device=dev)
        if lstm_sizes:
            I, Hl = lstm_sizes
            self.wcat = torch.zeros(4 * Hl, I + Hl, device=dev)
            self.bcat = torch.zeros(4 * Hl, device=dev)
            self.wcatT = torch.zeros(I + Hl, 4 * Hl, device=dev)
        self.packed_version = None
        self.version = 0
        self._probe = next(iter(views.values()))     # any parameter view: views of one flat buffer share its in-place version counter

    def pack(self):
        """Refresh the padded / stacked / transposed operand copies after the parameters changed (optimizer step, load)."""
        # `version` is bumped by the trainer's own writers (optimizer step kernels, checkpoint loader: raw-pointer writes torch does
        # not see); torch-level in-place writes to any parameter view (p.copy_(), load_state_dict on an inner module, dist.broadcast,
        # a torch optimizer) bump the shared version counter of the flat buffer the views alias — both are part of the key
        key = (self.version, self._probe._version)
        if self.packed_version == key:
            return
        v, nm = self.views, self.names
        with torch.no_grad():
            if self.kind == 'mlp':
                self.w1p[:, :self.obs_dim].copy_(v[nm['enc_w']])
            r = 0
            for w, b in nm['actors']:
                n = v[w].shape[0]
                self.w2v[r:r + n].copy_(v[w])
                self.b2v[r:r + n].copy_(v[b])
                r += n
            self.w2v[r].copy_(v[nm['value_w']][0])
            self.b2v[r].copy_(v[nm['value_b']][0])
            self.w2vT.copy_(self.w2v.t())
            if self.lstm:
                I, Hl = self.lstm
                self.wcat[:, :I].copy_(v[nm['w_ih']])
                self.wcat[:, I:].copy_(v[nm['w_hh']])
                torch.add(v[nm['b_ih']], v[nm['b_hh']], out=self.bcat)
                self.wcatT.copy_(self.wcat.t())
        if self.conv is not None:
            self.conv.version = self.version
            self.conv.pack()
        self.packed_version = key

    # ---------------------------------------------------------------------------------------------------------------- forward
    def encode(self, obs, rows, out, ldo):
        """Features (post-ReLU) of `rows` observation rows into out[rows][:F] (row stride ldo).  obs: f32 [rows][Kp] (mlp, zero padded)
        or uint8 frames [rows][frame_bytes] (cnn, at most conv.chunk rows)."""
        if self.kind == 'mlp':
            gemm_rows(obs, self.Kp, rows, self.Kp, self.w1p, self.Kp, self.F, out, ldo, EPI_BIAS_RELU, self.views[self.names['enc_b']])
        else:
            h = self.conv.forward(obs, rows)
            rows_perm(h, self.F, out, ldo, rows, self.F)

    def lstm_step(self, xh, gates, c_prev, c_out, h_out, ldh, h_out2=None, ldh2=0):
        """One nn.LSTM step on the rows of xh = [x | h_prev] [R][I + Hl]: gates [R][4 Hl] keeps the activated gates."""
        I, Hl = self.lstm
        R = xh.shape[0]
        gemm_rows(xh, I + Hl, R, I + Hl, self.wcat, I + Hl, 4 * Hl, gates, 4 * Hl, EPI_BIAS, self.bcat)
        _lib.check(_lib.lib().pfa_lstm_cell_forward(_lib.ptr(gates), _lib.ptr(c_prev), _lib.ptr(c_out), _lib.ptr(h_out), ldh, _lib.ptr(h_out2), ldh2,
                                                    R, Hl, _lib.stream_handle()), 'lstm_cell_forward')

    def head_outputs(self, feat, ldf, rows, out):
        """out [rows][NO] = feat W2v^T + b2v: logits of all heads, then the value, then zero columns."""
        gemm_rows(feat, ldf, rows, self.FH, self.w2v, self.FH, self.NO, out, self.NO, EPI_BIAS, self.b2v)


def _net_for_general(gp, conv_chunk=0):
    names = {}
    if gp.kind == 'mlp':
        names['enc_w'], names['enc_b'] = gp.name_of(gp.mlp.encoder.weight), gp.name_of(gp.mlp.encoder.bias)
        decs = list(gp.mlp.decoder) if gp.multidiscrete else [gp.mlp.decoder]
        names['actors'] = [(gp.name_of(d.weight), gp.name_of(d.bias)) for d in decs]
        names['value_w'], names['value_b'] = gp.name_of(gp.mlp.value_head.weight), gp.name_of(gp.mlp.value_head.bias)
        conv = None
    else:
        names['actors'] = [(gp.name_of(gp.cnn.actor.weight), gp.name_of(gp.cnn.actor.bias))]
        names['value_w'], names['value_b'] = gp.name_of(gp.cnn.value_fn.weight), gp.name_of(gp.cnn.value_fn.bias)
        from . import cnn
        conv = cnn.Engine(_ConvAdapter(gp), chunk=max(conv_chunk, 256))
    lstm_sizes = None
    if gp.lstm is not None:
        for role, attr in (('w_ih', 'weight_ih_l0'), ('w_hh', 'weight_hh_l0'), ('b_ih', 'bias_ih_l0'), ('b_hh', 'bias_hh_l0')):
            names[role] = gp.name_of(getattr(gp.lstm, attr))
        lstm_sizes = (gp.lstm.input_size, gp.lstm.hidden_size)
    return Net(gp.kind, gp.views, names, gp.obs_dim, gp.obs_stride, gp.features, gp.num_actions, gp.heads, lstm_sizes, gp.flat.device, conv)


def net_for_flat(fp):
    """The same launch sequences over the kernel-layout buffer of models.FlatParams (Default(128) [+ LSTM(128, 128)]): used for
    policy(obs, action=...) on the policies whose training runs in the fused kernels."""
    views = dict(fp.views)
    names = dict(enc_w='encoder.weight', enc_b='encoder.bias', value_w='value_head.weight', value_b='value_head.bias')
    names['actors'] = ([(f'decoder.{h}.weight', f'decoder.{h}.bias') for h in range(len(fp.nvec))] if fp.multidiscrete
                       else [('decoder.weight', 'decoder.bias')])
    lstm_sizes = None
    if fp.lstm is not None:
        for role, attr in (('w_ih', 'weight_ih_l0'), ('w_hh', 'weight_hh_l0'), ('b_ih', 'bias_ih_l0'), ('b_hh', 'bias_hh_l0')):
            views['recurrent.' + attr] = fp.lstm_views[attr]
            names[role] = 'recurrent.' + attr
        lstm_sizes = (fp.lstm.input_size, fp.lstm.hidden_size)
    H = int(fp.views['encoder.bias'].shape[0])
    return Net('mlp', views, names, fp.obs_dim, _round_up(fp.obs_dim, 16), H, fp.num_actions, fp.dims.heads, lstm_sizes, fp.flat.device)


class Evaluator:
    """policy(obs, action=...) — sample_logits with given actions (cleanrl.py:38-44) behind Policy.forward / RecurrentPolicy.forward
    (cleanrl.py:60-66,87-93): (action, logprob, entropy, value[, state]).  No autograd graph is attached: the gradients of these
    quantities live in clean_pufferl.train()'s kernels."""

    def __init__(self, net):
        self.net = net

    def _rows(self, x):
        net = self.net
        rows = x.shape[0]
        if net.kind == 'cnn':
            return x.reshape(rows, -1).to(torch.uint8).contiguous(), rows
        x2 = x.reshape(rows, -1)
        if x2.shape[1] != net.obs_dim:
            raise ValueError(f'observation rows of {x2.shape[1]} values, the policy reads {net.obs_dim}')
        src = torch.zeros(rows, net.Kp, dtype=torch.float32, device=net.dev)
        src[:, :net.obs_dim] = x2.float()
        return src, rows

    def forward(self, x, action=None, state=None, noise=None, key=None, row_offset=0, obs_rank=None):
        """x: (B, obs...) or, for a recurrent policy, (B, TT, obs...) like LSTMWrapper.forward (models.py:84-111), told apart by
        `obs_rank` = len(single_observation_space.shape) (None: x is (B, ...) with one step); state (h, c) of shape (1, B, Hl) or
        None.  action None -> samples (noise [rows][A] or the Philox key)."""
        net = self.net
        net.pack()
        L = _lib.lib()
        dev = net.dev
        if not x.is_cuda:
            x = x.to(dev)
        TT = 1
        B = x.shape[0]
        if net.lstm and obs_rank is not None:
            if x.dim() == obs_rank + 2:
                TT = x.shape[1]
                x = x.reshape(B * TT, *x.shape[2:])          # row b * TT + t, as models.py:99 flattens it
            elif x.dim() != obs_rank + 1:
                raise ValueError('Invalid input tensor shape', tuple(x.shape))
        src, rows = self._rows(x)
        F, FH, NO = net.F, net.FH, net.NO
        if net.lstm:
            I, Hl = net.lstm
            xh = torch.zeros(TT + 1, B, I + Hl, device=dev)
            feat = torch.empty(rows, F, device=dev)
            self._encode_chunks(src, rows, feat, F)
            rows_perm(feat, F, xh, I + Hl, rows, F, B, TT, True)             # (b, t) rows -> time-major, into the x columns
            c = torch.zeros(TT + 1, B, Hl, device=dev)
            if state is not None and state[0] is not None:
                xh[0, :, I:].copy_(state[0].reshape(B, Hl))
                c[0].copy_(state[1].reshape(B, Hl))
            hs = torch.empty(TT, B, Hl, device=dev)
            gates = torch.empty(B, 4 * Hl, device=dev)
            for t in range(TT):
                net.lstm_step(xh[t], gates, c[t], c[t + 1], hs[t], Hl, xh[t + 1, :, I:], I + Hl)
            head_in = torch.empty(rows, Hl, device=dev)
            rows_perm(hs, Hl, head_in, Hl, rows, Hl, B, TT, False)           # back to (b, t) row order, like models.py:107-108
            new_state = (hs[TT - 1].unsqueeze(0).clone(), c[TT].unsqueeze(0).clone())
        else:
            head_in = torch.empty(rows, F, device=dev)
            self._encode_chunks(src, rows, head_in, F)
            new_state = None
        out = torch.empty(rows, NO, device=dev)
        net.head_outputs(head_in, FH, rows, out)
        logprob = torch.empty(rows, device=dev)
        entropy = torch.empty(rows, device=dev)
        value = torch.empty(rows, device=dev)
        if action is None:
            actions = torch.empty(rows, dtype=torch.int64, device=dev)
            if noise is not None:
                noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            _lib.check(L.pfa_heads_rows_sample(_lib.ptr(out), NO, rows, net.A, net.heads, _lib.ptr(noise), C.byref(key) if key is not None else None,
                                               row_offset, _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(entropy), _lib.ptr(value),
                                               _lib.stream_handle()), 'heads_rows_sample')
            packed = actions
        else:
            a = action.to(dev)
            # given actions must name a choice of their head: the reference's gather (cleanrl.py:38-44) raises on anything else, and the
            # kernels' nibble packing would let a choice >= 16 spill into the neighbouring heads (one host round trip: not the rollout path)
            if net.heads:                                            # (rows, heads) choices -> the kernels' nibble packing
                a = a.reshape(rows, -1).to(torch.int64)
                sizes = torch.tensor([(net.heads >> (4 * h)) & 15 for h in range(a.shape[1])], device=dev, dtype=torch.int64)
                if sizes.numel() != a.shape[1] or int(sizes.min()) == 0 or bool(((a < 0) | (a >= sizes)).any()):
                    raise IndexError(f'action out of range for MultiDiscrete heads {sizes.tolist()}')
                shifts = torch.arange(0, 4 * a.shape[1], 4, device=dev, dtype=torch.int64)
                packed = (a << shifts).sum(dim=1)
            else:
                packed = a.reshape(rows).to(torch.int64)
                if rows and bool(((packed < 0) | (packed >= net.A)).any()):
                    raise IndexError(f'action out of range for Discrete({net.A})')
            packed = packed.contiguous()
            _lib.check(L.pfa_heads_rows_eval(_lib.ptr(out), NO, rows, net.A, net.heads, _lib.ptr(packed), _lib.ptr(logprob), _lib.ptr(entropy),
                                             _lib.ptr(value), _lib.stream_handle()), 'heads_rows_eval')
        return packed, logprob, entropy, value.unsqueeze(1), new_state

    def _encode_chunks(self, src, rows, out, ldo):
        net = self.net
        if net.kind == 'mlp':
            net.encode(src, rows, out, ldo)
            return
        step = 2048
        net.conv._alloc(min(rows, step))
        for lo in range(0, rows, step):
            m = min(step, rows - lo)
            net.encode(src[lo:lo + m], m, out[lo:lo + m], ldo)


USE_TILE_VIEW = True


def tile_view(gp):
    """pfa_mlp_view of a plain Default(hidden 64 / 128 / 256 / 512, one Discrete head of <= 15 actions, rows of <= 64 floats): the
    rollout-mode forward of such a policy runs the register-resident tile kernels of csrc/rollout.hip — the standalone forward for
    the protocol path (policy(obs), Engine.policy_step) and the persistent fused rollout on vector.Squared — straight from the
    module's tensors (their addresses never change: every parameter is a view of the flat buffer).  None for any other shape: the
    GEMM path.  (`USE_TILE_VIEW = False` on this module sends every shape down the GEMM path: the tests compare the two.)"""
    if getattr(gp, '_tile_view', False) is not False:
        return gp._tile_view
    view = None
    if (isinstance(gp, GeneralParams) and gp.kind == 'mlp' and gp.lstm is None and not gp.multidiscrete and gp.obs_stride in (16, 32, 64)
            and gp.num_actions <= 15 and USE_TILE_VIEW):
        m, v = gp.mlp, gp.views
        t = {k: v[gp.name_of(p)] for k, p in (('w1', m.encoder.weight), ('b1', m.encoder.bias), ('w2', m.decoder.weight), ('b2', m.decoder.bias),
                                              ('wv', m.value_head.weight), ('bv', m.value_head.bias))}
        cand = _lib.MlpView(t['w1'].data_ptr(), int(gp.obs_dim), int(gp.obs_dim), int(gp.obs_stride), int(gp.features), int(gp.num_actions), 0,
                            t['b1'].data_ptr(), t['w2'].data_ptr(), t['b2'].data_ptr(), t['wv'].data_ptr(), t['bv'].data_ptr())
        if _lib.lib().pfa_mlp_view_supported(C.byref(cand)):
            view = cand
    gp._tile_view = view
    return view


class Engine:
    """Rollout step and PPO update of a general policy for one (experience, vecenv) pair; interface of lstm.Engine / cnn.Engine
    as clean_pufferl drives them: policy_step, update, clip_adam, state."""

    def __init__(self, gp, experience=None, num_agents=0, frames_per_chunk=8192):
        self.gp = self.fp = gp
        self.dev = gp.flat.device
        self.experience = experience
        self.net = _net_for_general(gp, conv_chunk=min(frames_per_chunk, 2048))
        self.state = None
        self.norm_partials = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        self.frames_per_chunk = frames_per_chunk
        self.lstm_h = self.lstm_c = None
        self._step_rows = 0
        self._upd = None
        self.mlp_view = tile_view(gp)
        # ... and the fused forward + loss + backward kernel of csrc/ppo_wide.hip for the widths it is built for (64 / 256 / 512)
        self.wide_ws = self._wide_gview = None
        # (pfa_ppo_wide_grad tiles minibatches in 16-row blocks: any other minibatch size trains through the GEMM path of update())
        if (self.mlp_view is not None
                and (experience is None or experience.minibatch_size % 16 == 0)
                and _lib.lib().pfa_ppo_wide_supported(C.byref(self.mlp_view))):
            self.wide_ws = torch.empty(int(_lib.lib().pfa_ppo_wide_workspace_bytes(C.byref(self.mlp_view))), dtype=torch.uint8, device=self.dev)
        if num_agents:
            self.reset_state(num_agents)

    # ------------------------------------------------------------------------------------------------------------ rollout
    def reset_state(self, num_agents):
        if self.net.lstm:
            Hl = self.net.lstm[1]
            self.lstm_h = torch.zeros(1, num_agents, Hl, device=self.dev)      # Experience.lstm_h / lstm_c (clean_pufferl.py:407-412)
            self.lstm_c = torch.zeros(1, num_agents, Hl, device=self.dev)

    def _alloc_step(self, n):
        if n <= self._step_rows:
            return
        net, dev = self.net, self.dev
        self._step_rows = n
        self.s_out = torch.empty(n, net.NO, device=dev)
        if net.lstm:
            I, Hl = net.lstm
            self.s_xh = torch.zeros(n, I + Hl, device=dev)
            self.s_gates = torch.empty(n, 4 * Hl, device=dev)
            self.s_c = torch.empty(n, Hl, device=dev)
            self.s_h = torch.empty(n, Hl, device=dev)
        else:
            self.s_feat = torch.empty(n, net.F, device=dev)
        if net.kind == 'cnn':
            net.conv._alloc(min(n, self.frames_per_chunk))

    def policy_step(self, obs, n, noise, key, row_offset, actions, logprob, entropy, value, ids=None):
        """policy(obs) in rollout mode for the n rows of `obs` ([n][obs_stride] f32, or uint8 frames): encoder -> (one LSTM step on the
        state rows `ids`, default 0..n-1, updated in place) -> heads -> sample_logits."""
        net = self.net
        L = _lib.lib()
        if self.mlp_view is not None:       # a Default of one of the tile kernels' widths: the code the fused rollout runs, one launch
            _lib.check(L.pfa_mlp_view_forward_sample(_lib.ptr(obs), n, C.byref(self.mlp_view), _lib.ptr(noise), C.byref(key), row_offset,
                                                     _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(entropy), _lib.ptr(value),
                                                     _lib.stream_handle()), 'mlp_view_forward_sample')
            return
        net.pack()
        self._alloc_step(n)
        if net.lstm:
            I, Hl = net.lstm
            xh = self.s_xh[:n]
            self._encode(obs, n, xh, I + Hl)
            h_all, c_all = self.lstm_h[0], self.lstm_c[0]
            if ids is None and n == h_all.shape[0]:
                xh[:, I:].copy_(h_all)
                c_prev = c_all
            else:
                idx = ids if ids is not None else torch.arange(n, device=self.dev)
                xh[:, I:].copy_(h_all.index_select(0, idx))
                c_prev = c_all.index_select(0, idx)
            net.lstm_step(xh, self.s_gates[:n], c_prev, self.s_c[:n], self.s_h[:n], Hl)
            if ids is None and n == h_all.shape[0]:
                h_all.copy_(self.s_h[:n])
                c_all.copy_(self.s_c[:n])
            else:
                h_all.index_copy_(0, idx, self.s_h[:n])
                c_all.index_copy_(0, idx, self.s_c[:n])
            feat, ldf = self.s_h, Hl
        else:
            self._encode(obs, n, self.s_feat, net.F)
            feat, ldf = self.s_feat, net.F
        net.head_outputs(feat, ldf, n, self.s_out)
        _lib.check(L.pfa_heads_rows_sample(_lib.ptr(self.s_out), net.NO, n, net.A, net.heads, _lib.ptr(noise), C.byref(key), row_offset,
                                           _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(entropy), _lib.ptr(value), _lib.stream_handle()),
                   'heads_rows_sample')

    def _encode(self, obs, n, out, ldo):
        net = self.net
        if net.kind == 'mlp':
            net.encode(obs, n, out, ldo)
            return
        step = net.conv.chunk
        for lo in range(0, n, step):
            m = min(step, n - lo)
            net.encode(obs[lo:lo + m], m, out[lo:lo + m], ldo)

    # ------------------------------------------------------------------------------------------------------------- update
    def _alloc_update(self, R, Th):
        """Buffers of one chunk of Rc segments x Th steps (all segments of the minibatch for the MLP encoder; as many segments as
        fit frames_per_chunk frames for the conv encoder)."""
        net, dev = self.net, self.dev
        Rc = R if net.kind == 'mlp' else max(1, min(R, self.frames_per_chunk // Th))
        if self._upd is not None and self._upd['key'] == (R, Th, Rc):
            return self._upd
        u = dict(key=(R, Th, Rc), Rc=Rc)
        rows = Rc * Th
        L = _lib.lib()
        F, FH, NO = net.F, net.FH, net.NO
        u['out'] = torch.empty(rows, NO, device=dev)
        u['dout'] = torch.empty(rows, NO, device=dev)
        u['dhead'] = torch.empty(rows, FH, device=dev)
        u['g2v'] = torch.empty(NO, FH, device=dev)
        u['gb2v'] = torch.empty(NO, device=dev)
        u['loss_ws'] = torch.empty(L.pfa_heads_rows_loss_workspace_bytes(rows), dtype=torch.uint8, device=dev)
        ws = [L.pfa_igemm_weights_workspace_bytes(rows, FH, NO)]
        if net.kind == 'mlp':
            u['obs'] = torch.empty(rows, net.Kp, device=dev)
            u['g1'] = torch.empty(F, net.Kp, device=dev)
            u['gb1'] = torch.empty(F, device=dev)
            u['dpre'] = torch.empty(rows, F, device=dev)
            ws.append(L.pfa_igemm_weights_workspace_bytes(rows, net.Kp, F))
        else:
            net.conv._alloc(rows)
            u['frames'] = torch.empty(rows, net.conv.frame_bytes, dtype=torch.uint8, device=dev)
            u['dpre'] = torch.empty(rows, F, device=dev)          # segment-major, what cnn.Engine.backward reads
        if net.lstm:
            I, Hl = net.lstm
            u['xh'] = torch.zeros(Th + 1, Rc, I + Hl, device=dev)
            u['gates'] = torch.empty(Th, Rc, 4 * Hl, device=dev)
            u['c'] = torch.zeros(Th + 1, Rc, Hl, device=dev)
            u['hs'] = torch.empty(Th, Rc, Hl, device=dev)
            u['dG'] = torch.empty(Th, Rc, 4 * Hl, device=dev)
            u['dxh'] = torch.empty(Th, Rc, I + Hl, device=dev)
            u['dc'] = torch.empty(Rc, Hl, device=dev)
            u['gcat'] = torch.empty(4 * Hl, I + Hl, device=dev)
            u['gbcat'] = torch.empty(4 * Hl, device=dev)
            u['h0'] = torch.zeros(R, Hl, device=dev)              # state carried across the minibatches of an epoch, per segment slot
            u['c0'] = torch.zeros(R, Hl, device=dev)
            ws.append(L.pfa_igemm_weights_workspace_bytes(rows, I + Hl, 4 * Hl))
        else:
            u['feat'] = torch.empty(rows, F, device=dev)
        u['ws'] = torch.empty(max(ws), dtype=torch.uint8, device=dev)
        self._upd = u
        return u

    def update(self, mb, hp, adv_stats, global_mb_rows, grads, B):
        """Forward + PPO loss + backward for minibatch `mb` (clean_pufferl.py:179-244 up to loss.backward()): the flat gradient in
        named_parameters() order + the 16-float loss tail."""
        net, gp, exp = self.net, self.gp, self.experience
        L = _lib.lib()
        stream = _lib.stream_handle()
        if self.wide_ws is not None:          # Default(64 / 256 / 512): one fused launch + the fixed-order sum of its partials
            _lib.check(L.pfa_ppo_wide_grad(C.byref(exp.c), B, mb, C.byref(self.mlp_view), C.byref(self._grad_view(grads)),
                                           C.c_void_p(grads.data_ptr() + 4 * gp.count), C.byref(hp), _lib.ptr(adv_stats), global_mb_rows,
                                           _lib.ptr(self.wide_ws), stream), 'ppo_wide_grad')
            return
        net.pack()
        M, Th = exp.minibatch_size, exp.bptt_horizon
        R = M // Th
        u = self._alloc_update(R, Th)
        Rc = u['Rc']
        F, FH, NO = net.F, net.FH, net.NO
        gv = gp.split(grads[:gp.count])
        tail = grads[gp.count:gp.count + 16]
        fresh_state = mb == 0 or self.state is None          # lstm_state = None at the start of every epoch (clean_pufferl.py:176)
        if net.kind == 'mlp':                                 # the whole minibatch, time-major, in one go
            _lib.check(L.pfa_gather_obs_time_major(C.byref(exp.c), B, mb, C.byref(hp), net.Kp, _lib.ptr(u['obs']), stream), 'gather_obs')
        conv_gv = net.conv.cp.split(grads[:gp.count]) if net.kind == 'cnn' else None
        for ci, k0 in enumerate(range(0, R, Rc)):
            acc = ci > 0
            rc = min(Rc, R - k0)
            rows = rc * Th
            # ---- encoder -> time-major features ---------------------------------------------------------------------------
            if net.lstm:
                I, Hl = net.lstm
                xh = u['xh'][:, :rc] if rc == Rc else torch.zeros(Th + 1, rc, I + Hl, device=self.dev)
                ldx = I + Hl
                if net.kind == 'mlp':
                    net.encode(u['obs'], rows, xh, ldx)                     # rows already time-major
                else:
                    _lib.check(L.pfa_cnn_gather_frames(_lib.ptr(exp.obs), net.conv.frame_bytes, B, mb, C.byref(hp), k0 * Th, rows,
                                                       _lib.ptr(u['frames']), stream), 'gather_frames')
                    h = net.conv.forward(u['frames'], rows)                  # segment-major [rows][512]
                    rows_perm(h, F, xh, ldx, rows, F, rc, Th, True)
                c = u['c'][:, :rc] if rc == Rc else torch.zeros(Th + 1, rc, Hl, device=self.dev)
                hs = u['hs'][:, :rc] if rc == Rc else torch.empty(Th, rc, Hl, device=self.dev)
                gates = u['gates'][:, :rc] if rc == Rc else torch.empty(Th, rc, 4 * Hl, device=self.dev)
                if fresh_state:
                    xh[0, :, I:].zero_()
                    c[0].zero_()
                else:                                                        # carried across minibatches, detached (clean_pufferl.py:188-191)
                    xh[0, :, I:].copy_(u['h0'][k0:k0 + rc])
                    c[0].copy_(u['c0'][k0:k0 + rc])
                for t in range(Th):
                    net.lstm_step(xh[t], gates[t], c[t], c[t + 1], hs[t], Hl, xh[t + 1, :, I:], ldx)
                u['h0'][k0:k0 + rc].copy_(hs[Th - 1])
                u['c0'][k0:k0 + rc].copy_(c[Th])
                head_in, ldh = hs, Hl
            else:
                net.encode(u['obs'], rows, u['feat'], F)
                head_in, ldh = u['feat'], F
            # ---- heads + loss -----------------------------------------------------------------------------------------------
            net.head_outputs(head_in, ldh, rows, u['out'])
            _lib.check(L.pfa_heads_rows_loss(_lib.ptr(u['out']), NO, C.byref(exp.c), B, mb, k0 * Th, rows, rc, net.A, net.heads, C.byref(hp),
                                             _lib.ptr(adv_stats), global_mb_rows, _lib.ptr(u['dout']), NO, NO, _lib.ptr(tail), 1 if acc else 0,
                                             _lib.ptr(u['loss_ws']), stream), 'heads_rows_loss')
            gemm_weights(head_in, ldh, rows, FH, u['dout'], NO, NO, u['g2v'], acc, u['gb2v'], u['ws'])
            if net.lstm:
                gemm_rows(u['dout'], NO, rows, NO, net.w2vT, NO, FH, u['dhead'], FH)
                # ---- back-propagation through time ------------------------------------------------------------------------
                dG = u['dG'][:, :rc] if rc == Rc else torch.empty(Th, rc, 4 * Hl, device=self.dev)
                dxh = u['dxh'][:, :rc] if rc == Rc else torch.empty(Th, rc, I + Hl, device=self.dev)
                dc = u['dc'][:rc]
                dc.zero_()
                dhead = u['dhead'].view(Th, Rc, Hl)[:, :rc] if rc == Rc else u['dhead'][:rows].view(Th, rc, Hl)
                for t in range(Th - 1, -1, -1):
                    dh_b = dxh[t + 1, :, I:] if t < Th - 1 else None
                    _lib.check(L.pfa_lstm_cell_backward(_lib.ptr(dhead[t]), Hl, _lib.ptr(dh_b), ldx, _lib.ptr(dc), _lib.ptr(gates[t]), _lib.ptr(c[t]),
                                                        _lib.ptr(c[t + 1]), _lib.ptr(dG[t]), rc, Hl, stream), 'lstm_cell_backward')
                    gemm_rows(dG[t], 4 * Hl, rc, 4 * Hl, net.wcatT, 4 * Hl, I + Hl, dxh[t], ldx)
                gemm_weights(xh, ldx, rows, I + Hl, dG, 4 * Hl, 4 * Hl, u['gcat'], acc, u['gbcat'], u['ws'])
                # d loss / d (pre-ReLU encoder output): the x columns of dxh, masked by relu' of the features (the x columns of xh)
                if net.kind == 'mlp':
                    rows_perm(dxh, ldx, u['dpre'], F, rows, F, act=xh, lda=ldx)
                else:
                    rows_perm(dxh, ldx, u['dpre'], F, rows, F, rc, Th, False, act=xh, lda=ldx)     # and back to segment-major
            else:
                gemm_rows(u['dout'], NO, rows, NO, net.w2vT, NO, FH, u['dpre'], F, EPI_MASK, None, u['feat'], F)
            # ---- encoder backward -------------------------------------------------------------------------------------------
            if net.kind == 'mlp':
                gemm_weights(u['obs'], net.Kp, rows, net.Kp, u['dpre'], F, F, u['g1'], acc, u['gb1'], u['ws'])
            else:
                net.conv.backward(u['frames'], rows, u['dpre'], conv_gv, acc)
        self.state = True
        # ---- scatter the packed gradients into named_parameters() order -----------------------------------------------------
        nm = net.names
        with torch.no_grad():
            if net.kind == 'mlp':
                gv[nm['enc_w']].copy_(u['g1'][:, :net.obs_dim])
                gv[nm['enc_b']].copy_(u['gb1'])
            r = 0
            for w, b in nm['actors']:
                n = gv[w].shape[0]
                gv[w].copy_(u['g2v'][r:r + n])
                gv[b].copy_(u['gb2v'][r:r + n])
                r += n
            gv[nm['value_w']].copy_(u['g2v'][r:r + 1])
            gv[nm['value_b']].copy_(u['gb2v'][r:r + 1])
            if net.lstm:
                I, Hl = net.lstm
                gv[nm['w_ih']].copy_(u['gcat'][:, :I])
                gv[nm['w_hh']].copy_(u['gcat'][:, I:])
                gv[nm['b_ih']].copy_(u['gbcat'])
                gv[nm['b_hh']].copy_(u['gbcat'])

    def _grad_view(self, grads):
        """The six tensors of the policy inside the gradient buffer `grads` (GeneralParams layout: named_parameters() order, torch
        shapes): the parameter view's pointers moved by the distance between the two buffers."""
        if self._wide_gview is None or self._wide_gview[0] != grads.data_ptr():
            v, shift = self.mlp_view, grads.data_ptr() - self.gp.flat.data_ptr()
            gv = _lib.MlpView(v.w1 + shift, v.ldw1, v.obs_dim, v.obs_stride, v.hidden, v.num_actions, 0, v.b1 + shift, v.w2 + shift, v.b2 + shift,
                              v.wv + shift, v.bv + shift)
            self._wide_gview = (grads.data_ptr(), gv)
        return self._wide_gview[1]

    def clip_adam(self, grads, opt, max_grad_norm, loss_acc, loss_scale):
        """clip_grad_norm_ + optimizer.step() (clean_pufferl.py:240-244) on the flat buffer; the packed operand forms go stale."""
        L = _lib.lib()
        stream = _lib.stream_handle()
        gp = self.gp
        n = self.norm_partials.numel()
        _lib.check(L.pfa_sumsq_partials(_lib.ptr(grads), gp.count, _lib.ptr(self.norm_partials), n, stream), 'sumsq')
        opt.step_count += 1
        g = opt.param_groups[0]
        _lib.check(L.pfa_adam_clip_step(_lib.ptr(gp.flat), _lib.ptr(grads), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), gp.count,
                                        float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), opt.step_count,
                                        float(max_grad_norm), 1.0, C.c_void_p(grads.data_ptr() + 4 * gp.count), _lib.ptr(loss_acc), loss_scale,
                                        _lib.ptr(self.norm_partials), n, stream), 'adam')
        self.net.version += 1
