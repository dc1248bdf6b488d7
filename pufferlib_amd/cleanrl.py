"""frameworks.cleanrl.Policy (pufferlib/frameworks/cleanrl.py:50-66) over the HIP forward+sample kernel.

``policy(obs)`` (rollout mode, action=None) returns (actions, logprob, entropy, value) like the reference; the
multinomial draw is argmax(softmax(logits)/q) with q ~ Exp(1) from either an explicit ``noise`` tensor (parity with
torch.multinomial given its exponential draw) or the Philox stream keyed by ``seed`` (include/pufferlib_amd.h).
``policy(obs, action=...)`` (training mode, cleanrl.py:60-66,87-93) returns the log-probability of the GIVEN actions, the entropy and
the value through the GEMM path (pufferlib_amd.general.Evaluator) for every policy shape; no autograd graph is attached — the
gradients of these quantities are computed by pufferlib_amd.clean_pufferl.train()'s kernels.

Policy shapes outside the fused kernels' envelope (Default(hidden_size != 128), observation rows wider than 128 / 160 floats, more
than 15 logits, LSTMWrapper sizes other than (128, 128), LSTMWrapper over the NatureCNN) adopt a general.GeneralParams buffer and
run rollout and update through general.Engine."""
import ctypes as C

import torch

from . import _lib
from .models import ConvParams, FlatParams, find_cnn

KERNEL_STRIDES = (16, 32, 64, 96, 128)     # observation row strides (floats) every policy kernel is built for
RECURRENT_STRIDES = KERNEL_STRIDES + (160,)  # the recurrent path also takes MiniGrid-shaped 160-byte rows (SURVEY config C3)


def obs_stride_for(obs_dim, recurrent=False):
    """Row stride (floats) of a flat observation: the narrowest stride the fused kernels are built for, or — wider rows run in the
    GEMM path (general.py) — the next multiple of 16."""
    for s in (RECURRENT_STRIDES if recurrent else KERNEL_STRIDES):
        if obs_dim <= s:
            return s
    return (obs_dim + 15) // 16 * 16


def needs_general(policy_module, recurrent):
    """True when the policy's shape is outside what the fused kernels are instantiated for (see the module docstring).  A module's
    shape does not change: the answer is cached on it (forward() asks on every call)."""
    cache = policy_module.__dict__.setdefault('_pfa_needs_general', {})
    if recurrent not in cache:
        cache[recurrent] = _needs_general(policy_module, recurrent)
    return cache[recurrent]


def _needs_general(policy_module, recurrent):
    from .models import HIDDEN, decoder_heads, find_lstm, find_mlp
    lstm = find_lstm(policy_module)
    if find_cnn(policy_module) is not None:
        return lstm is not None
    mlp = find_mlp(policy_module)
    H, D = mlp.encoder.weight.shape
    nvec = decoder_heads(mlp)
    if H != HIDDEN or sum(nvec) > 15 or D > (RECURRENT_STRIDES if recurrent else KERNEL_STRIDES)[-1]:
        return True
    return lstm is not None and (lstm.input_size, lstm.hidden_size) != (HIDDEN, HIDDEN)


def _weights_changed(module, _incompatible_keys=None):
    """load_state_dict post hook of Policy / RecurrentPolicy: the parameters were overwritten in place, so every packed operand copy
    (GEMM-path Net, conv engine) is stale — the same version bump the optimizer step and the checkpoint loader make."""
    for obj in (getattr(getattr(module, 'gen_engine', None), 'net', None), getattr(getattr(module, '_evaluator', None), 'net', None),
                getattr(module, 'cnn_engine', None)):
        if obj is not None and hasattr(obj, 'version'):
            obj.version += 1


def _adopt(self, obs_stride, device, recurrent):
    """Shared by Policy / RecurrentPolicy: move the parameters into one flat device buffer (idempotent for the same stride/device)."""
    if needs_general(self.policy, recurrent):
        from . import general
        if (self._flat is None or not isinstance(self._flat, general.GeneralParams) or self._flat.flat.device != torch.device(device)
                or (self._flat.kind == 'mlp' and obs_stride and self._flat.obs_stride != obs_stride)):
            self._flat = general.GeneralParams(self.policy, device, obs_stride or None)
            self._evaluator = None
        return self._flat
    if (self._flat is None or self._flat.obs_stride != obs_stride or self._flat.flat.device != torch.device(device)):
        self._flat = FlatParams(self.policy, obs_stride, device)
        self._evaluator = None
    return self._flat


def _evaluator(self, device, recurrent):
    """general.Evaluator over this policy's parameter buffer (built on first use; re-packs its operand copies on every call: the
    fused update kernels change the parameters without telling it)."""
    from . import general
    if self._flat is None:
        D = general.find_mlp(self.policy).encoder.weight.shape[1] if find_cnn(self.policy) is None else 0
        self.adopt(obs_stride_for(int(D), recurrent) if D else 0, device)
    if getattr(self, '_evaluator', None) is None:
        if isinstance(self._flat, general.GeneralParams):
            eng = getattr(self, 'gen_engine', None)
            net = eng.net if eng is not None else general._net_for_general(self._flat)
        elif isinstance(self._flat, ConvParams):
            raise NotImplementedError('policy(obs, action=...) for the non-recurrent models.Convolutional: wrap it in LSTMWrapper or use train()')
        else:
            net = general.net_for_flat(self._flat)
        self._evaluator = general.Evaluator(net)
    eng = getattr(self, 'gen_engine', None)
    if not (isinstance(self._flat, general.GeneralParams) and eng is not None and self._evaluator.net is eng.net):
        # the fused 128-wide update kernels change the kernel-layout buffer behind this evaluator's back: re-pack its operand copies on
        # every call (so does a policy used on its own, whose tensors anybody may write).  A trainer's GEMM-path Net is told when its
        # weights change: the optimizer step (Engine.clip_adam), the checkpoint loader and load_state_dict (post hook) bump its version.
        self._evaluator.net.version += 1
    return self._evaluator


class Policy(torch.nn.Module):
    def __init__(self, policy, seed=0):
        super().__init__()
        self.policy = policy
        self.noise_seed = int(seed)
        self.noise_step = 0
        self._flat = None
        self.register_load_state_dict_post_hook(_weights_changed)

    def adopt(self, obs_stride, device):
        """Move the parameters into one flat device buffer (idempotent for the same stride/device)."""
        if find_cnn(self.policy) is not None:       # models.Convolutional: its own parameter layout and engine (cnn.py)
            if self._flat is None or self._flat.flat.device != torch.device(device):
                from . import cnn
                self._flat = ConvParams(self.policy, device)
                self.cnn_engine = cnn.Engine(self._flat, chunk=256)       # grows on demand (Engine._alloc)
            return self._flat
        return _adopt(self, obs_stride, device, False)

    @property
    def flat_params(self):
        return self._flat

    def __getstate__(self):
        """Pickles (torch.save of the whole module, clean_pufferl.py:517) carry the parameters only: the flat-buffer bookkeeping
        and the conv engine's activation buffers are rebuilt by adopt() on first use."""
        state = dict(self.__dict__)
        state['_flat'] = None
        state.pop('cnn_engine', None)
        state.pop('gen_engine', None)
        state.pop('_evaluator', None)
        return state

    def get_value(self, x, state=None):
        return self.forward(x)[3]

    def get_action_and_value(self, x, action=None, noise=None):
        return self.forward(x, action=action, noise=noise)

    def forward(self, x, action=None, noise=None):
        _lib.require_gpu()
        L = _lib.lib()
        if not x.is_cuda:
            x = x.cuda()
        rows = x.shape[0]
        x2 = x.reshape(rows, -1)
        D = x2.shape[1]
        general_shape = needs_general(self.policy, False)
        if action is not None or general_shape:
            # given actions (cleanrl.py:60-66 with action=...): log-prob / entropy / value of THOSE actions; or a policy shape that
            # runs in the GEMM path altogether
            if find_cnn(self.policy) is not None and not general_shape:
                raise NotImplementedError('policy(frames, action=...) for the non-recurrent models.Convolutional')
            if action is None and find_cnn(self.policy) is None:
                out = self._forward_tile(x2, rows, D, noise)     # Default(64 / 256 / 512): the tile code the fused rollout runs
                if out is not None:
                    return out
            ev = _evaluator(self, x.device, False)
            key = None
            if action is None and noise is None:
                key = _lib.NoiseKey(self.noise_seed, self.noise_step)
                self.noise_step += 1
            a, logprob, entropy, value, _ = ev.forward(x2, action=action, noise=noise, key=key)
            return (action if action is not None else self._flat.unpack_actions(a)), logprob, entropy, value
        if find_cnn(self.policy) is not None:
            return self._forward_cnn(x2, rows, noise)
        stride = obs_stride_for(D)
        if (x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) >= D and x2.stride(0) in KERNEL_STRIDES
                and x2.data_ptr() % 16 == 0):   # any other dtype takes the .float() copy below, like models.Default (models.py:50)
            stride = x2.stride(0)            # e.g. the vecenv's live buffer, already padded
            src = x2
        else:
            src = torch.zeros(rows, stride, dtype=torch.float32, device=x.device)
            src[:, :D] = x2.float()
        fp = self.adopt(stride, x.device)
        actions = torch.empty(rows, dtype=torch.int64, device=x.device)
        logprob = torch.empty(rows, dtype=torch.float32, device=x.device)
        entropy = torch.empty(rows, dtype=torch.float32, device=x.device)
        value = torch.empty(rows, dtype=torch.float32, device=x.device)
        key = _lib.NoiseKey(self.noise_seed, self.noise_step)
        if noise is not None:
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
            assert noise.shape == (rows, fp.num_actions)
        else:
            self.noise_step += 1
        _lib.check(L.pfa_mlp_forward_sample(_lib.ptr(src), rows, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(noise),
                                            C.byref(key), 0, _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(entropy),
                                            _lib.ptr(value), _lib.stream_handle()), 'mlp_forward_sample')
        return fp.unpack_actions(actions), logprob, entropy, value.unsqueeze(1)


def _forward_tile(self, x2, rows, D, noise):
    """policy(obs) in rollout mode for a Default of one of the tile kernels' widths (general.tile_view): one launch of
    pfa_mlp_view_forward_sample — the same code, bit for bit, as the persistent fused rollout.  None when the shape is not one of them."""
    from . import general
    fp = self._flat if self._flat is not None else self.adopt(obs_stride_for(D), x2.device)
    view = general.tile_view(fp)
    if view is None:
        return None
    if x2.shape[1] != fp.obs_dim:
        raise ValueError(f'observation rows of {x2.shape[1]} values, the policy reads {fp.obs_dim}')
    stride = fp.obs_stride
    if x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) == stride and x2.data_ptr() % 16 == 0:
        src = x2                             # e.g. the vecenv's live buffer, already padded
    else:
        src = torch.zeros(rows, stride, dtype=torch.float32, device=x2.device)
        src[:, :D] = x2.float()
    dev = x2.device
    actions = torch.empty(rows, dtype=torch.int64, device=dev)
    logprob = torch.empty(rows, dtype=torch.float32, device=dev)
    entropy = torch.empty(rows, dtype=torch.float32, device=dev)
    value = torch.empty(rows, dtype=torch.float32, device=dev)
    key = _lib.NoiseKey(self.noise_seed, self.noise_step)
    if noise is not None:
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        assert noise.shape == (rows, fp.num_actions)
    else:
        self.noise_step += 1
    _lib.check(_lib.lib().pfa_mlp_view_forward_sample(_lib.ptr(src), rows, C.byref(view), _lib.ptr(noise), C.byref(key), 0, _lib.ptr(actions),
                                                      _lib.ptr(logprob), _lib.ptr(entropy), _lib.ptr(value), _lib.stream_handle()),
               'mlp_view_forward_sample')
    return actions, logprob, entropy, value.unsqueeze(1)



def _forward_cnn(self, x2, rows, noise):
    """policy(frames) for models.Convolutional: uint8 (rows, framestack*84*84) -> (actions, logprob, entropy, value)."""
    cp = self.adopt(0, x2.device)
    if x2.shape[1] != cp.obs_dim:
        raise ValueError(f'expected frames of {cp.obs_dim} bytes, got rows of {x2.shape[1]}')
    frames = x2.to(torch.uint8).contiguous()       # the reference divides whatever it is given by 255 (models.py:152); frames are bytes
    eng = self.cnn_engine
    eng._alloc(min(rows, 8192))
    dev = x2.device
    actions = torch.empty(rows, dtype=torch.int64, device=dev)
    logprob = torch.empty(rows, dtype=torch.float32, device=dev)
    entropy = torch.empty(rows, dtype=torch.float32, device=dev)
    value = torch.empty(rows, dtype=torch.float32, device=dev)
    key = _lib.NoiseKey(self.noise_seed, self.noise_step)
    if noise is not None:
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        assert noise.shape == (rows, cp.num_actions)
    else:
        self.noise_step += 1
    eng.policy_step(frames, rows, noise, key, 0, actions, logprob, entropy, value)
    return actions, logprob, entropy, value.unsqueeze(1)


Policy._forward_cnn = _forward_cnn
Policy._forward_tile = _forward_tile


class RecurrentPolicy(torch.nn.Module):
    """frameworks.cleanrl.RecurrentPolicy (pufferlib/frameworks/cleanrl.py:69-93) over the HIP LSTM path.

    ``policy(obs, state)`` (rollout mode) returns (actions, logprob, entropy, value, (h, c)) with state tensors of shape
    (1, rows, 128) like nn.LSTM.  The training-mode forward lives in pufferlib_amd.clean_pufferl.train."""

    def __init__(self, policy, seed=0):
        super().__init__()
        self.policy = policy
        self.noise_seed = int(seed)
        self.noise_step = 0
        self._flat = None
        self.register_load_state_dict_post_hook(_weights_changed)

    @property
    def lstm(self):
        if hasattr(self.policy, 'recurrent'):
            return self.policy.recurrent
        elif hasattr(self.policy, 'lstm'):
            return self.policy.lstm
        raise ValueError('Policy must have a subnetwork named lstm or recurrent')

    def adopt(self, obs_stride, device):
        return _adopt(self, obs_stride, device, True)

    @property
    def flat_params(self):
        return self._flat

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_flat'] = None
        state.pop('gen_engine', None)
        state.pop('_evaluator', None)
        return state

    def get_action_and_value(self, x, state=None, action=None, noise=None):
        return self.forward(x, state=state, action=action, noise=noise)

    def forward(self, x, state=None, action=None, noise=None):
        from . import lstm as plstm
        _lib.require_gpu()
        if not x.is_cuda:
            x = x.cuda()
        if action is not None or needs_general(self.policy, True):
            # cleanrl.py:87-93: (B, obs...) or (B, TT, obs...) through encoder -> LSTM over TT steps -> heads, with the given actions
            # scored (or, for a policy shape outside the fused kernels, sampled) — the GEMM path
            ev = _evaluator(self, x.device, True)
            obs_shape = getattr(self.policy, 'obs_shape', None)
            key = None
            if action is None and noise is None:
                key = _lib.NoiseKey(self.noise_seed, self.noise_step)
                self.noise_step += 1
            a, logprob, entropy, value, new_state = ev.forward(x, action=action, state=state, noise=noise, key=key,
                                                               obs_rank=len(obs_shape) if obs_shape is not None else None)
            return (action if action is not None else self._flat.unpack_actions(a)), logprob, entropy, value, new_state
        rows = x.shape[0]
        x2 = x.reshape(rows, -1)
        D = x2.shape[1]
        stride = obs_stride_for(D, recurrent=True)
        if (x2.dtype == torch.float32 and x2.stride(1) == 1 and x2.stride(0) >= D and x2.stride(0) in RECURRENT_STRIDES
                and x2.data_ptr() % 16 == 0):
            stride = x2.stride(0)
            src = torch.as_strided(x2, (rows, stride), (stride, 1))
        else:
            src = torch.zeros(rows, stride, dtype=torch.float32, device=x.device)
            src[:, :D] = x2.float()
        fp = self.adopt(stride, x.device)
        if state is None:
            h = torch.zeros(rows, 128, device=x.device)
            c = torch.zeros(rows, 128, device=x.device)
        else:
            h, c = state[0][0].contiguous().clone(), state[1][0].contiguous().clone()
        key = _lib.NoiseKey(self.noise_seed, self.noise_step)
        if noise is not None:
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        else:
            self.noise_step += 1
        out = plstm.policy_step(fp, src, h, c, noise, key, 0)
        return out + ((h.unsqueeze(0), c.unsqueeze(0)),)
