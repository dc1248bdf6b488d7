"""Policy containers.  ``Default`` has the parameters, names and initialisation of pufferlib.models.Default
(models.py:12-62) so state_dicts are interchangeable with the reference, but its arithmetic runs in the HIP
kernels: once adopted by ``FlatParams`` every parameter is a *view* into one flat fp32 device buffer laid out as
include/pufferlib_amd.h describes (encoder rows padded to obs_stride), so module, kernels, checkpoints and the
optimizer always see the same bytes."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib

HIDDEN = 128


def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    """CleanRL's default layer initialisation (pufferlib/pytorch.py:193-197)."""
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


class Default(nn.Module):
    """Linear(obs -> 128) + ReLU, Linear(128 -> n_actions) (orthogonal, std 0.01), Linear(128 -> 1).  A MultiDiscrete action
    space gets one decoder Linear per head in an nn.ModuleList (models.py:29-35), state_dict keys ``decoder.<h>.weight``."""

    def __init__(self, env, hidden_size=HIDDEN):
        super().__init__()
        if hidden_size < 16 or hidden_size % 16 != 0:     # 128 runs in the fused kernels, any other width in the GEMM path (general.py)
            raise ValueError(f'pufferlib_amd.models.Default: hidden_size must be a multiple of 16 (got {hidden_size})')
        self.hidden_size = int(hidden_size)
        self.obs_dim = int(np.prod(env.single_observation_space.shape))
        self.encoder = nn.Linear(self.obs_dim, hidden_size)
        self.is_multidiscrete = hasattr(env.single_action_space, 'nvec')
        if self.is_multidiscrete:
            nvec = [int(n) for n in env.single_action_space.nvec]
            self.num_actions = sum(nvec)
            self.decoder = nn.ModuleList([layer_init(nn.Linear(hidden_size, n), std=0.01) for n in nvec])
        else:
            self.num_actions = int(env.single_action_space.n)
            self.decoder = layer_init(nn.Linear(hidden_size, self.num_actions), std=0.01)
        self.value_head = nn.Linear(hidden_size, 1)

    def forward(self, observations):
        raise RuntimeError('pufferlib_amd.models.Default is a parameter container: call it through '
                           'pufferlib_amd.cleanrl.Policy / pufferlib_amd.clean_pufferl (HIP kernels)')


class LSTMWrapper(nn.Module):
    """pufferlib.models.LSTMWrapper (models.py:64-111): policy.encode_observations -> nn.LSTM(input_size, hidden_size, 1) ->
    policy.decode_actions; LSTM weights orthogonal (gain 1), biases 0.  Parameter container like ``Default``.  (128, 128) over
    ``Default`` runs in the fused recurrent kernels; any other sizes — and the NatureCNN underneath, environments/atari/torch.py:4-6 —
    in the GEMM path (general.py).  As in the reference the heads of `policy` must read hidden_size features."""

    def __init__(self, env, policy, input_size=HIDDEN, hidden_size=HIDDEN, num_layers=1):
        super().__init__()
        if num_layers != 1:
            raise ValueError('pufferlib_amd.models.LSTMWrapper supports num_layers=1 (the reference default, models.py:65)')
        if hidden_size % 16 != 0 or input_size % 16 != 0:
            raise ValueError('pufferlib_amd.models.LSTMWrapper: input_size and hidden_size must be multiples of 16')
        self.obs_shape = env.single_observation_space.shape
        self.policy = policy
        self.input_size, self.hidden_size = input_size, hidden_size
        self.recurrent = nn.LSTM(input_size, hidden_size, num_layers)
        for name, param in self.recurrent.named_parameters():
            if 'bias' in name:
                nn.init.constant_(param, 0)
            elif 'weight' in name:
                nn.init.orthogonal_(param, 1.0)

    def forward(self, x, state):
        raise RuntimeError('pufferlib_amd.models.LSTMWrapper is a parameter container: call it through '
                           'pufferlib_amd.cleanrl.RecurrentPolicy / pufferlib_amd.clean_pufferl (HIP kernels)')


class Convolutional(nn.Module):
    """pufferlib.models.Convolutional (models.py:113-157), the CleanRL NatureCNN used for Atari: the same modules, names
    (``network.0/2/4/7``, ``actor``, ``value_fn``) and initialisation (layer_init: orthogonal sqrt(2), actor 0.01, value 1), so
    state_dicts are interchangeable with the reference.  Parameter container: the arithmetic runs in csrc/igemm.hip /
    csrc/cnn_heads.hip through pufferlib_amd.cnn.Engine."""

    def __init__(self, env, *args, framestack=4, flat_size=64 * 7 * 7, input_size=512, hidden_size=512, output_size=512,
                 channels_last=False, downsample=1, **kwargs):
        super().__init__()
        if channels_last or downsample != 1 or hidden_size != 512 or output_size != 512 or flat_size != 64 * 7 * 7:
            raise NotImplementedError('pufferlib_amd.models.Convolutional runs the Atari geometry: uint8 (framestack, 84, 84) frames, '
                                      'flat_size 3136, hidden 512')
        self.framestack = int(framestack)
        self.network = nn.Sequential(
            layer_init(nn.Conv2d(framestack, 32, 8, stride=4)), nn.ReLU(),
            layer_init(nn.Conv2d(32, 64, 4, stride=2)), nn.ReLU(),
            layer_init(nn.Conv2d(64, 64, 3, stride=1)), nn.ReLU(),
            nn.Flatten(),
            layer_init(nn.Linear(flat_size, hidden_size)), nn.ReLU(),
        )
        self.actor = layer_init(nn.Linear(hidden_size, env.single_action_space.n), std=0.01)
        self.value_fn = layer_init(nn.Linear(output_size, 1), std=1)

    def forward(self, observations):
        raise RuntimeError('pufferlib_amd.models.Convolutional is a parameter container: call it through '
                           'pufferlib_amd.cleanrl.Policy / pufferlib_amd.clean_pufferl (HIP kernels)')


def find_cnn(module):
    """The Convolutional-shaped submodule of a policy wrapper (ours or the reference's): network / actor / value_fn."""
    for m in module.modules():
        if all(hasattr(m, n) for n in ('network', 'actor', 'value_fn')) and isinstance(getattr(m, 'network'), nn.Sequential):
            return m
    return None


class ConvParams:
    """One flat fp32 device buffer holding the NatureCNN parameters in named_parameters() order (network.0.weight, .bias,
    network.2.*, network.4.*, network.7.*, actor.*, value_fn.*), every module parameter re-pointed at its view: module,
    kernels, optimizer and checkpoints see the same bytes.  The interface the trainer uses matches FlatParams."""
    multidiscrete = False

    def __init__(self, policy_module, device):
        net = find_cnn(policy_module)
        if net is None:
            raise ValueError('policy has no network/actor/value_fn (models.Convolutional shape)')
        self.net = net
        convs = [m for m in net.network if isinstance(m, nn.Conv2d)]
        fc = [m for m in net.network if isinstance(m, nn.Linear)]
        if len(convs) != 3 or len(fc) != 1:
            raise ValueError('expected three Conv2d and one Linear in network')
        want = [((32, None, 8, 8), (4, 4)), ((64, 32, 4, 4), (2, 2)), ((64, 64, 3, 3), (1, 1))]
        for cv, (shape, stride) in zip(convs, want):
            ws = tuple(cv.weight.shape)
            if ws[0] != shape[0] or ws[2:] != shape[2:] or (shape[1] is not None and ws[1] != shape[1]) or tuple(cv.stride) != stride \
                    or tuple(cv.padding) != (0, 0):
                raise NotImplementedError(f'conv layer {ws} stride {cv.stride}: only the NatureCNN geometry is built')
        if tuple(fc[0].weight.shape) != (512, 3136):
            raise NotImplementedError('Linear(3136, 512) expected behind the conv stack')
        self.framestack = int(convs[0].weight.shape[1])
        if (self.framestack * 8) % 4 != 0:
            raise NotImplementedError('framestack')
        self.num_actions = int(net.actor.weight.shape[0])
        if self.num_actions > 15:
            raise NotImplementedError('the head kernels take up to 15 actions')
        self.nvec = [self.num_actions]
        self.names = [n for n, _ in net.named_parameters()]
        sizes = [p.numel() for _, p in net.named_parameters()]
        self.count = int(sum(sizes))
        self.flat = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.views = self.split(self.flat)
        with torch.no_grad():
            for name, p in net.named_parameters():
                v = self.views[name]
                v.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = v
        self.obs_dim = self.obs_stride = self.framestack * 84 * 84

    def split(self, flat):
        out, o = {}, 0
        for name, p in self.net.named_parameters():
            n = p.numel()
            out[name] = flat[o:o + n].view(p.shape)
            o += n
        return out

    def flat_like(self):
        return torch.zeros_like(self.flat)

    def unpack_actions(self, packed):
        return packed


LSTM_KEYS = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']


def find_lstm(module):
    """The nn.LSTM of a recurrent policy wrapper (ours or the reference's LSTMWrapper: attribute ``recurrent``)."""
    for m in module.modules():
        if isinstance(m, nn.LSTM):
            return m
    return None


MLP_KEYS = ['encoder.weight', 'encoder.bias', 'decoder.weight', 'decoder.bias', 'value_head.weight', 'value_head.bias']


def decoder_heads(mlp):
    """[logits per head] of a Default-shaped module: one entry for a Discrete decoder, one per Linear of a MultiDiscrete
    ModuleList (models.py:29-35)."""
    if isinstance(mlp.decoder, nn.Linear):
        return [int(mlp.decoder.weight.shape[0])]
    return [int(d.weight.shape[0]) for d in mlp.decoder]


def find_mlp(module):
    """Locate the Default-shaped submodule inside a policy wrapper (ours or the reference's
    frameworks.cleanrl.Policy -> .policy).  Returns the module owning encoder/decoder/value_head."""
    for m in module.modules():
        if all(hasattr(m, n) for n in ('encoder', 'decoder', 'value_head')) and isinstance(getattr(m, 'encoder'), nn.Linear):
            if isinstance(m.decoder, nn.Linear) or (isinstance(m.decoder, nn.ModuleList) and len(m.decoder) >= 1
                                                    and all(isinstance(d, nn.Linear) for d in m.decoder)):
                return m
    raise ValueError('policy has no encoder/decoder/value_head Linear layers (models.Default shape)')


class FlatParams:
    """One flat fp32 device buffer holding the MLP parameters in kernel layout + views for the nn.Module."""

    def __init__(self, policy_module, obs_stride, device):
        mlp = find_mlp(policy_module)
        self.mlp = mlp
        H, D = mlp.encoder.weight.shape
        self.nvec = decoder_heads(mlp)
        self.multidiscrete = not isinstance(mlp.decoder, nn.Linear)
        A = sum(self.nvec)
        if H != HIDDEN:
            raise ValueError(f'hidden size must be {HIDDEN}')
        decs = list(mlp.decoder) if self.multidiscrete else [mlp.decoder]
        if not (mlp.value_head.weight.shape == (1, H) and all(d.weight.shape[1] == H for d in decs)):
            raise ValueError('unexpected head shapes')
        if A > 15 or (self.multidiscrete and (len(self.nvec) > 8 or max(self.nvec) > 15)):
            raise NotImplementedError(f'action heads {self.nvec}: the policy kernels take up to 15 logits in up to 8 heads')
        heads = sum(n << (4 * h) for h, n in enumerate(self.nvec)) if self.multidiscrete else 0
        self.dims = _lib.MlpDims(int(D), int(obs_stride), int(H), int(A), heads)
        self.obs_dim, self.obs_stride, self.num_actions = int(D), int(obs_stride), int(A)
        DP = self.obs_stride
        self.mlp_count = H * DP + H + A * H + A + H + 1
        self.lstm = find_lstm(policy_module)
        if self.lstm is not None:
            if (self.lstm.input_size, self.lstm.hidden_size, self.lstm.num_layers) != (H, H, 1) or self.lstm.bidirectional:
                raise ValueError(f'the recurrent policy must use nn.LSTM({H}, {H}, 1)')
        self.count = self.mlp_count + (8 * H * H + 8 * H if self.lstm is not None else 0)
        self.flat = torch.zeros(self.count, dtype=torch.float32, device=device)
        self.views = self.split_mlp(self.flat)
        self.lstm_views = self._lstm_views(self.flat)
        with torch.no_grad():
            for name, view in self.views.items():
                p = mlp.get_parameter(name)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view          # the module now aliases the flat buffer
            for name, view in self.lstm_views.items():
                p = getattr(self.lstm, name)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
            if self.lstm is not None:
                self.lstm._flat_weights = [getattr(self.lstm, n) for n in self.lstm._flat_weights_names]

    def _lstm_views(self, flat):
        if self.lstm is None:
            return {}
        H, o = HIDDEN, self.mlp_count
        v = {}
        v['weight_ih_l0'] = flat[o:o + 4 * H * H].view(4 * H, H); o += 4 * H * H
        v['weight_hh_l0'] = flat[o:o + 4 * H * H].view(4 * H, H); o += 4 * H * H
        v['bias_ih_l0'] = flat[o:o + 4 * H]; o += 4 * H
        v['bias_hh_l0'] = flat[o:o + 4 * H]; o += 4 * H
        assert o == self.count
        return v

    def encoder_weight_padded(self, flat=None):
        """encoder.weight with its pad columns: [128][obs_stride] contiguous (what the GEMMs use)."""
        flat = self.flat if flat is None else flat
        return flat[:HIDDEN * self.obs_stride].view(HIDDEN, self.obs_stride)

    def unpack_actions(self, packed):
        """Kernel action words -> what the reference hands to envs: [rows] for a Discrete head, [rows, heads] for MultiDiscrete
        (head h's choice sits in bits 4h..4h+3 of the word, include/pufferlib_amd.h: pfa_mlp_dims.heads)."""
        if not self.multidiscrete:
            return packed
        shifts = torch.arange(0, 4 * len(self.nvec), 4, device=packed.device, dtype=packed.dtype)
        return (packed.unsqueeze(-1) >> shifts) & 15

    def flat_like(self):
        return torch.zeros_like(self.flat)

    def split_mlp(self, flat):
        """Views of a flat vector with the MLP parameters' names and shapes, in the reference's named_parameters order.  The
        decoder block is [A][H] weights then [A] biases; a MultiDiscrete head owns a run of rows / entries of it."""
        H, D, DP, A = HIDDEN, self.obs_dim, self.obs_stride, self.num_actions
        out = {}
        out['encoder.weight'] = flat[:H * DP].view(H, DP)[:, :D]
        out['encoder.bias'] = flat[H * DP:H * DP + H]
        o = H * DP + H
        w, b = flat[o:o + A * H].view(A, H), flat[o + A * H:o + A * H + A]
        if self.multidiscrete:
            r = 0
            for h, n in enumerate(self.nvec):
                out[f'decoder.{h}.weight'], out[f'decoder.{h}.bias'] = w[r:r + n], b[r:r + n]
                r += n
        else:
            out['decoder.weight'], out['decoder.bias'] = w, b
        o += A * H + A
        out['value_head.weight'] = flat[o:o + H].view(1, H)
        out['value_head.bias'] = flat[o + H:o + H + 1]
        assert o + H + 1 == self.mlp_count
        return out

    def split(self, flat):
        """Views of another flat vector (gradients, Adam moments) with the parameters' shapes."""
        out = self.split_mlp(flat)
        for k, v in self._lstm_views(flat).items():
            out['recurrent.' + k] = v
        return out
