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
    """Linear(obs -> 128) + ReLU, Linear(128 -> n_actions) (orthogonal, std 0.01), Linear(128 -> 1)."""

    def __init__(self, env, hidden_size=HIDDEN):
        super().__init__()
        if hidden_size != HIDDEN:
            raise ValueError(f'pufferlib_amd.models.Default supports hidden_size={HIDDEN} only')
        self.obs_dim = int(np.prod(env.single_observation_space.shape))
        self.num_actions = int(env.single_action_space.n)
        self.encoder = nn.Linear(self.obs_dim, hidden_size)
        self.decoder = layer_init(nn.Linear(hidden_size, self.num_actions), std=0.01)
        self.value_head = nn.Linear(hidden_size, 1)
        self.is_multidiscrete = False

    def forward(self, observations):
        raise RuntimeError('pufferlib_amd.models.Default is a parameter container: call it through '
                           'pufferlib_amd.cleanrl.Policy / pufferlib_amd.clean_pufferl (HIP kernels)')


class LSTMWrapper(nn.Module):
    """pufferlib.models.LSTMWrapper (models.py:64-111): policy.encode_observations -> nn.LSTM(128, 128, 1) ->
    policy.decode_actions; LSTM weights orthogonal (gain 1), biases 0.  Parameter container like ``Default``."""

    def __init__(self, env, policy, input_size=HIDDEN, hidden_size=HIDDEN, num_layers=1):
        super().__init__()
        if (input_size, hidden_size, num_layers) != (HIDDEN, HIDDEN, 1):
            raise ValueError(f'pufferlib_amd.models.LSTMWrapper supports nn.LSTM({HIDDEN}, {HIDDEN}, 1) only')
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


LSTM_KEYS = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']


def find_lstm(module):
    """The nn.LSTM of a recurrent policy wrapper (ours or the reference's LSTMWrapper: attribute ``recurrent``)."""
    for m in module.modules():
        if isinstance(m, nn.LSTM):
            return m
    return None


MLP_KEYS = ['encoder.weight', 'encoder.bias', 'decoder.weight', 'decoder.bias', 'value_head.weight', 'value_head.bias']


def find_mlp(module):
    """Locate the Default-shaped submodule inside a policy wrapper (ours or the reference's
    frameworks.cleanrl.Policy -> .policy).  Returns the module owning encoder/decoder/value_head."""
    for m in module.modules():
        if all(hasattr(m, n) for n in ('encoder', 'decoder', 'value_head')) and isinstance(getattr(m, 'encoder'), nn.Linear):
            if isinstance(m.decoder, nn.Linear):
                return m
    raise ValueError('policy has no encoder/decoder/value_head Linear layers (models.Default shape)')


class FlatParams:
    """One flat fp32 device buffer holding the MLP parameters in kernel layout + views for the nn.Module."""

    def __init__(self, policy_module, obs_stride, device):
        mlp = find_mlp(policy_module)
        self.mlp = mlp
        H, D = mlp.encoder.weight.shape
        A = mlp.decoder.weight.shape[0]
        if H != HIDDEN:
            raise ValueError(f'hidden size must be {HIDDEN}')
        if not (mlp.value_head.weight.shape == (1, H) and mlp.decoder.weight.shape[1] == H):
            raise ValueError('unexpected head shapes')
        self.dims = _lib.MlpDims(int(D), int(obs_stride), int(H), int(A))
        self.obs_dim, self.obs_stride, self.num_actions = int(D), int(obs_stride), int(A)
        DP = self.obs_stride
        self.mlp_count = H * DP + H + A * H + A + H + 1
        self.lstm = find_lstm(policy_module)
        if self.lstm is not None:
            if (self.lstm.input_size, self.lstm.hidden_size, self.lstm.num_layers) != (H, H, 1) or self.lstm.bidirectional:
                raise ValueError(f'the recurrent policy must use nn.LSTM({H}, {H}, 1)')
        self.count = self.mlp_count + (8 * H * H + 8 * H if self.lstm is not None else 0)
        self.flat = torch.zeros(self.count, dtype=torch.float32, device=device)
        o = 0
        self.views = {}
        self.views['encoder.weight'] = self.flat[o:o + H * DP].view(H, DP)[:, :D]; o += H * DP
        self.views['encoder.bias'] = self.flat[o:o + H]; o += H
        self.views['decoder.weight'] = self.flat[o:o + A * H].view(A, H); o += A * H
        self.views['decoder.bias'] = self.flat[o:o + A]; o += A
        self.views['value_head.weight'] = self.flat[o:o + H].view(1, H); o += H
        self.views['value_head.bias'] = self.flat[o:o + 1]; o += 1
        assert o == self.mlp_count
        self.lstm_views = self._lstm_views(self.flat)
        with torch.no_grad():
            for name, view in self.views.items():
                mod, attr = name.split('.')
                p = getattr(getattr(mlp, mod), attr)
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

    def flat_like(self):
        return torch.zeros_like(self.flat)

    def split(self, flat):
        """Views of another flat vector (gradients, Adam moments) with the parameters' shapes."""
        H, D, DP, A = HIDDEN, self.obs_dim, self.obs_stride, self.num_actions
        o = 0
        out = {}
        out['encoder.weight'] = flat[o:o + H * DP].view(H, DP)[:, :D]; o += H * DP
        out['encoder.bias'] = flat[o:o + H]; o += H
        out['decoder.weight'] = flat[o:o + A * H].view(A, H); o += A * H
        out['decoder.bias'] = flat[o:o + A]; o += A
        out['value_head.weight'] = flat[o:o + H].view(1, H); o += H
        out['value_head.bias'] = flat[o:o + 1]
        for k, v in self._lstm_views(flat).items():
            out['recurrent.' + k] = v
        return out
