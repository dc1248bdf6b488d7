"""CPU ORACLE (test infrastructure, NOT product code): torch-fp32 restatement of the policy
forward / sampling / PPO update part of the hot path.

The floating-point arithmetic is owned by PyTorch in the reference too, so this oracle uses the
same torch CPU ops in the same order; it restates the *control flow and data layout* of

  pufferlib/models.py:41-62          Default.forward (encode -> relu -> decoder / value_head)
  pufferlib/models.py:84-111         LSTMWrapper.forward
  pufferlib/models.py:113-157        Convolutional (NatureCNN) encode / decode
  pufferlib/frameworks/cleanrl.py:12-47   log_prob / entropy / sample_logits
  clean_pufferl.py:76-154            evaluate (rollout + Experience.store, :436-450)
  clean_pufferl.py:157-271           train (sort :452-464, GAE, flatten :466-482, minibatch loop)

Parity status: PINNED — tests/test_oracle_golden.py replays tests/golden/ppo_{mlp,lstm,cnn}.npz (outputs of
the unmodified reference) through this file and requires identical actions / experience buffers and
post-update weights, Adam moments and losses.

The env side is oracle/c_oracle.SquaredSerial.  Action sampling takes the exponential noise ``q``
explicitly: torch.multinomial(p, 1) on CPU == argmax(p / q) with q ~ Exp(1).
"""
import numpy as np
import torch

from . import c_oracle


class Policy:
    """Weights of models.Default (+ optional nn.LSTM) held as torch leaf tensors, in the
    reference's ``named_parameters`` order so that clip_grad_norm_/Adam see the same sequence."""

    MLP_NAMES = ['encoder.weight', 'encoder.bias', 'decoder.weight', 'decoder.bias',
                 'value_head.weight', 'value_head.bias']
    LSTM_NAMES = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']

    def __init__(self, weights, recurrent=False):
        """``weights``: dict name -> array using the short names above.  A MultiDiscrete policy (models.py:29-35: one
        decoder Linear per head) is recognised by its ``decoder.<h>.weight`` keys; parameters keep the reference's
        named_parameters order."""
        self.recurrent = recurrent
        self.heads = None
        if 'decoder.0.weight' in weights:
            nh = len([k for k in weights if k.startswith('decoder.') and k.endswith('.weight')])
            self.heads = [int(np.asarray(weights[f'decoder.{h}.weight']).shape[0]) for h in range(nh)]
            dec = [f'decoder.{h}.{w}' for h in range(nh) for w in ('weight', 'bias')]
            self.names = ['encoder.weight', 'encoder.bias'] + dec + ['value_head.weight', 'value_head.bias']
        else:
            self.names = list(self.MLP_NAMES)
        self.names += list(self.LSTM_NAMES) if recurrent else []
        self.params = [torch.tensor(np.asarray(weights[n]), dtype=torch.float32, requires_grad=True)
                       for n in self.names]

    @classmethod
    def from_reference_state_dict(cls, sd, prefix=''):
        """Accepts the reference's state_dict keys (cleanrl.Policy -> 'policy.encoder.weight';
        RecurrentPolicy(LSTMWrapper) -> 'policy.policy.encoder.weight', 'policy.recurrent.weight_ih_l0')."""
        keys = [k[len(prefix):] for k in sd if k.startswith(prefix)]
        recurrent = any('recurrent.' in k for k in keys)
        w = {}
        base = 'policy.policy.' if recurrent else 'policy.'
        for k in keys:
            if k.startswith(base) and 'recurrent.' not in k:
                w[k[len(base):]] = sd[prefix + k]
        if recurrent:
            for n in cls.LSTM_NAMES:
                w[n] = sd[prefix + 'policy.recurrent.' + n]
        return cls(w, recurrent)

    def p(self, name):
        return self.params[self.names.index(name)]

    def state_arrays(self):
        return {n: p.detach().numpy().copy() for n, p in zip(self.names, self.params)}

    # models.py:46-51
    def encode(self, obs):
        obs = obs.reshape(obs.shape[0], -1).float()
        return torch.relu(torch.nn.functional.linear(obs, self.p('encoder.weight'), self.p('encoder.bias')))

    # models.py:53-62
    def decode(self, hidden):
        value = torch.nn.functional.linear(hidden, self.p('value_head.weight'), self.p('value_head.bias'))
        if self.heads is not None:      # models.py:55-58: a list of per-head logits
            return [torch.nn.functional.linear(hidden, self.p(f'decoder.{h}.weight'), self.p(f'decoder.{h}.bias'))
                    for h in range(len(self.heads))], value
        logits = torch.nn.functional.linear(hidden, self.p('decoder.weight'), self.p('decoder.bias'))
        return logits, value

    def lstm(self, x, state):
        """nn.LSTM(H, H, 1) on (TT, B, H) with gate order i,f,g,o (models.py:76,103-105)."""
        w_ih, w_hh = self.p('weight_ih_l0'), self.p('weight_hh_l0')
        b_ih, b_hh = self.p('bias_ih_l0'), self.p('bias_hh_l0')
        TT, B, _ = x.shape
        H = w_hh.shape[1]            # nn.LSTM(input_size, hidden_size): the state has hidden_size columns, x has input_size
        if state is None:
            h = torch.zeros(B, H)
            c = torch.zeros(B, H)
        else:
            h, c = state[0][0], state[1][0]
        outs = []
        for t in range(TT):
            gates = torch.nn.functional.linear(x[t], w_ih, b_ih) + torch.nn.functional.linear(h, w_hh, b_hh)
            i, f, g, o = gates.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs), (h.unsqueeze(0), c.unsqueeze(0))

    def forward(self, obs, state=None):
        """Default.forward, or LSTMWrapper.forward when recurrent: obs (B, ...) or (B, TT, ...)."""
        if not self.recurrent:
            hidden = self.encode(obs)
            logits, value = self.decode(hidden)
            return logits, value, None
        if state is None and obs.dim() == 2:
            B, TT = obs.shape[0], 1
        elif obs.dim() == 2:
            B, TT = obs.shape[0], 1
        else:
            B, TT = obs.shape[:2]
        hidden = self.encode(obs.reshape(B * TT, -1))
        H = hidden.shape[1]
        hidden = hidden.reshape(B, TT, H).transpose(0, 1)
        hidden, state = self.lstm(hidden, state)
        hidden = hidden.transpose(0, 1).reshape(B * TT, H)
        logits, value = self.decode(hidden)
        return logits, value, state


class ConvPolicy:
    """Weights of models.Convolutional (pufferlib/models.py:113-157, the CleanRL NatureCNN) as torch leaf tensors in the
    reference's named_parameters order.  Same interface as ``Policy`` for the Trainer below."""
    NAMES = ['network.0.weight', 'network.0.bias', 'network.2.weight', 'network.2.bias', 'network.4.weight', 'network.4.bias',
             'network.7.weight', 'network.7.bias', 'actor.weight', 'actor.bias', 'value_fn.weight', 'value_fn.bias']
    recurrent = False
    heads = None

    def __init__(self, weights, framestack=4, dtype=torch.float32):
        """dtype=torch.float64 (with torch.set_default_dtype(torch.float64) around the Trainer) gives the same arithmetic in double
        precision: the yardstick for how far fp32 summation order alone moves a result."""
        self.names = list(self.NAMES)
        self.framestack = framestack
        self.params = [torch.tensor(np.asarray(weights[n]), dtype=dtype, requires_grad=True) for n in self.names]

    p = Policy.p
    state_arrays = Policy.state_arrays

    # models.py:150-154: observations / 255.0 through Conv(8,4) ReLU Conv(4,2) ReLU Conv(3,1) ReLU Flatten Linear ReLU
    def encode(self, obs):
        F = torch.nn.functional
        x = obs.reshape(obs.shape[0], self.framestack, 84, 84) / 255.0
        x = torch.relu(F.conv2d(x, self.p('network.0.weight'), self.p('network.0.bias'), stride=4))
        x = torch.relu(F.conv2d(x, self.p('network.2.weight'), self.p('network.2.bias'), stride=2))
        x = torch.relu(F.conv2d(x, self.p('network.4.weight'), self.p('network.4.bias'), stride=1))
        return torch.relu(F.linear(x.flatten(1), self.p('network.7.weight'), self.p('network.7.bias')))

    # models.py:156-157
    def decode(self, hidden):
        F = torch.nn.functional
        return F.linear(hidden, self.p('actor.weight'), self.p('actor.bias')), F.linear(hidden, self.p('value_fn.weight'), self.p('value_fn.bias'))

    def forward(self, obs, state=None):
        logits, value = self.decode(self.encode(obs))
        return logits, value, None


class RecurrentConvPolicy(ConvPolicy):
    """environments/atari/torch.py:4-6: ``Recurrent`` = pufferlib.models.LSTMWrapper(input_size=512, hidden_size=512) over the
    NatureCNN ``Policy`` — encode_observations -> nn.LSTM -> decode_actions (models.py:84-111).  Parameters in the reference's
    named_parameters order: the wrapped policy's first, then the LSTM's."""
    recurrent = True
    LSTM_NAMES = Policy.LSTM_NAMES

    def __init__(self, weights, framestack=4):
        self.names = list(ConvPolicy.NAMES) + list(self.LSTM_NAMES)
        self.framestack = framestack
        self.params = [torch.tensor(np.asarray(weights[n]), dtype=torch.float32, requires_grad=True) for n in self.names]

    lstm = Policy.lstm

    def forward(self, obs, state=None):
        """obs (B, F*84*84) [one step] or (B, TT, F*84*84)."""
        if obs.dim() == 2:
            B, TT = obs.shape[0], 1
        else:
            B, TT = obs.shape[:2]
        hidden = self.encode(obs.reshape(B * TT, -1).float())
        H = hidden.shape[1]
        hidden = hidden.reshape(B, TT, H).transpose(0, 1)
        hidden, state = self.lstm(hidden, state)
        hidden = hidden.transpose(0, 1).reshape(B * TT, -1)
        logits, value = self.decode(hidden)
        return logits, value, state


def sample_logits(logits, action=None, noise=None):
    """cleanrl.py:25-47.  ``noise`` replaces torch.multinomial's internal exponential draw:
    action = argmax(softmax(logits) / noise).  A list of logits is the MultiDiscrete branch: one draw per head (noise
    columns concatenated in head order), actions [batch, heads], log-probabilities and entropies summed over heads."""
    if isinstance(logits, (list, tuple)):
        batch = logits[0].shape[0]
        cols = np.cumsum([0] + [l.shape[1] for l in logits])
        acts, logprob, entropy = [], 0.0, 0.0
        for h, l in enumerate(logits):
            a_h = None if action is None else action.reshape(batch, -1)[:, h]
            n_h = None if noise is None else noise[:, cols[h]:cols[h + 1]]
            a_h, lp, en = sample_logits(l, a_h, n_h)
            acts.append(a_h)
            logprob, entropy = logprob + lp, entropy + en
        return torch.stack(acts, dim=1), logprob, entropy
    normalized = logits - logits.logsumexp(dim=-1, keepdim=True)
    if action is None:
        probs = torch.softmax(logits, dim=-1)
        action = (probs / noise).argmax(dim=-1)
    action = action.long()
    logprob = normalized.gather(-1, action.unsqueeze(-1)).squeeze(-1)
    min_real = torch.finfo(normalized.dtype).min
    clamped = torch.clamp(normalized, min=min_real)
    entropy = -(clamped * torch.softmax(clamped, dim=-1)).sum(-1)
    return action, logprob, entropy


class Trainer:
    """clean_pufferl.create/evaluate/train restated for one Serial vecenv whose recv() returns all
    N envs every step (so Experience.store appends whole steps; the (env_id, step) sort is a transpose)."""

    def __init__(self, policy, vec, *, batch_size, minibatch_size, bptt_horizon, update_epochs,
                 learning_rate, gamma, gae_lambda, clip_coef, vf_coef, vf_clip_coef, max_grad_norm,
                 ent_coef, total_timesteps, anneal_lr=True, norm_adv=True, clip_vloss=True, seed=1,
                 target_kl=None):
        self.policy, self.vec = policy, vec
        self.N = vec.num_envs
        self.B, self.mbs, self.horizon, self.epochs = batch_size, minibatch_size, bptt_horizon, update_epochs
        assert batch_size % minibatch_size == 0 and minibatch_size % bptt_horizon == 0
        self.nmb = batch_size // minibatch_size
        self.rows = minibatch_size // bptt_horizon
        self.lr0, self.gamma, self.lam = learning_rate, gamma, gae_lambda
        self.clip, self.vf_coef, self.vf_clip = clip_coef, vf_coef, vf_clip_coef
        self.max_grad_norm, self.ent_coef = max_grad_norm, ent_coef
        self.total_timesteps, self.anneal_lr = total_timesteps, anneal_lr
        self.norm_adv, self.clip_vloss, self.target_kl = norm_adv, clip_vloss, target_kl
        self.opt = torch.optim.Adam(policy.params, lr=learning_rate, eps=1e-5)  # clean_pufferl.py:54-55
        self.global_step = 0
        self.epoch = 0
        D = int(np.prod(vec.observations.shape[1:]))
        self.obs = torch.zeros(batch_size, D)
        self.actions = np.zeros((batch_size,) + ((len(policy.heads),) if policy.heads is not None else ()), np.int64)
        self.logprobs = np.zeros(batch_size, np.float32)
        self.rewards = np.zeros(batch_size, np.float32)
        self.dones = np.zeros(batch_size, np.float32)
        self.values = np.zeros(batch_size, np.float32)
        self.lstm_h = self.lstm_c = None
        if policy.recurrent:
            H = policy.p('weight_hh_l0').shape[1]
            self.lstm_h = torch.zeros(1, self.N, H)
            self.lstm_c = torch.zeros(1, self.N, H)
        vec.async_reset(seed)  # clean_pufferl.py:39
        self.stats = {}
        self.losses = {}

    def evaluate(self, noise):
        """noise: (T, N, A) exponential variates, one slab per rollout step."""
        ptr, step = 0, 0
        infos = {'episode_return': [], 'episode_length': [], 'score': []}
        while ptr < self.B:
            o, r, d, t, info, env_id, mask = self.vec.recv()
            self.global_step += int(mask.sum())
            o_t = torch.as_tensor(np.ascontiguousarray(o)).reshape(self.N, -1)
            with torch.no_grad():
                if self.policy.recurrent:
                    logits, value, (h, c) = self.policy.forward(o_t, (self.lstm_h, self.lstm_c))
                    self.lstm_h, self.lstm_c = h, c
                else:
                    logits, value, _ = self.policy.forward(o_t)
                action, logprob, _ = sample_logits(logits, noise=torch.as_tensor(noise[step]))
            n = min(self.N, self.B - ptr)
            self.obs[ptr:ptr + n] = o_t[:n]
            self.values[ptr:ptr + n] = value.flatten().numpy()[:n]
            self.actions[ptr:ptr + n] = action.numpy()[:n]
            self.logprobs[ptr:ptr + n] = logprob.numpy()[:n]
            self.rewards[ptr:ptr + n] = r[:n]
            self.dones[ptr:ptr + n] = d[:n]
            ptr += n
            step += 1
            for i in info:
                for k in infos:
                    infos[k].append(i[k])
            self.vec.send(action.numpy())
        self.stats = {k: float(np.mean(v)) for k, v in infos.items() if len(v)}
        return self.stats

    def train(self):
        N, B, nmb, rows, hz, mbs = self.N, self.B, self.nmb, self.rows, self.horizon, self.mbs
        T = B // N
        # Experience.sort_training_data: argsort by (env_id, step) of a step-major store == transpose
        idxs = np.arange(B).reshape(T, N).T.reshape(-1)
        adv_np = c_oracle.compute_gae(self.dones[idxs], self.values[idxs], self.rewards[idxs], self.gamma, self.lam)
        b_idxs = torch.as_tensor(idxs.reshape(rows, nmb, hz).transpose(1, 0, 2))   # (nmb, rows, hz)
        b_flat = b_idxs.reshape(nmb, mbs)
        adv = torch.from_numpy(adv_np)
        b_adv = adv.reshape(rows, nmb, hz).transpose(0, 1).reshape(nmb, mbs)
        b_obs = self.obs[b_idxs]
        b_act = torch.as_tensor(self.actions)[b_idxs]
        b_logp = torch.as_tensor(self.logprobs)[b_idxs]
        b_val = torch.as_tensor(self.values)[b_flat]
        b_ret = b_adv + b_val
        self.b_advantages, self.b_returns, self.b_idxs = b_adv, b_ret, b_idxs
        returns_np = adv_np + self.values            # (sic) clean_pufferl.py:476, mis-aligned on purpose

        L = dict(policy_loss=0.0, value_loss=0.0, entropy=0.0, old_approx_kl=0.0, approx_kl=0.0, clipfrac=0.0)
        for epoch in range(self.epochs):
            state = None
            for mb in range(nmb):
                obs, atn = b_obs[mb], b_act[mb]
                if self.policy.recurrent:
                    logits, newvalue, state = self.policy.forward(obs, state)
                    state = (state[0].detach(), state[1].detach())
                    _, newlogprob, entropy = sample_logits(logits, action=atn.reshape(-1))
                else:
                    logits, newvalue, _ = self.policy.forward(obs.reshape(mbs, -1))
                    _, newlogprob, entropy = sample_logits(logits, action=atn.reshape(mbs, -1) if self.policy.heads else atn.reshape(-1))
                logratio = newlogprob - b_logp[mb].reshape(-1)
                ratio = logratio.exp()
                with torch.no_grad():
                    old_approx_kl = (-logratio).mean()
                    approx_kl = ((ratio - 1) - logratio).mean()
                    clipfrac = ((ratio - 1.0).abs() > self.clip).float().mean()
                a = b_adv[mb].reshape(-1)
                if self.norm_adv:
                    a = (a - a.mean()) / (a.std() + 1e-8)
                pg_loss = torch.max(-a * ratio, -a * torch.clamp(ratio, 1 - self.clip, 1 + self.clip)).mean()
                newvalue = newvalue.view(-1)
                ret, val = b_ret[mb], b_val[mb]
                if self.clip_vloss:
                    v_unclipped = (newvalue - ret) ** 2
                    v_clipped = val + torch.clamp(newvalue - val, -self.vf_clip, self.vf_clip)
                    v_loss = 0.5 * torch.max(v_unclipped, (v_clipped - ret) ** 2).mean()
                else:
                    v_loss = 0.5 * ((newvalue - ret) ** 2).mean()
                entropy_loss = entropy.mean()
                loss = pg_loss - self.ent_coef * entropy_loss + v_loss * self.vf_coef
                self.opt.zero_grad()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.policy.params, self.max_grad_norm)
                self.opt.step()
                L['policy_loss'] += pg_loss.item() / nmb
                L['value_loss'] += v_loss.item() / nmb
                L['entropy'] += entropy_loss.item() / nmb
                L['old_approx_kl'] += old_approx_kl.item() / nmb
                L['approx_kl'] += approx_kl.item() / nmb
                L['clipfrac'] += clipfrac.item() / nmb
            if self.target_kl is not None and approx_kl > self.target_kl:
                break
        if self.anneal_lr:
            frac = 1.0 - self.global_step / self.total_timesteps
            self.opt.param_groups[0]['lr'] = frac * self.lr0
        var_y = np.var(returns_np)
        L['explained_variance'] = np.nan if var_y == 0 else 1 - np.var(returns_np - self.values) / var_y
        self.epoch += 1
        self.losses = L
        return L

    def adam_moments(self):
        st = self.opt.state_dict()['state']
        m = {n: st[i]['exp_avg'].numpy().copy() for i, n in enumerate(self.policy.names)}
        v = {n: st[i]['exp_avg_sq'].numpy().copy() for i, n in enumerate(self.policy.names)}
        return m, v
