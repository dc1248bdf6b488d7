"""Diagnostic: HIP conv-policy update vs the torch oracle in fp32 and in fp64 on the same device rollout (which of the two fp32
results is closer to the double-precision value of each loss?)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import cnn_golden  # noqa: E402
from oracle import ppo_torch  # noqa: E402
from pufferlib_amd import clean_pufferl  # noqa: E402
from test_gpu_cnn_ppo import _trainer  # noqa: E402

n, horizon, nmb, bptt, epochs = 8, 8, 2, 4, 2
B = n * horizon
hp = [1e-3, 0.97, 0.9, 0.2, 0.5, 0.2, 0.5, 0.02]
start = cnn_golden.start_weights(cnn_golden.container())
vec, pol, data = _trainer(n, horizon, B // nmb, bptt, epochs, B * 10, hp, 3, start=start, episode_length=5)
sm = lambda x: x.view(n, horizon, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
keys = ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac', 'explained_variance')
for it in range(2):
    clean_pufferl.evaluate(data)
    e = data.experience
    w = {k[len('policy.'):]: v.cpu().numpy().copy() for k, v in pol.state_dict().items()}
    res = {}
    for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        torch.set_default_dtype(dt)
        opol = ppo_torch.ConvPolicy(w, dtype=dt)
        tr = ppo_torch.Trainer(opol, cnn_golden.ReplayVec.blank(n), batch_size=B, minibatch_size=B // nmb, bptt_horizon=bptt, update_epochs=epochs,
                               learning_rate=data.optimizer.param_groups[0]['lr'], gamma=hp[1], gae_lambda=hp[2], clip_coef=hp[3], vf_coef=hp[4],
                               vf_clip_coef=hp[5], max_grad_norm=hp[6], ent_coef=hp[7], total_timesteps=B * 10, seed=3)
        tr.obs = torch.as_tensor(sm(e.obs)).to(dt)
        tr.actions = sm(e.actions).astype(np.int64)
        tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (e.logprobs, e.rewards, e.dones, e.values))
        tr.global_step = data.global_step
        # fresh Adam state per iteration in this diagnostic: copy the device optimizer's moments so every leg starts alike
        Lo = tr.train() if it == 0 else None
        res[name] = Lo
        torch.set_default_dtype(torch.float32)
    clean_pufferl.train(data)
    L = data.losses
    if it == 0:
        print('loss            hip            torch32        torch64        |hip-64|   |t32-64|')
        for k in keys:
            h, a, b = getattr(L, k), res['f32'][k], res['f64'][k]
            print(f'{k:18s} {h: .8e} {a: .8e} {b: .8e} {abs(h - b):.2e} {abs(a - b):.2e}')
