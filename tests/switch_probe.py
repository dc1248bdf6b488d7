"""Worker of tests/test_gpu_switches.py: one process under ONE non-default PFA_* setting (the environment is the caller's):
  (a) the reference golden replay (ppo_mlp.npz or ppo_lstm.npz: create / evaluate / train, two iterations) at the tests' tolerance;
  (b) two full-size iterations (4096 envs x 128 steps, 4 minibatches x 4 epochs, the bench workload) from fixed seeds; advantages,
      losses, episode statistics and the final weights go to --out for the caller to compare with the default-settings run."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--policy', choices=['mlp', 'lstm'], default='mlp')
    a = ap.parse_args()
    golden = os.path.join(HERE, 'golden')
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from test_gpu_ppo import _config
    if a.policy == 'mlp':
        import test_gpu_ppo
        test_gpu_ppo.test_create_evaluate_train_replays_golden(golden)
    else:
        import test_gpu_lstm
        test_gpu_lstm.test_create_evaluate_train_replays_golden_lstm(golden)
    n, horizon = (4096, 128) if a.policy == 'mlp' else (1024, 64)
    hp = [2.5e-4, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
    torch.manual_seed(11)
    vec = vector.make(vector.make_squared, num_envs=n, backend=vector.Squared, obs_stride=64)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if a.policy == 'lstm' else cleanrl.Policy(base)
    data = clean_pufferl.create(_config(n, horizon, n * horizon // 4, 16, 4, n * horizon * 16, hp, seed=9), vec, pol)
    out = {}
    for it in range(2):
        stats, _ = clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        out[f'{it}.stats'] = np.array([stats.get('episode_return', np.nan), stats.get('episode_length', np.nan), stats.get('score', np.nan)])
        out[f'{it}.losses'] = np.array([data.losses[k] for k in ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac',
                                                                   'explained_variance')])
        out[f'{it}.advantages'] = data.experience.advantages.cpu().numpy().copy()
        out[f'{it}.flat'] = data.flat_params.flat.cpu().numpy().copy()
        out[f'{it}.actions'] = data.experience.actions.cpu().numpy().copy()
    np.savez(a.out, **out)


if __name__ == '__main__':
    main()
