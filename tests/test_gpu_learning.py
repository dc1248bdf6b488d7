"""Learning regression (the reference's ocean envs are "learnable tasks", run_baselines.sh): PPO through
create/evaluate/train must actually solve the device-resident envs.  Thresholds are far inside what the runs reach
(Squared: score 0.008 -> 0.999; Stochastic: 0.957 -> 0.997) so they only trip on real regressions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(kind, recurrent, updates, n=1024, horizon=128, **envkw):
    from pufferlib_amd import clean_pufferl, cleanrl, models, namespace, vector
    torch.manual_seed(0)
    if kind == 'squared':
        vec = vector.make(vector.make_squared, env_kwargs=envkw, num_envs=n, backend=vector.Squared)
    elif kind == 'bandit':
        vec = vector.make(vector.make_bandit, env_kwargs=envkw, num_envs=n, backend=vector.Bandit)
    else:
        vec = vector.make(vector.make_stochastic, env_kwargs=envkw, num_envs=n, backend=vector.Stochastic)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent else cleanrl.Policy(base)
    B = n * horizon
    cfg = namespace(env=kind, seed=1, torch_deterministic=True, cpu_offload=False, device='cuda', total_timesteps=B * updates,
                    learning_rate=2.5e-3, anneal_lr=True, gamma=0.95, gae_lambda=0.9, update_epochs=4, norm_adv=True,
                    clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01,
                    target_kl=None, batch_size=B, minibatch_size=B // 4, bptt_horizon=16, compile=False,
                    checkpoint_interval=0, data_dir='/tmp/pfa_experiments', exp_id='learn')
    data = clean_pufferl.create(cfg, vec, pol)
    first = last = None
    for _ in range(updates):
        stats, _ = clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        if first is None and 'score' in stats:
            first = stats['score']
        last = stats.get('score', last)
    assert np.isfinite(data.losses.value_loss)
    return first, last


def test_ppo_solves_squared_with_the_mlp_policy():
    first, last = _train('squared', False, 80, distance_to_target=3, num_targets=1)
    assert first < 0.1 and last > 0.95, (first, last)


def test_ppo_solves_squared_with_the_recurrent_policy():
    first, last = _train('squared', True, 40, distance_to_target=3, num_targets=1)
    assert first < 0.1 and last > 0.9, (first, last)


@pytest.mark.parametrize('p', [0.7, 0.3])
def test_ppo_learns_the_stochastic_policy(p):
    first, last = _train('stochastic', False, 60, p=p)
    assert first < 0.97 and last > 0.99, (first, last)          # score = 1 - (p - action-0 fraction)^2 at the episode end


def test_ppo_finds_the_bandit_arm_under_reward_noise():
    first, last = _train('bandit', False, 40, horizon=32)
    assert first < 0.3 and last > 0.8, (first, last)            # score = fraction of pulls on the solution arm (chance 0.1)


def test_ppo_learns_three_action_heads_on_the_host_path():
    """MultiDiscrete([3, 4, 2]) on a host vecenv whose reward is the fraction of heads matching an observation-derived target:
    chance is (1/3 + 1/4 + 1/2) / 3 = 0.36; the per-head gradient has to be right for every head to move."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from host_vecenv import HostMultiHead
    from pufferlib_amd import clean_pufferl, cleanrl, models, namespace
    torch.manual_seed(0)
    n, horizon, updates = 128, 32, 40
    vec = HostMultiHead(n, [3, 4, 2], obs_dim=8)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    B = n * horizon
    cfg = namespace(env='multihead', seed=1, torch_deterministic=True, cpu_offload=False, device='cuda', total_timesteps=B * updates,
                    learning_rate=5e-3, anneal_lr=True, gamma=0.9, gae_lambda=0.8, update_epochs=4, norm_adv=True,
                    clip_coef=0.2, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.2, max_grad_norm=0.5, ent_coef=0.005,
                    target_kl=None, batch_size=B, minibatch_size=B // 4, bptt_horizon=16, compile=False,
                    checkpoint_interval=0, data_dir='/tmp/pfa_experiments', exp_id='learn_md')
    data = clean_pufferl.create(cfg, vec, pol)
    first = last = None
    for _ in range(updates):
        stats, _ = clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
        first = stats['score'] if first is None else first
        last = stats['score']
    assert first < 0.45 and last > 0.6, (first, last)
