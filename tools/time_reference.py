"""Times the UNMODIFIED reference (PufferLib 1.0.1 at /root/reference: clean_pufferl.create/evaluate/train over
pufferlib.vector.Serial or Multiprocessing, ocean squared, models.Default) on the headline configuration of bench.py —
4096 envs x 128 steps, 4 minibatches x 4 epochs, bptt 16 — on this machine's CPU cores.  Build container only (the reference
does not travel to the GPU box); gymnasium / gym / pettingzoo come from tests/shims.  Writes one JSON line.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py [--backend serial|multiprocessing] [--workers W] [--iters K]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(REPO, 'tests', 'shims'), '/root/reference']

import warnings  # noqa: E402
warnings.filterwarnings('ignore')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=['serial', 'multiprocessing'], default='serial')
    ap.add_argument('--workers', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--horizon', type=int, default=128)
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    import torch
    torch.set_num_threads(args.threads)
    import pufferlib
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import pufferlib.environments.ocean as ocean
    import clean_pufferl

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil             # (the monitor thread polls torch.cuda / psutil; not part of the hot path)
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None
    N, T = args.envs, args.horizon
    B = N * T
    config = pufferlib.namespace(
        env='squared', seed=1, torch_deterministic=True, cpu_offload=False, device='cpu', total_timesteps=B * 1000,
        learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95, update_epochs=4, norm_adv=True, clip_coef=0.1,
        clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=B,
        minibatch_size=B // 4, bptt_horizon=16, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
        data_dir='/tmp/ref_timing', exp_id='timing')
    if args.backend == 'serial':
        vec = pufferlib.vector.make(ocean.env_creator('squared'), num_envs=N, backend=pufferlib.vector.Serial)
    else:
        vec = pufferlib.vector.make(ocean.env_creator('squared'), num_envs=N, num_workers=args.workers, batch_size=N,
                                    backend=pufferlib.vector.Multiprocessing)
    torch.manual_seed(1)
    policy = pufferlib.frameworks.cleanrl.Policy(pufferlib.models.Default(vec.driver_env, hidden_size=128))
    data = clean_pufferl.create(config, vec, policy)
    t_eval = t_train = 0.0
    clean_pufferl.evaluate(data)                   # warm-up iteration (first-call costs: pyximport of c_gae, allocator)
    clean_pufferl.train(data)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        a = time.perf_counter()
        clean_pufferl.evaluate(data)
        b = time.perf_counter()
        clean_pufferl.train(data)
        t_eval += b - a
        t_train += time.perf_counter() - b
    dt = time.perf_counter() - t0
    vec.close()
    cpu = 'unknown'
    for line in open('/proc/cpuinfo'):
        if line.startswith('model name'):
            cpu = line.split(':', 1)[1].strip()
            break
    print(json.dumps(dict(what='the unmodified reference (clean_pufferl + pufferlib.vector.%s + c_gae), squared d=3 nt=1, MLP 128'
                               % ('Serial' if args.backend == 'serial' else f'Multiprocessing x{args.workers} workers'),
                          value=args.iters * B / dt, unit='env_steps/s', envs=N, horizon=T, iterations=args.iters,
                          evaluate_s_per_iter=t_eval / args.iters, train_s_per_iter=t_train / args.iters,
                          cores_available=os.cpu_count(), torch_threads=args.threads, cpu_model=cpu,
                          box='build container (no GPU)')))


if __name__ == '__main__':
    main()
