"""Times the UNMODIFIED reference (PufferLib 1.0.1: clean_pufferl.create / evaluate / train over pufferlib.vector.Serial or
Multiprocessing, c_gae.pyx through pyximport) on this machine's CPU cores, on the configurations BASELINE.md section 3 names:

    c1   squared d=3 nt=1,   64 envs x 128 steps, models.Default(128), 4 minibatches x 4 epochs, bptt 16
    c2   squared d=3 nt=1, 4096 envs x 128 steps, models.Default(128)            (the configuration bench.py's headline is quoted on)
    c4   "C4-policy": envs of Atari-shaped uint8 (4, 84, 84) frames (uniform [0, 255], 100-step episodes, 4 actions: a gymnasium env
         defined HERE, wrapped by the reference's EpisodeStats + GymnasiumPufferEnv), pufferlib.models.Convolutional (models.py:113-157,
         the NatureCNN) behind frameworks.cleanrl.Policy, horizon 32, 4 minibatches x 4 epochs — a bounded sample (default 256 envs) of
         BASELINE configs[3]'s 8192
    c3   "C3-policy": 4096 envs of MiniGrid-shaped 160-byte uint8 rows (uniform [0, 10], 100-step episodes, 7 actions: a gymnasium
         env defined HERE, wrapped by the reference's EpisodeStats + GymnasiumPufferEnv), LSTMWrapper(Default(128)) 128, bptt 16

The reference is imported read-only from --reference (default: the staged copy `_refstage/` that tools/gpu_jobs/with_reference.sh
ships to the GPU box, else /root/reference); gymnasium / gym / pettingzoo come from tests/shims.  One JSON line per run; with
--out the line is also appended to that file (tools/gpu_jobs/with_reference.sh collects them into
profiles/r06_reference_cpu_on_gpu_box.json).

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py --config c2 [--backend serial|multiprocessing] [--workers W]
                                                             [--threads T] [--iters K] [--reference DIR] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True

import warnings  # noqa: E402
warnings.filterwarnings('ignore')


def cpu_model():
    for line in open('/proc/cpuinfo'):
        if line.startswith('model name'):
            return line.split(':', 1)[1].strip()
    return 'unknown'


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except Exception:
        return os.cpu_count() or 1


def make_rows_env(obs_bytes=160, num_actions=7, episode_length=100):
    """A gymnasium env with the shapes of config C3: the simulator itself (minigrid 2.3.1) is third-party and absent."""
    import gymnasium
    import numpy as np
    import pufferlib.emulation
    import pufferlib.postprocess

    class Rows(gymnasium.Env):
        def __init__(self):
            self.observation_space = gymnasium.spaces.Box(low=0, high=255, shape=(obs_bytes,), dtype=np.uint8)
            self.action_space = gymnasium.spaces.Discrete(num_actions)
            self.rs = np.random.RandomState(0)
            self.render_mode = 'ansi'
            self.tick = 0

        def reset(self, seed=None):
            if seed is not None:
                self.rs = np.random.RandomState(seed)
            self.tick = 0
            return self.rs.randint(0, 11, obs_bytes).astype(np.uint8), {}

        def step(self, action):
            self.tick += 1
            done = self.tick >= episode_length
            return self.rs.randint(0, 11, obs_bytes).astype(np.uint8), float(self.rs.randint(0, 2)), done, False, {}

    env = Rows()
    env = pufferlib.postprocess.EpisodeStats(env)
    return pufferlib.emulation.GymnasiumPufferEnv(env=env)


def make_frames_env(num_actions=4, episode_length=100):
    """A gymnasium env with the shapes of config C4 (the emulator itself, ale-py, is third-party and absent)."""
    import gymnasium
    import numpy as np
    import pufferlib.emulation
    import pufferlib.postprocess

    class FramesEnv(gymnasium.Env):
        def __init__(self):
            self.observation_space = gymnasium.spaces.Box(low=0, high=255, shape=(4, 84, 84), dtype=np.uint8)
            self.action_space = gymnasium.spaces.Discrete(num_actions)
            self.rs = np.random.RandomState(0)
            self.render_mode = 'rgb_array'
            self.tick = 0

        def reset(self, seed=None):
            if seed is not None:
                self.rs = np.random.RandomState(seed)
            self.tick = 0
            return self.rs.randint(0, 256, (4, 84, 84), dtype=np.uint8), {}

        def step(self, action):
            self.tick += 1
            done = self.tick >= episode_length
            return self.rs.randint(0, 256, (4, 84, 84), dtype=np.uint8), float(self.rs.randint(0, 2)), done, False, {}

    env = FramesEnv()
    env = pufferlib.postprocess.EpisodeStats(env)
    return pufferlib.emulation.GymnasiumPufferEnv(env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', choices=['c1', 'c2', 'c3', 'c4'], default='c2')
    ap.add_argument('--backend', choices=['serial', 'multiprocessing'], default='serial')
    ap.add_argument('--workers', type=int, default=None, help='Multiprocessing: worker processes (default: physical cores, capped so that envs % workers == 0)')
    ap.add_argument('--envs', type=int, default=None)
    ap.add_argument('--horizon', type=int, default=None)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--threads', type=int, default=None, help='torch intra-op threads (default: physical cores)')
    ap.add_argument('--reference', default=None)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    ref = args.reference or (os.path.join(REPO, '_refstage') if os.path.exists(os.path.join(REPO, '_refstage', 'demo.py')) else '/root/reference')
    sys.path[:0] = [os.path.join(REPO, 'tests', 'shims'), ref]
    phys = physical_cores()
    threads = args.threads or phys
    import torch
    torch.set_num_threads(threads)
    import pufferlib
    import pufferlib.vector
    import pufferlib.models
    import pufferlib.frameworks.cleanrl
    import pufferlib.environments.ocean as ocean
    import clean_pufferl
    assert os.path.abspath(clean_pufferl.__file__).startswith(os.path.abspath(ref)), clean_pufferl.__file__

    class _NoUtil:
        def __init__(self, *a, **k):
            self.cpu_util = self.cpu_mem = self.gpu_util = self.gpu_mem = [0]

        def stop(self):
            pass

    clean_pufferl.Utilization = _NoUtil             # (the monitor thread polls torch.cuda / psutil; not part of the hot path)
    clean_pufferl.print_dashboard = lambda *a, **k: None
    clean_pufferl.save_checkpoint = lambda data: None
    N = args.envs or {'c1': 64, 'c2': 4096, 'c3': 4096, 'c4': 256}[args.config]
    T = args.horizon or (32 if args.config == 'c4' else 128)
    B = N * T
    config = pufferlib.namespace(
        env='squared', seed=1, torch_deterministic=True, cpu_offload=False, device='cpu', total_timesteps=B * 1000,
        learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95, update_epochs=4, norm_adv=True, clip_coef=0.1,
        clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01, target_kl=None, batch_size=B,
        minibatch_size=B // 4, bptt_horizon=16, compile=False, compile_mode='reduce-overhead', checkpoint_interval=10 ** 9,
        data_dir='/tmp/ref_timing', exp_id='timing')
    creator = {'c3': make_rows_env, 'c4': make_frames_env}.get(args.config) or ocean.env_creator('squared')
    workers = None
    t_build = time.perf_counter()
    if args.backend == 'serial':
        vec = pufferlib.vector.make(creator, num_envs=N, backend=pufferlib.vector.Serial)
    else:
        workers = args.workers or phys
        while N % workers:
            workers -= 1
        vec = pufferlib.vector.make(creator, num_envs=N, num_workers=workers, batch_size=N, backend=pufferlib.vector.Multiprocessing)
    t_build = time.perf_counter() - t_build
    torch.manual_seed(1)
    if args.config == 'c3':
        policy = pufferlib.frameworks.cleanrl.RecurrentPolicy(
            pufferlib.models.LSTMWrapper(vec.driver_env, pufferlib.models.Default(vec.driver_env, hidden_size=128), input_size=128, hidden_size=128))
    elif args.config == 'c4':
        policy = pufferlib.frameworks.cleanrl.Policy(pufferlib.models.Convolutional(vec.driver_env, framestack=4, flat_size=64 * 7 * 7))
    else:
        policy = pufferlib.frameworks.cleanrl.Policy(pufferlib.models.Default(vec.driver_env, hidden_size=128))
    data = clean_pufferl.create(config, vec, policy)
    t_eval = t_train = 0.0
    clean_pufferl.evaluate(data)                   # warm-up iteration (first-call costs: pyximport of c_gae, allocator)
    clean_pufferl.train(data)
    prof0 = {k: getattr(data.profile, k).elapsed for k in ('env', 'eval_forward', 'eval_misc', 'train_forward', 'learn', 'train_misc')}
    t0 = time.perf_counter()
    for _ in range(args.iters):
        a = time.perf_counter()
        clean_pufferl.evaluate(data)
        b = time.perf_counter()
        clean_pufferl.train(data)
        t_eval += b - a
        t_train += time.perf_counter() - b
    dt = time.perf_counter() - t0
    prof = {k + '_s_per_iter': round((getattr(data.profile, k).elapsed - prof0[k]) / args.iters, 4) for k in prof0}
    vec.close()
    what = {'c1': 'C1: squared d=3 nt=1, models.Default(128)', 'c2': 'C2: squared d=3 nt=1, models.Default(128)',
            'c3': 'C3-policy: 160-byte uint8 rows (uniform [0,10], 100-step episodes, 7 actions), LSTMWrapper(Default(128)) 128',
            'c4': 'C4-policy: uint8 (4,84,84) frames (uniform [0,255], 100-step episodes, 4 actions), models.Convolutional (NatureCNN)'}[args.config]
    line = dict(config=args.config,
                what='the unmodified reference (clean_pufferl.create/evaluate/train + pufferlib.vector.%s + c_gae.pyx), %s'
                     % ('Serial' if args.backend == 'serial' else f'Multiprocessing x{workers} workers', what),
                backend=args.backend, workers=workers, value=args.iters * B / dt, unit='env_steps/s', envs=N, horizon=T, batch=B,
                minibatches=4, epochs=4, bptt=16, iterations=args.iters, warmup_iterations=1,
                evaluate_s_per_iter=t_eval / args.iters, train_s_per_iter=t_train / args.iters, profile=prof, vec_build_s=round(t_build, 2),
                cores_used=(1 if args.backend == 'serial' else workers), torch_threads=threads,
                cores_physical=phys, cores_logical=os.cpu_count(), cpu_model=cpu_model(),
                torch=torch.__version__, date=time.strftime('%Y-%m-%d'), reference_dir=os.path.relpath(ref, REPO) if ref.startswith(REPO) else ref)
    s = json.dumps(line)
    print(s, flush=True)
    if args.out:
        with open(args.out, 'a') as f:
            f.write(s + '\n')
    os._exit(0)   # (Multiprocessing workers are daemons spinning on shared memory; do not wait for interpreter teardown)


if __name__ == '__main__':
    main()
