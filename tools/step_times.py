"""Developer tool (GPU box): wall time of every evaluate() + train() iteration of the headline workload from a cold start — what the
first iterations after create() cost against the steady state (clock ramp, lazily created buffers / events, the first tape rounds).
    python tools/step_times.py [iterations]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    torch.cuda.set_device(0)
    vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=bench.D, num_targets=bench.NT), num_envs=bench.NUM_ENVS,
                      backend=vector.Squared, obs_stride=64)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(bench.make_config(bench.NUM_ENVS * bench.HORIZON * (n + 8) * 2), vec, pol)
    torch.cuda.synchronize()
    out = []
    for i in range(n):
        t0 = time.perf_counter()
        clean_pufferl.evaluate(data)
        t1 = time.perf_counter()
        clean_pufferl.train(data)
        t2 = time.perf_counter()
        out.append((round((t1 - t0) * 1e6, 1), round((t2 - t1) * 1e6, 1)))
    tot = [a + b for a, b in out]
    print('iteration us (evaluate, train):', out[:12])
    for lo, hi in ((0, 5), (5, 10), (10, 25), (25, 45), (45, n)):
        if hi <= n:
            seg = tot[lo:hi]
            print(f'iterations {lo:3d}..{hi - 1:3d}: mean {sum(seg) / len(seg):8.1f} us  min {min(seg):8.1f}  max {max(seg):8.1f}')
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'step_times.json'), 'w'))


if __name__ == '__main__':
    main()
