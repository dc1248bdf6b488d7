"""Timeline of the fused MLP rollout kernel (developer tool, not product code): builds a -DPFA_PROBES variant of csrc/rollout.hip
into tools/_probe/ and prints, for workgroup 0 at the bench shape (4096 envs x 128 steps, 64-float rows), where each wave's
time goes inside a step (s_memtime ticks = shader clocks).
    python tools/probe_rollout.py build     (build container: cross-compiles)
    python tools/probe_rollout.py           (GPU box)"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tools', '_probe')
SO = os.path.join(OUT, 'librollout_probe.so')


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(REPO, 'pufferlib_amd', 'csrc')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DPFA_PROBES', '-x', 'hip',
           os.path.join(src, 'rollout.hip'), os.path.join(src, 'common.cpp'), '-o', SO]
    subprocess.check_call(cmd)
    return SO


def main():
    import numpy as np
    import torch
    from pufferlib_amd import _lib, clean_pufferl, cleanrl, models, vector
    sys.path.insert(0, REPO)
    import bench
    L = C.CDLL(SO)
    vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=3, num_targets=1), num_envs=4096, backend=vector.Squared,
                      obs_stride=64)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(bench.make_config(4096 * 128 * 100), vec, pol)
    for _ in range(3):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    fp, ex = data.flat_params, data.experience
    T0, STEPS = 40, 12
    tr = torch.zeros(4, STEPS, 8, dtype=torch.int64, device='cuda')
    L.pfa_probe_set_rollout_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.pfa_rollout_mlp_squared.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
    vec.ensure_tape(128)
    torch.cuda.synchronize()
    key = _lib.NoiseKey(1, 12345)
    args = (vec.state.data_ptr(), C.byref(vec.cfg), fp.flat.data_ptr(), C.byref(fp.dims), C.byref(ex.c), None, C.byref(key), 0,
            vec.obs_buf.data_ptr(), vec.rewards.data_ptr(), vec.terminals_u8.data_ptr(), vec.truncations_u8.data_ptr(), vec.masks_u8.data_ptr(),
            None)
    assert L.pfa_probe_set_rollout_trace(tr.data_ptr(), T0, STEPS) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert L.pfa_rollout_mlp_squared(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    print(f'kernel {e0.elapsed_time(e1) * 1e3:.1f} us for 128 steps (default stream, probe build)')
    t = tr.cpu().numpy()
    names = ['store obs', 'forward', 'barrier 1', 'noise', 'sample', 'env step + scalars', 'barrier 2']
    print('ticks per phase (wave: mean over steps %d..%d)' % (T0, T0 + STEPS - 1))
    for w in range(4):
        d = np.diff(t[w, :, :8], axis=1).astype(float)                    # [steps][7]
        step = np.diff(t[w, :, 0]).astype(float)
        print(f'  wave {w}: step {step.mean():7.0f} | ' + ' | '.join(f'{n} {d[:, i].mean():6.0f}' for i, n in enumerate(names)))
    d = np.diff(t[0, :, :8], axis=1)
    for j in range(STEPS):
        print('   wave 0 step', T0 + j, d[j].tolist())


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        print(build())
    else:
        if not os.path.exists(SO):
            build()
        main()
