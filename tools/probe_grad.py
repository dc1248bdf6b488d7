"""Ablation probe of the fused PPO gradient kernel (developer tool, not product code).

Builds a -DPFA_PROBES variant of csrc/ppo_update.hip into tools/_probe/, then times the kernel at the bench
shape (131 072 rows x 64 floats) with parts compiled out, to see where the time goes:
    python tools/probe_grad.py            (on the GPU box)
"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tools', '_probe')


def build():
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, 'libprobe.so')
    src = os.path.join(REPO, 'pufferlib_amd', 'csrc')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DPFA_PROBES',
           '-x', 'hip', os.path.join(src, 'ppo_update.hip'), os.path.join(src, 'common.cpp'), os.path.join(src, 'dist.cpp'), '-ldl', '-o', so]
    subprocess.check_call(cmd)
    return so


def main():
    import torch
    from pufferlib_amd import _lib
    so = os.path.join(OUT, 'libprobe.so')
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    N, T, DP, A, NMB = 4096, 128, 64, 8, 4
    B = N * T
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(B, DP, device=dev, generator=g)
    exp = _lib.Experience(obs.data_ptr(), *(t.data_ptr() for t in (
        torch.randint(0, A, (B,), device=dev, dtype=torch.int32, generator=g),
        torch.full((B,), -2.0794, device=dev), torch.randn(B, device=dev, generator=g),
        torch.randn(B, device=dev, generator=g), torch.zeros(B, device=dev),
        torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g))), T)
    keep = [obs]
    dims = _lib.MlpDims(49, DP, 128, A)
    hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, NMB, 16)
    P = 128 * DP + 128 + A * 128 + A + 128 + 1
    params = torch.randn(P, device=dev, generator=g) * 0.05
    grads = torch.zeros(P + 8, device=dev)
    ws = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
    L.pfa_probe_grad.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_void_p]
    names = {0: 'full', 1: 'no loss math', 2: 'no backward', 4: 'no forward', 6: 'no fwd+bwd (loss, staging, reduce)',
             7: 'staging + reduce only', 8: 'no workgroup reduce', 15: 'staging only'}
    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    # fixed cost vs per-tile cost: shrink the minibatch so every pair gets 1, 2, 4, 8 tiles (grid stays 256 WGs)
    for tiles_per_pair in (1, 2, 4, 8):
        rows = 16 * 1024 * tiles_per_pair
        hp2 = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, B // rows, 16)
        st2 = torch.tensor([[0.0, float(rows)]] * (B // rows), dtype=torch.float64, device=dev)
        us = timed(lambda: L.pfa_probe_grad(C.byref(exp), B, 0, params.data_ptr(), C.byref(dims), C.byref(hp2), st2.data_ptr(),
                                            grads.data_ptr(), ws.data_ptr(), 0, None))
        print(f'tiles/pair={tiles_per_pair}  rows={rows:7d}  {us:8.1f} us')
    for abl in (0, 1, 2, 4, 6, 7):
        def run():
            rc = L.pfa_probe_grad(C.byref(exp), B, 1, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(),
                                  grads.data_ptr(), ws.data_ptr(), abl, None)
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f'ABL={abl:2d} {names[abl]:40s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        print(build())
    else:
        main()
