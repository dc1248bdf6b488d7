"""A/B timing of compile-time variants of the MLP update kernels (developer tool, not product code).

    python tools/variant_bench.py build  name1:-DFLAG=1,-DOTHER=2  name2:...     (here, no GPU: hipcc cross-compiles)
    python tools/variant_bench.py run                                            (on the GPU box)

`build` compiles csrc/ppo_update.hip once per variant with the given -D flags and links it with the product's other objects
into tools/_probe/lib_<name>.so (travels with the gpurun snapshot).  `run` drives every variant through the PRODUCT entry point
pfa_ppo_mlp_train on the bench shape (131 072-row minibatches, 16 optimizer steps per call), reports microseconds per optimizer
step (wall, HIP events around the whole call) and per kernel (the library's own event brackets), and compares the parameters
after 2 updates with the first variant's (max abs difference), so a faster variant that computes something else shows up.
"""
import ctypes as C
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tools', '_probe')


def build(specs):
    from pufferlib_amd import _lib
    _lib.build()
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, 'lib_*.so')):
        os.remove(f)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(':')
        flags = [f for f in flags.split(',') if f]
        obj = os.path.join(OUT, f'ppo_update_{name}.o')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', *flags,
               os.path.join(_lib.CSRC, 'ppo_update.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [os.path.join(_lib.LIB_DIR, os.path.splitext(s)[0] + '.o') for s in _lib.SOURCES if s != 'ppo_update.hip'] + [obj]
        so = os.path.join(OUT, f'lib_{name}.so')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + objs + ['-ldl'])
        os.remove(obj)
        print(so)


def run():
    import torch
    from pufferlib_amd import _lib
    N, T, DP, A, NMB = 4096, 128, 64, 8, 4
    B = N * T
    dev = 'cuda'
    res = {}
    ref = None
    for so in sorted(glob.glob(os.path.join(OUT, 'lib_*.so'))):
        name = os.path.basename(so)[4:-3]
        L = C.CDLL(so)
        for fn, (restype, argtypes) in _lib._SIGNATURES.items():
            if hasattr(L, fn):
                getattr(L, fn).restype, getattr(L, fn).argtypes = restype, argtypes
        g = torch.Generator(device=dev).manual_seed(0)
        obs = torch.randn(B, DP, device=dev, generator=g)
        obs[:, 49:] = 0
        bufs = (torch.randint(0, A, (B,), device=dev, dtype=torch.int32, generator=g),
                torch.full((B,), -2.0794, device=dev), torch.randn(B, device=dev, generator=g),
                torch.randn(B, device=dev, generator=g), torch.zeros(B, device=dev),
                torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g))
        exp = _lib.Experience(obs.data_ptr(), *(t.data_ptr() for t in bufs), T)
        dims = _lib.MlpDims(49, DP, 128, A, 0)
        hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, NMB, 16)
        P = 128 * DP + 128 + A * 128 + A + 128 + 1
        params = torch.randn(P, device=dev, generator=g) * 0.05
        params[:128 * DP].view(128, DP)[:, 49:] = 0
        m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
        grads = torch.zeros(P + 16, device=dev)
        losses = torch.zeros(8, dtype=torch.float64, device=dev)
        ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)) + (1 << 20), dtype=torch.uint8, device=dev)
        stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
        step = [0]

        def train():
            rc = L.pfa_ppo_mlp_train(C.byref(exp), B, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), grads.data_ptr(),
                                     m.data_ptr(), v.data_ptr(), step[0], 2.5e-4, .9, .999, 1e-5, .5, 4, losses.data_ptr(), ws.data_ptr(), 0, None)
            assert rc == 0, (name, rc, L.pfa_last_error())
            step[0] += 16
        train()
        train()
        torch.cuda.synchronize()
        snap = params.clone()
        if ref is None:
            ref = snap
        diff = float((snap - ref).abs().max())
        for _ in range(3):
            train()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            train()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / reps / 16 * 1e3
        L.pfa_timing_reset()
        L.pfa_timing_enable(2)
        for _ in range(5):
            train()
        torch.cuda.synchronize()
        L.pfa_timing_enable(0)
        parts = {}
        for k in ('ppo_mlp_grad', 'ppo_reduce', 'adam_clip', 'ppo_reduce_adam', 'ppo_pack'):
            n, ms = C.c_int64(0), C.c_double(0.0)
            L.pfa_timing_read(k.encode(), C.byref(n), C.byref(ms))
            if n.value:
                parts[k] = round(ms.value / n.value * 1e3, 2)
        res[name] = dict(us_per_opt_step=round(wall, 2), kernels_us=parts, max_abs_param_diff_vs_first=diff,
                         finite=bool(torch.isfinite(params).all()))
        print(name, res[name], flush=True)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, 'gpurun_out', 'variant_bench.json'), 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run()
