"""A/B timing of compile-time variants of the other-width update kernel csrc/ppo_wide.hip (developer tool, not product code).

    python tools/wide_variant_bench.py build  name1:-DFLAG=1,-DOTHER=2  name2:...     (here, no GPU: hipcc cross-compiles)
    python tools/wide_variant_bench.py run [hidden ...]                              (on the GPU box; default 64 256 512)

`build` compiles ppo_wide.hip once per variant with the given -D flags and links it with the product's other objects into
tools/_probe/libw_<name>.so (travels with the gpurun snapshot).  `run` drives every variant through the PRODUCT entry point
pfa_ppo_wide_grad on the bench shape (7x7 grid rows, 131 072-row minibatches, Default(hidden)), reports microseconds per launch from
the library's own event brackets (gradient kernel and the fixed-order sum of its partials) and from events around 16 back-to-back calls,
and compares the gradient with the first variant's (max abs difference relative to the largest entry), so a faster variant that
computes something else shows up.
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
    for f in glob.glob(os.path.join(OUT, 'libw_*.so')):
        os.remove(f)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(':')
        flags = [f for f in flags.split(',') if f]
        obj = os.path.join(OUT, f'ppo_wide_{name}.o')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', *flags,
               '-I' + os.path.join(REPO, 'include'), os.path.join(_lib.CSRC, 'ppo_wide.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        objs = [os.path.join(_lib.LIB_DIR, os.path.splitext(s)[0] + '.o') for s in _lib.SOURCES if s != 'ppo_wide.hip'] + [obj]
        so = os.path.join(OUT, f'libw_{name}.so')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + objs + ['-ldl'])
        os.remove(obj)
        print(so)


def run(hiddens):
    import torch
    from pufferlib_amd import _lib
    N, T, DP, D, A, NMB = 4096, 128, 64, 49, 8, 4
    B = N * T
    dev = 'cuda'
    res = {}
    for H in hiddens:
        ref = None
        for so in sorted(glob.glob(os.path.join(OUT, 'libw_*.so'))):
            name = os.path.basename(so)[5:-3]
            L = C.CDLL(so)
            for fn, (restype, argtypes) in _lib._SIGNATURES.items():
                if hasattr(L, fn):
                    getattr(L, fn).restype, getattr(L, fn).argtypes = restype, argtypes
            g = torch.Generator(device=dev).manual_seed(0)
            obs = torch.randn(B, DP, device=dev, generator=g)
            obs[:, D:] = 0
            bufs = (torch.randint(0, A, (B,), device=dev, dtype=torch.int32, generator=g),
                    torch.full((B,), -2.0794, device=dev), torch.randn(B, device=dev, generator=g),
                    torch.randn(B, device=dev, generator=g), torch.zeros(B, device=dev),
                    torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g))
            exp = _lib.Experience(obs.data_ptr(), *(t.data_ptr() for t in bufs), T)
            hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, NMB, 16)
            shapes = [(H, D), (H,), (A, H), (A,), (1, H), (1,)]        # named_parameters() order of models.Default
            count = sum(int(torch.tensor(s).prod()) for s in shapes)
            flat = torch.randn(count, device=dev, generator=g) * 0.05
            grads = torch.zeros(count + 16, device=dev)

            def view(base):
                ptrs, o = [], 0
                for s in shapes:
                    ptrs.append(base.data_ptr() + 4 * o)
                    o += int(torch.tensor(s).prod())
                return _lib.MlpView(ptrs[0], D, D, DP, H, A, 0, ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5])
            pv, gv = view(flat), view(grads)
            assert L.pfa_ppo_wide_supported(C.byref(pv)), (name, H)
            ws = torch.zeros(int(L.pfa_ppo_wide_workspace_bytes(C.byref(pv))) + (1 << 20), dtype=torch.uint8, device=dev)
            stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)

            def step(mb):
                rc = L.pfa_ppo_wide_grad(C.byref(exp), B, mb, C.byref(pv), C.byref(gv), grads.data_ptr() + 4 * count, C.byref(hp),
                                         stats.data_ptr(), B // NMB, ws.data_ptr(), 0)
                assert rc == 0, (name, rc, L.pfa_last_error())
            step(1)
            torch.cuda.synchronize()
            snap = grads.clone()
            if ref is None:
                ref = snap
            diff = float((snap - ref).abs().max() / ref.abs().max())
            for _ in range(8):
                step(0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 64
            e0.record()
            for i in range(reps):
                step(i % NMB)
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / reps * 1e3
            L.pfa_timing_reset()
            L.pfa_timing_enable(2)
            for i in range(32):
                step(i % NMB)
            torch.cuda.synchronize()
            L.pfa_timing_enable(0)
            parts = {}
            for k in ('ppo_wide_grad', 'ppo_wide_reduce'):
                n, ms = C.c_int64(0), C.c_double(0.0)
                L.pfa_timing_read(k.encode(), C.byref(n), C.byref(ms))
                if n.value:
                    parts[k] = round(ms.value / n.value * 1e3, 2)
            flop = 310.0 * H * (B // NMB)
            res[f'{name}/h{H}'] = dict(us_per_call_wall=round(wall, 2), kernels_us=parts, rel_grad_diff_vs_first=diff,
                                       finite=bool(torch.isfinite(grads).all()),
                                       frac_of_fp32_mfma_peak=round(flop / (parts.get('ppo_wide_grad', wall) * 1e-6) / 157.3e12, 3))
            print(f'{name}/h{H}', res[f'{name}/h{H}'], flush=True)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, 'gpurun_out', 'wide_variant_bench.json'), 'w'), indent=1)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run([int(x) for x in sys.argv[2:]] or [64, 256, 512])
