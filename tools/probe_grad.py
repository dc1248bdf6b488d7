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
           '-x', 'hip', os.path.join(src, 'ppo_update.hip'), os.path.join(src, 'common.cpp'), os.path.join(src, 'dist.cpp'), os.path.join(src, 'p2p.hip'), '-ldl', '-o', so]
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
    grads = torch.zeros(P + 16, device=dev)
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


def product():
    """Times the PRODUCT entry points on the bench shape: pfa_ppo_mlp_grad (gradient + reduce launches) and the whole
    pfa_ppo_mlp_train loop of one update (16 optimizer steps), kernel times from the library's own HIP-event brackets."""
    import torch
    from pufferlib_amd import _lib
    L = _lib.lib()
    N, T, DP, A, NMB = 4096, 128, 64, 8, 4
    B = N * T
    dev = 'cuda'
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
    m = torch.zeros(P, device=dev)
    v = torch.zeros(P, device=dev)
    grads = torch.zeros(P + 16, device=dev)
    losses = torch.zeros(8, dtype=torch.float64, device=dev)
    ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)), dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)

    def grad():
        _lib.check(L.pfa_ppo_mlp_grad(C.byref(exp), B, 1, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), B // NMB,
                                      grads.data_ptr(), ws.data_ptr(), None), 'grad')

    def train():
        _lib.check(L.pfa_ppo_mlp_train(C.byref(exp), B, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), grads.data_ptr(),
                                       m.data_ptr(), v.data_ptr(), 0, 2.5e-4, .9, .999, 1e-5, .5, 4, losses.data_ptr(), ws.data_ptr(), 0, None),
                   'train')

    def kernel_ms(name):
        n, ms = C.c_int64(0), C.c_double(0.0)
        L.pfa_timing_read(name.encode(), C.byref(n), C.byref(ms))
        return n.value, ms.value
    for fn, label, per in ((grad, 'pfa_ppo_mlp_grad', 1), (train, 'pfa_ppo_mlp_train (16 steps)', 16)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / reps * 1e3
        L.pfa_timing_reset()
        L.pfa_timing_enable(2)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        L.pfa_timing_enable(0)
        parts = {k: kernel_ms(k) for k in ('ppo_mlp_grad', 'ppo_reduce', 'adam_clip')}
        print(f'{label}: {wall:8.1f} us per call ({wall / per:6.1f} us per optimizer step)   '
              + '  '.join(f'{k} {ms / max(n, 1) * 1e3:6.1f} us x{n // reps}' for k, (n, ms) in parts.items()))
    print('params finite:', bool(torch.isfinite(params).all()), ' losses', losses.cpu().numpy()[:6])


def trace():
    """Timeline of workgroup 0 over its tiles (s_memtime stamps, probe build only)."""
    import torch
    from pufferlib_amd import _lib
    so = os.path.join(OUT, 'libprobe.so')
    L = C.CDLL(so)
    N, T, DP, A, NMB = 4096, 128, 64, 8, 4
    B = N * T
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(B, DP, device=dev, generator=g)
    obs[:, 49:] = 0
    bufs = (torch.randint(0, A, (B,), device=dev, dtype=torch.int32, generator=g),
            torch.full((B,), -2.0794, device=dev), torch.randn(B, device=dev, generator=g),
            torch.randn(B, device=dev, generator=g), torch.zeros(B, device=dev),
            torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g))
    exp = _lib.Experience(obs.data_ptr(), *(t.data_ptr() for t in bufs), T)
    dims = _lib.MlpDims(49, DP, 128, A)
    hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, NMB, 16)
    P = 128 * DP + 128 + A * 128 + A + 128 + 1
    params = torch.randn(P, device=dev, generator=g) * 0.05
    grads = torch.zeros(P + 16, device=dev)
    ws = torch.zeros(16 << 20, dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
    L.pfa_probe_grad.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_void_p]
    L.pfa_probe_set_trace.argtypes = [C.c_void_p, C.c_int]
    J = 8
    tr = torch.zeros(8, J, 8, dtype=torch.int64, device=dev)

    def run():
        assert L.pfa_probe_grad(C.byref(exp), B, 1, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(),
                                grads.data_ptr(), ws.data_ptr(), 0, None) == 0
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    assert L.pfa_probe_set_trace(tr.data_ptr(), J) == 0
    run()
    torch.cuda.synchronize()
    L.pfa_probe_set_trace(None, 0)
    t = tr.cpu().numpy()
    t0 = t[t > 0].min()
    print('producer stamps: 0 top, 1 staged, 2 fwd+heads done, 3 loss done (at alpha), 4 past alpha, 5 past beta   [ticks from kernel start]')
    for w in (0, 1):
        for j in range(J):
            r = [int(x - t0) if x > 0 else -1 for x in t[w, j, :6]]
            print(f'  P wave {w} tile {j}: {r}   stage {r[1]-r[0]} fwd {r[2]-r[1]} loss {r[3]-r[2]} wait_alpha {r[4]-r[3]} publish+beta {r[5]-r[4]}'
                  + (f' dW2+loop {int(t[w, j + 1, 0] - t0) - r[5]}' if j + 1 < J else ''))
    print('consumer stamps: 0 top, 1 bwd(j-1) done, 2 past alpha, 3 past beta')
    for w in (4, 5):
        for j in range(J):
            r = [int(x - t0) if x > 0 else -1 for x in t[w, j, :4]]
            print(f'  C wave {w} tile {j}: {r}   bwd {r[1]-r[0]} wait_alpha {r[2]-r[1]} wait_beta {r[3]-r[2]}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        print(build())
    elif len(sys.argv) > 1 and sys.argv[1] == 'trace':
        trace()
    elif len(sys.argv) > 1 and sys.argv[1] == 'product':
        product()

    else:
        main()
