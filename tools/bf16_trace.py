"""Developer tool: timeline of the bf16-path gradient kernel (csrc/ppo_bf16.hpp) from s_memtime stamps of workgroups 0 and 256
(lane 0 of every wave) in a probe build.
    python tools/bf16_trace.py build        (here: builds tools/_probe/libt_bf16.so with -DPFA_BF16_TRACE)
    python tools/bf16_trace.py run          (GPU box)
Stamps per tile: 0 top, 1 staged X, 2 after barrier A, 3 forward done, 4 heads + patch written, 5 after barrier B, 6 loss done,
7 after barrier C, 8 dW2v + dh done, 9 dh pieces written, 10 dW1 done."""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'tools', '_probe')
SO = os.path.join(OUT, 'libt_bf16.so')


def build():
    from pufferlib_amd import _lib
    _lib.build()
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, 'ppo_update_trace.o')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', '-DPFA_BF16_TRACE',
                           os.path.join(_lib.CSRC, 'ppo_update.hip'), '-o', obj])
    objs = [os.path.join(_lib.LIB_DIR, os.path.splitext(s)[0] + '.o') for s in _lib.SOURCES if s != 'ppo_update.hip'] + [obj]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', SO] + objs + ['-ldl'])
    os.remove(obj)
    print(SO)


def run():
    import torch
    from pufferlib_amd import _lib
    L = C.CDLL(SO)
    for fn, (restype, argtypes) in _lib._SIGNATURES.items():
        if hasattr(L, fn):
            getattr(L, fn).restype, getattr(L, fn).argtypes = restype, argtypes
    A, N, T, DP, NMB = 8, 4096, 128, 64, 4
    B = N * T
    dev = 'cuda'
    g = torch.Generator(device=dev).manual_seed(0)
    obs = torch.randn(B, DP, device=dev, generator=g)
    obs[:, 49:] = 0
    bufs = (torch.randint(0, A, (B,), device=dev, dtype=torch.int32, generator=g), torch.full((B,), -2.0794, device=dev),
            torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g), torch.zeros(B, device=dev),
            torch.randn(B, device=dev, generator=g), torch.randn(B, device=dev, generator=g))
    exp = _lib.Experience(obs.data_ptr(), *(t.data_ptr() for t in bufs), T)
    dims = _lib.MlpDims(49, DP, 128, A, 0)
    hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, NMB, 16)
    P = 128 * DP + 128 + A * 128 + A + 128 + 1
    params = torch.randn(P, device=dev, generator=g) * 0.05
    ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)) + (1 << 20), dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
    gr = torch.zeros(P + 16, device=dev)
    assert L.pfa_igemm_set_products(1) == 0
    TILES = 8
    trace = torch.zeros(2 * 4 * TILES * 16, dtype=torch.int64, device=dev)
    for it in range(3):
        trace.zero_()
        L.pfa_probe_bf16_trace.argtypes = [C.c_void_p, C.c_int]
        assert L.pfa_probe_bf16_trace(trace.data_ptr(), TILES) == 0
        rc = L.pfa_ppo_mlp_grad(C.byref(exp), B, 1, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), B // NMB, gr.data_ptr(), ws.data_ptr(), 0)
        assert rc == 0, L.pfa_last_error()
        torch.cuda.synchronize()
    t = trace.cpu().numpy().reshape(2, 4, TILES, 16)
    t0 = t[t > 0].min()
    names = ['top', 'staged', 'barA', 'fwd', 'heads', 'barB', 'loss', 'barC', 'dw2+dh', 'dhsplit', 'dw1']
    print('s_memtime ticks (100 MHz: 1 tick = 10 ns = ~24 shader cycles); rows = tiles, columns = stamps relative to the launch\'s first stamp')
    for wg in range(2):
        for wv in range(4):
            print(f'--- workgroup {"0" if wg == 0 else "256"} wave {wv}')
            for j in range(TILES):
                row = t[wg, wv, j, :11]
                print(f'  tile {j}: ' + ' '.join(f'{names[k]}={int(row[k] - t0) if row[k] else -1:5d}' for k in range(11)))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    import numpy as np
    np.save(os.path.join(REPO, 'gpurun_out', 'bf16_trace.npy'), t - t0)


if __name__ == '__main__':
    build() if sys.argv[1] == 'build' else run()
