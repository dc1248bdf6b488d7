"""Developer tool (GPU box): launches the fused MLP gradient step N times on the bench shape in one product form, for rocprofv3 runs
(kernel trace or --pmc passes) that should see nothing else.
    python tools/grad_loop.py <products 0|1> [launches] [num_actions]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from pufferlib_amd import _lib
    L = _lib.lib()
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    A = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    N, T, DP, NMB = 4096, 128, 64, 4
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
    params[:128 * DP].view(128, DP)[:, 49:] = 0
    ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)) + (1 << 20), dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
    gr = torch.zeros(P + 16, device=dev)
    _lib.check(L.pfa_igemm_set_products(mode), 'set_products')
    for i in range(n):
        _lib.check(L.pfa_ppo_mlp_grad(C.byref(exp), B, i % NMB, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), B // NMB,
                                      gr.data_ptr(), ws.data_ptr(), 0), 'grad')
    torch.cuda.synchronize()
    _lib.check(L.pfa_igemm_set_products(0), 'set_products')
    print('ok', float(gr.abs().sum()))


if __name__ == '__main__':
    main()
