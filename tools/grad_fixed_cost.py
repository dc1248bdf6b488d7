"""Developer tool (GPU box): launch time of the fused MLP gradient step against the number of tiles a workgroup runs, both product
forms — the intercept is the launch's fixed cost (prologue: weight fragments / tables; epilogue: the partial), the slope the time per
round of tiles.  Minibatches of 1/32 .. 1/4 of the 524 288-row bench batch.
    python tools/grad_fixed_cost.py"""
import ctypes as C
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from pufferlib_amd import _lib
    L = _lib.lib()
    A, N, T, DP = 8, 4096, 128, 64
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
    P = 128 * DP + 128 + A * 128 + A + 128 + 1
    params = torch.randn(P, device=dev, generator=g) * 0.05
    params[:128 * DP].view(128, DP)[:, 49:] = 0
    gr = torch.zeros(P + 16, device=dev)
    out = {}
    for mode in (0, 1):
        _lib.check(L.pfa_igemm_set_products(mode), 'set_products')
        res = {}
        for nmb in (32, 16, 8, 4):
            hp = _lib.PpoHparams(.1, .1, .5, .01, 1, 1, nmb, 16)
            ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)) + (1 << 20), dtype=torch.uint8, device=dev)
            stats = torch.tensor([[0.0, float(B // nmb)]] * nmb, dtype=torch.float64, device=dev)

            def launch(i):
                _lib.check(L.pfa_ppo_mlp_grad(C.byref(exp), B, i % nmb, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), B // nmb,
                                              gr.data_ptr(), ws.data_ptr(), 0), 'grad')
            for i in range(8):
                launch(i)
            torch.cuda.synchronize()
            L.pfa_timing_reset()
            L.pfa_timing_select(b'ppo_mlp_grad')
            L.pfa_timing_enable(1)
            for i in range(40):
                launch(i)
            torch.cuda.synchronize()
            L.pfa_timing_enable(0)
            n, ms = C.c_int64(0), C.c_double(0.0)
            L.pfa_timing_read(b'ppo_mlp_grad', C.byref(n), C.byref(ms))
            res[B // nmb] = round(ms.value / max(n.value, 1) * 1e3, 2)
        out['fp32' if mode == 0 else 'bf16x6'] = res
    _lib.check(L.pfa_igemm_set_products(0), 'set_products')
    for k, r in out.items():
        rows = sorted(r)
        slope = (r[rows[-1]] - r[rows[0]]) / (rows[-1] - rows[0])
        print(k, r, 'us; per 16 384 rows', round(slope * 16384, 2), 'us; intercept', round(r[rows[0]] - slope * rows[0], 2), 'us')
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'grad_fixed_cost.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
