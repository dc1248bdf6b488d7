"""Developer tool (GPU box): the fused MLP gradient step in both product forms on the bench shape — pfa_igemm_set_products(0): exact fp32
MFMA chains (csrc/ppo_update.hip), (1): six bf16 partial products per fp32 product (csrc/ppo_bf16.hpp).  Prints the largest
difference between the two flat gradients (relative to the largest entry, per parameter tensor), the loss sums of both, and the
time per launch of both from the library's own event brackets and from events around whole pfa_ppo_mlp_train calls.

    python tools/bf16_grad_check.py [num_actions]
"""
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
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N, T, DP, NMB = 4096, 128, 64, 4
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
    params0 = torch.randn(P, device=dev, generator=g) * 0.05
    params0[:128 * DP].view(128, DP)[:, 49:] = 0
    ws = torch.zeros(L.pfa_ppo_workspace_bytes(C.byref(dims), B, C.byref(hp)) + (1 << 20), dtype=torch.uint8, device=dev)
    stats = torch.tensor([[0.0, float(B // NMB)]] * NMB, dtype=torch.float64, device=dev)
    out = {}
    grads = {}
    for mode in (0, 1):
        _lib.check(L.pfa_igemm_set_products(mode), 'set_products')
        gr = torch.zeros(P + 16, device=dev)
        _lib.check(L.pfa_ppo_mlp_grad(C.byref(exp), B, 1, params0.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), B // NMB,
                                      gr.data_ptr(), ws.data_ptr(), 0), 'grad')
        torch.cuda.synchronize()
        grads[mode] = gr.clone()
        # timing through the product's train loop (16 optimizer steps per call)
        params, m, v = params0.clone(), torch.zeros(P, device=dev), torch.zeros(P, device=dev)
        losses = torch.zeros(8, dtype=torch.float64, device=dev)
        step = [0]

        def train():
            rc = L.pfa_ppo_mlp_train(C.byref(exp), B, params.data_ptr(), C.byref(dims), C.byref(hp), stats.data_ptr(), gr.data_ptr(),
                                     m.data_ptr(), v.data_ptr(), step[0], 2.5e-4, .9, .999, 1e-5, .5, 4, losses.data_ptr(), ws.data_ptr(), 0, None)
            assert rc == 0, (rc, L.pfa_last_error())
            step[0] += 16
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
        for k in ('ppo_mlp_grad', 'ppo_reduce_adam'):
            n, ms = C.c_int64(0), C.c_double(0.0)
            L.pfa_timing_read(k.encode(), C.byref(n), C.byref(ms))
            if n.value:
                parts[k] = round(ms.value / n.value * 1e3, 2)
        out['fp32' if mode == 0 else 'bf16x6'] = dict(us_per_opt_step=round(wall, 2), kernels_us=parts, finite=bool(torch.isfinite(params).all()),
                                                     params_after=params.clone())
    _lib.check(L.pfa_igemm_set_products(0), 'set_products')
    g0, g1 = grads[0], grads[1]
    names = [('w1', 0, 128 * DP), ('b1', 128 * DP, 128 * DP + 128), ('w2', 128 * DP + 128, 128 * DP + 128 + A * 128),
             ('b2', 128 * DP + 128 + A * 128, 128 * DP + 128 + A * 128 + A), ('wv', 128 * DP + 128 + A * 128 + A, P - 1), ('bv', P - 1, P), ('loss sums', P, P + 16)]
    diffs = {}
    for nme, lo, hi in names:
        a, b = g0[lo:hi], g1[lo:hi]
        diffs[nme] = dict(max_abs=float(a.abs().max()), max_abs_diff=float((a - b).abs().max()), rel=float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)))
    pd = float((out['fp32']['params_after'] - out['bf16x6']['params_after']).abs().max())
    for k in out:
        del out[k]['params_after']
    res = dict(num_actions=A, grad_diff=diffs, max_abs_param_diff_after_updates=pd, timing=out)
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, 'gpurun_out', f'bf16_grad_check_a{A}.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
