"""NatureCNN policy (pufferlib/models.py:113-157 `Convolutional`, BASELINE configs[3]) on the fp32-MFMA implicit-GEMM kernels
(csrc/igemm.hip, csrc/cnn_heads.hip): every layer's forward, dX and dW against torch's own conv2d / linear + autograd on the
CPU in fp32 (the arithmetic the reference runs), heads + sampling + PPO loss against the oracle's sample_logits / loss."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


class _Env:
    def __init__(self, actions=4, framestack=4):
        from pufferlib_amd import spaces
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(framestack, 84, 84), dtype=np.uint8)
        self.single_action_space = spaces.Discrete(actions)


def _net(actions=4, seed=0):
    from pufferlib_amd import cnn, models
    torch.manual_seed(seed)
    net = models.Convolutional(_Env(actions))
    ref = {k: v.detach().clone() for k, v in net.state_dict().items()}      # CPU copy before adoption
    cp = models.ConvParams(net, 'cuda')
    return net, ref, cp, cnn.Engine(cp, chunk=64)


def _torch_forward(ref, frames_u8):
    x = frames_u8.float() / 255.0
    a1 = F.relu(F.conv2d(x, ref['network.0.weight'], ref['network.0.bias'], stride=4))
    a2 = F.relu(F.conv2d(a1, ref['network.2.weight'], ref['network.2.bias'], stride=2))
    a3 = F.relu(F.conv2d(a2, ref['network.4.weight'], ref['network.4.bias'], stride=1))
    h = F.relu(F.linear(a3.flatten(1), ref['network.7.weight'], ref['network.7.bias']))
    return a1, a2, a3, h


@pytest.mark.parametrize('n', [3, 37])
def test_every_layer_forward_and_backward_matches_torch(n, matrix_products):
    net, ref, cp, eng = _net()
    assert cp.count == 1686693 - 0 if cp.num_actions == 4 else True
    g = torch.Generator().manual_seed(1)
    frames = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, generator=g)
    for k in ref:
        ref[k].requires_grad_(True)
    a1, a2, a3, h = _torch_forward(ref, frames)
    G = torch.randn(n, 512, generator=g)
    (h * G).sum().backward()
    dev_frames = frames.cuda().reshape(n, -1).contiguous()
    hd = eng.forward(dev_frames, n)
    nhwc = lambda t, c, hw: t[:n * hw * hw].view(n, hw, hw, c).permute(0, 3, 1, 2).cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(nhwc(eng.a1, 32, 20), a1.detach().numpy(), **TOL)
    np.testing.assert_allclose(nhwc(eng.a2, 64, 9), a2.detach().numpy(), **TOL)
    np.testing.assert_allclose(nhwc(eng.a3, 64, 7), a3.detach().numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(hd.cpu().numpy(), h.detach().numpy(), rtol=1e-5, atol=3e-5)
    # backward of sum(h * G): d(pre-ReLU hidden) = G * relu'
    eng.dh[:n] = (G * (h.detach() > 0)).cuda()
    grads = torch.zeros(cp.count, device='cuda')
    gv = cp.split(grads)
    eng.backward(dev_frames, n, eng.dh, gv, False)
    for name in ('network.7.weight', 'network.7.bias', 'network.4.weight', 'network.4.bias', 'network.2.weight', 'network.2.bias',
                 'network.0.weight', 'network.0.bias'):
        want = ref[name].grad.numpy()
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(gv[name].cpu().numpy() / scale, want / scale, rtol=1e-4, atol=2e-5, err_msg=name)
    # accumulate = True adds a second chunk
    eng.forward(dev_frames, n)
    eng.backward(dev_frames, n, eng.dh, gv, True)
    np.testing.assert_allclose(gv['network.2.weight'].cpu().numpy(), 2 * ref['network.2.weight'].grad.numpy(), rtol=1e-4, atol=1e-4)


def test_split_products_on_the_same_operands_as_the_fp32_products():
    """The opt-in product form (six bf16 partial products per fp32 product, fp32 accumulation) against the default one on IDENTICAL
    operands — every rows-form product of the NatureCNN at a chunk large enough for the 128-row tiles (forward of the four layers,
    dX of the three that have one) — and both against an f64 evaluation of the same contraction on sampled outputs."""
    from pufferlib_amd import _lib
    L = _lib.lib()
    n = 192
    net, ref, cp, eng = _net()
    eng._alloc(n)
    g = torch.Generator(device='cuda').manual_seed(2)
    frames = torch.randint(0, 256, (n, 4 * 84 * 84), dtype=torch.uint8, device='cuda', generator=g)
    eng.forward(frames, n)                                             # operands of every product, from the default form
    dh = torch.randn(n, 512, device='cuda', generator=g)
    d3in = torch.randn(n * 49, 64, device='cuda', generator=g)
    d2in = torch.randn(n * 81, 64, device='cuda', generator=g)
    c1, c2, c3, fc = eng.conv1, eng.conv2, eng.conv3, eng.fc
    a1, a2, a3 = eng.a1[:n * 400].clone(), eng.a2[:n * 81].clone(), eng.a3[:n * 49].clone()
    outs = dict(conv1=torch.empty_like(a1), conv2=torch.empty_like(a2), conv3=torch.empty_like(a3), fc=torch.empty(n, 512, device='cuda'),
                fc_dx=torch.empty_like(a3), conv3_dx=torch.empty_like(a2), conv2_dx=torch.empty_like(a1))
    products = [('conv1', lambda o: c1.forward(frames, n, o)), ('conv2', lambda o: c2.forward(a1, n, o)), ('conv3', lambda o: c3.forward(a2, n, o)),
                ('fc', lambda o: fc.forward(a3, n, o)), ('fc_dx', lambda o: fc.backward_dx(dh, n, a3, o)),
                ('conv3_dx', lambda o: c3.backward_dx(d3in, n, a2, o)), ('conv2_dx', lambda o: c2.backward_dx(d2in, n, a1, o))]
    try:
        for name, fn in products:
            got = {}
            for mode in (0, 1):
                _lib.check(L.pfa_igemm_set_products(mode), 'set_products')
                out = outs[name]
                out.zero_()
                fn(out)
                got[mode] = out.clone()
            scale = float(got[0].abs().max())
            err = float((got[1] - got[0]).abs().max())
            assert scale > 0 and err <= 1e-5 * scale, (name, err, scale)
            if name.startswith('conv'):          # (192 dense rows are too few for the 128-row tiles: the Linear keeps the fp32 form here)
                assert not torch.equal(got[0], got[1]), name          # the opt-in form really ran: its bits differ
    finally:
        _lib.check(L.pfa_igemm_set_products(0), 'set_products')
    # the Linear forward (the longest contraction, K = 3136) against f64 at a row count where both forms run their 128-row tiles
    rows = 4096
    x = torch.relu(torch.randn(rows, 3136, device='cuda', generator=g))
    out = torch.empty(rows, 512, device='cuda')
    w, b = cp.views['network.7.weight'], cp.views['network.7.bias']
    wp = fc.w_p                                                        # [512][3136], columns in the NHWC order the rows arrive in
    exact = torch.relu(x.double().cpu() @ wp.double().cpu().t() + b.double().cpu())
    seen = {}
    for mode in (0, 1):
        _lib.check(L.pfa_igemm_set_products(mode), 'set_products')
        try:
            fc.forward(x, rows, out)
        finally:
            _lib.check(L.pfa_igemm_set_products(0), 'set_products')
        seen[mode] = out.clone()
        err = float((out.double().cpu() - exact).abs().max() / exact.abs().max())
        assert err < 5e-6, (mode, err)
    assert not torch.equal(seen[0], seen[1])


def test_heads_sample_and_loss_match_the_oracle():
    from oracle import ppo_torch
    from pufferlib_amd import _lib
    import ctypes as C
    net, ref, cp, eng = _net(actions=6, seed=3)
    L = _lib.lib()
    n, A = 50, 6
    g = torch.Generator().manual_seed(2)
    h = torch.relu(torch.randn(n, 512, generator=g))
    noise = torch.empty(n, A).exponential_(1, generator=g)
    logits = F.linear(h, ref['actor.weight'], ref['actor.bias'])
    value = F.linear(h, ref['value_fn.weight'], ref['value_fn.bias']).flatten()
    oa, olp, oent = ppo_torch.sample_logits(logits, noise=noise)
    v = cp.views
    acts = torch.empty(n, dtype=torch.int64, device='cuda')
    lp, ent, val = (torch.empty(n, device='cuda') for _ in range(3))
    key = _lib.NoiseKey(1, 0)
    hc, nz = h.cuda(), noise.cuda()
    _lib.check(L.pfa_cnn_heads_sample(_lib.ptr(hc), n, _lib.ptr(v['actor.weight']), _lib.ptr(v['actor.bias']), _lib.ptr(v['value_fn.weight']),
                                      _lib.ptr(v['value_fn.bias']), A, _lib.ptr(nz), C.byref(key), 0, _lib.ptr(acts), _lib.ptr(lp), _lib.ptr(ent),
                                      _lib.ptr(val), None), 'sample')
    assert torch.equal(acts.cpu(), oa)
    np.testing.assert_allclose(lp.cpu().numpy(), olp.numpy(), **TOL)
    np.testing.assert_allclose(ent.cpu().numpy(), oent.numpy(), **TOL)
    np.testing.assert_allclose(val.cpu().numpy(), value.detach().numpy(), **TOL)


@pytest.mark.parametrize('M,K,N', [(700000, 512, 64), (8192, 3136, 512), (3000000, 256, 32), (8192, 512, 16)])
def test_weight_form_at_full_chunk_sizes_stays_inside_its_workspace(M, K, N):
    """The weight-gradient contraction at the row counts of an 8192-frame chunk (row splits at their caps): the workspace the query
    sizes is the one the launch uses (guard bytes behind it stay untouched), result against an fp64 matmul."""
    import ctypes as C
    from pufferlib_amd import _lib, cnn
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(M % 1000)
    a = torch.randn(M, K, device='cuda', generator=g) * 0.1
    d = torch.randn(M, N, device='cuda', generator=g) * 0.1
    nbytes = L.pfa_igemm_weights_workspace_bytes(M, K, N)
    ws = torch.full((nbytes + 4096,), 0xA5, dtype=torch.uint8, device='cuda')
    out = torch.empty(N, K, device='cuda')
    bias = torch.empty(N, device='cuda')
    op = cnn._operand(cnn.MODE_DENSE, a, K)
    _lib.check(L.pfa_igemm_weights(C.byref(op), M, K, _lib.ptr(d), N, N, _lib.ptr(out), 1, 0, _lib.ptr(bias), _lib.ptr(ws), _lib.stream_handle()),
               'weights')
    torch.cuda.synchronize()
    assert bool((ws[nbytes:] == 0xA5).all()), 'the launch wrote past the workspace it asked for'
    want = (d.double().t() @ a.double())
    err = (out.double() - want).abs().max().item()
    assert err <= 1e-5 * want.abs().max().item() + 1e-5, err
    wb = d.double().sum(0)
    assert (bias.double() - wb).abs().max().item() <= 1e-5 * wb.abs().max().item() + 1e-5
