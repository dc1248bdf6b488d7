"""HIP GAE (pfa_gae_f32) vs the reference's c_gae outputs (golden) and the C oracle."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = RTOL = 1e-5   # SURVEY.md hard part 5: allclose(atol=1e-5, rtol=1e-5), never pure relative


def hip_gae(dones, values, rewards, gamma, lam, want_returns=False):
    import torch
    from pufferlib_amd import _lib
    L = _lib.lib()
    dev = 'cuda'
    d = torch.as_tensor(np.ascontiguousarray(dones, np.float32)).to(dev)
    v = torch.as_tensor(np.ascontiguousarray(values, np.float32)).to(dev)
    r = torch.as_tensor(np.ascontiguousarray(rewards, np.float32)).to(dev)
    n = d.numel()
    adv = torch.full((n,), float('nan'), device=dev)
    ret = torch.full((n,), float('nan'), device=dev)
    ws = torch.zeros(max(1, L.pfa_gae_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    _lib.check(L.pfa_gae_f32(_lib.ptr(d), _lib.ptr(v), _lib.ptr(r), _lib.ptr(adv), _lib.ptr(ret), n, gamma, lam,
                             _lib.ptr(ws), _lib.stream_handle()), 'gae')
    return (adv.cpu().numpy(), ret.cpu().numpy()) if want_returns else adv.cpu().numpy()


def test_golden_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'gae.npz'))
    for c in sorted({k.split('_')[0] for k in g.files}):
        gamma, lam = g[c + '_gl']
        adv, ret = hip_gae(g[c + '_dones'], g[c + '_values'], g[c + '_rewards'], gamma, lam, want_returns=True)
        np.testing.assert_allclose(adv, g[c + '_adv'], rtol=RTOL, atol=ATOL, err_msg=c)
        np.testing.assert_allclose(ret, g[c + '_adv'] + g[c + '_values'], rtol=RTOL, atol=ATOL, err_msg=c)
        assert adv[-1] == 0.0


@pytest.fixture(params=['self-starting window', 'f64-seeded window'])
def gae_form(request, monkeypatch):
    """Both single-array forms of csrc/gae.hip: one launch whose walkers start from 0 far enough behind their items (the default
    where gamma lambda lets the 1024-element window do it), and pass 1's f64 chunk maps seeding a shorter warm-up (PFA_GAE_SELF=0;
    what runs for gamma lambda > 0.984)."""
    monkeypatch.setenv('PFA_GAE_SELF', '1' if request.param.startswith('self') else '0')
    return request.param


@pytest.mark.parametrize('n,p_done', [(524288, 0.0), (524288, 0.01), (524288, 0.25), (2048, 0.1), (2049, 0.1),
                                       (65537, 0.5), (3, 0.0), (1023, 0.0), (1025, 0.02), (9, 0.0)])
def test_vs_oracle_full_size(n, p_done, gae_form):
    from oracle import c_oracle
    rng = np.random.RandomState(n % 1000 + int(p_done * 100))
    d = (rng.rand(n) < p_done).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    want = c_oracle.compute_gae(d, v, r, 0.99, 0.95)
    got = hip_gae(d, v, r, 0.99, 0.95)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    # tighter than the contract: a few ulps of the largest advantage
    assert np.abs(got - want).max() <= 8 * np.finfo(np.float32).eps * max(1.0, np.abs(want).max())
    # round 5: the parallel kernel lands on the reference's OWN fp32 sequence (warm-up contraction, csrc/gae.hip gae_exact_kernel):
    # every advantage is the bit pattern c_gae's sequential loop produces
    assert np.array_equal(got, want), (int((got != want).sum()), float(np.abs(got - want).max()))


@pytest.mark.parametrize('gamma,lam', [(0.99, 0.95), (0.9, 0.8), (0.997, 0.97), (0.995, 0.985), (1.0, 1.0), (0.5, 0.0)])
def test_bit_identical_to_the_sequential_loop_for_other_discounts(gamma, lam, gae_form):
    """gamma lambda up to ~0.985 (here 0.967): the warm-up (ln 1e-7 / ln(gamma lambda) elements, at most 1024) has contracted the
    start error away and the result is c_gae's bit pattern; at gamma = lambda = 1 nothing contracts and the kernel is what every scan
    is: within a few ulps."""
    from oracle import c_oracle
    n = 300000
    rng = np.random.RandomState(7)
    d = (rng.rand(n) < 0.002).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    want = c_oracle.compute_gae(d, v, r, gamma, lam)
    got = hip_gae(d, v, r, gamma, lam)
    if gamma * lam < 0.981:
        assert np.array_equal(got, want), (int((got != want).sum()), float(np.abs(got - want).max()))
    else:
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-4 * max(1.0, float(np.abs(want).max())))


def test_linearity_property():
    """GAE is linear in (rewards, values) for fixed dones: adv(a x + b y) = a adv(x) + b adv(y)."""
    rng = np.random.RandomState(3)
    n = 524288
    d = (rng.rand(n) < 0.02).astype(np.float32)
    v1, r1 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    v2, r2 = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    a1, a2 = hip_gae(d, v1, r1, .99, .95), hip_gae(d, v2, r2, .99, .95)
    a12 = hip_gae(d, 2 * v1 - v2, 2 * r1 - r2, .99, .95)
    np.testing.assert_allclose(a12, 2 * a1 - a2, rtol=1e-4, atol=1e-4)


def test_empty_is_noop():
    from pufferlib_amd import _lib
    L = _lib.lib()
    assert L.pfa_gae_f32(None, None, None, None, None, 0, .99, .95, None, None) == 0


@pytest.mark.parametrize('n,shards,p_done,gl', [(524288, 8, 0.01, (0.99, 0.95)), (524288, 8, 0.0, (0.99, 0.95)), (12288, 3, 0.2, (0.99, 0.95)),
                                                  (4096, 8, 0.0, (0.99, 0.95)), (4100, 2, 0.0, (0.99, 0.95)), (1600, 8, 0.02, (0.995, 0.97)),
                                                  (7, 7, 0.3, (0.99, 0.95)), (16, 8, 0.5, (0.9, 0.8)), (64 * 48, 4, 0.05, (0.99, 0.95)),
                                                  (262144, 8, 0.003, (0.995, 0.985)), (6000, 3, 0.0, (0.995, 0.985))])
def test_sharded_halo_form_is_the_flat_scan_bit_for_bit(n, shards, p_done, gl):
    """The form clean_pufferl runs data parallel (round 6): every shard publishes the bit patterns of its first min(m, H) rows
    (pfa_gae_halo_publish, next to extra sums that ride along), the host sums the zero-padded buffers (the all-reduce), every shard
    drops the rows that follow it behind its arrays (pfa_gae_halo_unpack: from ONE later shard when m >= H, from several when the
    shards are shorter than the halo — 4096 / 8 = 512 < 544, 7 / 7 = one row each) and runs the single-rank kernel over n + halo rows
    (pfa_gae_halo_f32).  Advantages and returns == the single flat scan, and == c_gae, as bit patterns; with sums: the per-minibatch
    advantage sums and explained-variance sums of all shards add up to the flat batch's."""
    import torch
    from oracle import c_oracle
    from pufferlib_amd import _lib
    L = _lib.lib()
    gamma, lam = gl
    rng = np.random.RandomState(n % 977 + shards)
    d = (rng.rand(n) < p_done).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    v[n // 3] = -0.0                                                  # a sign bit an arithmetic gather would lose
    r[(n // shards) % n] = -0.0                                       # ... in a published row
    want = np.asarray(c_oracle.compute_gae(d, v, r, gamma, lam), np.float32)
    flat = hip_gae(d, v, r, gamma, lam)
    m = n // shards
    H = int(L.pfa_gae_halo_rows(gamma, lam))
    assert H > 0 and H % 8 == 0 and H <= 2056
    hp = min(m, H)
    st = _lib.stream_handle()
    size = 3 + 3 * shards * hp
    bufs, total = [], torch.zeros(size, dtype=torch.float64, device='cuda')
    for q in range(shards):
        t = [torch.cat([torch.as_tensor(x[q * m:(q + 1) * m].copy()), torch.full((H,), float('nan'))]).cuda() for x in (d, v, r)]   # + halo room
        extra = torch.tensor([1.0, 2.0 * q, -0.5], dtype=torch.float64, device='cuda')
        out = torch.full((size,), float('nan'), dtype=torch.float64, device='cuda')
        _lib.check(L.pfa_gae_halo_publish(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), m, gamma, lam, _lib.ptr(extra), 3, _lib.ptr(out), q,
                                          shards, st), 'publish')
        o = out[3:].view(shards, 3 * hp)
        assert bool(torch.isfinite(out).all()) and all(float(o[k].abs().sum()) == 0.0 for k in range(shards) if k != q)
        total += out
        bufs.append(t)
    assert total[:3].tolist() == [float(shards), float(shards * (shards - 1)), -0.5 * shards]
    # sums: minibatches of 2 x bptt 8 where the shard allows it (its rows as m / 8 "envs" of 8 steps... any partition the kernel accepts)
    N_loc, nmb, bptt = (m // 16, 2, 16) if m % 32 == 0 else (0, 0, 0)
    with_sums = N_loc > 0 and L.pfa_gae_sums_supported(m, N_loc, nmb, bptt) == 1
    got = np.empty(n, np.float32)
    stats_sum, ev_sum = np.zeros((max(nmb, 1), 2)), np.zeros(4)
    for q, t in enumerate(bufs):
        halo_len = L.pfa_gae_halo_unpack(_lib.ptr(total[3:]), q, shards, m, gamma, lam, _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), st)
        assert halo_len == min(H, (shards - 1 - q) * m)
        for x, src in zip(t, (d, v, r)):                              # the rows that follow the shard in the flat batch, bit for bit
            assert np.array_equal(x[m:m + halo_len].cpu().numpy().view(np.uint32), src[(q + 1) * m:(q + 1) * m + halo_len].view(np.uint32))
        adv = torch.full((m,), float('nan'), device='cuda')
        ret = torch.full((m,), float('nan'), device='cuda')
        ws = torch.zeros(max(16, L.pfa_gae_sums_workspace_bytes(m, max(nmb, 1))), dtype=torch.uint8, device='cuda')
        stats = torch.full((max(nmb, 1), 2), float('nan'), dtype=torch.float64, device='cuda')
        ev4 = torch.full((4,), float('nan'), dtype=torch.float64, device='cuda')
        zero8 = torch.ones(8, dtype=torch.float64, device='cuda')
        _lib.check(L.pfa_gae_halo_f32(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(adv), _lib.ptr(ret), m, halo_len, gamma, lam,
                                      N_loc, nmb, bptt, _lib.ptr(stats) if with_sums else None, _lib.ptr(ev4), _lib.ptr(zero8), _lib.ptr(ws), st),
                   'halo gae')
        got[q * m:(q + 1) * m] = adv.cpu().numpy()
        assert np.array_equal(ret.cpu().numpy(), got[q * m:(q + 1) * m] + v[q * m:(q + 1) * m])
        if with_sums:
            assert float(zero8.abs().sum()) == 0.0
            a = got[q * m:(q + 1) * m].astype(np.float64)
            seg = a.reshape(m // bptt, bptt)
            for k in range(nmb):
                rows = seg[k::nmb].ravel()
                np.testing.assert_allclose(stats[k].cpu().numpy(), [rows.sum(), (rows * rows).sum()], rtol=1e-12, atol=1e-9)
            yp = v[q * m:(q + 1) * m].reshape(N_loc, m // N_loc).T.ravel().astype(np.float64)
            yt = a + yp
            np.testing.assert_allclose(ev4.cpu().numpy(), [yt.sum(), (yt * yt).sum(), a.sum(), (a * a).sum()], rtol=1e-12, atol=1e-8)
    assert np.array_equal(got.view(np.uint32), flat.view(np.uint32))          # the single-rank kernel's bits
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))          # ... which are c_gae's
    assert got[-1] == 0.0


def test_halo_form_is_refused_outside_the_self_starting_window():
    from pufferlib_amd import _lib
    L = _lib.lib()
    assert L.pfa_gae_halo_rows(0.99, 0.95) == 544 and L.pfa_gae_halo_rows(0.9, 0.8) % 8 == 0
    assert L.pfa_gae_halo_rows(0.995, 0.985) == 1640               # gamma lambda 0.98: the 2048-element window (1632 warm-up elements + 8)
    assert L.pfa_gae_halo_rows(0.999, 0.99) == 0                   # the f64-carry form (pfa_gae_shard_publish / fold / pass2) serves these
    assert L.pfa_gae_halo_publish(None, None, None, 16, 0.999, 0.99, None, 0, None, 0, 2, None) != 0
    assert b'self-starting' in L.pfa_last_error()


@pytest.mark.parametrize('n,shards,p_done', [(524288, 8, 0.01), (12288, 3, 0.2), (4100, 2, 0.0), (4098, 2, 0.05), (7, 7, 0.3), (16, 8, 0.5)])
def test_sharded_one_exchange_form_equals_flat_scan(n, shards, p_done):
    """The form clean_pufferl runs data parallel since round 4: every shard publishes six numbers from its OWN rows
    (pfa_gae_shard_publish, next to extra sums that ride along), the host sums the zero-padded buffers (the all-reduce), every
    shard folds the gathered numbers (pfa_gae_shard_fold: later shards' maps completed with their last elements, halo row, patched
    last block aggregate) and finishes with pfa_gae_shard_pass2.  4098 = a shard of 2049 rows: its last element sits alone in a
    block of the n-element pass; 7 / 7 and 16 / 8: shards of one and two rows."""
    import torch
    from oracle import c_oracle
    from pufferlib_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(n % 977 + shards)
    d = (rng.rand(n) < p_done).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    want = c_oracle.compute_gae(d, v, r, 0.99, 0.95)
    m = n // shards
    st = _lib.stream_handle()
    bufs, total = [], torch.zeros(3 + 6 * shards, dtype=torch.float64, device='cuda')
    for q in range(shards):
        t = [torch.cat([torch.as_tensor(x[q * m:(q + 1) * m].copy()), torch.full((1,), float('nan'))]).cuda() for x in (d, v, r)]   # + the halo slot
        ws = torch.zeros(max(16, L.pfa_gae_workspace_bytes(m)), dtype=torch.uint8, device='cuda')
        extra = torch.tensor([1.0, 2.0 * q, -0.5], dtype=torch.float64, device='cuda')
        out = torch.full((3 + 6 * shards,), float('nan'), dtype=torch.float64, device='cuda')
        _lib.check(L.pfa_gae_shard_publish(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), m, 0.99, 0.95, _lib.ptr(ws), _lib.ptr(extra), 3,
                                           _lib.ptr(out), q, shards, st), 'publish')
        assert int((out[3:] != 0).sum()) <= 6 and bool(torch.isfinite(out).all())      # zero outside this rank's six numbers
        total += out
        bufs.append((t, ws))
    assert total[:3].tolist() == [float(shards), float(shards * (shards - 1)), -0.5 * shards]
    got = np.empty(n, np.float32)
    for q, (t, ws) in enumerate(bufs):
        has_next = int(q < shards - 1)
        carry = torch.full((1,), float('nan'), dtype=torch.float64, device='cuda')
        _lib.check(L.pfa_gae_shard_fold(_lib.ptr(total[3:]), q, shards, m, 0.99, 0.95, _lib.ptr(ws), _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]),
                                        _lib.ptr(carry), st), 'fold')
        adv = torch.full((m,), float('nan'), device='cuda')
        ret = torch.full((m,), float('nan'), device='cuda')
        _lib.check(L.pfa_gae_shard_pass2(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(adv), _lib.ptr(ret), m,
                                         has_next, 0.99, 0.95, _lib.ptr(ws), _lib.ptr(carry) if has_next else None, st), 'pass2')
        got[q * m:(q + 1) * m] = adv.cpu().numpy()
        if has_next:       # the fold left the next shard's first row behind this shard's arrays
            assert [float(x[m]) for x in t] == [float(d[(q + 1) * m]), float(v[(q + 1) * m]), float(r[(q + 1) * m])]
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    assert np.abs(got - want).max() <= 8 * np.finfo(np.float32).eps * max(1.0, np.abs(want).max())
    assert got[-1] == 0.0


@pytest.mark.parametrize('N,T,nmb,bptt', [(4096, 128, 4, 16), (64, 128, 4, 16), (48, 32, 2, 8), (5, 16, 1, 16), (256, 64, 8, 32),
                                          (1000, 24, 1, 8), (8, 128, 32, 8)])
def test_one_pass_gae_with_sums_equals_the_separate_entry_points(N, T, nmb, bptt, gae_form):
    """pfa_gae_sums_f32 = pfa_gae_f32 (advantages / returns bit for bit) + the per-minibatch advantage sums (numpy f64 over the
    minibatch's rows, clean_pufferl.py:455-457 partition) + the explained-variance sums over the storage-order values
    (clean_pufferl.py:266-270)."""
    import torch
    from pufferlib_amd import _lib
    L = _lib.lib()
    n = N * T
    assert L.pfa_gae_sums_supported(n, N, nmb, bptt) == 1
    rng = np.random.RandomState(N + T)
    d = (rng.rand(n) < 0.05).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    want_adv, want_ret = hip_gae(d, v, r, 0.99, 0.95, want_returns=True)
    dev = 'cuda'
    dt, vt, rt = (torch.as_tensor(x).to(dev) for x in (d, v, r))
    adv = torch.full((n,), float('nan'), device=dev)
    ret = torch.full((n,), float('nan'), device=dev)
    stats = torch.full((nmb, 2), float('nan'), dtype=torch.float64, device=dev)
    ev4 = torch.full((4,), float('nan'), dtype=torch.float64, device=dev)
    zero8 = torch.ones(8, dtype=torch.float64, device=dev)
    ws = torch.zeros(L.pfa_gae_sums_workspace_bytes(n, nmb), dtype=torch.uint8, device=dev)
    for _ in range(2):   # (twice: the workspace is reused as it is)
        _lib.check(L.pfa_gae_sums_f32(_lib.ptr(dt), _lib.ptr(vt), _lib.ptr(rt), _lib.ptr(adv), _lib.ptr(ret), n, 0.99, 0.95, N, nmb, bptt,
                                      _lib.ptr(stats), _lib.ptr(ev4), _lib.ptr(zero8), _lib.ptr(ws), _lib.stream_handle()), 'gae_sums')
    assert np.array_equal(adv.cpu().numpy(), want_adv) and np.array_equal(ret.cpu().numpy(), want_ret)
    assert float(zero8.abs().sum()) == 0.0
    a = want_adv.astype(np.float64)
    seg = a.reshape(n // bptt, bptt)
    for m in range(nmb):
        rows = seg[m::nmb].ravel()
        np.testing.assert_allclose(stats[m].cpu().numpy(), [rows.sum(), (rows * rows).sum()], rtol=1e-12, atol=1e-9)
    yp = v.reshape(N, T).T.ravel().astype(np.float64)          # storage (step-major) order of the env-major value buffer
    yt = a + yp
    np.testing.assert_allclose(ev4.cpu().numpy(), [yt.sum(), (yt * yt).sum(), a.sum(), (a * a).sum()], rtol=1e-12, atol=1e-8)


def test_one_pass_gae_with_sums_refuses_partitions_it_cannot_bin():
    from pufferlib_amd import _lib
    L = _lib.lib()
    assert L.pfa_gae_sums_supported(8 * 128, 8, 8, 4) == 0       # bptt 4: a thread's 8 rows span two segments (ocean's config)
    assert L.pfa_gae_sums_supported(96 * 48, 96, 3, 16) == 0     # 3 minibatches: the cycle is no power of two
    assert L.pfa_gae_sums_supported(4096 * 128, 4096, 4, 16) == 1
