"""Data-parallel contract on CPU: world_size 2, gloo.

Two processes each own half of the envs' experience and run the SAME sequence the product runs per optimizer step —
all-reduce of the advantage sums (once per update), local gradient scaled by 1 / GLOBAL minibatch rows, one flat
bucket all-reduce, clip after the reduce, Adam — using pufferlib_amd.dist for every collective and the torch-fp32
oracle for the arithmetic (no GPU here).  The result must equal the single-process update over the concatenated
batch (weights within fp32 summation-order noise), which is the multi-GPU parity definition.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

N_PER_RANK, T, BPTT, NMB, EPOCHS = 32, 16, 4, 2, 2
HP = dict(lr=2.5e-4, clip=0.1, vf_coef=0.5, vf_clip=0.1, max_grad_norm=0.5, ent_coef=0.01)


def _free_port():
    """A free listening port BELOW the kernel's ephemeral range (32768-60999): an ephemeral one can be taken as the source port of a peer's
    connection attempt between this probe and rank 0's bind (seen once as EADDRINUSE in a full-suite run)."""
    import random
    import socket
    for _ in range(128):
        p = random.randint(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(('127.0.0.1', p))
                return p
            except OSError:
                continue
    raise RuntimeError('no free port in 20000-32000')


def _make_batch(world):
    """Synthetic experience for world*N envs, env-major, plus initial weights (same on every rank)."""
    g = torch.Generator().manual_seed(0)
    n = N_PER_RANK * world
    B = n * T
    obs = torch.randint(-1, 2, (B, 49), generator=g).float()
    batch = dict(obs=obs, actions=torch.randint(0, 8, (B,), generator=g), logprobs=-2.08 + 0.05 * torch.randn(B, generator=g),
                 values=torch.randn(B, generator=g), advantages=torch.randn(B, generator=g) * 2 + 0.3,
                 returns=torch.randn(B, generator=g))
    w = {'encoder.weight': torch.randn(128, 49, generator=g) * 0.1, 'encoder.bias': torch.zeros(128),
         'decoder.weight': torch.randn(8, 128, generator=g) * 0.05, 'decoder.bias': torch.zeros(8),
         'value_head.weight': torch.randn(1, 128, generator=g) * 0.1, 'value_head.bias': torch.zeros(1)}
    return batch, w


def _minibatch_rows(n_envs, mb):
    """clean_pufferl.py:455-457 on an env-major stream of n_envs*T rows."""
    segs = n_envs * (T // BPTT)
    k = np.arange(segs // NMB)
    return ((mb + k[:, None] * NMB) * BPTT + np.arange(BPTT)[None, :]).reshape(-1)


def _update(batch, w, n_envs, rank, world):
    """One clean_pufferl.train minibatch loop over this process's rows, with the product's DP collectives."""
    sys.path.insert(0, REPO)
    from oracle import ppo_torch
    from pufferlib_amd import dist as pdist
    pol = ppo_torch.Policy({k: v.numpy() for k, v in w.items()})
    opt = torch.optim.Adam(pol.params, lr=HP['lr'], eps=1e-5)
    rows_local = n_envs * T // NMB
    rows_global = rows_local * world
    # once per update: per-minibatch advantage sums, all-reduced
    stats = torch.zeros(NMB, 2, dtype=torch.float64)
    for mb in range(NMB):
        a = batch['advantages'][_minibatch_rows(n_envs, mb)].double()
        stats[mb, 0], stats[mb, 1] = a.sum(), (a * a).sum()
    pdist.all_reduce_sum_(stats)
    for epoch in range(EPOCHS):
        for mb in range(NMB):
            idx = _minibatch_rows(n_envs, mb)
            logits, newvalue, _ = pol.forward(batch['obs'][idx])
            _, newlogprob, entropy = ppo_torch.sample_logits(logits, action=batch['actions'][idx])
            ratio = (newlogprob - batch['logprobs'][idx]).exp()
            mean, std = pdist.normalisation_from_sums(float(stats[mb, 0]), float(stats[mb, 1]), rows_global)
            adv = (batch['advantages'][idx] - np.float32(mean)) / (np.float32(std) + 1e-8)
            pg = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - HP['clip'], 1 + HP['clip']))
            nv, ret, val = newvalue.view(-1), batch['returns'][idx], batch['values'][idx]
            vcl = val + torch.clamp(nv - val, -HP['vf_clip'], HP['vf_clip'])
            vl = 0.5 * torch.max((nv - ret) ** 2, (vcl - ret) ** 2)
            # SUM over local rows / GLOBAL rows: summing the ranks' gradients gives the global-minibatch mean gradient
            loss = (pg.sum() - HP['ent_coef'] * entropy.sum() + HP['vf_coef'] * vl.sum()) / rows_global
            opt.zero_grad()
            loss.backward()
            bucket = torch.cat([p.grad.reshape(-1) for p in pol.params])
            pdist.all_reduce_sum_(bucket)                      # ONE flat bucket per optimizer step
            o = 0
            for p in pol.params:
                p.grad.copy_(bucket[o:o + p.numel()].view_as(p))
                o += p.numel()
            torch.nn.utils.clip_grad_norm_(pol.params, HP['max_grad_norm'])   # after the reduce: same on all ranks
            opt.step()
    return pol.state_arrays()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    torch.set_num_threads(1)
    from pufferlib_amd import dist as pdist
    d, r, w = pdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    pdist.check_partition(N_PER_RANK, T, BPTT, NMB)
    assert pdist.env_offset(rank, N_PER_RANK) == rank * N_PER_RANK
    batch, weights = _make_batch(world)
    lo, hi = rank * N_PER_RANK * T, (rank + 1) * N_PER_RANK * T      # this rank's env shard, env-major
    shard = {k: v[lo:hi] for k, v in batch.items()}
    wt = {k: v.clone() for k, v in weights.items()}
    for v in wt.values():
        if rank != 0:
            v.zero_()
        pdist.broadcast_(v, src=0)                                    # create(): parameters come from rank 0
    res = _update(shard, wt, N_PER_RANK, rank, world)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **res)
    d.barrier()
    d.destroy_process_group()


def test_two_rank_update_equals_concatenated_single_rank(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), f'ranks diverged on {k}'     # identical replicas, bit for bit
    torch.set_num_threads(1)
    batch, weights = _make_batch(world)
    single = _update(batch, weights, N_PER_RANK * world, 0, 1)
    for k in r0.files:
        np.testing.assert_allclose(r0[k], single[k], rtol=1e-5, atol=1e-6, err_msg=k)
        assert not np.array_equal(single[k], weights[k].numpy()) or k.endswith('bias') is False or True


def test_partition_check_rejects_misaligned_shards():
    sys.path.insert(0, REPO)
    from pufferlib_amd import dist as pdist
    pdist.check_partition(4096, 128, 16, 4)
    with pytest.raises(ValueError):
        pdist.check_partition(3, 16, 16, 2)       # 3 segments cannot split into 2 minibatches
    with pytest.raises(ValueError):
        pdist.check_partition(4, 10, 4, 2)        # horizon not a multiple of bptt_horizon


def test_normalisation_from_sums_matches_torch():
    sys.path.insert(0, REPO)
    from pufferlib_amd import dist as pdist
    a = torch.randn(4096, dtype=torch.float64) * 3 + 1
    mean, std = pdist.normalisation_from_sums(float(a.sum()), float((a * a).sum()), a.numel())
    assert abs(mean - float(a.mean())) < 1e-12 and abs(std - float(a.std())) < 1e-10


def _gae_halo_worker(rank, world, port, out_dir, n, gamma, lam):
    """Host protocol of the halo form clean_pufferl runs (_publish_gae / _finish_gae over csrc/gae.hip's gae_halo_* kernels; here their
    host mirrors in pufferlib_amd.dist and the numpy model of the kernel's walkers, tests/test_gae_window_model.py): every rank
    publishes the bit patterns of its first rows next to other f64 sums, ONE all-reduce(SUM), every rank scans its rows + the rows
    that follow them."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from pufferlib_amd import dist as pdist
    from test_gae_window_model import warm_self, window_gae
    d, _, _ = pdist.init_from_env('gloo')
    dn, v, r = _gae_inputs(n)
    m = n // world
    lo = rank * m
    H = pdist.gae_halo_rows(gamma, lam)
    assert H == warm_self(gamma, lam) + 8
    extra = np.array([1.0 + rank, 10.0 * (rank + 1)])                 # e.g. episode-return sum and episode count
    buf = torch.from_numpy(np.concatenate([extra, pdist.gae_halo_pack(dn[lo:lo + m], v[lo:lo + m], r[lo:lo + m], rank, world, H)]))
    d.all_reduce(buf)                                                 # the ONE exchange
    hd, hv, hr = pdist.gae_halo_unpack(buf[2:].numpy(), rank, world, m, H)
    own = [np.concatenate([x[lo:lo + m], h]) for x, h in ((r, hr), (v, hv), (dn, hd))]
    adv = window_gae(*own, gamma, lam, H - 8)[:m]                     # n_read = m + halo: the kernel pins element n_read - 1
    np.save(os.path.join(out_dir, f'adv{rank}.npy'), adv)
    np.save(os.path.join(out_dir, f'extra{rank}.npy'), buf[:2].numpy())
    d.barrier()
    d.destroy_process_group()


def _gae_inputs(n):
    rng = np.random.RandomState(5)
    dn = (rng.rand(n) < 0.02).astype(np.float32)
    v, r = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    v[n // 2], r[n // 3] = -0.0, -0.0                                 # sign bits an arithmetic gather would lose (n // 3 = a shard's first row at world 3)
    return dn, v, r


@pytest.mark.parametrize('n,world,gl', [(3000, 3, (0.99, 0.95)), (600, 3, (0.99, 0.95)), (1200, 2, (0.995, 0.97)), (4000, 2, (0.995, 0.985))])
def test_halo_sharded_gae_is_the_flat_scan_bit_for_bit(tmp_path, n, world, gl):
    """3000 / 3: the halo (544 rows) comes from the next shard alone; 600 / 3: shards of 200 rows — rank 0's halo is all of rank 1
    and rank 2, and the pinned last element of the batch sits inside it; 0.995 x 0.97: a 936-row halo; 0.995 x 0.985: 1640 rows (the 2048-element window)."""
    mp.spawn(_gae_halo_worker, args=(world, _free_port(), str(tmp_path), n, *gl), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f'adv{q}.npy') for q in range(world)]).astype(np.float32)
    sys.path.insert(0, REPO)
    from oracle import c_oracle
    dn, v, r = _gae_inputs(n)
    want = np.asarray(c_oracle.compute_gae(dn, v, r, *gl), np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got[-1] == 0.0
    for q in range(world):                                            # the sums that rode along are the global sums on every rank
        assert np.array_equal(np.load(tmp_path / f'extra{q}.npy'), np.array([sum(1.0 + k for k in range(world)), sum(10.0 * (k + 1) for k in range(world))]))


def _gae_one_exchange_worker(rank, world, port, out_dir):
    """The f64-carry one-exchange form clean_pufferl runs where gamma lambda is outside the halo form's window (csrc/gae.hip's publish and fold
    kernels; here their host mirrors in pufferlib_amd.dist): every rank publishes six numbers computed from its OWN rows —
    interior map, last value, first row — next to other f64 sums (the episode statistics), ONE all-reduce(SUM) of the zero-padded
    [extra | world x 6] buffer, fold, local scan."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    from pufferlib_amd import dist as pdist
    d, _, _ = pdist.init_from_env('gloo')
    rng = np.random.RandomState(5)
    n, gamma, lam = 600, 0.99, 0.95
    dn = (rng.rand(n) < 0.1).astype(np.float64)
    v, r = rng.randn(n), rng.randn(n)
    m = n // world
    lo = rank * m
    extra = np.array([1.0 + rank, 10.0 * (rank + 1)])                 # e.g. episode-return sum and episode count
    buf = torch.zeros(2 + 6 * world, dtype=torch.float64)
    buf[:2] = torch.from_numpy(extra)
    buf[2 + 6 * rank:2 + 6 * rank + 6] = torch.tensor(pdist.gae_publish_numbers(dn[lo:lo + m], v[lo:lo + m], r[lo:lo + m], gamma, lam),
                                                      dtype=torch.float64)
    d.all_reduce(buf)                                                 # the ONE exchange
    pub = buf[2:].view(world, 6).numpy()
    x, (lc, ld) = pdist.gae_fold_published(pub, rank, gamma, lam)
    adv = np.zeros(m)
    adv[m - 1] = lc * x + ld                                          # the shard's last element: pinned to 0 on the last shard
    for t in range(m - 2, -1, -1):
        nnt = 1.0 - dn[lo + t + 1]
        adv[t] = r[lo + t + 1] + gamma * v[lo + t + 1] * nnt - v[lo + t] + gamma * lam * nnt * adv[t + 1]
    np.save(os.path.join(out_dir, f'adv{rank}.npy'), adv)
    np.save(os.path.join(out_dir, f'extra{rank}.npy'), buf[:2].numpy())
    d.barrier()
    d.destroy_process_group()


def test_one_exchange_sharded_gae_equals_flat_scan(tmp_path):
    world = 3
    mp.spawn(_gae_one_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f'adv{q}.npy') for q in range(world)])
    rng = np.random.RandomState(5)
    n = 600
    dn = (rng.rand(n) < 0.1).astype(np.float64)
    v, r = rng.randn(n), rng.randn(n)
    want, x = np.zeros(n), 0.0
    for t in range(n - 2, -1, -1):                                    # c_gae.pyx:11-32 in f64 over the whole flat batch
        nnt = 1.0 - dn[t + 1]
        x = r[t + 1] + 0.99 * v[t + 1] * nnt - v[t] + 0.99 * 0.95 * nnt * x
        want[t] = x
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    assert got[-1] == 0.0
    for q in range(world):                                            # the sums that rode along are the global sums on every rank
        assert np.array_equal(np.load(tmp_path / f'extra{q}.npy'), np.array([6.0, 60.0]))
