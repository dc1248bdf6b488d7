"""Data parallel on real kernels: two ranks share the one GPU of the test box and talk through gloo
(device tensors; PFA_ALLREDUCE=torch: the torch.distributed fallback path), each owning a shard of the envs.  The contract:
the update of R ranks on their shards == the update of ONE process on the rank-major concatenation of those shards
(GAE as a single flat scan across the shard boundary, global-minibatch advantage normalisation, summed gradients, one clip
norm), both ranks hold bit-identical parameters after every step, and episode statistics / step counts are global.
(Env trajectories themselves are per-process like the reference's: the `random.sample` stream is process-global, so a
rank is one reference process with seeds seed + r*N + i — config C5 — not a slice of a 2N-env process.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(__file__))
pytestmark = pytest.mark.gpu

HP = [2.5e-3, 0.99, 0.95, 0.1, 0.5, 0.1, 0.5, 0.01]
N_PER_RANK, HORIZON, ITERS = 32, 16, 2


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


CONV = 'conv'      # third policy kind next to recurrent = False / True: models.Convolutional on vector.Frames (4 envs x 8 steps per rank)
WIDE = 'wide'      # fourth: models.Default(hidden_size=256) — the GEMM-path engine (general.py) under data parallelism


def _sizes(recurrent):
    return (4, 8, 4) if recurrent == CONV else (N_PER_RANK, HORIZON, 8)       # envs per rank, horizon, bptt


def _loop(n_local, world, recurrent, inject=None):
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from test_gpu_ppo import _config
    torch.manual_seed(5)
    n_rank, HORIZON, bptt = _sizes(recurrent)
    if recurrent == CONV:
        vec = vector.make(vector.make_frames, env_kwargs=dict(episode_length=3), num_envs=n_local, backend=vector.Frames)
        pol = cleanrl.Policy(models.Convolutional(vec.driver_env, framestack=4))
    else:
        vec = vector.make(vector.make_squared, num_envs=n_local, backend=vector.Squared)
        base = models.Default(vec.driver_env, hidden_size=256 if recurrent == WIDE else 128)
        pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent is True else cleanrl.Policy(base)
    B = n_local * HORIZON
    data = clean_pufferl.create(_config(n_local, HORIZON, B // 2, bptt, 2, n_rank * 2 * HORIZON * 8, HP, seed=21,
                                        env='frames' if recurrent == CONV else 'squared'), vec, pol)
    out = {}
    for it in range(ITERS):
        stats, _ = clean_pufferl.evaluate(data)
        e = data.experience
        if inject is not None:          # replace the rollout by the given rank-major concatenation of shard rollouts
            for name, arr in inject[it].items():
                getattr(e, name).copy_(torch.as_tensor(arr).to(getattr(e, name).device))
        for name in ('obs', 'actions', 'logprobs', 'values', 'rewards', 'dones'):
            out[f'{it}.{name}'] = getattr(e, name).cpu().numpy().copy()
        clean_pufferl.train(data)
        out[f'{it}.advantages'] = e.advantages.cpu().numpy().copy()
        out[f'{it}.flat'] = data.flat_params.flat.cpu().numpy().copy()
        out[f'{it}.stats'] = np.array([stats.get('episode_return', np.nan), stats.get('score', np.nan)])
        out[f'{it}.losses'] = np.array([data.losses[k] for k in ('policy_loss', 'value_loss', 'entropy', 'approx_kl')])
        out[f'{it}.global_step'] = np.array([data.global_step])
    return out


TRANSPORTS = {
    # torch.distributed (gloo on device tensors) for every collective: the fallback path
    'torch': dict(PFA_ALLREDUCE='torch'),
    # the one-shot peer-mapped all-reduce (csrc/p2p.hip) for every collective of the update: IPC-mapped slots work between two
    # processes on one device exactly as between two devices
    'p2p': dict(PFA_ALLREDUCE='p2p'),
    # ... with the MLP policy's optimizer-step exchange as an all-reduce launch of its own instead of inside the reduce + Adam launch
    'p2p-unfused': dict(PFA_ALLREDUCE='p2p', PFA_FUSED_DP='0'),
    # ... with the sharded GAE and its two exchanges in train() instead of at the end of evaluate() (what a host vecenv's evaluate,
    # which does not publish, and callers that rewrite the experience between the two calls get)
    'p2p-late-gae': dict(PFA_ALLREDUCE='p2p', PFA_EARLY_GAE='0'),
    # ... with the f64-carry shard form of the GAE (what gamma * lambda > 0.984 selects: six numbers per rank instead of the halo rows; a few
    # ulps from the flat scan at the shard ends instead of its bits)
    'p2p-f64-gae': dict(PFA_ALLREDUCE='p2p', PFA_GAE_SELF='0'),
    # the native RCCL communicator: two ranks on ONE device are refused by RCCL ("duplicate GPU"); the refusal must be clean
    # on both ranks (no hang, torch's own RCCL instance unharmed) and the run must continue on the fallback path
    'rccl-refused': dict(PFA_ALLREDUCE='rccl'),
}


def _worker(rank, world, port, recurrent, out_dir, transport='torch'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0', **TRANSPORTS[transport])
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    sys.path.insert(0, os.path.dirname(__file__))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = _loop(_sizes(recurrent)[0], world, recurrent)
    from pufferlib_amd import dist as pdist
    res['native'] = np.array([int(pdist.native_ready()), int(pdist._native.get('p2p', False)), int(pdist._native.get('rccl', False))])
    from pufferlib_amd import _lib
    res['ll_calls'] = np.array([int(_lib.lib().pfa_p2p_ll_calls())])
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **res)
    dist.barrier()
    pdist.finalize_native()
    dist.destroy_process_group()


def _spawn(fn, args, world, timeout_s=240):
    """mp.spawn with a deadline: a collective that never completes must fail the test, not hang the box."""
    import time
    ctx = mp.spawn(fn, args=args, nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > timeout_s:
            for p in ctx.processes:
                p.kill()
            raise AssertionError(f'ranks still running after {timeout_s} s')


def _check_against_single_process(tmp_path, world, recurrent, exact_gae=True):
    N_PER_RANK, HORIZON, _ = _sizes(recurrent)
    r = [np.load(tmp_path / f'rank{q}.npz') for q in range(world)]
    names = ('obs', 'actions', 'logprobs', 'values', 'rewards', 'dones')
    inject = [{n: np.concatenate([r[q][f'{it}.{n}'] for q in range(world)]) for n in names} for it in range(ITERS)]
    single = _loop(N_PER_RANK * world, 1, recurrent, inject=inject)
    rows = N_PER_RANK * HORIZON
    for it in range(ITERS):
        for q in range(world):      # one flat GAE scan across the shard boundaries: the halo form leaves the single scan's own bits
            if exact_gae:
                assert np.array_equal(r[q][f'{it}.advantages'].view(np.uint32), single[f'{it}.advantages'][q * rows:(q + 1) * rows].view(np.uint32)), (it, q)
            else:
                np.testing.assert_allclose(r[q][f'{it}.advantages'], single[f'{it}.advantages'][q * rows:(q + 1) * rows], rtol=1e-5, atol=2e-6)
            assert np.array_equal(r[0][f'{it}.flat'], r[q][f'{it}.flat'])              # replicas stay bit-identical
            assert np.array_equal(r[0][f'{it}.stats'], r[q][f'{it}.stats'], equal_nan=True)   # all-reduced episode stats
            assert np.array_equal(r[0][f'{it}.losses'], r[q][f'{it}.losses'])
        np.testing.assert_allclose(r[0][f'{it}.flat'], single[f'{it}.flat'], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(r[0][f'{it}.losses'], single[f'{it}.losses'], rtol=1e-5, atol=1e-5)
        assert r[0][f'{it}.global_step'][0] == single[f'{it}.global_step'][0] == (it + 1) * world * rows
    return r


def test_eight_ranks_full_update_equals_single_process_run(tmp_path):
    """BASELINE configs[4]'s rank count before it meets eight GPUs: eight processes share the one GPU, each owning 32 envs; every
    optimizer step's exchange runs INSIDE the reduce + Adam launch (flag-in-data over the IPC-mapped peer buffers), the episode
    statistics and the sharded GAE's numbers ride one all-reduce, the advantage and explained-variance sums another.  After every
    evaluate -> train all eight ranks hold identical bits, equal (fp32 summation order) to ONE process training on the rank-major
    concatenation of the eight shard rollouts, with advantages equal to the single flat scan."""
    world = 8
    _spawn(_worker, (world, _free_port(), False, str(tmp_path), 'p2p'), world, timeout_s=420)
    r = _check_against_single_process(tmp_path, world, False)
    for q in range(world):
        assert tuple(int(x) for x in r[q]['native']) == (1, 1, 0)
        assert int(r[q]['ll_calls'][0]) == 4 + ITERS * 2 * 2          # self-test + iterations x epochs x minibatches, nothing else
    assert not np.array_equal(r[0]['0.obs'], r[7]['0.obs'])
    # SURVEY 8e's parity definition for the env side: rank q is one reference process over ITS shard — a Serial(Squared) of N envs
    # seeded seed + q N (+ env index: make_seeds, vector.py:639-641), the shared random.sample stream process-local — so its
    # trajectory under its own recorded actions is bit-exact against the C oracle seeded that way, continuously across updates
    from oracle import c_oracle
    N, T = N_PER_RANK, HORIZON
    for q in range(world):
        ovec = c_oracle.SquaredSerial(N, 3, 1)
        ovec.async_reset(21 + q * N)
        for it in range(ITERS):
            obs = r[q][f'{it}.obs'].reshape(N, T, -1)[:, :, :49]
            act, rew, don = (r[q][f'{it}.{k}'].reshape(N, T) for k in ('actions', 'rewards', 'dones'))
            for t in range(T):
                o, rr, dd = ovec.recv()[:3]
                assert np.array_equal(obs[:, t], o.reshape(N, -1)), (q, it, t)
                assert np.array_equal(rew[:, t].view(np.uint32), rr.view(np.uint32)) and np.array_equal(don[:, t] != 0, dd), (q, it, t)
                ovec.send(act[:, t].astype(np.int64))


def test_two_ranks_equal_single_process_run_with_the_gradient_step_in_the_bf16x6_form(tmp_path, monkeypatch):
    """The opt-in product form under data parallelism: the bf16-path gradient kernel (csrc/ppo_bf16.hpp) leaves 512 instead of 256
    partials per launch in front of the reduce + Adam launch that carries the exchange; ranks inherit PFA_MATRIX_PRODUCTS, the
    single-process reference run of this process switches with set_matrix_products."""
    import ctypes as C
    import pufferlib_amd
    from pufferlib_amd import _lib
    world = 2
    N, T, _ = _sizes(False)
    assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(_lib.MlpDims(49, 64, 128, 8, 0)), N * T // 2) == 0
    monkeypatch.setenv('PFA_MATRIX_PRODUCTS', 'bf16x6')
    pufferlib_amd.set_matrix_products('bf16x6')
    try:
        assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(_lib.MlpDims(49, 64, 128, 8, 0)), N * T // 2) == 1   # the rank-local minibatch takes it
        _spawn(_worker, (world, _free_port(), False, str(tmp_path), 'p2p'), world)
        r = _check_against_single_process(tmp_path, world, False)
    finally:
        pufferlib_amd.set_matrix_products('fp32')
    assert tuple(int(x) for x in r[0]['native']) == (1, 1, 0)


@pytest.mark.parametrize('recurrent,transport', [(False, 'torch'), (True, 'torch'), (False, 'p2p'), (True, 'p2p'), (False, 'rccl-refused'),
                                                 (CONV, 'torch'), (CONV, 'p2p'), (WIDE, 'torch'), (WIDE, 'p2p'), (False, 'p2p-unfused'), (False, 'p2p-late-gae'), (True, 'p2p-late-gae'),
                                                 (False, 'p2p-f64-gae'), (CONV, 'p2p-f64-gae')])
def test_two_ranks_on_one_gpu_equal_single_process_run(tmp_path, recurrent, transport):
    world = 2
    N_PER_RANK, HORIZON, _ = _sizes(recurrent)
    _spawn(_worker, (world, _free_port(), recurrent, str(tmp_path), transport), world)
    r = [np.load(tmp_path / f'rank{q}.npz') for q in range(world)]
    native = [tuple(int(x) for x in r[q]['native']) for q in range(world)]
    want_native = {'torch': (0, 0, 0), 'p2p': (1, 1, 0), 'p2p-unfused': (1, 1, 0), 'p2p-late-gae': (1, 1, 0), 'p2p-f64-gae': (1, 1, 0), 'rccl-refused': (0, 0, 0)}[transport]
    if recurrent == CONV and transport in ('p2p', 'p2p-f64-gae'):
        want_native = (0, 0, 0)          # the 6.7 MB bucket is over the peer path's 1 MiB cap: the run stays on torch.distributed
    assert native[0] == native[1] == want_native, native
    # the fused MLP update exchanges inside its reduce + Adam launch: one flag-in-data exchange per optimizer step (+ 4 in the self-test)
    steps = ITERS * 2 * 2 if (transport in ('p2p', 'p2p-late-gae', 'p2p-f64-gae') and recurrent is False) else 0
    assert int(r[0]['ll_calls'][0]) == int(r[1]['ll_calls'][0]) == (4 + steps if want_native[1] else 0)
    # (the single-process reference run of THIS process uses the default, bit-exact form; the f64-carry ranks are a few ulps off at shard ends)
    _check_against_single_process(tmp_path, world, recurrent, exact_gae=transport != 'p2p-f64-gae')
    # the shards really are different envs (seeds seed + r*N + i), not replicas of each other
    assert not np.array_equal(r[0]['0.obs'], r[1]['0.obs'])


def _p2p_worker(rank, world, port, out_dir, sizes=(1, 7, 9497, 9497, 153752, 262144, 3, 9497), reps=4):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', PFA_WAIT_TIMEOUT_MS='20000')
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pufferlib_amd import _lib, dist as pdist
    assert pdist.init_p2p(1 << 20)
    L = _lib.lib()
    ok = True
    g = torch.Generator(device='cuda').manual_seed(100 + rank)
    for it, n in enumerate(list(sizes) * reps):       # many calls: both phases are reused
        for dtype, fn in ((torch.float32, L.pfa_p2p_all_reduce_f32), (torch.float64, L.pfa_p2p_all_reduce_f64)):
            if dtype == torch.float64 and n > 131072:
                continue
            x = torch.randn(n, device='cuda', dtype=dtype, generator=g)
            mine = x.clone()
            _lib.check(fn(_lib.ptr(x), n, _lib.stream_handle()), 'p2p all-reduce')
            both = [torch.zeros_like(mine) for _ in range(world)]
            both[rank] = mine
            ref = torch.stack(both)
            dist.all_reduce(ref)                                       # gloo: every rank's contribution, then the rank-order sum
            want = ref[0]
            for q in range(1, world):
                want = want + ref[q]
            ok = ok and bool(torch.equal(x, want))
            if dtype == torch.float32 and n <= (1 << 18) + 2304:       # the flag-in-data form of the same exchange (csrc/p2p_ll.hpp)
                y = mine.clone()
                _lib.check(L.pfa_p2p_ll_all_reduce_f32(_lib.ptr(y), n, _lib.stream_handle()), 'p2p ll all-reduce')
                ok = ok and bool(torch.equal(y, want))
    assert L.pfa_p2p_status() == 0
    info = pdist.transport_info()
    assert info['p2p'] and info['p2p_world'] == world and info['p2p_calls'] > 0 and info['p2p_selftest'] is True, info
    np.save(os.path.join(out_dir, f'ok{rank}.npy'), np.array([int(ok)]))
    dist.barrier()
    L.pfa_p2p_close()
    dist.destroy_process_group()


def test_one_shot_peer_all_reduce_is_the_rank_order_sum_bit_for_bit(tmp_path):
    """csrc/p2p.hip between two processes on the one GPU: f32 and f64 buckets from 1 element to 1 MB, 64 calls (phase reuse),
    result == slot 0 + slot 1 in rank order on BOTH ranks, exactly."""
    world = 2
    _spawn(_p2p_worker, (world, _free_port(), str(tmp_path)), world)
    assert all(int(np.load(tmp_path / f'ok{q}.npy')[0]) == 1 for q in range(world))


def test_eight_ranks_on_one_gpu_run_the_full_slot_and_flag_logic(tmp_path):
    """R = 8 — the node size the peer path is built for — before it ever meets xGMI: eight processes on the one GPU, every rank
    pushing into seven peers' slots and waiting on seven flags per chunk, MLP- and LSTM-sized buckets, phase reuse; rank-order sums
    bit for bit on every rank."""
    world = 8
    _spawn(_p2p_worker, (world, _free_port(), str(tmp_path), (1, 9497, 153752, 5, 9497), 3), world, timeout_s=420)
    assert all(int(np.load(tmp_path / f'ok{q}.npy')[0]) == 1 for q in range(world))


def _p2p_skew_worker(rank, world, port, out_dir, calls=300):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', PFA_WAIT_TIMEOUT_MS='20000')
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pufferlib_amd import _lib, dist as pdist
    assert pdist.init_p2p(1 << 16)
    L = _lib.lib()
    n = 9497                                            # the MLP policy's bucket
    idx = torch.arange(n, device='cuda')
    rs = np.random.RandomState(1000 + rank)             # every rank stalls at different calls, for different lengths
    outs = []
    for j in range(calls):
        if rs.rand() < 0.3:
            torch.cuda._sleep(int(rs.randint(1, 400)) * 10000)      # up to ~2 ms of GPU-side delay in front of this rank's push
        x = (((idx + 31 * j) % 97) * (rank + 1)).float()
        fn = L.pfa_p2p_ll_all_reduce_f32 if j % 3 == 2 else L.pfa_p2p_all_reduce_f32       # both forms, interleaved
        _lib.check(fn(_lib.ptr(x), n, _lib.stream_handle()), 'p2p all-reduce')
        outs.append(x)                                  # no host synchronisation between calls: ranks run ahead of each other
    torch.cuda.synchronize()
    ok = L.pfa_p2p_status() == 0
    scale = world * (world + 1) // 2
    for j, x in enumerate(outs):
        ok = ok and bool(torch.equal(x, (((idx + 31 * j) % 97) * scale).float()))
    np.save(os.path.join(out_dir, f'ok{rank}.npy'), np.array([int(ok)]))
    dist.barrier()
    L.pfa_p2p_close()
    dist.destroy_process_group()


def test_back_to_back_peer_all_reduces_under_rank_skew(tmp_path):
    """What training does to the peer path: hundreds of all-reduces enqueued back to back with no host synchronisation in between,
    every rank stalling at different calls (GPU-side sleeps of up to ~2 ms in front of its push) — the two-phase slot reuse must hold
    a fast rank back exactly when a slow peer has not read the previous round yet.  Integer-valued buckets: every result is known
    exactly without a second collective."""
    world = 4
    _spawn(_p2p_skew_worker, (world, _free_port(), str(tmp_path)), world, timeout_s=300)
    assert all(int(np.load(tmp_path / f'ok{q}.npy')[0]) == 1 for q in range(world))


def _p2p_lost_peer_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', PFA_WAIT_TIMEOUT_MS='700')
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pufferlib_amd import _lib, dist as pdist
    assert pdist.init_p2p(1 << 16)                     # self-test passes: both ranks are there
    L = _lib.lib()
    res = dict(status_before=L.pfa_p2p_status())
    if rank == 0:                                      # rank 1 never joins this all-reduce
        x = torch.ones(5000, device='cuda')
        _lib.check(L.pfa_p2p_all_reduce_f32(_lib.ptr(x), x.numel(), _lib.stream_handle()), 'p2p all-reduce')
        torch.cuda.synchronize()                       # returns: the wait is bounded
        res.update(status=L.pfa_p2p_status(), all_nan=bool(torch.isnan(x).all()))
        try:
            pdist._native['p2p'] = True
            pdist.raise_if_peer_lost()
            res['raised'] = False
        except RuntimeError:
            res['raised'] = True
    # recovery (round 5): the ranks' sequence numbers differ now (rank 1 skipped a call); reset_p2p agrees on a new epoch on every rank
    pdist._native['p2p'] = True
    recovered = pdist.reset_p2p()
    y = torch.full((3000,), float(rank + 1), device='cuda')
    _lib.check(L.pfa_p2p_all_reduce_f32(_lib.ptr(y), y.numel(), _lib.stream_handle()), 'p2p all-reduce after reset')
    z = torch.full((3000,), float(rank + 1), device='cuda')
    _lib.check(L.pfa_p2p_ll_all_reduce_f32(_lib.ptr(z), z.numel(), _lib.stream_handle()), 'p2p ll all-reduce after reset')
    torch.cuda.synchronize()
    res.update(recovered=recovered, status_after=L.pfa_p2p_status(), sum_ok=bool((y == 3.0).all() and (z == 3.0).all()))
    np.savez(os.path.join(out_dir, f'lost{rank}.npz'), **{k: np.array([int(v)]) for k, v in res.items()})
    dist.barrier()
    L.pfa_p2p_close()
    dist.destroy_process_group()


def test_lost_peer_is_an_error_not_a_stale_sum(tmp_path):
    """a rank that never arrives must not leave the others with whatever was in the slots.  The wait is bounded
    in wall-clock time, the bucket comes back all-NaN, the status word is raised and clean_pufferl's readback check raises."""
    _spawn(_p2p_lost_peer_worker, (2, _free_port(), str(tmp_path)), 2, timeout_s=180)
    r = np.load(tmp_path / 'lost0.npz')
    assert int(r['status_before'][0]) == 0 and int(r['status'][0]) == 1 and int(r['all_nan'][0]) == 1 and int(r['raised'][0]) == 1
    for q in range(2):   # ... and pufferlib_amd.dist.reset_p2p() brings the path back on both ranks
        r = np.load(tmp_path / f'lost{q}.npz')
        assert int(r['recovered'][0]) == 1 and int(r['status_after'][0]) == 0 and int(r['sum_ok'][0]) == 1, (q, dict(r))


def _p2p_status_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', PFA_WAIT_TIMEOUT_MS='2000')
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pufferlib_amd import _lib, dist as pdist
    assert pdist.init_p2p(1 << 16)
    pdist._native['p2p'] = True
    L = _lib.lib()
    if rank == 1:
        _lib.check(L.pfa_p2p_debug_set_status(1), 'set_status')      # what a timed-out wait leaves on ONE rank
    x = torch.full((2000,), float(rank + 1), device='cuda')
    _lib.check(L.pfa_p2p_ll_all_reduce_f32(_lib.ptr(x), x.numel(), _lib.stream_handle()), 'll all-reduce')
    torch.cuda.synchronize()
    res = dict(status=L.pfa_p2p_status())
    try:
        pdist.raise_if_peer_lost()
        res['raised'] = 0
    except RuntimeError:
        res['raised'] = 1
    res['recovered'] = int(pdist.reset_p2p())
    res['status_after'] = L.pfa_p2p_status()
    np.savez(os.path.join(out_dir, f'st{rank}.npz'), **{k: np.array([int(v)]) for k, v in res.items()})
    dist.barrier()
    L.pfa_p2p_close()
    dist.destroy_process_group()


def test_a_timeout_on_one_rank_is_an_error_on_every_rank_in_the_same_update(tmp_path):
    """VERDICT round 4 (weak 4 / next 6b): only the rank whose wait ran out used to raise; a peer that was merely late saw a complete
    exchange and trained on.  The ranks' status words ride every flag-in-data exchange (csrc/p2p_ll.hpp: ll_status_exchange, also inside
    ppo_reduce_adam_kernel), so the next exchange raises the word on every rank: 1 where the wait ran out, 2 where a peer reported it."""
    world = 3
    _spawn(_p2p_status_worker, (world, _free_port(), str(tmp_path)), world, timeout_s=240)
    got = [np.load(tmp_path / f'st{q}.npz') for q in range(world)]
    assert [int(g['status'][0]) for g in got] == [2, 1, 2]
    assert all(int(g['raised'][0]) == 1 and int(g['recovered'][0]) == 1 and int(g['status_after'][0]) == 0 for g in got)


def test_injected_rank_skew_shows_up_as_peer_wait_on_the_ranks_that_were_on_time(tmp_path):
    """one rank's host thread is held back in front of every evaluate() (tools/dp_jitter.py); the in-kernel
    peer-wait telemetry (csrc/p2p_ll.hpp ll_wait_report, csrc/p2p.hip; pufferlib_amd.dist.wait_stats) must say so — the rank that was on
    time waits about the injected time per step (in the first exchange after the rollout: the small all-reduce at the end of evaluate()),
    the late rank does not — the iteration takes about that much longer, and the replicas stay bit-identical."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), 'tools'))
    import dp_jitter
    res = dp_jitter.run(world=2, envs=256, horizon=32, iters=12, warmup=2, skews=(0, 2000), late_rank=1, out=str(tmp_path / 'jitter.json'),
                        timeout_s=400)
    assert res['replicas_identical'] and res['p2p_status'] == 0
    quiet, skewed = res['legs']
    assert skewed['injected_skew_us'] == 2000
    on_time, late = skewed['wait_us_per_step']
    assert on_time > quiet['wait_us_per_step'][0] + 1000, (quiet, skewed)         # rank 0 stands waiting for rank 1 ...
    assert on_time > late + 1000, (quiet, skewed)                                  # ... which itself finds its peer's data there
    assert skewed['ms_per_step'] > quiet['ms_per_step'] + 1.0, (quiet, skewed)     # nothing hides a late rank: the step grows by about the skew
