"""Data-parallel plumbing (no reference counterpart): one process per GPU under
``python -m torch.distributed.run``; ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm), "gloo" in the
CPU tests.

Sharding contract
  * rank r of R owns global envs [r*N, (r+1)*N): seeds ``seed + global index`` (vector.py:639-641 semantics) and Philox
    noise rows = global env index.  A rank is one reference PROCESS (SURVEY config C5): the reset-target stream of
    ``random.sample`` is process-global in the reference, so after their first episode the envs of a rank follow that
    rank's own stream — a rank is not a slice of one big process;
  * every rank keeps its own env-major experience and applies the reference's minibatch partition locally; because
    ``N * (T / bptt_horizon) % num_minibatches == 0`` (checked) local minibatch m is exactly this rank's share of
    global minibatch m;
  * GAE is the reference's single scan over the rank-major flat batch, bit for bit: every rank's first rows travel (as bit
    patterns, next to the episode statistics) in ONE all-reduce at the end of evaluate(), every rank runs the single-rank kernel
    over its shard + the rows that follow it (csrc/gae.hip gae_halo_*; gamma lambda > 0.984: six numbers per rank and an f64 carry,
    a few ulps at the shard ends);
  * then ONE all-reduce(SUM) of the per-minibatch advantage sums [nmb][2] and the explained-variance sums (f64) so every rank
    normalises with the global-minibatch mean / unbiased std — also at the end of evaluate(); per optimizer step ONE all-reduce(SUM) of the flat bucket
    [gradient (already divided by the GLOBAL minibatch rows) | 8 loss sums]; the clip norm is taken after it, so
    every rank applies the identical Adam step to identical parameters (broadcast once at create()).
"""
import math
import os


def world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def init_from_env(backend='nccl', device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return world()
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return None, 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device(f'cuda:{local}')
        pin_rank(local, device_index=local)
    dist.init_process_group(backend, rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']), **kw)
    return world()


def _parse_cpulist(text):
    """'0-63,128-191' -> sorted list of CPU numbers (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def _gpu_numa_node(device_index):
    """NUMA node of HIP device `device_index` from the amdgpu driver's sysfs entry of its PCI function; -1 = unknown (no NUMA
    information in a VM / container, no GPU)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        with open(f'/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/numa_node') as f:
            return int(f.read().strip())
    except Exception:
        return -1


def plan_affinity(local_rank, local_world, nodes, node_cpus, allowed):
    """Pure planning step of pin_rank (unit-tested on the CPU): `nodes[i]` = NUMA node of local rank i's GPU (-1 unknown),
    `node_cpus[n]` = CPUs of node n, `allowed` = the process's current affinity mask.  A rank gets its share of the CPUs of its GPU's
    node, split evenly among the local ranks whose GPUs sit on the same node (whole cores of a 2-socket host stay with the GPUs next
    to them; two ranks never compete for one core); without NUMA information, its share of the allowed CPUs.  Never empty: a share
    that would be empty falls back to the whole candidate set."""
    allowed = sorted(allowed)
    node = nodes[local_rank] if 0 <= local_rank < len(nodes) else -1
    pool = sorted(set(node_cpus.get(node, [])) & set(allowed)) if node >= 0 else []
    if pool:
        peers = [i for i in range(local_world) if nodes[i] == node]
    else:
        pool, peers, node = allowed, list(range(local_world)), -1
    k, n = peers.index(local_rank), len(peers)
    per = len(pool) // n
    share = pool[k * per:(k + 1) * per] if per >= 1 else pool
    return node, share or pool


def pin_rank(local_rank=None, local_world=None, device_index=None):
    """Pin this rank's process (all of its threads created from here on) to the CPU cores next to its GPU: os.sched_setaffinity to
    its share of the NUMA node of the device (plan_affinity).  An iteration makes two host round trips with polled waits of ~2 ms;
    on a 2-socket, 256-thread host an unpinned rank migrates between sockets, its waits and launches jitter, and with an exchange
    inside every optimizer step every rank waits for the slowest one 16 times per update — host jitter becomes rank skew.
    PFA_RANK_AFFINITY=0 leaves the mask alone.  Returns (and remembers for bench.py's `dist` block) {numa_node, cpus, first, last}."""
    import torch
    info = dict(numa_node=-1, cpus=0, first=-1, last=-1, pinned=False)
    try:
        if os.environ.get('PFA_RANK_AFFINITY', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
            raise RuntimeError('off')
        local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))) if local_world is None else local_world
        ndev = max(torch.cuda.device_count(), 1)
        nodes = [_gpu_numa_node(i % ndev) for i in range(local_world)]
        if device_index is not None:
            nodes[local_rank] = _gpu_numa_node(device_index)
        node_cpus = {}
        for n in set(x for x in nodes if x >= 0):
            try:
                with open(f'/sys/devices/system/node/node{n}/cpulist') as f:
                    node_cpus[n] = _parse_cpulist(f.read())
            except OSError:
                pass
        allowed = os.sched_getaffinity(0)
        node, share = plan_affinity(local_rank, local_world, nodes, node_cpus, allowed)
        if local_world > 1 or node >= 0:
            os.sched_setaffinity(0, share)
            info.update(pinned=True)
        info.update(numa_node=node, cpus=len(share), first=share[0], last=share[-1])
    except Exception:
        pass
    _native['affinity'] = info
    return info


def wait_stats(reset=False):
    """Peer-wait telemetry of the peer path (csrc/p2p_ll.hpp ll_wait_report, csrc/p2p.hip): mean microseconds a launch stood
    waiting for its slowest peer, for the optimizer steps' flag-in-data exchange and for the small flag-based all-reduces, and how
    many workgroups reported.  Synchronises the device."""
    import ctypes as C
    from . import _lib
    out = (C.c_int64 * 4)()
    _lib.check(_lib.lib().pfa_p2p_wait_stats(out, 1 if reset else 0), 'p2p_wait_stats')
    us = lambda ticks, n: (ticks / n / 100.0) if n else 0.0       # 100 MHz ticks
    return dict(grad_exchange_wait_us=us(out[0], out[1]), grad_exchange_workgroups=int(out[1]),
                small_exchange_wait_us=us(out[2], out[3]), small_exchange_chunks=int(out[3]))


def env_offset(rank, envs_per_rank):
    return rank * envs_per_rank


def all_reduce_sum_(tensor):
    d, _, w = world()
    if w > 1:
        d.all_reduce(tensor)
    return tensor


def broadcast_(tensor, src=0):
    d, _, w = world()
    if w > 1:
        d.broadcast(tensor, src=src)
    return tensor


def check_partition(envs_per_rank, horizon, bptt_horizon, num_minibatches):
    """Local minibatch m must be this rank's share of global minibatch m (see module docstring)."""
    segments = envs_per_rank * (horizon // bptt_horizon)
    if horizon % bptt_horizon != 0 or segments % num_minibatches != 0:
        raise ValueError('data-parallel sharding needs envs_per_rank * (horizon / bptt_horizon) divisible by '
                         f'num_minibatches (got {envs_per_rank} x {horizon}/{bptt_horizon} vs {num_minibatches})')


def normalisation_from_sums(s1, s2, count):
    """mean and unbiased std of a (global) minibatch from sum / sum of squares / row count — the formula the fused
    update kernel applies (csrc/ppo_update.hip) to match ``adv.mean()`` / ``adv.std()`` (clean_pufferl.py:212-213)."""
    mean = s1 / count
    var = max((s2 - s1 * mean) / (count - 1.0), 0.0)
    return mean, math.sqrt(var)


def gae_halo_rows(gamma, lam):
    """Host mirror of pfa_gae_halo_rows (csrc/gae.hip): rows a shard needs from behind its end so that every walker of the
    self-starting window sits on the flat scan's rounded sequence — warm-up ((gamma lambda)^warm <= 1e-7 * 2^-24, a multiple of 8)
    + 8; 0 when that exceeds the kernel's largest (2048-element) window (gamma lambda > 0.984)."""
    import numpy as np
    gl = float(np.float32(gamma) * np.float32(lam))
    if not gl > 0.0:
        return 16
    if gl >= 0.999:
        return 0
    w = math.ceil(math.log(1e-7 * 2.0 ** -24) / math.log(gl) / 8.0) * 8
    return int(max(w, 8)) + 8 if w <= 2048 else 0


def gae_halo_pack(dones, values, rewards, rank, world, H):
    """Host mirror of gae_halo_publish_kernel: [world][3][min(n, H)] f64, this rank's first rows as BIT PATTERNS (exact under a
    SUM with the other ranks' zeros, sign of zero included), zeros elsewhere."""
    import numpy as np
    hp = min(len(values), H)
    out = np.zeros((world, 3, hp), np.float64)
    for k, x in enumerate((dones, values, rewards)):
        out[rank, k] = np.ascontiguousarray(x[:hp], np.float32).view(np.uint32).astype(np.float64)
    return out.reshape(-1)


def gae_halo_unpack(gathered, rank, world, n, H):
    """Host mirror of gae_halo_unpack_kernel: the (dones, values, rewards) rows that follow shard `rank` in the rank-major flat
    batch, min(H, (world - 1 - rank) n) of them — from several later shards when those are shorter than the halo."""
    import numpy as np
    hp = min(n, H)
    g = np.asarray(gathered, np.float64).reshape(world, 3, hp)
    halo_len = min(H, (world - 1 - rank) * n)
    i = np.arange(halo_len)
    q, row = rank + 1 + i // n, i % n
    return tuple(g[q, k, row].astype(np.uint32).view(np.float32) for k in range(3))


def gae_publish_numbers(dones, values, rewards, gamma, lam):
    """Host mirror (numpy f64) of what a rank publishes for the one-exchange data-parallel GAE (csrc/gae.hip
    gae_shard_publish_kernel): the affine map (C, D) of its elements 0 .. n-2 — every one of them reads its successor inside the
    shard — then values[n-1] and the shard's first row (done, value, reward).  Used by the CPU protocol test; the product runs the
    device kernels."""
    n = len(values)
    C, D = 1.0, 0.0
    for t in range(n - 2, -1, -1):
        nnt = 1.0 - dones[t + 1]
        coef, delta = gamma * lam * nnt, rewards[t + 1] + gamma * values[t + 1] * nnt - values[t]
        C, D = coef * C, delta + coef * D          # compose(f_t, acc): f_t is applied last
    return [C, D, float(values[n - 1]), float(dones[0]), float(values[0]), float(rewards[0])]


def gae_fold_published(pub, rank, gamma, lam):
    """Host mirror of gae_shard_fold_kernel: from the gathered [world][6] numbers, (carry-in of `rank`'s shard = the advantage of
    the first element after it, the (coef, delta) of this shard's last element).  The last element of the last shard is pinned
    to advantage 0 (the map x -> 0)."""
    world = len(pub)

    def last_map(q):
        if q == world - 1:
            return 0.0, 0.0
        d1, v1, r1 = pub[q + 1][3], pub[q + 1][4], pub[q + 1][5]
        nnt = 1.0 - d1
        return gamma * lam * nnt, r1 + gamma * v1 * nnt - pub[q][2]

    x = 0.0
    for q in range(world - 1, rank, -1):
        lc, ld = last_map(q)
        c, d = pub[q][0] * lc, pub[q][0] * ld + pub[q][1]     # interior o last
        x = c * x + d
    return x, last_map(rank)


_native = dict(ready=False, world=1, rccl=False, p2p=False, p2p_selftest=None, p2p_reason=None)


def native_ready():
    return _native['ready']


P2P_MAX_BUCKET = 1 << 20


def init_p2p(bucket_bytes):
    """Open the one-shot peer-mapped all-reduce (csrc/p2p.hip) for buckets of up to ``bucket_bytes``: every rank allocates its
    fine-grained slot buffer, the 64-byte IPC handles travel through the torch.distributed group, every rank maps its peers.
    One node, world size <= 8.  Returns True when EVERY rank succeeded (MIN all-reduce), else closes again everywhere."""
    import ctypes as C
    import numpy as np
    import torch
    from . import _lib
    d, rank, w = world()
    if w <= 1 or w > 8:
        return False
    L = _lib.lib()
    # dmabuf IPC has to be chosen before HIP initialises: pufferlib_amd/__init__.py sets the variable at import and records what it
    # found.  A rank whose process cannot share memory any more reports failure here, and EVERY rank then stays off the peer path
    # (the flag travels with the handles below), with the reason logged once.
    from . import IPC_MODE
    handle = np.zeros(64, np.uint8)
    ipc_ok = bool(IPC_MODE['ok']) and os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'
    if not ipc_ok:
        _native['p2p_reason'] = IPC_MODE['reason'] or 'HSA_ENABLE_IPC_MODE_LEGACY was changed after import'
        import warnings
        warnings.warn('pufferlib_amd: peer-mapped all-reduce disabled on rank %d: %s; the update uses RCCL / torch.distributed'
                      % (rank, _native['p2p_reason']))
    ok = int(ipc_ok and L.pfa_p2p_alloc(int(bucket_bytes), w, handle.ctypes.data_as(C.c_void_p)) == 0)
    table = torch.zeros(w, 65, dtype=torch.int32, device='cuda')     # 64 handle bytes + the rank's success flag
    table[rank, :64] = torch.from_numpy(handle.astype(np.int32)).cuda()
    table[rank, 64] = ok
    d.all_reduce(table)
    host = table.cpu().numpy()
    ok = int(host[:, 64].min())
    if ok:
        handles = np.ascontiguousarray(host[:, :64].astype(np.uint8))
        ok = int(L.pfa_p2p_open(handles.ctypes.data_as(C.c_void_p), rank, w) == 0)
    flag = torch.tensor([ok], dtype=torch.int32, device='cuda')
    d.all_reduce(flag, op=d.ReduceOp.MIN)
    ok = int(flag.item())
    if not ok:
        L.pfa_p2p_close()
        return False
    d.barrier()                                                       # nobody pushes into a buffer that is not mapped yet
    # Self-test before anything depends on the transport: a few all-reduces of integer-valued data (exact in any order) against
    # torch.distributed's result.  The kernel's spins are bounded, so a peer that cannot be reached shows up as a status word or
    # a wrong sum here — and the update then stays on RCCL / torch.distributed — instead of as a hang in the first optimizer step.
    good = 1
    for n, dtype, fn in ((1024, torch.float32, L.pfa_p2p_all_reduce_f32), (37, torch.float64, L.pfa_p2p_all_reduce_f64),
                         (min(int(bucket_bytes) // 4, 1 << 18), torch.float32, L.pfa_p2p_all_reduce_f32)):
        x = ((torch.arange(n, device='cuda') % 251) * (rank + 1)).to(dtype)
        want = x.clone()
        d.all_reduce(want)
        if fn(x.data_ptr(), n, _lib.stream_handle()) != 0:
            good = 0
        torch.cuda.synchronize()
        if good and (L.pfa_p2p_status() != 0 or not torch.equal(x, want)):
            good = 0
    # ... and the flag-in-data form (csrc/p2p_ll.hpp) the fused optimizer step uses, twice per phase
    for j in range(4):
        n = min(int(bucket_bytes) // 4 + 2303, 1 << 16)     # (the slot's last entry carries the ranks' status words)
        x = (((torch.arange(n, device='cuda') + 7 * j) % 127) * (rank + 1)).float()
        want = x.clone()
        d.all_reduce(want)
        if L.pfa_p2p_ll_all_reduce_f32(x.data_ptr(), n, _lib.stream_handle()) != 0:
            good = 0
        torch.cuda.synchronize()
        if good and (L.pfa_p2p_status() != 0 or not torch.equal(x, want)):
            good = 0
    flag = torch.tensor([good], dtype=torch.int32, device='cuda')
    d.all_reduce(flag, op=d.ReduceOp.MIN)
    _native['p2p_selftest'] = bool(int(flag.item()))
    if not int(flag.item()):
        L.pfa_p2p_close()
        return False
    return True


def init_native(force_single=False, bucket_bytes=0, small_bytes=0):
    """Native collectives for the update: the optimizer-step all-reduce (and the few small reductions around it) are enqueued
    from native code on the compute stream, with no stream hand-off.  Two transports; ``PFA_ALLREDUCE`` names the ones allowed
    (comma list; default ``p2p,rccl``; ``torch`` = neither: every collective goes through torch.distributed):
      ``rccl``   this process's own RCCL communicator inside libpufferlib_amd.so (csrc/dist.cpp); the 128-byte id travels from
                 rank 0 through the already-initialised torch.distributed group;
      ``p2p``    the one-shot peer-mapped all-reduce (csrc/p2p.hip) for every bucket of up to ``bucket_bytes`` (<= 1 MiB) — one hop
                 over the xGMI mesh instead of a ring; it is opened only if its self-test against torch.distributed passes on every
                 rank; larger buckets go to RCCL (when allowed), else torch.distributed.
    ``small_bytes``: the largest of the update's other (f64) exchanges — the GAE halo rows next to the episode statistics — so that the
    peer path's slots hold it too and no exchange of an iteration needs a second transport.
    Returns True when a native transport is up on EVERY rank (agreement by MIN all-reduce); otherwise all ranks use
    torch.distributed collectives.  ``force_single`` builds a 1-rank RCCL communicator without a process group (tests)."""
    import ctypes as C
    import numpy as np
    import torch
    from . import _lib
    d, rank, w = world()
    mode = [m.strip() for m in os.environ.get('PFA_ALLREDUCE', 'p2p,rccl').lower().split(',')]
    p2p_ok = False
    # one hop pays for latency-bound buckets; a multi-MB bucket (the conv policy's 6.7 MB) is bandwidth-bound and stays on RCCL
    if 'p2p' in mode and w > 1 and 0 < bucket_bytes <= P2P_MAX_BUCKET:
        p2p_ok = init_p2p(max(int(bucket_bytes), int(small_bytes), 65536))
        _native.update(p2p=p2p_ok)
    if 'rccl' not in mode:
        _native.update(ready=p2p_ok, world=w if p2p_ok else 1)
        return p2p_ok
    if w == 1 and not force_single:
        return False
    L = _lib.lib()
    ok = 1
    ident = np.zeros(129, np.uint8)                   # 128-byte id + rank 0's success flag, so nobody waits on a dead id
    if rank == 0:
        ok = int(L.pfa_dist_unique_id(ident.ctypes.data_as(C.c_void_p)) == 0)
        ident[128] = ok
    if w > 1:
        t = torch.from_numpy(ident).cuda()
        d.broadcast(t, src=0)
        ident = t.cpu().numpy()
        ok = int(ident[128])
    if ok and L.pfa_dist_init(ident.ctypes.data_as(C.c_void_p), rank, w) != 0:
        ok = 0
    if w > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device='cuda')
        d.all_reduce(flag, op=d.ReduceOp.MIN)
        ok = int(flag.item())
        if not ok:
            L.pfa_dist_finalize()
    _native.update(ready=bool(ok) or p2p_ok, world=w, rccl=bool(ok))
    return bool(ok) or p2p_ok


def transport_info():
    """What the update's collectives actually run on (bench.py prints it; tests assert on it): RCCL communicator and its
    ncclCommCount, the peer path, how many all-reduces each has carried so far, and the peer path's status word."""
    import ctypes as C
    from . import _lib
    out = (C.c_int64 * 8)()
    _lib.check(_lib.lib().pfa_dist_info(out), 'dist_info')
    return dict(native=bool(_native['ready']), rccl=bool(out[0]), rccl_nranks=int(out[1]), p2p=bool(out[2]), p2p_world=int(out[3]),
                p2p_slot_bytes=int(out[4]), p2p_calls=int(out[5]), rccl_calls=int(out[6]), p2p_status=int(out[7]),
                p2p_selftest=_native.get('p2p_selftest'), p2p_ll_calls=int(_lib.lib().pfa_p2p_ll_calls()),
                p2p_reason=_native.get('p2p_reason'))


def reset_p2p():
    """Collective recovery after a raised status word (a timed-out exchange: `raise_if_peer_lost` raised on every rank): agree on a
    sequence number past every rank's, clear the status words, and check the path with one flag-in-data all-reduce.  The replicas'
    PARAMETERS are not repaired here — a timed-out rank holds NaN: reload a checkpoint (clean_pufferl.try_load_checkpoint) or
    broadcast rank 0's before training on.  Returns True when the path answers again on every rank; False closes nothing (the caller
    may retry or set PFA_ALLREDUCE=rccl and re-create)."""
    import torch
    from . import _lib
    d, rank, w = world()
    if not _native.get('p2p') or w <= 1:
        return False
    L = _lib.lib()
    torch.cuda.synchronize()
    seq = torch.tensor([int(L.pfa_p2p_seq())], dtype=torch.int64, device='cuda')
    d.all_reduce(seq, op=d.ReduceOp.MAX)
    d.barrier()                                        # nobody is still inside an exchange of the old epoch
    _lib.check(L.pfa_p2p_reset(int(seq.item()) + 64), 'p2p_reset')
    d.barrier()
    x = (torch.arange(4096, device='cuda') % 97 * (rank + 1)).float()
    want = x.clone()
    d.all_reduce(want)
    ok = int(L.pfa_p2p_ll_all_reduce_f32(x.data_ptr(), x.numel(), _lib.stream_handle()) == 0)
    torch.cuda.synchronize()
    ok = int(ok and L.pfa_p2p_status() == 0 and torch.equal(x, want))
    flag = torch.tensor([ok], dtype=torch.int32, device='cuda')
    d.all_reduce(flag, op=d.ReduceOp.MIN)
    return bool(int(flag.item()))


def raise_if_peer_lost():
    """The peer all-reduce's waits are bounded (csrc/p2p.hip): when one ran out the bucket was filled with NaN and the status word
    raised.  Called where train()/evaluate() read their results back, so a dead or stalled rank ends the run with an error."""
    from . import _lib
    st = _lib.lib().pfa_p2p_status() if _native.get('p2p') else 0
    if st > 0:
        raise RuntimeError(('data-parallel all-reduce over the peer path timed out waiting for a rank (PFA_WAIT_TIMEOUT_MS)' if st == 1 else
                            'a peer rank reported a timed-out exchange over the peer path (its replica holds NaN)')
                           + '; the gradients of this update are invalid on every rank.  Recovery: restore the parameters '
                             '(try_load_checkpoint / broadcast) and call pufferlib_amd.dist.reset_p2p() on every rank')


def finalize_native():
    from . import _lib
    if _native['ready']:
        if _native.get('p2p'):
            _lib.lib().pfa_p2p_close()
        _lib.lib().pfa_dist_finalize()
        _native.update(ready=False, world=1, rccl=False, p2p=False, p2p_selftest=None)
