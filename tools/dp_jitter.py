"""Skew budget of the data-parallel iteration, measured: R ranks (one process each, sharing the visible device(s) through gloo + the
peer-mapped path, as tests/test_gpu_dp.py runs them) train the BASELINE-shaped MLP workload while ONE rank's host thread is held
back by a fixed time in front of every evaluate() — what an unpinned, migrating or pre-empted rank process does to the others.
Per injected skew: ms per iteration (MAX over ranks), throughput, and every rank's in-kernel peer wait (csrc/p2p_ll.hpp
ll_wait_report / csrc/p2p.hip; pufferlib_amd.dist.wait_stats) split into the optimizer steps' exchanges and the two small all-reduces
at the end of evaluate().  The iteration has no slack to hide a late rank in (every exchange is a barrier among the ranks), so the
injected time shows up (a) as wait on the ranks that were on time and (b) one-to-one in the iteration time: the budget for >= 6x at
8 ranks (t_8 <= 8/6 t_1) is the figure this prints as `budget_us_per_step`.

    python tools/dp_jitter.py --world 2 --envs 4096 --horizon 128 --iters 20 --skews 0,50,100,200,500 --out gpurun_out/r06_dp_jitter.json
"""
import argparse
import json
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

HP = dict(learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01)
EPOCHS, NMB, BPTT = 4, 4, 16


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


def _hold(us):
    """Busy wait (time.sleep's granularity is of the order of the skews injected here)."""
    if us <= 0:
        return
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


def worker(rank, world, port, envs, horizon, iters, warmup, skews, late_rank, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY='0', PFA_ALLREDUCE='p2p')
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pufferlib_amd import clean_pufferl, cleanrl, models, namespace, vector
    from pufferlib_amd import dist as pdist
    aff = pdist.pin_rank(rank, world, device_index=rank % torch.cuda.device_count())
    torch.manual_seed(5)
    vec = vector.make(vector.make_squared, num_envs=envs, backend=vector.Squared, obs_stride=64)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    B = envs * horizon
    total_iters = (iters + warmup) * len(skews) + 24
    cfg = namespace(env='squared', seed=1, torch_deterministic=True, device='cuda', total_timesteps=B * world * total_iters * 2, anneal_lr=True,
                    update_epochs=EPOCHS, norm_adv=True, clip_vloss=True, target_kl=None, batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT,
                    checkpoint_interval=0, data_dir='/tmp/pfa_jitter', exp_id='jitter', **HP)
    data = clean_pufferl.create(cfg, vec, pol)
    assert data.native_dp, 'the peer path did not come up'
    legs = []
    for leg_no, skew in enumerate(skews):
        for _ in range(warmup + (12 if leg_no == 0 else 0)):     # (the first iterations after create() run slower: clocks, allocator, first-touch)
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        pdist.wait_stats(reset=True)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            if rank == late_rank:
                _hold(skew)
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ws = pdist.wait_stats(reset=True)
        t = torch.zeros(world, 4, dtype=torch.float64)
        t[rank] = torch.tensor([dt, ws['grad_exchange_wait_us'], ws['small_exchange_wait_us'], float(ws['grad_exchange_workgroups'])], dtype=torch.float64)
        dist.all_reduce(t)
        rows = t.tolist()
        dmax = max(r[0] for r in rows)
        legs.append(dict(injected_skew_us=skew, late_rank=late_rank, ms_per_step=dmax / iters * 1e3, value=world * B * iters / dmax,
                         rank_ms_per_step=[round(r[0] / iters * 1e3, 4) for r in rows],
                         grad_exchange_wait_us=[round(r[1], 2) for r in rows], small_exchange_wait_us=[round(r[2], 2) for r in rows],
                         wait_us_per_step=[round(r[1] * EPOCHS * NMB + r[2] * 2, 1) for r in rows]))
    # replicas identical after all of it (the late rank changes when things happen, never what is computed)
    chk = torch.zeros(world, 2, dtype=torch.float64)
    flat = data.flat_params.flat.double()
    chk[rank] = torch.tensor([float(flat.sum()), float((flat * flat).sum())], dtype=torch.float64)
    dist.all_reduce(chk)
    same = bool((chk == chk[0]).all()) and bool(torch.isfinite(chk).all())
    from pufferlib_amd import _lib
    status = int(_lib.lib().pfa_p2p_status())
    if rank == 0:
        base = legs[0]['ms_per_step']
        for leg in legs:
            leg['added_ms_per_step'] = round(leg['ms_per_step'] - base, 4)
        out = dict(world=world, envs_per_rank=envs, horizon=horizon, iters=iters, devices_visible=torch.cuda.device_count(),
                   ranks_share_devices=torch.cuda.device_count() < world, affinity_rank0=aff, replicas_identical=same, p2p_status=status,
                   # >= 6x weak scaling at 8 ranks: t_8 <= 8/6 t_1, so everything data parallelism adds per step — exchanges, waits,
                   # rank skew — has t_1 / 3 to fit in; t_1 here = this box's single-rank figure if given, else the zero-skew leg
                   legs=legs)
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, 'w') as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out))
    dist.barrier()
    pdist.finalize_native()
    dist.destroy_process_group()


def run(world=2, envs=4096, horizon=128, iters=20, warmup=3, skews=(0, 50, 100, 200, 500), late_rank=None, out='gpurun_out/dp_jitter.json',
        timeout_s=900):
    import torch.multiprocessing as mp
    late_rank = world - 1 if late_rank is None else late_rank
    ctx = mp.spawn(worker, args=(world, _free_port(), envs, horizon, iters, warmup, tuple(skews), late_rank, out), nprocs=world, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > timeout_s:
            for p in ctx.processes:
                p.kill()
            raise RuntimeError(f'ranks still running after {timeout_s} s')
    with open(out) as f:
        return json.load(f)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--horizon', type=int, default=128)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--skews', default='0,50,100,200,500')
    ap.add_argument('--out', default='gpurun_out/dp_jitter.json')
    a = ap.parse_args()
    run(a.world, a.envs, a.horizon, a.iters, a.warmup, [int(x) for x in a.skews.split(',')], out=a.out)
