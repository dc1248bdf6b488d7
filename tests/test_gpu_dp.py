"""Data parallel on real kernels: two ranks share the one GPU of the test box and talk through gloo
(device tensors, torch.distributed fallback path: PFA_NATIVE_RCCL=0), each owning a shard of the envs.  The contract:
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
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loop(n_local, world, recurrent, inject=None):
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    from test_gpu_ppo import _config
    torch.manual_seed(5)
    vec = vector.make(vector.make_squared, num_envs=n_local, backend=vector.Squared)
    base = models.Default(vec.driver_env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, base)) if recurrent else cleanrl.Policy(base)
    B = n_local * HORIZON
    data = clean_pufferl.create(_config(n_local, HORIZON, B // 2, 8, 2, N_PER_RANK * 2 * HORIZON * 8, HP, seed=21), vec, pol)
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


def _worker(rank, world, port, recurrent, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', PFA_NATIVE_RCCL='0')
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    sys.path.insert(0, os.path.dirname(__file__))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = _loop(N_PER_RANK, world, recurrent)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('recurrent', [False, True])
def test_two_ranks_on_one_gpu_equal_single_process_run(tmp_path, recurrent):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), recurrent, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f'rank{q}.npz') for q in range(world)]
    names = ('obs', 'actions', 'logprobs', 'values', 'rewards', 'dones')
    inject = [{n: np.concatenate([r[q][f'{it}.{n}'] for q in range(world)]) for n in names} for it in range(ITERS)]
    single = _loop(N_PER_RANK * world, 1, recurrent, inject=inject)
    rows = N_PER_RANK * HORIZON
    for it in range(ITERS):
        for q in range(world):      # one flat GAE scan across the shard boundary
            np.testing.assert_allclose(r[q][f'{it}.advantages'], single[f'{it}.advantages'][q * rows:(q + 1) * rows],
                                       rtol=1e-5, atol=2e-6)
        assert np.array_equal(r[0][f'{it}.flat'], r[1][f'{it}.flat'])                  # replicas stay bit-identical
        np.testing.assert_allclose(r[0][f'{it}.flat'], single[f'{it}.flat'], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(r[0][f'{it}.losses'], single[f'{it}.losses'], rtol=1e-5, atol=1e-5)
        assert np.array_equal(r[0][f'{it}.stats'], r[1][f'{it}.stats'])                # all-reduced episode stats
        assert r[0][f'{it}.global_step'][0] == single[f'{it}.global_step'][0] == (it + 1) * world * rows
    # the shards really are different envs (seeds seed + r*N + i), not replicas of each other
    assert not np.array_equal(r[0]['0.obs'], r[1]['0.obs'])
