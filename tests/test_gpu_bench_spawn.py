"""`python bench.py --gpus N` as a PLAIN command (the form the driver's scaling run uses): the script spawns its own N ranks under
torch.distributed.run.  The test box has one GPU, so the two ranks share it over gloo (bench.py picks that itself and says so in
the line): a functional run of the N-rank path end to end, not a scaling number."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, timeout=600):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_plain_python_invocation_spawns_its_ranks():
    d = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-breakdown', '--no-extra', '--sustained-seconds', '0.2'])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 2 * 4096 * 128 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and d['sustained_value'] > 0
    dist = d['dist']
    assert dist['ranks_share_devices'] is True and dist['backend'] == 'gloo'
    # two ranks on one device: RCCL refuses ("duplicate GPU"), the peer path opens (IPC between processes) and carries the bucket
    assert dist['transport']['grad_bucket'] in ('p2p', 'torch') and dist['p2p_status'] in (0, -1)
    if dist['transport']['grad_bucket'] == 'p2p':
        assert dist['p2p_selftest_passed'] is True and dist['allreduce_calls']['p2p'] > 0
        # the optimizer steps' exchanges ran inside the reduce + Adam launch (flag-in-data), and the per-transport A/B is in the line
        assert dist['allreduce_calls']['p2p_flag_in_data'] > 0
        assert set(dist['transports']) >= {'p2p_fused', 'p2p_launch'} and all(t['value'] > 0 for t in dist['transports'].values())
        assert dist['transports']['p2p_fused']['launches_per_optimizer_step'] == 2
    assert dist['collectives_per_step']['gradient_exchanges'] == 16 and dist['collectives_per_step']['small_all_reduces'] == 2
    assert len(d['rank_ms_per_step']['per_rank']) == 2 and d['rank_ms_per_step']['max'] >= d['rank_ms_per_step']['min'] > 0


def test_single_gpu_line_is_unchanged_in_shape():
    d = _run(['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extra', '--sustained-seconds', '0.2'])
    assert d['n_gpus'] == 1 and 'dist' not in d
    assert d['roofline']['kernel'] == 'ppo_mlp_grad' and 0 < d['roofline']['frac'] < 1 and d['roofline']['frac_executed'] < d['roofline']['frac']
    assert d['roofline_hbm']['bound'] == 'hbm' and 0 < d['roofline_hbm']['frac'] < 1
    # traffic is quoted only from a PMC summary stamped with this build: either it is there with its source, or it is null with the reason
    r = d['roofline']
    assert r['traffic_build'] and ((r['traffic'] is not None and r['traffic_source']) or (r['traffic'] is None and r['traffic_source'] is None and r['traffic_note']))
    assert set(d['profile_ms_per_step']) >= {'eval_time', 'env_time', 'eval_forward_time', 'eval_misc_time', 'train_time', 'train_forward_time', 'learn_time', 'train_misc_time'}
