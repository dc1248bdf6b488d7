"""The reference's UNMODIFIED demo.py training loop (demo.py:153-201; loop :189-192; config merge :22-99; make_policy :101-109)
END TO END on the device engine: `python -m pufferlib_amd.demo --reference <stage> -- --env squared --mode train --vec serial`.

Needs a GPU *and* the reference's files.  The GPU box has no /root/reference, so the files arrive as the git-ignored staging
directory `_refstage/` that tools/gpu_jobs/with_reference.sh ships with one gpurun job (tools/stage_reference.py) and removes
afterwards; without it (the driver's round-end run) the tests skip.  The recorded outcome of the staged run is
profiles/r06_demo_end_to_end.json.

demo.py swallows exceptions (`except Exception: print; os._exit(0)`, demo.py:196-198), so the exit code proves nothing: the
driver below records what the loop did and the test asserts on the record.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(REPO, '_refstage')

DRIVER = r'''
import json, sys
sys.dont_write_bytecode = True
from pufferlib_amd import demo, clean_pufferl as ours, readback
record = dict(iterations=[], created=None, closed=False)
OUT = sys.argv[2]            # (demo.main replaces sys.argv with demo.py's own)
_create, _train, _close = ours.create, ours.train, ours.close

def create(config, vecenv, policy, *a, **k):
    data = _create(config, vecenv, policy, *a, **k)
    import clean_pufferl, pufferlib.vector
    record['created'] = dict(trainer_module=sys.modules['clean_pufferl'].__name__, trainer_file=clean_pufferl.__file__,
                             vecenv=type(vecenv).__module__ + '.' + type(vecenv).__name__, num_envs=int(vecenv.num_envs),
                             policy=type(policy).__module__ + '.' + type(policy).__name__,
                             inner=type(getattr(policy, 'policy', policy)).__module__ + '.' + type(getattr(policy, 'policy', policy)).__name__,
                             device=str(config.device), batch_size=int(config.batch_size), minibatch_size=int(config.minibatch_size),
                             bptt_horizon=int(config.bptt_horizon), learning_rate=float(config.learning_rate),
                             total_timesteps=int(config.total_timesteps), serial_is_factory=hasattr(pufferlib.vector.Serial, 'host_backend'))
    print('[driver] created', record['created'], file=sys.stderr, flush=True)
    return data

def train(data):
    stats = dict(readback.materialize(data.stats))
    r = _train(data)
    if len(record['iterations']) < 3:
        print('[driver] train', len(record['iterations']), int(data.global_step), stats, file=sys.stderr, flush=True)
    record['iterations'].append(dict(global_step=int(data.global_step), epoch=int(data.epoch),
                                     score=stats.get('score'), episode_return=stats.get('episode_return'),
                                     episode_length=stats.get('episode_length'),
                                     policy_loss=float(data.losses.policy_loss), value_loss=float(data.losses.value_loss),
                                     entropy=float(data.losses.entropy)))
    return r

def close(data):
    record['closed'] = True
    record['final_global_step'] = int(data.global_step)
    r = _close(data)
    json.dump(record, open(OUT, 'w'))
    return r

def loud(fn):            # demo.py:196-198 prints a swallowed exception through rich and os._exit(0)s; say it on stderr first
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except BaseException:
            import traceback
            print('[driver] exception inside', fn.__name__, 'after', len(record['iterations']), 'iterations', file=sys.stderr, flush=True)
            traceback.print_exc(file=sys.stderr)
            sys.stderr.flush()
            raise
    wrapped.__name__ = fn.__name__
    return wrapped

ours.create, ours.train, ours.close, ours.evaluate = loud(create), loud(train), loud(close), loud(ours.evaluate)
demo.main(['--reference', sys.argv[1], '--'] + sys.argv[3:])
'''


def run_demo(tmp_path, extra, timeout=900):
    out = tmp_path / 'record.json'
    if not (tmp_path / 'config.yaml').exists():          # demo.py:22 opens config.yaml relative to the cwd
        import shutil
        shutil.copy(os.path.join(STAGE, 'config.yaml'), tmp_path / 'config.yaml')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([REPO, os.path.join(REPO, 'tests', 'shims')]), PYTHONDONTWRITEBYTECODE='1',
               PYTHONUNBUFFERED='1')      # demo.py prints a swallowed exception and os._exit(0)s: unbuffered, or the traceback is lost
    r = subprocess.run([sys.executable, '-X', 'faulthandler', '-c', DRIVER, STAGE, str(out)] + extra, cwd=tmp_path, env=env, capture_output=True, text=True,
                       timeout=timeout)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'r06_demo_process_output.txt'), 'a') as f:   # (pytest elides long assertion messages)
        f.write(f'==== {extra}\n---- returncode {r.returncode}\n---- stdout\n{r.stdout[-20000:]}\n---- stderr\n{r.stderr[-20000:]}\n')
    assert out.exists(), ('demo.py did not reach clean_pufferl.close()', r.stdout[-1500:], r.stderr[-1500:])
    return json.load(open(out)), r


needs_stage = pytest.mark.skipif(not os.path.exists(os.path.join(STAGE, 'demo.py')),
                                 reason='needs the staged reference (tools/gpu_jobs/with_reference.sh)')


@needs_stage
def test_unmodified_demo_py_trains_squared_end_to_end_on_the_device_engine(tmp_path):
    # ocean's config section: use_rnn True (LSTMWrapper + RecurrentPolicy of the REFERENCE's classes), 8 envs, batch 1024,
    # minibatch 128, bptt 4, lr 0.017, device cpu -> cuda on the command line (config.yaml:498-509)
    rec, r = run_demo(tmp_path, ['--env', 'squared', '--mode', 'train', '--vec', 'serial', '--train.device', 'cuda',
                                 '--train.total-timesteps', '200000'])
    c = rec['created']
    assert c['trainer_module'] == 'pufferlib_amd.clean_pufferl' and c['serial_is_factory'], c
    assert c['vecenv'] == 'pufferlib_amd.vector.Squared' and c['num_envs'] == 8, c
    assert c['policy'] == 'pufferlib.frameworks.cleanrl.RecurrentPolicy' and c['inner'] == 'pufferlib.models.LSTMWrapper', c
    assert (c['batch_size'], c['minibatch_size'], c['bptt_horizon']) == (1024, 128, 4) and abs(c['learning_rate'] - 0.017) < 1e-12, c
    its = rec['iterations']
    assert rec['closed'] and rec['final_global_step'] == 1024 * (len(its) + 1)   # demo.py:200: one more evaluate() after the loop
    assert len(its) == -(-200000 // 1024), len(its)                       # demo.py:189: while global_step < total_timesteps
    assert [i['global_step'] for i in its] == [1024 * (k + 1) for k in range(len(its))]
    assert all(all(v == v for v in (i['policy_loss'], i['value_loss'], i['entropy'])) for i in its)
    scores = [i['score'] for i in its if i['score'] is not None]
    first, last = scores[:10], scores[-10:]
    assert sum(last) / len(last) > sum(first) / len(first) + 0.3, (first, last)   # the policy learned to reach the target
    rets = [i['episode_return'] for i in its if i['episode_return'] is not None]
    assert sum(rets[-10:]) / 10 > sum(rets[:10]) / 10
    ckpts = [p for p in os.listdir(tmp_path / 'experiments' / os.listdir(tmp_path / 'experiments')[0])]
    assert 'trainer_state.pt' in ckpts and any(p.startswith('model_') for p in ckpts), ckpts   # close() -> save_checkpoint
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(dict(command='python -m pufferlib_amd.demo --reference _refstage -- --env squared --mode train --vec serial '
                           '--train.device cuda --train.total-timesteps 200000', created=c, iterations=len(its),
                   final_global_step=rec['final_global_step'], score_first10=first, score_last10=last,
                   episode_return_first10=rets[:10], episode_return_last10=rets[-10:], checkpoints=sorted(ckpts),
                   stdout_tail=r.stdout[-600:]),
              open(os.path.join(REPO, 'gpurun_out', 'r06_demo_end_to_end.json'), 'w'), indent=1)


@needs_stage
def test_unmodified_demo_py_with_the_mlp_policy_and_4096_envs(tmp_path):
    # BASELINE configs[1] through demo.py: the yaml's use_rnn cannot be switched off from the command line, so the
    # run uses a cwd config.yaml whose ocean section says use_rnn: False — demo.py:22 reads config.yaml from the cwd
    import yaml
    cfg = yaml.safe_load(open(os.path.join(STAGE, 'config.yaml')))
    cfg['ocean']['use_rnn'] = False
    yaml.safe_dump(cfg, open(tmp_path / 'config.yaml', 'w'))
    rec, r = run_demo(tmp_path, ['--env', 'squared', '--mode', 'train', '--vec', 'serial', '--train.device', 'cuda',
                                 '--train.num-envs', '4096', '--train.env-batch-size', '4096', '--train.batch-size', '524288',
                                 '--train.minibatch-size', '131072', '--train.bptt-horizon', '16', '--train.learning-rate', '0.00025',
                                 '--train.total-timesteps', '10485760'])
    c = rec['created']
    assert c['policy'] == 'pufferlib.frameworks.cleanrl.Policy' and c['inner'] == 'pufferlib.models.Default', c
    assert c['vecenv'] == 'pufferlib_amd.vector.Squared' and c['num_envs'] == 4096
    its = rec['iterations']
    assert len(its) == 20 and rec['final_global_step'] == 21 * 524288      # demo.py:200: one more evaluate() after the loop
    scores = [i['score'] for i in its if i['score'] is not None]
    assert scores[-1] > scores[0]
