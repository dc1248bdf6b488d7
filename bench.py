"""bench.py — end-to-end PPO throughput of the device-resident engine on BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is ONE pass of the hot path over one batch: clean_pufferl.evaluate() (128-step rollout of 4096 Squared
envs per GPU: env step + MLP forward + sample + store) followed by clean_pufferl.train() (GAE + 4 epochs x 4
minibatches of fused fwd/loss/bwd + clip + Adam) = 524 288 env steps per GPU.  Workload = BASELINE.json
configs[1] ("squared env, 4096 envs, 64-dim flat obs, MLP policy, 1xMI355X"); with N > 1 every rank runs that
workload on its own env shard (weak scaling, configs[4] at N = 8) and all-reduces one flat gradient bucket per
optimizer step over RCCL.  Inputs are synthetic by construction (the env IS the generator) and resident in HBM.

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     dominant kernel ppo_mlp_grad (fp32 MFMA bound): algorithmic FLOPs per launch / average launch
               duration measured with HIP events on the launch stream over the timed region (every 17th launch
               bracketed: the event packets serialise the queue)
  cpu_baseline kind "reference": the UNMODIFIED reference's Serial CPU path timed on this box model by tools/gpu_jobs/
               with_reference.sh (profiles/r06_reference_cpu_on_gpu_box.json; used when the box fingerprint matches), with the
               live timing of the CPU oracle port (C env + torch-fp32 policy/update, oracle/) on a bounded sample of the same
               workload on this box's host cores riding along (`port_live`; it IS the baseline when no record matches)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

NUM_ENVS = 4096          # per GPU
HORIZON = 128
NMB = 4
EPOCHS = 4
BPTT = 16
SUSTAINED_CAP = 20000    # most steps the sustained leg may run: total_timesteps (the lr-anneal horizon) is sized to hold them
D, NT = 3, 1             # obs 7x7 = 49 floats, padded to a 64-float row (256 B)
# SURVEY.md §8d algorithmic figures (MLP, obs row 64 f32, 8 actions, hidden 128)
FLOP_PER_ROW_UPDATE = 39680          # fwd 18 688 + bwd 20 992 per row per epoch (the 64-float padded row)
FLOP_PER_ROW_USEFUL = 2 * (49 * 128 + 128 * 9) + 2 * 49 * 128 + 2 * 2 * 128 * 9   # the same on the 49 real columns: 32 000
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity headline figure is not a roof)
# NatureCNN (workload c4; SURVEY 8d: ~18.7 MFLOP per forward sample), products of the rows-form kernel per frame:
#   forward  conv1 2*400*256*32 + conv2 2*81*512*64 + conv3 2*49*576*64 + fc 2*3136*512          = 18 685 952
#   dX       fc 2*3136*512 + conv3 (= its forward) + conv2 (= its forward); conv1 needs none      = 12 132 352
# and of the weight-form kernel: dW of the four layers (= the forward products) + the heads' 2*512*16
CNN_FWD_FLOP = 2 * (400 * 256 * 32 + 81 * 512 * 64 + 49 * 576 * 64 + 3136 * 512)
CNN_DX_FLOP = 2 * (3136 * 512 + 49 * 576 * 64 + 81 * 512 * 64)
CNN_DW_FLOP = CNN_FWD_FLOP + 2 * 512 * 16


def make_config(total_timesteps, env='squared'):
    from pufferlib_amd import namespace
    B = NUM_ENVS * HORIZON
    return namespace(env=env, seed=1, torch_deterministic=True, device='cuda', total_timesteps=total_timesteps,
                     learning_rate=2.5e-4, anneal_lr=True, gamma=0.99, gae_lambda=0.95, update_epochs=EPOCHS,
                     norm_adv=True, clip_coef=0.1, clip_vloss=True, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5,
                     ent_coef=0.01, target_kl=None, batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT,
                     checkpoint_interval=0, data_dir='/tmp/pfa_bench', exp_id='bench')


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _oracle_trainer(n, seed=1):
    import numpy as np
    import torch
    from oracle import c_oracle, ppo_torch
    B = n * HORIZON
    vec = c_oracle.SquaredSerial(n, D, NT)
    torch.manual_seed(1)
    w = {'encoder.weight': torch.randn(128, 49) * 0.1, 'encoder.bias': torch.zeros(128),
         'decoder.weight': torch.randn(8, 128) * 0.01, 'decoder.bias': torch.zeros(8),
         'value_head.weight': torch.randn(1, 128) * 0.1, 'value_head.bias': torch.zeros(1)}
    pol = ppo_torch.Policy({k: v.numpy() for k, v in w.items()})
    tr = ppo_torch.Trainer(pol, vec, batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT, update_epochs=EPOCHS,
                           learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5,
                           vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01, total_timesteps=B * 1000, seed=seed)
    return tr, np.random.default_rng(0)


def reference_cpu_baseline(config_key, port):
    """cpu_baseline with kind = "reference": the UNMODIFIED reference's own CPU path (clean_pufferl + pufferlib.vector.Serial + c_gae.pyx),
    timed on a box of THIS model by tools/gpu_jobs/with_reference.sh and recorded in profiles/r06_reference_cpu_on_gpu_box.json
    (the reference cannot be read at bench time: /root/reference does not exist on the GPU box; the job ships it once in a git-ignored
    staging directory).  Used only when the record's box fingerprint (CPU model + logical core count) is this box's; otherwise, and
    when the file is absent, the live port timing `port` is returned unchanged.  The live port timing of this run rides along."""
    path = os.path.join(REPO, 'profiles', 'r06_reference_cpu_on_gpu_box.json')
    try:
        doc = json.load(open(path))
        box, s = doc['box'], doc['summary'][config_key]
    except Exception:
        return port
    import torch
    if box.get('cpu_model') != _cpu_model() or box.get('cores_logical') != (os.cpu_count() or 1) or box.get('torch') != torch.__version__:
        return dict(port, reference_record_refused=f"{os.path.relpath(path, REPO)} was taken on {box.get('cpu_model')} x{box.get('cores_logical')} "
                                                   f"with torch {box.get('torch')}, this box is {_cpu_model()} x{os.cpu_count()} with torch {torch.__version__}")
    ser, best = s['serial'], s['best']
    return dict(value=ser['value'], unit='env_steps/s', cores=max(int(ser['torch_threads']), int(ser['cores_used'])), cores_available=box['cores_logical'],
                cores_physical=box['cores_physical'], cpu_model=box['cpu_model'], kind='reference',
                sample=f"{ser['iterations']} evaluate+train iteration(s) of {ser['envs']} envs x {ser['horizon']} steps after 1 warm-up: {ser['what']}; "
                       f"env stepping on 1 core (Serial), torch on {ser['torch_threads']} threads",
                measured=f"recorded {box.get('date')} on a box of this model (fingerprint matched: CPU model + logical cores + torch version), not re-timed in "
                         'this run; the live port timing of THIS run is `port_live`',
                source=os.path.relpath(path, REPO) + ' (tools/gpu_jobs/with_reference.sh -> tools/time_reference.py)',
                breakdown_s_per_iter=dict(evaluate=ser['evaluate_s_per_iter'], train=ser['train_s_per_iter'], **ser['profile']),
                best_vectoriser=dict(value=best['value'], backend=best['backend'], workers=best['workers'], torch_threads=best['torch_threads']),
                port_live={k: port[k] for k in ('value', 'unit', 'cores', 'kind', 'sample') if k in port})


def cpu_baseline(budget_s=24.0):
    """The CPU oracle port (oracle/: C restatement of Serial(Squared) on 1 core + the torch-fp32 restatement of the policy and
    of clean_pufferl.train) timed on the SAME configuration the GPU number is quoted on — 4096 envs x 128 steps, 4 minibatches
    x 4 epochs — for 1 warm-up + as many evaluate+train iterations as fit the budget (at least 1).  The unmodified reference's own
    timing on this box model is reference_cpu_baseline() above; this live port timing rides along with it."""
    import torch
    cores_avail = os.cpu_count() or 1
    # threads actually used: the reference's torch CPU path scales poorly past a socket's worth of cores on these tiny GEMMs
    # (256 threads measured 800x slower than 16 on the GPU box), so cap and report the cap
    cores = min(cores_avail, 16)
    torch.set_num_threads(cores)
    n = NUM_ENVS
    B = n * HORIZON
    tr, rng = _oracle_trainer(n)

    def one():
        tr.evaluate(rng.exponential(size=(HORIZON, n, 8)).astype('float32'))
        tr.train()

    t0 = time.perf_counter()
    one()  # warm-up
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    iters = 0
    while True:
        one()
        iters += 1
        if time.perf_counter() - t0 + warm > budget_s or iters >= 50:
            break
    dt = time.perf_counter() - t0
    return dict(value=iters * B / dt, unit='env_steps/s', cores=cores, cores_available=cores_avail, cpu_model=_cpu_model(), kind='port',
                sample=f'{iters} evaluate+train iteration(s) of {n} envs x {HORIZON} steps (batch {B}, {NMB} minibatches x {EPOCHS} '
                       f'epochs: the bench configuration itself) after 1 warm-up; C env on 1 core, torch-fp32 on {cores} threads')


def cpu_baseline_c4(n=32, horizon=16):
    """The oracle port of the conv-policy path (torch-fp32 ConvPolicy + the restated trainer) on a bounded sample of the c4
    workload's shape: `n` envs x `horizon` steps of uint8 (4, 84, 84) frames (numpy generator), 4 minibatches x 4 epochs."""
    import numpy as np
    import torch
    from oracle import ppo_torch
    cores_avail = os.cpu_count() or 1
    cores = min(cores_avail, 16)
    torch.set_num_threads(cores)
    B = n * horizon
    rs = np.random.RandomState(0)

    class FrameVec:
        num_envs = n
        observations = np.zeros((n, 4, 84, 84), np.uint8)

        def async_reset(self, seed):
            pass

        def recv(self):
            self.observations[:] = rs.randint(0, 256, self.observations.shape, dtype=np.uint8)
            return (self.observations, rs.randint(0, 2, n).astype(np.float32), np.zeros(n, bool), np.zeros(n, bool), [], np.arange(n), np.ones(n, bool))

        def send(self, actions):
            pass

    shapes = {'network.0.weight': (32, 4, 8, 8), 'network.0.bias': (32,), 'network.2.weight': (64, 32, 4, 4), 'network.2.bias': (64,),
              'network.4.weight': (64, 64, 3, 3), 'network.4.bias': (64,), 'network.7.weight': (512, 3136), 'network.7.bias': (512,),
              'actor.weight': (4, 512), 'actor.bias': (4,), 'value_fn.weight': (1, 512), 'value_fn.bias': (1,)}
    w = {k: (rs.standard_normal(sh) * (0.01 if len(sh) == 1 else 1.0 / np.sqrt(np.prod(sh[1:])))).astype(np.float32) for k, sh in shapes.items()}
    tr = ppo_torch.Trainer(ppo_torch.ConvPolicy(w), FrameVec(), batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT, update_epochs=EPOCHS,
                           learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5,
                           ent_coef=0.01, total_timesteps=B * 1000, seed=1)

    def one():
        tr.evaluate(rs.exponential(size=(horizon, n, 4)).astype('float32'))
        tr.train()

    one()
    t0 = time.perf_counter()
    iters = 0
    while True:
        one()
        iters += 1
        if time.perf_counter() - t0 > 15.0 or iters >= 20:
            break
    dt = time.perf_counter() - t0
    return dict(value=iters * B / dt, unit='env_steps/s', cores=cores, cores_available=cores_avail, cpu_model=_cpu_model(), kind='port',
                sample=f'{iters} evaluate+train iteration(s) of {n} envs x {horizon} steps of uint8 (4,84,84) frames (batch {B}, {NMB} minibatches x '
                       f'{EPOCHS} epochs) after 1 warm-up; torch-fp32 NatureCNN on {cores} threads')


def cpu_baseline_c3(n=256, horizon=128):
    """The oracle port of the recurrent path (torch-fp32 Default(128) + nn.LSTM(128, 128) restatement + the restated trainer with
    its bptt segments) on a bounded sample of the c3 workload's shape — BASELINE.md section 3's "C3-policy": `n` envs x `horizon`
    steps of 160-float rows (numpy generator, 100-step episodes), 7 actions, 4 minibatches x 4 epochs, bptt 16."""
    import numpy as np
    import torch
    from oracle import ppo_torch
    cores_avail = os.cpu_count() or 1
    cores = min(cores_avail, 16)
    torch.set_num_threads(cores)
    B = n * horizon
    rs = np.random.RandomState(0)

    class RowVec:
        num_envs = n
        observations = np.zeros((n, 160), np.float32)
        tick = np.zeros(n, np.int64)

        def async_reset(self, seed):
            pass

        def recv(self):
            self.observations[:] = rs.randint(0, 11, self.observations.shape).astype(np.float32)
            self.tick += 1
            done = self.tick % 100 == 0
            return (self.observations, rs.randint(0, 2, n).astype(np.float32), done, np.zeros(n, bool), [], np.arange(n), np.ones(n, bool))

        def send(self, actions):
            pass

    shapes = {'encoder.weight': (128, 160), 'encoder.bias': (128,), 'decoder.weight': (7, 128), 'decoder.bias': (7,), 'value_head.weight': (1, 128),
              'value_head.bias': (1,), 'weight_ih_l0': (512, 128), 'weight_hh_l0': (512, 128), 'bias_ih_l0': (512,), 'bias_hh_l0': (512,)}
    w = {k: (rs.standard_normal(sh) * (0.01 if len(sh) == 1 else 1.0 / np.sqrt(sh[-1]))).astype(np.float32) for k, sh in shapes.items()}
    tr = ppo_torch.Trainer(ppo_torch.Policy(w, recurrent=True), RowVec(), batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT, update_epochs=EPOCHS,
                           learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5,
                           ent_coef=0.01, total_timesteps=B * 1000, seed=1)

    def one():
        tr.evaluate(rs.exponential(size=(horizon, n, 7)).astype('float32'))
        tr.train()

    one()
    t0 = time.perf_counter()
    iters = 0
    while True:
        one()
        iters += 1
        if time.perf_counter() - t0 > 15.0 or iters >= 20:
            break
    dt = time.perf_counter() - t0
    return dict(value=iters * B / dt, unit='env_steps/s', cores=cores, cores_available=cores_avail, cpu_model=_cpu_model(), kind='port',
                sample=f'{iters} evaluate+train iteration(s) of {n} envs x {horizon} steps of 160-float rows (batch {B}, {NMB} minibatches x '
                       f'{EPOCHS} epochs, bptt {BPTT}) after 1 warm-up; numpy row generator, torch-fp32 MLP 128 + LSTM 128 on {cores} threads')


def self_check(data, pol):
    """One more (untimed) evaluate + train, the train replayed by the torch-fp32 oracle trainer on the device rollout's
    experience at the FULL bench size (one epoch = 4 optimizer steps over all 524 288 rows, to bound the host time): losses and
    post-update weights must agree within north_star's 1e-5.  The oracle is the checker here, never the thing timed."""
    import numpy as np
    import torch
    from oracle import c_oracle, ppo_torch
    from pufferlib_amd import clean_pufferl
    n, T = NUM_ENVS, HORIZON
    B = n * T
    clean_pufferl.evaluate(data)
    exp = data.experience
    w0 = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    opol = ppo_torch.Policy(w0)
    epochs_saved = data.config.update_epochs
    data.config.update_epochs = 1
    tr = ppo_torch.Trainer(opol, c_oracle.SquaredSerial(n, D, NT), batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT,
                           update_epochs=1, learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5,
                           vf_clip_coef=0.1, max_grad_norm=0.5, ent_coef=0.01, total_timesteps=data.config.total_timesteps, seed=1)
    sm = lambda x: x.view(n, T, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
    tr.obs = torch.as_tensor(sm(exp.obs)[:, :49].copy())
    tr.actions = sm(exp.actions).astype(np.int64)
    tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (exp.logprobs, exp.rewards, exp.dones, exp.values))
    tr.global_step = data.global_step
    tr.opt.param_groups[0]['lr'] = data.optimizer.param_groups[0]['lr']
    # the oracle starts from fresh Adam moments; give it the device's (the trainer has been running)
    m, v = data.flat_params.split(data.optimizer.exp_avg), data.flat_params.split(data.optimizer.exp_avg_sq)
    for i, name in enumerate(opol.names):
        p = opol.params[i]
        tr.opt.state[p] = dict(step=torch.tensor(float(data.optimizer.step_count)), exp_avg=m[name].detach().cpu().clone(),
                               exp_avg_sq=v[name].detach().cpu().clone())
    Lo = tr.train()
    clean_pufferl.train(data)
    data.config.update_epochs = epochs_saved
    keys = ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')
    loss_err = max(abs(float(getattr(data.losses, k)) - float(Lo[k])) for k in keys)
    sd = pol.state_dict()
    w_err = max(float(np.abs(sd['policy.' + k].cpu().numpy() - arr).max()) for k, arr in opol.state_arrays().items())
    adv_err = float(np.abs(exp.advantages.cpu().numpy() - c_oracle.compute_gae(
        exp.dones.cpu().numpy(), exp.values.cpu().numpy(), exp.rewards.cpu().numpy(), 0.99, 0.95)).max())
    assert loss_err <= 1e-5 and w_err <= 1e-5 and adv_err <= 1e-5, (loss_err, w_err, adv_err)
    return dict(rows=B, optimizer_steps=NMB, max_abs_loss_err=loss_err, max_abs_weight_err=w_err, max_abs_advantage_err=adv_err,
                tolerance=1e-5, checker='oracle/ppo_torch.py + oracle/puffer_oracle.c')


def reference_replay_check():
    """The headline configuration against the UNMODIFIED REFERENCE itself: one whole iteration of BASELINE
    configs[1] — 4096 envs x 128 steps, 4 minibatches x 4 epochs = 16 optimizer steps — recorded from /root/reference's
    clean_pufferl.create / evaluate / train by tests/golden/make_golden.py (`big`: tests/golden/ppo_c2_mlp.npz, digest form) and
    replayed HERE on the HIP path from the same initial weights: actions / observations / rewards / dones bit for bit, losses,
    advantages, values and the updated weights within north_star's 1e-5.  The multinomial noise is regenerated as the reference drew
    it (torch.manual_seed(1), one exponential_ per step; checked against the recorded digest) — a box whose torch draws other
    numbers replays the recorded actions instead.  Untimed; the fixture is data (it travels with the repository)."""
    import hashlib
    import numpy as np
    import torch
    from pufferlib_amd import clean_pufferl, cleanrl, models, namespace, vector
    path = os.path.join(REPO, 'tests', 'golden', 'ppo_c2_mlp.npz')
    if not os.path.exists(path):
        return dict(skipped='tests/golden/ppo_c2_mlp.npz not found')
    g = np.load(path)
    n, T, mbs, bptt, epochs, total, iters = (int(x) for x in g['config'])
    lr, gamma, lam, clip, vf_coef, vf_clip, mgn, ent = (float(x) for x in g['hparams'])
    B = n * T

    def digest(a, samples=64):
        f = np.asarray(a, np.float64).reshape(-1)
        idx = np.linspace(0, f.size - 1, min(samples, f.size)).astype(np.int64)
        return np.concatenate([[f.sum(), np.abs(f).sum()], f[idx]])
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
    sm = lambda x: x.view(n, T, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()   # noqa: E731
    vec = vector.make(vector.make_squared, num_envs=n, backend=vector.Squared)
    pol = cleanrl.Policy(models.Default(vec.driver_env))
    pol.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith('w0.')})
    cfg = namespace(env='squared', seed=1, torch_deterministic=True, device='cuda', total_timesteps=total, learning_rate=lr, anneal_lr=True,
                    gamma=gamma, gae_lambda=lam, update_epochs=epochs, norm_adv=True, clip_coef=clip, clip_vloss=True, vf_coef=vf_coef,
                    vf_clip_coef=vf_clip, max_grad_norm=mgn, ent_coef=ent, target_kl=None, batch_size=B, minibatch_size=mbs, bptt_horizon=bptt,
                    checkpoint_interval=0, data_dir='/tmp/pfa_bench', exp_id='replay')
    data = clean_pufferl.create(cfg, vec, pol)
    state = torch.get_rng_state()
    torch.manual_seed(1)
    nz = np.stack([torch.empty(n, 8).exponential_(1).numpy() for _ in range(T)])
    torch.set_rng_state(state)
    how = 'regenerated from torch.manual_seed(1)'
    if not np.array_equal(digest(nz), g['it0.noise_digest']):
        nz = np.ones((T, n, 8), np.float32)
        np.put_along_axis(nz, g['it0.actions'].reshape(T, n, 1).astype(np.int64), np.float32(1e-30), axis=2)
        how = 'recorded actions (this torch draws other exponentials than the recording box)'
    data.noise = torch.as_tensor(nz)
    stats, _ = clean_pufferl.evaluate(data)
    exp = data.experience
    exact = (np.array_equal(sm(exp.actions), g['it0.actions'].astype(np.int32))
             and sha(sm(exp.obs)[:, :49].astype(np.int8)) == str(g['it0.obs_sha'])
             and sha(sm(exp.rewards).astype(np.float32)) == str(g['it0.rewards_sha'])
             and sha(sm(exp.dones).astype(np.float32)) == str(g['it0.dones_sha'])
             and data.global_step == int(g['it0.global_step']))
    stats_err = float(np.max(np.abs(np.array([stats['episode_return'], stats['episode_length'], stats['score']]) - g['it0.stats'])))

    def derr(got, key):      # largest error over the digest's 64 samples, and of the mean
        d, want = digest(got), g[key]
        return max(float(np.abs(d[2:] - want[2:]).max()), abs(float(d[0] - want[0])) / np.asarray(got).size)
    val_err = max(derr(sm(exp.values), 'it0.values'), derr(sm(exp.logprobs), 'it0.logprobs'))
    clean_pufferl.train(data)
    idx = torch.stack([exp.minibatch_rows_index(m) for m in range(exp.num_minibatches)])
    adv_err = max(derr(exp.advantages[idx].cpu().numpy(), 'it0.advantages'), derr(exp.returns[idx].cpu().numpy(), 'it0.returns'))
    L = data.losses
    got = np.array([L.policy_loss, L.value_loss, L.entropy, L.old_approx_kl, L.approx_kl, L.clipfrac, L.explained_variance])
    loss_err = float(np.max(np.abs(got - g['it0.losses'])))
    w_err = max(derr(v.cpu().numpy(), f'it0.w.{k}') for k, v in pol.state_dict().items())
    assert exact, 'integer side differs from the reference recording'
    assert max(val_err, adv_err, loss_err, w_err) <= 1e-5 and stats_err <= 1e-9, (val_err, adv_err, loss_err, w_err, stats_err)
    return dict(fixture='tests/golden/ppo_c2_mlp.npz (the unmodified reference, tests/golden/make_golden.py big)', rows=B, optimizer_steps=epochs * (B // mbs),
                actions_obs_rewards_dones_bit_exact=bool(exact), action_noise=how, max_abs_logprob_value_err=val_err,
                max_abs_advantage_return_err=adv_err, max_abs_loss_err=loss_err, max_abs_weight_err=w_err, tolerance=1e-5)


class _StubVec:
    """What oracle.ppo_torch.Trainer asks of a vecenv when its experience is injected (never stepped)."""

    def __init__(self, n, shape, dtype):
        import numpy as np
        self.num_envs = n
        self.observations = np.zeros((n,) + tuple(shape), dtype)

    def async_reset(self, seed):
        pass


def _inject_adam(tr, opol, data, key_of):
    """The oracle trainer starts from fresh Adam moments; give it the device's (the trainer has been running)."""
    import torch
    m, v = data.flat_params.split(data.optimizer.exp_avg), data.flat_params.split(data.optimizer.exp_avg_sq)
    for i, name in enumerate(opol.names):
        p = opol.params[i]
        tr.opt.state[p] = dict(step=torch.tensor(float(data.optimizer.step_count)), exp_avg=m[key_of(name)].detach().cpu().to(p.dtype).clone(),
                               exp_avg_sq=v[key_of(name)].detach().cpu().to(p.dtype).clone())


def self_check_c3(data, pol):
    """The c3 leg at the size it is timed: one more (untimed) evaluate + train with ONE epoch
    (4 optimizer steps over all 524 288 rows of 160 floats, LSTM state carried over the minibatches), the train replayed by the
    torch-fp32 oracle trainer on the device rollout's experience: losses and post-update weights within north_star's 1e-5."""
    import numpy as np
    import torch
    from oracle import c_oracle, ppo_torch
    from pufferlib_amd import clean_pufferl
    n, T = NUM_ENVS, HORIZON
    B = n * T
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    clean_pufferl.evaluate(data)
    exp = data.experience
    opol = ppo_torch.Policy.from_reference_state_dict({k: v.detach().cpu().clone() for k, v in pol.state_dict().items()})
    epochs_saved = data.config.update_epochs
    data.config.update_epochs = 1
    tr = ppo_torch.Trainer(opol, _StubVec(n, (160,), np.float32), batch_size=B, minibatch_size=B // NMB, bptt_horizon=BPTT, update_epochs=1,
                           learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5,
                           ent_coef=0.01, total_timesteps=data.config.total_timesteps, seed=1)
    sm = lambda x: x.view(n, T, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
    tr.obs = torch.as_tensor(sm(exp.obs)[:, :160].copy())
    tr.actions = sm(exp.actions).astype(np.int64)
    tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (exp.logprobs, exp.rewards, exp.dones, exp.values))
    tr.global_step = data.global_step
    tr.opt.param_groups[0]['lr'] = data.optimizer.param_groups[0]['lr']
    _inject_adam(tr, opol, data, lambda name: ('recurrent.' + name) if name.endswith('_l0') else name)
    t0 = time.perf_counter()
    Lo = tr.train()
    host_s = time.perf_counter() - t0
    clean_pufferl.train(data)
    data.config.update_epochs = epochs_saved
    keys = ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')
    loss_err = max(abs(float(getattr(data.losses, k)) - float(Lo[k])) for k in keys)
    sd = pol.state_dict()
    w_err = max(float(np.abs(sd[('policy.recurrent.' + k) if k.endswith('_l0') else ('policy.policy.' + k)].cpu().numpy() - arr).max())
                for k, arr in opol.state_arrays().items())
    adv_err = float(np.abs(exp.advantages.cpu().numpy() - c_oracle.compute_gae(
        exp.dones.cpu().numpy(), exp.values.cpu().numpy(), exp.rewards.cpu().numpy(), 0.99, 0.95)).max())
    assert loss_err <= 1e-5 and w_err <= 1e-5 and adv_err <= 1e-5, (loss_err, w_err, adv_err)
    return dict(rows=B, optimizer_steps=NMB, max_abs_loss_err=loss_err, max_abs_weight_err=w_err, max_abs_advantage_err=adv_err,
                tolerance=1e-5, oracle_seconds=round(host_s, 1), checker='oracle/ppo_torch.py (torch fp32, LSTM state carried over the minibatches) + oracle/puffer_oracle.c')


def self_check_c4():
    """The c4 leg at the minibatch size it is timed: ONE optimizer step over a 65 536-frame minibatch
    — cnn.Engine walks it in eight 8192-frame chunks, as in the timed loop — forward + PPO loss + backward through all four layers
    + clip + Adam, against the oracle trainer in DOUBLE precision on the same uint8 frames (2048 envs x 32 steps of a fresh
    vector.Frames; the host time of the f64 replay is what bounds the size: ~1 minute).  fp32 torch convolutions on the host differ
    from double precision by as much as the kernels do, so the yardstick is f64."""
    import numpy as np
    import torch
    from oracle import ppo_torch
    from pufferlib_amd import clean_pufferl, cleanrl, models, vector
    n, T = 2048, 32
    B = n * T
    vec = vector.make(vector.make_frames, env_kwargs=dict(framestack=4, num_actions=4, episode_length=100), num_envs=n, backend=vector.Frames)
    torch.manual_seed(3)
    pol = cleanrl.Policy(models.Convolutional(vec.driver_env, framestack=4))
    cfg = make_config(B * 10, env='frames')
    cfg.batch_size, cfg.minibatch_size, cfg.update_epochs = B, B, 1
    data = clean_pufferl.create(cfg, vec, pol)
    assert data.cnn_engine.chunk == 8192
    start = {k[len('policy.'):]: v.detach().cpu().numpy().copy() for k, v in pol.state_dict().items()}
    clean_pufferl.evaluate(data)
    e = data.experience
    sm = lambda x: x.view(n, T, *x.shape[1:]).transpose(0, 1).reshape(B, *x.shape[1:]).cpu().numpy()  # noqa: E731
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    try:
        torch.set_default_dtype(torch.float64)
        opol = ppo_torch.ConvPolicy(start, dtype=torch.float64)
        tr = ppo_torch.Trainer(opol, _StubVec(n, (1,), np.uint8), batch_size=B, minibatch_size=B, bptt_horizon=BPTT, update_epochs=1,
                               learning_rate=2.5e-4, gamma=0.99, gae_lambda=0.95, clip_coef=0.1, vf_coef=0.5, vf_clip_coef=0.1, max_grad_norm=0.5,
                               ent_coef=0.01, total_timesteps=B * 10, seed=1)
        tr.obs = torch.as_tensor(sm(e.obs)).to(torch.float64)
        tr.actions = sm(e.actions).astype(np.int64)
        tr.logprobs, tr.rewards, tr.dones, tr.values = (sm(x).copy() for x in (e.logprobs, e.rewards, e.dones, e.values))
        tr.global_step = data.global_step
        Lo = tr.train()
    finally:
        torch.set_default_dtype(torch.float32)
    host_s = time.perf_counter() - t0
    clean_pufferl.train(data)
    keys = ('policy_loss', 'value_loss', 'entropy', 'old_approx_kl', 'approx_kl', 'clipfrac')
    loss_err = max(abs(float(getattr(data.losses, k)) - float(Lo[k])) for k in keys)
    sd = pol.state_dict()
    errs = {k: float(np.abs(sd['policy.' + k].cpu().numpy() - arr).max()) for k, arr in opol.state_arrays().items()}
    w_err = max(errs.values())
    assert loss_err <= 1e-5 and w_err <= 1e-5, (loss_err, errs)
    return dict(frames=B, chunks=B // 8192, optimizer_steps=1, max_abs_loss_err=loss_err, max_abs_weight_err=w_err,
                max_abs_weight_err_by_tensor={k: float('%.3g' % v) for k, v in errs.items()}, tolerance=1e-5, oracle_seconds=round(host_s, 1),
                checker='oracle/ppo_torch.py ConvPolicy in float64')


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


def self_spawn(args):
    """`python bench.py --gpus N` as a plain command (no torchrun): re-run this file as N ranks, one per GPU, under
    torch.distributed.run on 127.0.0.1 and hand its exit code back.  The torchrun form the contract names keeps working (it
    arrives with WORLD_SIZE set and never gets here).  On a box with fewer GPUs than ranks the ranks share devices and the
    process group is gloo (PFA_DIST_BACKEND=gloo): a functional run of the N-rank path, flagged `ranks_share_devices` in the
    JSON line — not a scaling measurement."""
    import subprocess
    import torch
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py: no GPU visible')
    if ndev < args.gpus and 'PFA_DIST_BACKEND' not in env:
        print(f'[bench] {args.gpus} ranks on {ndev} device(s): ranks share devices, process group gloo (not a scaling run)', file=sys.stderr)
        env['PFA_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def source_build_id():
    from pufferlib_amd import _lib
    return _lib.source_hash()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from profiles/pmc_summary.json — only when that file was generated (tools/make_pmc_summary.py,
    same gpurun job as the rocprofv3 --pmc passes) from the kernel sources this library was built from; a summary of another
    build is refused (traffic null, the reason in traffic_source) instead of silently going stale."""
    pmc = os.path.join(REPO, 'profiles', 'pmc_summary.json')
    if not os.path.exists(pmc):
        return None, 'no profiles/pmc_summary.json'
    try:
        doc = json.load(open(pmc))
    except Exception as e:
        return None, f'unreadable pmc_summary.json: {e}'
    have, want = doc.get('_build'), source_build_id()
    entry = doc.get(kernel, {})
    if have != want:
        return None, f'refused: pmc_summary.json was collected on kernel-source build {have}, this library is build {want}'
    if entry.get('carried_over'):
        return None, f'refused: the {kernel} entry of pmc_summary.json was carried over from an earlier build'
    return entry.get('hbm_bytes_per_launch'), f"{entry.get('source', 'profiles/pmc_summary.json')} [build {have}]"


def pmc_mfma_utilisation(kernel, waves_per_simd=2):
    """north_star's "MFMA utilisation" from the SQ counters of the same stamped summary: SQ_VALU_MFMA_BUSY_CYCLES over the SIMD
    cycles of the launch (SQ_WAVE_CYCLES counts quad-cycles per wave; `waves_per_simd` waves share a SIMD).  Measured UNDER the
    counters (the launch runs ~30 % longer then), so it is a floor of the undisturbed figure."""
    pmc = os.path.join(REPO, 'profiles', 'pmc_summary.json')
    try:
        doc = json.load(open(pmc))
        e = doc.get(kernel, {})
        if doc.get('_build') != source_build_id() or e.get('carried_over'):
            return None
        busy, wave = e.get('sq_valu_mfma_busy_cycles'), e.get('sq_wave_cycles')
        if not busy or not wave:
            return None
        return dict(sq_valu_mfma_busy_cycles=busy, simd_cycles=wave * 4.0 / waves_per_simd, frac=busy / (wave * 4.0 / waves_per_simd),
                    sq_insts_mfma=e.get('sq_insts_mfma'), sq_insts_valu=e.get('sq_insts_valu'), launch_us_under_pmc=e.get('dur_us_under_pmc'))
    except Exception:
        return None


def extra_workload(flags, timeout_s=600):
    """A short run of one of the side workloads (BASELINE configs[2] / configs[3]) as its own process; its JSON line, trimmed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--no-breakdown', '--no-extra'] + flags
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][-1]
        d = json.loads(line)
        return {k: d[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'config', 'roofline', 'cpu_baseline',
                                  'profile_ms_per_step', 'self_check', 'kernel_ms_per_step') if k in d}
    except Exception as e:  # a side workload never takes the headline line down with it
        return dict(flags=flags, error=f'{type(e).__name__}: {e}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--self-check', action='store_true', help='run the oracle self-check of the headline workload even with --no-cpu-baseline')
    ap.add_argument('--policy', choices=['mlp', 'lstm'], default='mlp',
                    help="'lstm' = LSTMWrapper(128) on the same envs (BASELINE configs[2]'s policy; not the headline metric)")
    ap.add_argument('--hidden', type=int, default=128,
                    help="models.Default(hidden_size=...) on the squared workload: 128 = the fused kernels (headline); any other multiple of 16 runs "
                         "the width-general GEMM path (pufferlib_amd/general.py) — a side workload, not the metric")
    ap.add_argument('--horizon', type=int, default=None, help='rollout steps per batch (default 128; 32 for c4)')
    ap.add_argument('--products', choices=['fp32', 'bf16x6'], default='fp32',
                    help="how csrc/igemm.hip's rows form multiplies (conv / GEMM-path workloads): 'fp32' = v_mfma_f32_16x16x4_f32 (default, what "
                         "every number of the line is quoted on); 'bf16x6' = each fp32 operand as three bf16 pieces, six partial products per product "
                         "on the bf16 matrix path with fp32 accumulation (as close to f64 as the fp32 chain, other bits) — a labelled side measurement")
    ap.add_argument('--workload', choices=['squared', 'c3', 'c4'], default='squared',
                    help="'c3' = BASELINE configs[2] / SURVEY config C3: MiniGrid-shaped 160-byte rows, 7 actions, 100-step episodes from the "
                         "device-side synthetic generator (the simulator is third-party: env parity unpinned), LSTM(128) policy, bptt 16; "
                         "not the headline metric; 'c4' = BASELINE configs[3] / SURVEY config C4: 8192 envs of uint8 (4,84,84) frames from the "
                         "device-side generator (Atari is a third-party emulator: env parity unpinned), NatureCNN policy; not the headline metric")
    ap.add_argument('--no-breakdown', action='store_true', help='skip the extra (untimed) per-kernel breakdown pass')
    ap.add_argument('--no-transport-ab', action='store_true', help='N > 1: skip the extra K-step legs per transport (p2p fused / p2p launch / rccl)')
    ap.add_argument('--no-extra', action='store_true', help="skip the short configs[2] / configs[3] side runs appended to the N = 1 headline line")
    ap.add_argument('--sustained-seconds', type=float, default=3.0,
                    help='after the K timed steps: an extra leg of at least this many seconds of the same loop, reported as sustained_value (0 = skip)')
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and 'RANK' not in os.environ:
        sys.exit(self_spawn(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # PFA_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than ranks (ranks then share
        # devices); the driver's runs use the default: nccl (= RCCL), one rank per GPU
        backend = os.environ.get('PFA_DIST_BACKEND', 'nccl')
        ndev = torch.cuda.device_count()
        if ndev < world and backend == 'nccl':
            raise SystemExit(f'bench.py: {world} ranks but {ndev} GPU(s) visible; RCCL needs one device per rank '
                             '(PFA_DIST_BACKEND=gloo runs the ranks on shared devices for a functional check)')
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        from pufferlib_amd import dist as pdist
        pdist.pin_rank(local_rank, world, device_index=dev_index)      # this rank's share of the cores next to its GPU (PFA_RANK_AFFINITY=0: off)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{dev_index}'))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)

    from pufferlib_amd import _lib, clean_pufferl, cleanrl, models, vector
    global NUM_ENVS, HORIZON
    if args.workload == 'c4':
        NUM_ENVS, HORIZON = 8192, 32
    if args.horizon:
        HORIZON = args.horizon
    L = _lib.lib()
    _lib.check(L.pfa_igemm_set_products(1 if args.products == 'bf16x6' else 0), 'set_products')
    K, W = args.steps, args.warmup
    per_gpu = NUM_ENVS * HORIZON
    cnn_lstm = False
    if args.workload == 'c4':
        cnn_lstm = args.policy == 'lstm'       # environments/atari/torch.py:4-6: LSTMWrapper(512, 512) over the NatureCNN (GEMM path, general.py)
        args.policy = 'cnn'
        vec = vector.make(vector.make_frames, env_kwargs=dict(framestack=4, num_actions=4, episode_length=100), num_envs=NUM_ENVS,
                          backend=vector.Frames)
    elif args.workload == 'c3':
        args.policy = 'lstm'
        vec = vector.make(vector.make_synthetic, env_kwargs=dict(obs_values=160, num_actions=7, episode_length=100, obs_high=10),
                          num_envs=NUM_ENVS, backend=vector.Synthetic)
    else:
        vec = vector.make(vector.make_squared, env_kwargs=dict(distance_to_target=D, num_targets=NT), num_envs=NUM_ENVS,
                          backend=vector.Squared, obs_stride=64)
    if args.workload == 'squared' and args.policy == 'mlp' and args.hidden != 128:
        args.policy = 'wide'
        pol = cleanrl.Policy(models.Default(vec.driver_env, hidden_size=args.hidden))
    elif args.policy == 'lstm':
        pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Default(vec.driver_env)))
    elif args.policy == 'cnn' and cnn_lstm:
        pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(vec.driver_env, models.Convolutional(vec.driver_env, framestack=4), input_size=512, hidden_size=512))
    elif args.policy == 'cnn':
        pol = cleanrl.Policy(models.Convolutional(vec.driver_env, framestack=4))
    else:
        pol = cleanrl.Policy(models.Default(vec.driver_env))
    data = clean_pufferl.create(make_config(per_gpu * world * (K + W + 8 + SUSTAINED_CAP) * 2, env={'c4': 'frames', 'c3': 'synthetic'}.get(args.workload, 'squared')),
                                vec, pol)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    # dominant kernel of the update: the fused fwd/loss/bwd kernel (MLP) or the BPTT kernel (LSTM)
    # (a wide Default on rows of <= 64 floats trains through the fused kernel of csrc/ppo_wide.hip; other wide shapes through the GEMM path)
    wide_fused = args.policy == 'wide' and getattr(data.gen_engine, 'wide_ws', None) is not None
    dominant = {'mlp': 'ppo_mlp_grad', 'lstm': 'lstm_seq_bwd', 'cnn': 'igemm_rows', 'wide': 'ppo_wide_grad' if wide_fused else 'igemm_rows'}[args.policy]
    L.pfa_timing_select(dominant.encode())
    # The event pair around a launch serialises the queue (~7 us of dispatch bubble each: 16 bracketed launches per step cost
    # the headline 7 %, measured).  The dominant kernel's launches all have the same shape in the MLP / LSTM updates, so every
    # 17th one is bracketed (17 is coprime to the 16 launches of a step: every epoch x minibatch position gets sampled, ~19 launches
    # of a 20-step region; rounds 2-5 bracketed every 5th, which tools/step_times.py shows as ~20 us of every 1.27 ms step — the
    # same loop without the brackets, from a cold start); the conv update's launches differ in shape and are all bracketed.
    event_stride = 1 if (args.policy in ('cnn', 'wide') and not wide_fused) else 17
    L.pfa_timing_stride(event_stride)
    L.pfa_timing_reset()
    L.pfa_timing_enable(1)   # dominant kernel only
    def profile_snapshot():
        # the reference's six section timers + the two wall timers (clean_pufferl.py:328-339), cumulative seconds
        pr, tm = data.profile, getattr(data, '_timers', {})
        snap = {k: getattr(pr, k).elapsed for k in ('env', 'eval_forward', 'eval_misc', 'train_forward', 'learn', 'train_misc')}
        snap.update({k: tm[k].elapsed for k in ('evaluate', 'train') if k in tm})
        return snap

    if world > 1 and data.native_dp:
        from pufferlib_amd import dist as pdist
        pdist.wait_stats(reset=True)             # peer-wait telemetry of the timed region only
    barrier()
    prof0 = profile_snapshot()
    t0 = time.perf_counter()
    for _ in range(K):
        clean_pufferl.evaluate(data)
        clean_pufferl.train(data)
    barrier()
    dt = time.perf_counter() - t0
    prof1 = profile_snapshot()
    peer_wait = pdist.wait_stats(reset=True) if (world > 1 and data.native_dp) else None
    L.pfa_timing_enable(0)
    # clean_pufferl.Profile's breakdown over the timed region, ms per step.  The section timers bracket kernel ENQUEUES (the device
    # runs asynchronously); eval_time / train_time are the wall timers of evaluate() / train(), each ending in one stream sync.
    names_ = dict(evaluate='eval_time', env='env_time', eval_forward='eval_forward_time', eval_misc='eval_misc_time', train='train_time',
                  train_forward='train_forward_time', learn='learn_time', train_misc='train_misc_time')
    profile_ms = {names_[k]: round((prof1[k] - prof0.get(k, 0.0)) / K * 1e3, 4) for k in prof1}
    rank_ms = [dt / K * 1e3]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device='cuda')
        t[rank] = dt
        dist.all_reduce(t)                       # every rank's own wall time of the timed region
        rank_ms = [float(x) / K * 1e3 for x in t.cpu()]
        dt = max(float(x) for x in t.cpu())      # the contract's MAX over ranks
        # per rank: where it is pinned, and how long its launches stood waiting for the slowest peer (in-kernel wall-clock ticks,
        # csrc/p2p_ll.hpp ll_wait_report) — the rank that arrives last waits for the transport only, the others for the skew on top
        from pufferlib_amd import dist as pdist
        aff = pdist._native.get('affinity') or {}
        pr = torch.zeros(world, 8, dtype=torch.float64, device='cuda')
        pr[rank, :4] = torch.tensor([aff.get('numa_node', -1), aff.get('cpus', 0), aff.get('first', -1), aff.get('last', -1)], dtype=torch.float64)
        if peer_wait is not None:
            pr[rank, 4:] = torch.tensor([peer_wait['grad_exchange_wait_us'], peer_wait['grad_exchange_workgroups'],
                                         peer_wait['small_exchange_wait_us'], peer_wait['small_exchange_chunks']], dtype=torch.float64)
        dist.all_reduce(pr)
        per_rank = pr.cpu().tolist()

    # sustained leg (untimed by the contract, reported next to it): the same loop for >= --sustained-seconds, so that a
    # sampler with a seconds-scale period (the driver's rocm-smi poll) sees the device busy; iteration count agreed on rank 0
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = min(max(int(args.sustained_seconds / max(dt / K, 1e-6)) + 1, K), SUSTAINED_CAP)
        if world > 1:
            tn = torch.tensor([n_sus], dtype=torch.int64, device='cuda')
            dist.broadcast(tn, src=0)
            n_sus = int(tn.item())
        barrier()
        ts = time.perf_counter()
        for _ in range(n_sus):
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        barrier()
        dts = time.perf_counter() - ts
        if world > 1:
            tt = torch.tensor([dts], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt.item())
        sustained = dict(steps=n_sus, seconds=dts, value=world * per_gpu * n_sus / dts, ms_per_step=dts / n_sus * 1e3)

    # the same K steps with the two per-iteration host readbacks deferred (PFA_LAZY_READBACK=1: evaluate() / train() hand back lazy
    # containers that resolve on first access; opt-in because C-level dict consumers see an unresolved container as empty) — what the
    # two host round trips of the default mode cost, reported next to the headline, never as the headline
    deferred = None
    if world == 1 and args.policy == 'mlp' and args.workload == 'squared' and args.sustained_seconds > 0:
        saved_lazy = os.environ.get('PFA_LAZY_READBACK')
        os.environ['PFA_LAZY_READBACK'] = '1'
        for _ in range(2):
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        barrier()
        tl = time.perf_counter()
        for _ in range(K):
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        float(data.losses.policy_loss)          # resolves the last readback
        barrier()
        dtl = time.perf_counter() - tl
        if saved_lazy is None:
            os.environ.pop('PFA_LAZY_READBACK', None)
        else:
            os.environ['PFA_LAZY_READBACK'] = saved_lazy
        deferred = dict(value=per_gpu * K / dtl, ms_per_step=dtl / K * 1e3, steps=K, how='PFA_LAZY_READBACK=1 (opt-in; pufferlib_amd/readback.py)')

    def kernel_ms(name):
        n, ms = C.c_int64(0), C.c_double(0.0)
        _lib.check(L.pfa_timing_read(name.encode(), C.byref(n), C.byref(ms)), 'timing_read')
        return n.value, ms.value

    grad_launches, grad_total_ms = kernel_ms(dominant)   # HIP events over the timed region, launch stream
    breakdown = {}
    if rank == 0 and not args.no_breakdown:
        # separate, untimed pass with every instrumented kernel bracketed by events (the extra event packets cost
        # a few us per launch, so they stay out of the timed region)
        KB = min(K, 5)
        L.pfa_timing_reset()
        L.pfa_timing_enable(2)
    if not args.no_breakdown:
        for _ in range(min(K, 5)):
            clean_pufferl.evaluate(data)
            clean_pufferl.train(data)
        barrier()
    if rank == 0 and not args.no_breakdown:
        L.pfa_timing_enable(0)
        names = (('philox_exp_noise', 'rollout_mlp_squared', 'squared_tape', 'gae', 'ppo_mlp_grad', 'ppo_reduce_adam', 'ppo_reduce', 'adam_clip') if args.policy == 'mlp'
                 else ('philox_exp_noise', 'rollout_mlp_squared', 'squared_tape', 'gae', 'ppo_wide_grad', 'ppo_wide_reduce', 'adam_clip') if wide_fused
                 else ('igemm_rows', 'igemm_weights', 'gae', 'adam_clip') if args.policy in ('cnn', 'wide')
                 else (('rollout_lstm_synth' if args.workload == 'c3' else 'rollout_lstm_squared'), 'squared_tape', 'gae', 'lstm_seq_fwd',
                       'lstm_seq_bwd', 'gemm_tn', 'adam_clip'))
        for name in names:
            n, ms = kernel_ms(name)
            breakdown[name] = dict(launches_per_step=n // KB, ms_per_step=round(ms / KB, 4))

    # Data parallel: the same K-step loop once per transport that is up, so that ONE run of `--gpus N` holds the A/B (every rank takes
    # the same legs: what is up was agreed on by MIN all-reduce at create()).  p2p_fused = the optimizer step's exchange inside the
    # reduce + Adam launch (the default, = the headline loop above); p2p_launch = the peer all-reduce as a launch of its own between
    # ppo_reduce and adam_clip; rccl = ncclAllReduce on the native communicator.  Per-collective figures: HIP events around every
    # instrumented launch in a short extra pass (they cost the queue a few us each, so they stay out of the timed loops).
    transports = None
    if world > 1 and args.policy == 'mlp' and data.native_dp and not args.no_transport_ab:
        from pufferlib_amd import dist as pdist
        info0 = pdist.transport_info()
        legs = []
        if info0['p2p']:
            legs += [('p2p_fused', '1', 1), ('p2p_launch', '0', 1)]
        if info0['rccl']:
            legs += [('rccl', '0', 0)]
        transports = {}
        saved_env = os.environ.get('PFA_FUSED_DP')
        for name, fused_dp, p2p_on in legs:
            os.environ['PFA_FUSED_DP'] = fused_dp
            L.pfa_p2p_enable(p2p_on)
            for _ in range(2):
                clean_pufferl.evaluate(data)
                clean_pufferl.train(data)
            barrier()
            tl = time.perf_counter()
            for _ in range(K):
                clean_pufferl.evaluate(data)
                clean_pufferl.train(data)
            barrier()
            dtl = torch.tensor([time.perf_counter() - tl], dtype=torch.float64, device='cuda')
            dist.all_reduce(dtl, op=dist.ReduceOp.MAX)
            KC = 3
            L.pfa_timing_reset()
            L.pfa_timing_enable(2)
            for _ in range(KC):
                clean_pufferl.evaluate(data)
                clean_pufferl.train(data)
            barrier()
            L.pfa_timing_enable(0)
            per_call = {}
            for kn in ('ppo_reduce_adam', 'ppo_reduce', 'p2p_all_reduce', 'rccl_all_reduce', 'adam_clip'):
                n_, ms_ = kernel_ms(kn)
                if n_:
                    per_call[kn] = dict(calls_per_step=n_ // KC, us_per_call=round(ms_ / n_ * 1e3, 2))
            leg_dt = float(dtl.item())
            transports[name] = dict(value=world * per_gpu * K / leg_dt, ms_per_step=leg_dt / K * 1e3, steps=K,
                                    launches_per_optimizer_step=2 if name == 'p2p_fused' else 4, rank0_events=per_call)
        L.pfa_p2p_enable(1)
        if saved_env is None:
            os.environ.pop('PFA_FUSED_DP', None)
        else:
            os.environ['PFA_FUSED_DP'] = saved_env

    import torch as _t
    assert bool(_t.isfinite(data.flat_params.flat).all()), 'non-finite weights after the timed loop'
    assert all(_t.isfinite(_t.tensor(float(v))) for k, v in data.losses.items() if k != 'explained_variance'), dict(data.losses)
    if rank == 0:
        value = world * per_gpu * K / dt
        launches, total_ms = grad_launches, grad_total_ms
        avg_ms = total_ms / max(launches, 1)
        rows_per_launch = per_gpu // NMB
        # algorithmic flop per minibatch row: DESIGN.md section 4 (MLP fwd+bwd) / section 7 (BPTT product [dxe | dh] = dG Wcat)
        flop_row = FLOP_PER_ROW_UPDATE if args.policy == 'mlp' else 2 * 512 * 256
        achieved = flop_row * rows_per_launch / (avg_ms * 1e-3) / 1e12 if launches else 0.0
        if args.policy == 'cnn':
            # the rows-form kernel runs every forward and dX product of the step (launches of different shapes): algorithmic flop of
            # all of them over their summed duration.  Rollout: B forwards; update: EPOCHS x B x (forward + dX)
            step_flop = per_gpu * (CNN_FWD_FLOP + EPOCHS * (CNN_FWD_FLOP + CNN_DX_FLOP))
            if cnn_lstm:   # + the LSTM's gate product [x | h] Wcat^T (1024 x 2048) forward, its transpose in the BPTT, and the heads' two
                lstm_f = 2 * 1024 * 2048
                step_flop += per_gpu * (lstm_f + 2 * 512 * 16 + EPOCHS * (2 * lstm_f + 2 * 2 * 512 * 16))
            achieved = step_flop * K / (total_ms * 1e-3) / 1e12 if launches else 0.0
            flop_row, rows_per_launch = step_flop * K / max(launches, 1), 1
        if wide_fused:
            # SURVEY 8d's per-row figure at another width: forward 2 (64 H + 9 H), backward 2 (9 H + 9 H + 64 H) = 310 H (39 680 at H = 128)
            flop_row = 310 * args.hidden
            achieved = flop_row * rows_per_launch / (avg_ms * 1e-3) / 1e12 if launches else 0.0
        elif args.policy == 'wide':
            # rows-form launches of the GEMM path: encoder + heads forward in the rollout and in every epoch, d feature = dout W2v per epoch
            Hh, Kp_, NO_ = args.hidden, 64, 16
            step_flop = per_gpu * ((2 * Kp_ * Hh + 2 * Hh * NO_) * (1 + EPOCHS) + EPOCHS * 2 * NO_ * Hh)
            achieved = step_flop * K / (total_ms * 1e-3) / 1e12 if launches else 0.0
            flop_row, rows_per_launch = step_flop * K / max(launches, 1), 1
        traffic, traffic_source = pmc_traffic(dominant)
        wide_how = 'fused kernels of csrc/ppo_wide.hip + tile-kernel rollout' if wide_fused else 'GEMM path, general.py'
        roof_peak = PEAK_FP32_MFMA_TFLOPS if args.products == 'fp32' else PEAK_BF16_MFMA_TFLOPS / 6.0
        out = {
            'metric': (f'env steps/sec end-to-end PPO (rollout+GAE+update), {NUM_ENVS} envs'
                       + {'c3': ' [configs[2] workload]', 'c4': ' [configs[3] workload]'}.get(args.workload, '')
                       + (f' [side workload: hidden {args.hidden}]' if args.policy == 'wide' else '')
                       + (' [fused gradient step as six bf16 partial products per fp32 product, fp32 accumulate: opt-in product form, not the metric]'
                          if args.products == 'bf16x6' and args.policy == 'mlp' and args.workload == 'squared' else
                          ' [rows-form products as six bf16 partial products, fp32 accumulate]' if args.products == 'bf16x6' else '')),
            'value': value, 'unit': 'env_steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': dt / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # what the products are computed in: fp32 MFMA everywhere by default; with --products bf16x6 the rows form of csrc/igemm.hip
            # multiplies three-piece bf16 splits of the fp32 operands (six partial products, fp32 accumulation)
            'dtype': ('f32' if args.products == 'fp32' else
                      'f32 operands as 3 x bf16, 6 partial products, f32 accumulate (fused gradient step, csrc/ppo_bf16.hpp); f32 elsewhere'
                      if args.policy == 'mlp' and args.workload == 'squared' else
                      'f32 operands as 3 x bf16, 6 partial products, f32 accumulate (rows form); f32 elsewhere'),
            'data': 'synthetic',
            'config': {'workload': (f'synthetic Atari-shaped frames (uint8 (4,84,84), 4 actions, 100-step episodes), {NUM_ENVS} envs/GPU x {HORIZON} steps, '
                                    f'NatureCNN (models.Convolutional){" + LSTMWrapper(512, 512)" if cnn_lstm else ""}, {NMB} minibatches x {EPOCHS} epochs, bptt {BPTT} (BASELINE configs[3]; env parity '
                                    'unpinned: third-party emulator' + (', sharded' if world > 1 else '') + ')')
                       if args.workload == 'c4' else
                       (f'synthetic MiniGrid-shaped rows (160 bytes as 160 f32, 7 actions, 100-step episodes), {NUM_ENVS} envs/GPU x '
                                    f'{HORIZON} steps, MLP 128 + LSTM 128, {NMB} minibatches x {EPOCHS} epochs, bptt {BPTT} (BASELINE configs[2]; '
                                    'env parity unpinned: third-party simulator' + (', sharded' if world > 1 else '') + ')')
                       if args.workload == 'c3' else
                       f'squared d={D} nt={NT}, {NUM_ENVS} envs/GPU x {HORIZON} steps, obs 49->64 f32 rows, '
                       f'{"MLP 128" if args.policy == "mlp" else f"MLP {args.hidden} ({wide_how})" if args.policy == "wide" else "MLP 128 + LSTM 128 (bptt 16)"}, {NMB} minibatches x {EPOCHS} epochs, bptt {BPTT} ({"BASELINE configs[1]" if args.policy != "wide" else "the configs[1] env with a wider policy: side workload, not the metric"}'
                       + (', sharded as configs[4]' if world > 1 else '') + ')',
                       'global_batch': world * per_gpu, 'parallelism': f'dp{world}'},
            # bf16x6: `achieved` counts the fp32 products' algorithmic flop; every one of them is six partial products on the bf16 matrix
            # pipe, so the roof it runs against is that pipe's dense peak / 6 (2.5 PFLOP/s / 6 = 416.7 TFLOP/s of fp32-equivalent products)
            'roofline': {'bound': 'mfma', 'kernel': dominant, 'achieved': achieved, 'peak': roof_peak,
                         'peak_is': ('fp32 MFMA dense peak (v_mfma_f32_16x16x4_f32)' if args.products == 'fp32' else
                                     'bf16 MFMA dense peak 2500 TFLOP/s / 6 partial products per fp32 product'),
                         'products': args.products,
                         'unit': 'TFLOP/s', 'frac': achieved / roof_peak,
                         'frac_of_fp32_peak': achieved / PEAK_FP32_MFMA_TFLOPS,
                         'frac_useful': (achieved * FLOP_PER_ROW_USEFUL / FLOP_PER_ROW_UPDATE / PEAK_FP32_MFMA_TFLOPS
                                         if args.policy == 'mlp' else None),   # on the 49 real columns (32 000 FLOP/row)
                         'traffic': traffic, 'traffic_source': traffic_source if traffic is not None else None,
                         'traffic_note': None if traffic is not None else traffic_source, 'traffic_build': source_build_id(),
                         'avg_launch_ms': avg_ms, 'launches': launches, 'bracketed': f'every {event_stride}. launch of the timed region' if event_stride > 1 else 'every launch of the timed region',
                         # ppo_mlp_grad's launch carries its own HIP events (hipExtLaunchKernelGGL start/stop: the dispatch's begin and end,
                         # what rocprofv3's kernel trace reports); the other kernels are bracketed by events recorded on their stream
                         'timing': ('HIP events attached to the dispatch (hip_ext.h start/stop)' if args.policy == 'mlp'
                                    else 'HIP events recorded on the launch stream before / after the launch'),
                         'flop_per_launch': flop_row * rows_per_launch},
            'kernel_ms_per_step': breakdown,
            'profile_ms_per_step': profile_ms,
            'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms), 'per_rank': [round(x, 4) for x in rank_ms]},
        }
        if args.policy == 'mlp':
            # the MFMA instructions the instantiated kernel really issues per 16-row tile (csrc/ppo_update.hip: forward KKU x 8, heads
            # 32, dW2v 32, dh DHK x 8, dW1 KTM x 32): the padded SURVEY figure above is the contract's `frac`, this is the executed one
            import ctypes as _C
            grad_path = int(L.pfa_ppo_mlp_grad_path(_C.byref(data.flat_params.dims), per_gpu // NMB))
            out['roofline']['gradient_kernel'] = ('ppo_mlp_grad_bf16_kernel (csrc/ppo_bf16.hpp)' if grad_path == 1 else
                                                  'ppo_mlp_grad_kernel (csrc/ppo_update.hip)')
            if grad_path == 0:
                mfma_tile = int(L.pfa_ppo_mlp_grad_mfma_per_tile(49, 64, 8))
                out['roofline']['mfma_per_tile'] = mfma_tile
                out['roofline']['mfma_utilisation_pmc'] = pmc_mfma_utilisation('ppo_mlp_grad')
                out['roofline']['frac_executed'] = (mfma_tile * 2048 / 16 * rows_per_launch / (avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS
                                                    if launches else 0.0)
            else:
                # v_mfma_f32_16x16x32_bf16 (16 384 FLOP each) per 32-row tile: 4 waves x (48 forward + 12 heads + 12 dW2v + 24 dh + 48 dW1)
                out['roofline']['mfma_per_tile'] = 576
                out['roofline']['frac_executed'] = (576 * 16384 / 32 * rows_per_launch / (avg_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS
                                                    if launches else 0.0)
                out['roofline']['frac_executed_is'] = 'issued bf16 MFMA flop / the bf16 dense peak (2500 TFLOP/s)'
                t_bf, src_bf = pmc_traffic('ppo_mlp_grad_bf16')
                out['roofline']['traffic'] = t_bf
                out['roofline']['traffic_source'] = src_bf if t_bf is not None else None
                out['roofline']['traffic_note'] = None if t_bf is not None else src_bf
        # north_star asks for the HBM side next to the MFMA side: the rollout kernel is the path's HBM-facing kernel (it writes the
        # experience rows, SURVEY 8d: 280 B per env step), and the end-to-end figure is 1420 B per env step
        roll = breakdown.get({'mlp': 'rollout_mlp_squared', 'lstm': 'rollout_lstm_synth' if args.workload == 'c3' else 'rollout_lstm_squared'}.get(args.policy, ''))
        if roll and roll['ms_per_step'] > 0 and args.workload == 'squared':
            gbs = 280.0 * per_gpu / (roll['ms_per_step'] * 1e-3) / 1e9
            out['roofline_hbm'] = {'bound': 'hbm', 'kernel': 'rollout_mlp_squared' if args.policy == 'mlp' else 'rollout_lstm_squared',
                                   'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                                   'bytes_per_launch': 280 * per_gpu, 'avg_launch_ms': roll['ms_per_step'],
                                   'timing': 'HIP events in the untimed breakdown pass',
                                   'traffic': pmc_traffic('rollout_mlp_squared')[0] if args.policy == 'mlp' else None,
                                   'end_to_end': {'bytes_per_env_step': 1420, 'achieved': 1420.0 * value / world / 1e9, 'unit': 'GB/s per GPU',
                                                  'frac': 1420.0 * value / world / 1e9 / 8000.0}}
        if sustained is not None:
            out['sustained_value'] = sustained['value']
            out['sustained'] = sustained
        if deferred is not None:
            out['deferred_readback'] = deferred
        if world > 1:
            from pufferlib_amd import dist as pdist
            info = pdist.transport_info()
            bucket_bytes = int(data.grads.numel()) * 4
            on_p2p = info['p2p'] and bucket_bytes <= info['p2p_slot_bytes']
            out['dist'] = {'backend': os.environ.get('PFA_DIST_BACKEND', 'nccl'), 'devices_visible': torch.cuda.device_count(),
                           'ranks_share_devices': torch.cuda.device_count() < world,
                           'native_collectives': bool(data.native_dp),
                           'transport': {'grad_bucket': ('p2p' if on_p2p else 'rccl' if (data.native_dp and info['rccl']) else 'torch'),
                                         'grad_bucket_bytes': bucket_bytes,
                                         'small_reductions': ('p2p' if info['p2p'] else 'rccl' if (data.native_dp and info['rccl']) else 'torch')},
                           # rank r: its CPU share (cores of its GPU's NUMA node, split among the ranks on that node) and the mean time its
                           # launches stood waiting for the slowest peer; per step = per exchange x exchanges.  min over ranks ~ transport
                           # latency, the spread above it = rank skew (what pinning and the skew budget below are about)
                           'affinity': [{'numa_node': int(x[0]), 'cpus': int(x[1]), 'first': int(x[2]), 'last': int(x[3])} for x in per_rank],
                           'rank_ms_per_step': [round(x, 4) for x in rank_ms],
                           'peer_wait': None if peer_wait is None else {
                               'grad_exchange_wait_us': [round(x[4], 2) for x in per_rank],
                               'small_exchange_wait_us': [round(x[6], 2) for x in per_rank],
                               'wait_us_per_step': [round(x[4] * EPOCHS * NMB + x[6] * 2, 1) for x in per_rank],
                               'what': 'in-kernel 100 MHz wall-clock ticks a workgroup spent spinning for peer data, longest lane, mean '
                                       'over the workgroups of the timed region'},
                           'rccl_nranks': info['rccl_nranks'], 'p2p_selftest_passed': info['p2p_selftest'], 'p2p_status': info['p2p_status'],
                           'allreduce_calls': {'p2p': info['p2p_calls'], 'p2p_flag_in_data': info['p2p_ll_calls'], 'rccl_native': info['rccl_calls']},
                           # what one step (evaluate + train) exchanges, and the budget >= 6x weak scaling at 8 ranks leaves for it:
                           # t_N <= 8/6 t_1, i.e. everything data parallelism adds (exchanges, rank skew, waits) <= t_1 / 3 per step
                           'collectives_per_step': {'gradient_exchanges': EPOCHS * NMB, 'small_all_reduces': 2,
                                                    'what': f'{EPOCHS * NMB} optimizer-step exchanges of the {bucket_bytes}-byte bucket (inside the reduce + '
                                                            'Adam launch on the fused peer path: 2 launches per optimizer step, else an all-reduce of its '
                                                            'own: 4); at the end of evaluate(): 1 all-reduce of [episode-statistic sums | the GAE halo rows of every '
                                                            'rank], then GAE, then 1 of [advantage sums | explained-variance sums]; train() holds none but the '
                                                            'optimizer steps\''},
                           'transports': transports}
        if world == 1 and args.no_cpu_baseline and args.self_check and args.policy == 'mlp':
            out['self_check'] = self_check(data, pol)
            if args.products == 'fp32':
                out['self_check']['reference_replay'] = reference_replay_check()
        if world == 1 and not args.no_cpu_baseline:
            if args.policy == 'mlp':
                out['self_check'] = self_check(data, pol)
                if args.products == 'fp32':
                    out['self_check']['reference_replay'] = reference_replay_check()
            elif args.workload == 'c3':
                out['self_check'] = self_check_c3(data, pol)
            elif args.workload == 'c4' and not cnn_lstm and args.products == 'fp32':
                del data, vec, pol
                torch.cuda.empty_cache()
                out['self_check'] = self_check_c4()
                data = vec = pol = None
            if args.workload == 'c4' and not cnn_lstm:
                out['cpu_baseline'] = reference_cpu_baseline('c4', cpu_baseline_c4())
            elif args.workload == 'c3':
                out['cpu_baseline'] = reference_cpu_baseline('c3', cpu_baseline_c3())
            elif args.policy == 'mlp':          # the headline configuration
                out['cpu_baseline'] = reference_cpu_baseline('c2', cpu_baseline())
        if world == 1 and args.workload == 'squared' and args.policy == 'mlp' and not args.no_extra:
            # BASELINE configs[2] / configs[3] as short side runs (own processes, after everything of the headline is measured):
            # not the metric, but driver-run instead of builder-run numbers for the recurrent and the conv path
            del data, vec, pol
            torch.cuda.empty_cache()
            nb = ['--no-cpu-baseline']
            out['extra_workloads'] = [# the headline workload with the gradient step in the opt-in product form (csrc/ppo_bf16.hpp) + its oracle self-check
                                      extra_workload(['--products', 'bf16x6', '--steps', '20', '--warmup', '3', '--sustained-seconds', '0', '--self-check'] + nb),
                                      extra_workload(['--workload', 'c3', '--steps', '10', '--warmup', '2', '--sustained-seconds', '0']),      # + its cpu_baseline
                                      extra_workload(['--workload', 'c4', '--steps', '5', '--warmup', '1', '--sustained-seconds', '0']),                # + its cpu_baseline
                                      # the same with the rows-form products on the bf16 matrix path (six-term split, fp32 accumulate)
                                      extra_workload(['--workload', 'c4', '--products', 'bf16x6', '--steps', '5', '--warmup', '1', '--sustained-seconds', '0'] + nb),
                                      # widths outside the 128-wide fused kernels on the headline env (tile-kernel rollout + GEMM-path update)
                                      extra_workload(['--hidden', '256', '--steps', '5', '--warmup', '2', '--sustained-seconds', '0'] + nb),
                                      # the recurrent NatureCNN of environments/atari/torch.py:4-6 (GEMM path, general.py)
                                      extra_workload(['--workload', 'c4', '--policy', 'lstm', '--steps', '5', '--warmup', '1', '--sustained-seconds', '0'] + nb)]
        print(json.dumps(out), flush=True)
    if world > 1:
        from pufferlib_amd import dist as pdist
        dist.barrier()
        pdist.finalize_native()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
