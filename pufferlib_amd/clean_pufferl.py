"""PPO trainer with the surface of the reference's clean_pufferl.py, device-resident end to end.

  create(config, vecenv, policy, optimizer=None, wandb=None)   clean_pufferl.py:30-73
  evaluate(data)   rollout: T x {recv, policy, store, send}      clean_pufferl.py:76-154
  train(data)      GAE + the epoch/minibatch PPO loop            clean_pufferl.py:157-292
  close(data), save_checkpoint, try_load_checkpoint, Experience, Profile, make_losses, seed_everything

``data`` is the same namespace (config, vecenv, policy, uncompiled_policy, optimizer, experience, profile,
losses, wandb, global_step, epoch, stats, msg, last_log_time, utilization), so the reference's demo.py loop
``while data.global_step < total: evaluate(data); train(data)`` runs unchanged.

What is different underneath: there is no host round-trip per step.  For a
pufferlib_amd.vector.Squared vecenv and an MLP policy evaluate() is ONE persistent kernel (csrc/rollout.hip) and
train() is one native call that enqueues, per optimizer step, the gradient kernel (fused fwd / loss / bwd) and the reduce + clip +
Adam kernel; GAE and the advantage statistics (2 launches) already ran at the end of evaluate(), under the host's wait for the episode
statistics.
Experience is stored env-major on device, which is the order the reference gets after sort_training_data
(clean_pufferl.py:452-464), so there is no sort and no gather.

Data parallel (no reference counterpart): one process per GPU; rank r owns envs
[r*N, (r+1)*N) (seeds seed + global index), parameters are broadcast from rank 0 at create(), and every
optimizer step all-reduces one flat bucket (gradient + 8 loss sums as float pairs) — inside the reduce + Adam launch over the
peer-mapped path, else RCCL; at the end of evaluate() two small all-reduces (episode statistics + the GAE halo rows; the
advantage-normalisation and explained-variance sums) so that every rank scans the flat batch's own bits and normalises with the
global-minibatch mean/std, and train() holds no collective but its optimizer steps'.

Interface mirroring, stated plainly: `Profile` / `Profile.update` (the timer fields the dashboard and wandb logging read,
clean_pufferl.py:341-367), the `Utilization` sampling loop, the `rollout()` viewer's render / step lines and the tmp-then-rename
checkpoint write follow the reference's own statements closely — they ARE the surface `demo.py` and the dashboard consume, and
there is nothing device-specific to redesign in them.  Everything that computes is this package's own.
"""
import ctypes as C
import os
import random
import time
from collections import deque
from threading import Thread

import numpy as np
import torch

from . import _lib, readback, utils
from . import dist as pdist
from .cleanrl import Policy, RecurrentPolicy
from .models import FlatParams  # noqa: F401  (re-exported: callers of create() type-check the trainer's parameter buffer against it)
from .namespace import namespace
from .vector import Bandit, Frames, Memory, Multiagent, Spaces, Squared, Stochastic, Synthetic


def seed_everything(seed, torch_deterministic=True):
    """clean_pufferl.py:596-601"""
    random.seed(seed)
    np.random.seed(seed)
    if seed is not None:
        torch.manual_seed(seed)
    torch.backends.cudnn.deterministic = torch_deterministic


def make_losses():
    return readback.LazyLosses(policy_loss=0, value_loss=0, entropy=0, old_approx_kl=0, approx_kl=0, clipfrac=0,
                     explained_variance=0)


def _dist():
    return pdist.world()


class HipAdam:
    """torch.optim.Adam(lr, eps=1e-5)-shaped handle (clean_pufferl.py:54-55) over the flat device buffers that
    csrc/ppo_update.hip's adam_clip_kernel updates.  Exposes param_groups[0]['lr'] (written by the lr anneal,
    clean_pufferl.py:261-264) and a torch-compatible state_dict for checkpoints (clean_pufferl.py:520-527)."""

    def __init__(self, flat_params, lr, betas=(0.9, 0.999), eps=1e-5):
        self.fp = flat_params
        self.exp_avg = flat_params.flat_like()
        self.exp_avg_sq = flat_params.flat_like()
        self.step_count = 0
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                                  params=list(range(len(flat_params.split(flat_params.flat)))))]

    def zero_grad(self):
        pass

    def state_dict(self):
        m, v = self.fp.split(self.exp_avg), self.fp.split(self.exp_avg_sq)
        state = {}
        if self.step_count > 0:
            for i, k in enumerate(m):
                state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m[k].clone(), exp_avg_sq=v[k].clone())
        return dict(state=state, param_groups=[dict(self.param_groups[0])])

    def load_state_dict(self, sd):
        m, v = self.fp.split(self.exp_avg), self.fp.split(self.exp_avg_sq)
        for i, k in enumerate(m):
            if i in sd['state']:
                m[k].copy_(sd['state'][i]['exp_avg'])
                v[k].copy_(sd['state'][i]['exp_avg_sq'])
                self.step_count = int(sd['state'][i]['step'])
        self.param_groups[0]['lr'] = sd['param_groups'][0]['lr']


GAE_HALO_MAX = 2056


class Experience:
    """clean_pufferl.Experience (clean_pufferl.py:380-482) as device tensors in env-major order:
    row (env e, step t) at flat index e*T + t."""

    def __init__(self, batch_size, bptt_horizon, minibatch_size, obs_stride, num_envs, device, obs_bytes=None):
        if minibatch_size is None:
            minibatch_size = batch_size
        num_minibatches = batch_size / minibatch_size
        self.num_minibatches = int(num_minibatches)
        if self.num_minibatches != num_minibatches:
            raise ValueError('batch_size must be divisible by minibatch_size')
        minibatch_rows = minibatch_size / bptt_horizon
        self.minibatch_rows = int(minibatch_rows)
        if self.minibatch_rows != minibatch_rows:
            raise ValueError('minibatch_size must be divisible by bptt_horizon')
        if batch_size % num_envs != 0:
            raise ValueError('batch_size must be divisible by the number of envs (whole rollout steps)')
        self.horizon = batch_size // num_envs
        if self.horizon % bptt_horizon != 0:
            # every env must contribute whole bptt segments
            raise ValueError('batch_size / num_envs must be divisible by bptt_horizon')
        B = batch_size
        if obs_bytes is not None:     # frame observations stay bytes (the conv kernels read uint8): rows of obs_bytes, a multiple of 16
            self.obs = torch.zeros(B, obs_bytes, dtype=torch.uint8, device=device)
        else:
            self.obs = torch.zeros(B, obs_stride, dtype=torch.float32, device=device)
        self.actions = torch.zeros(B, dtype=torch.int32, device=device)
        self.logprobs = torch.zeros(B, dtype=torch.float32, device=device)
        # room behind rewards / dones / values for the halo of the data-parallel GAE: the rows that follow this rank's shard in the
        # rank-major flat batch (csrc/gae.hip gae_halo_*: at most the self-starting window's 2048 + 8)
        self._rdv = torch.zeros(3, B + GAE_HALO_MAX, dtype=torch.float32, device=device)
        self.rewards, self.dones, self.values = self._rdv[0, :B], self._rdv[1, :B], self._rdv[2, :B]
        self.advantages = torch.zeros(B, dtype=torch.float32, device=device)
        self.returns = torch.zeros(B, dtype=torch.float32, device=device)
        self.lstm_h = self.lstm_c = None
        self.batch_size, self.bptt_horizon, self.minibatch_size = B, bptt_horizon, minibatch_size
        self.num_envs, self.device = num_envs, device
        self.ptr = 0
        self.step = 0
        self.c = _lib.Experience(self.obs.data_ptr(), self.actions.data_ptr(), self.logprobs.data_ptr(),
                                 self.values.data_ptr(), self.rewards.data_ptr(), self.dones.data_ptr(),
                                 self.advantages.data_ptr(), self.returns.data_ptr(), self.horizon)

    @property
    def full(self):
        return self.ptr >= self.batch_size

    def minibatch_rows_index(self, mb):
        """Flat env-major rows of minibatch `mb`: segments {mb + k*nmb} of bptt_horizon rows (clean_pufferl.py:455-457)."""
        k = torch.arange(self.minibatch_rows, device=self.device)
        h = torch.arange(self.bptt_horizon, device=self.device)
        return ((mb + k[:, None] * self.num_minibatches) * self.bptt_horizon + h[None, :]).reshape(-1)


class Profile:
    """clean_pufferl.Profile (clean_pufferl.py:306-367): same fields; the six section timers bracket kernel
    *enqueues* (the device runs asynchronously), while SPS / eval_time / train_time come from the
    evaluate()/train() wall timers, which end in one stream sync each."""
    SPS = 0
    uptime = 0
    remaining = 0
    eval_time = 0
    env_time = 0
    eval_forward_time = 0
    eval_misc_time = 0
    train_time = 0
    train_forward_time = 0
    learn_time = 0
    train_misc_time = 0

    def __init__(self):
        self.start = time.time()
        self.env = utils.Profiler()
        self.eval_forward = utils.Profiler()
        self.eval_misc = utils.Profiler()
        self.train_forward = utils.Profiler()
        self.learn = utils.Profiler()
        self.train_misc = utils.Profiler()
        self.prev_steps = 0

    def __iter__(self):
        for k in ('SPS', 'uptime', 'remaining', 'eval_time', 'env_time', 'eval_forward_time', 'eval_misc_time',
                  'train_time', 'train_forward_time', 'learn_time', 'train_misc_time'):
            yield k, getattr(self, k)

    # epoch_time / update: deliberate mirror of clean_pufferl.py:342-366 — the reference's dashboard and wandb logging read exactly
    # these fields with exactly this update rule.  Interface, not hot path.
    @property
    def epoch_time(self):
        return self.train_time + self.eval_time

    def update(self, data, interval_s=1):
        global_step = data.global_step
        if global_step == 0:
            return True
        uptime = time.time() - self.start
        if uptime - self.uptime < interval_s:
            return False
        self.SPS = (global_step - self.prev_steps) / (uptime - self.uptime)
        self.prev_steps = global_step
        self.uptime = uptime
        self.remaining = (data.config.total_timesteps - global_step) / max(self.SPS, 1e-9)
        self.eval_time = data._timers['evaluate'].elapsed
        self.eval_forward_time = self.eval_forward.elapsed
        self.env_time = self.env.elapsed
        self.eval_misc_time = self.eval_misc.elapsed
        self.train_time = data._timers['train'].elapsed
        self.train_forward_time = self.train_forward.elapsed
        self.learn_time = self.learn.elapsed
        self.train_misc_time = self.train_misc.elapsed
        return True


def gpu_busy_percent(device_index=None):
    """What the reference reads with torch.cuda.utilization() (clean_pufferl.py:501), from the amdgpu driver's own counter:
    /sys/bus/pci/devices/<domain:bus:device.0>/gpu_busy_percent of the current HIP device (no amdsmi module needed).
    0 when the counter cannot be read (no GPU, container without sysfs)."""
    try:
        props = torch.cuda.get_device_properties(torch.cuda.current_device() if device_index is None else device_index)
        path = f'/sys/bus/pci/devices/{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0/gpu_busy_percent'
        with open(path) as f:
            return int(f.read().strip())
    except Exception:
        return 0


class Utilization(Thread):
    """clean_pufferl.Utilization (clean_pufferl.py:484-507) — sampling thread for the dashboard; optional."""

    def __init__(self, delay=1, maxlen=20):
        super().__init__(daemon=True)
        self.cpu_mem, self.cpu_util = deque(maxlen=maxlen), deque(maxlen=maxlen)
        self.gpu_util, self.gpu_mem = deque(maxlen=maxlen), deque(maxlen=maxlen)
        self.delay, self.stopped = delay, False
        self.start()

    def run(self):
        import psutil
        while not self.stopped:
            self.cpu_util.append(psutil.cpu_percent())
            mem = psutil.virtual_memory()
            self.cpu_mem.append(mem.active / mem.total)
            try:
                free, total = torch.cuda.mem_get_info()
                self.gpu_mem.append(free / total)
            except Exception:
                self.gpu_mem.append(0)
            self.gpu_util.append(gpu_busy_percent())
            time.sleep(self.delay)

    def stop(self):
        self.stopped = True


class _NoUtilization:
    cpu_mem = cpu_util = gpu_util = gpu_mem = (0,)

    def stop(self):
        pass


def _cfg(config, key, default):
    try:
        return config[key]
    except (KeyError, AttributeError):
        return getattr(config, key, default)


def create(config, vecenv, policy, optimizer=None, wandb=None):
    seed_everything(config.seed, _cfg(config, 'torch_deterministic', True))
    profile = Profile()
    losses = make_losses()
    utilization = Utilization() if _cfg(config, 'utilization_thread', False) else _NoUtilization()
    n_params = sum(p.numel() for p in policy.parameters())
    msg = f'Model Size: {n_params} parameters'

    host_mode = not isinstance(vecenv, (Squared, Stochastic, Memory, Bandit, Multiagent, Spaces, Synthetic))   # anything else speaks the recv/send protocol on the host
    if not isinstance(policy, (Policy, RecurrentPolicy)):
        from .models import find_lstm
        if find_lstm(policy) is not None:             # e.g. the reference's RecurrentPolicy(LSTMWrapper(Default))
            policy = RecurrentPolicy(getattr(policy, 'policy', policy), seed=config.seed)
        else:
            policy = Policy(policy, seed=config.seed)     # e.g. the reference's cleanrl.Policy(models.Default)
    recurrent = isinstance(policy, RecurrentPolicy)
    from .models import find_cnn
    conv = find_cnn(policy) is not None
    if conv:
        space = vecenv.single_observation_space
        if not (host_mode or isinstance(vecenv, Frames)):
            raise NotImplementedError('models.Convolutional reads uint8 (framestack, 84, 84) frames: a host vecenv or vector.Frames')
        if np.dtype(space.dtype) != np.uint8 or tuple(space.shape[-2:]) != (84, 84):
            raise NotImplementedError(f'models.Convolutional reads uint8 (framestack, 84, 84) frames, the env shows {space.dtype} {space.shape}')
    if isinstance(vecenv, Frames) and not conv:
        raise NotImplementedError('vector.Frames shows uint8 (framestack, 84, 84) observations: use models.Convolutional')
    if recurrent and isinstance(vecenv, Stochastic):
        raise NotImplementedError('the device-resident Stochastic vecenv has a fused rollout for the MLP policy only '
                                  '(ocean.Stochastic: "do not use a policy with memory", ocean.py:534)')
    dist, rank, world = _dist()
    env_offset = rank * vecenv.num_agents
    host_bridge = None
    if host_mode:
        from . import hostpath
        _lib.require_gpu()
        device = torch.device('cuda', torch.cuda.current_device())
        obs_values = int(np.prod(vecenv.single_observation_space.shape))
        obs_stride = obs_values if conv else hostpath.obs_stride_for(obs_values, recurrent)     # frames stay bytes
        host_bridge = hostpath.HostBridge(vecenv, obs_stride, device, frames=conv)
    else:
        device = vecenv.device
        obs_stride = vecenv.obs_stride
        vecenv.info_mode = 'lazy'
        vecenv.env_offset = env_offset
    vecenv.async_reset(config.seed + env_offset)          # clean_pufferl.py:39; env i of rank r gets seed + r*N + i
    fp = policy.adopt(obs_stride, device)
    if fp.multidiscrete and not (host_mode or isinstance(vecenv, Spaces)):
        raise NotImplementedError('MultiDiscrete action heads run on the host-vecenv path and on vector.Spaces; the other device-resident envs take one Discrete head')
    native_dp = False
    if world > 1:
        dist.broadcast(fp.flat, src=0)
        # RCCL communicator inside the native library and / or the one-shot peer path for this policy's bucket, else torch fallback
        # (the other exchange the slots must hold: episode statistics + the GAE halo rows of every rank as f64 bit patterns, _publish_gae)
        native_dp = pdist.init_native(bucket_bytes=(fp.count + 16) * 4,
                                      small_bytes=8 * (16 + 3 * world * min(config.batch_size, GAE_HALO_MAX)))
    elif _cfg(config, 'force_native_dp', False):
        native_dp = pdist.init_native(force_single=True)   # 1-rank communicator: exercises the DP code path in tests
    policy.noise_seed = int(config.seed)

    total_agents = vecenv.num_agents
    if world > 1:
        nmb_ = config.batch_size // (_cfg(config, 'minibatch_size', None) or config.batch_size)
        pdist.check_partition(total_agents, config.batch_size // total_agents, config.bptt_horizon, nmb_)
    experience = Experience(config.batch_size, config.bptt_horizon, _cfg(config, 'minibatch_size', None),
                            obs_stride, total_agents, device, obs_bytes=fp.obs_dim if conv else None)
    optimizer = HipAdam(fp, lr=config.learning_rate, eps=1e-5)
    lstm_engine = cnn_engine = gen_engine = None
    from . import general
    if isinstance(fp, general.GeneralParams):
        # a policy shape outside the fused kernels (wide Default, wide / conv LSTM): GEMM-path engine for rollout steps and updates
        gen_engine = general.Engine(fp, experience, total_agents)
        policy.gen_engine = gen_engine
        policy._evaluator = None
        if gen_engine.lstm_h is not None:
            experience.lstm_h, experience.lstm_c = gen_engine.lstm_h, gen_engine.lstm_c
    elif conv:
        cnn_engine = policy.cnn_engine
        cnn_engine.experience = experience
        cnn_engine._alloc(min(8192, max(experience.minibatch_size, total_agents)))
    if recurrent and gen_engine is None:
        from . import lstm as plstm
        lstm_engine = plstm.Engine(fp, experience, vecenv)
        experience.lstm_h, experience.lstm_c = lstm_engine.lstm_h, lstm_engine.lstm_c   # clean_pufferl.py:407-412

    L = _lib.lib()
    hp = _make_hparams(config, experience)
    # (the conv engine owns its gradient workspaces; the shared one then only serves adv_stats / GAE / log sums: any MLP shape sizes it)
    # (likewise the GEMM-path engine: a wide flat observation would otherwise reserve 256 fused-kernel partials of 128 x obs_stride floats)
    dims = _lib.MlpDims(64, 64, 128, 4, 0) if (conv or gen_engine is not None) else fp.dims
    ws_bytes = max(L.pfa_ppo_workspace_bytes(C.byref(dims), config.batch_size, C.byref(hp)),
                   L.pfa_gae_workspace_bytes(config.batch_size),
                   L.pfa_gae_sums_workspace_bytes(config.batch_size, experience.num_minibatches))
    dp_sums = torch.zeros(2 * experience.num_minibatches + 4, dtype=torch.float64, device=device)
    data = namespace(
        config=config, vecenv=vecenv, policy=policy, uncompiled_policy=policy, optimizer=optimizer,
        experience=experience, profile=profile, losses=losses, wandb=wandb, global_step=0, epoch=0, stats={},
        msg=msg, last_log_time=0, utilization=utilization,
        # engine state
        flat_params=fp, rank=rank, world_size=world, native_dp=native_dp, lstm_engine=lstm_engine, cnn_engine=cnn_engine, gen_engine=gen_engine,
        env_offset=env_offset, host_bridge=host_bridge,
        workspace=torch.zeros(ws_bytes, dtype=torch.uint8, device=device),
        grads=torch.zeros(fp.count + 16, dtype=torch.float32, device=device),   # gradient + 8 loss sums as (hi, lo) float pairs
        # [adv_stats (nmb x 2) | the four explained-variance sums]: one allocation, so that data parallel they are ONE all-reduce
        dp_sums=dp_sums, adv_stats=dp_sums[:2 * experience.num_minibatches].view(experience.num_minibatches, 2),
        # data parallel: the sharded GAE's block aggregates (kept from the publish at the end of evaluate() to train()), the
        # exchange buffer [episode-statistic sums | world x 6 GAE numbers] and the carry-in
        gae_ws=torch.zeros(max(int(L.pfa_gae_workspace_bytes(config.batch_size)), 16), dtype=torch.uint8, device=device) if world > 1 else None,
        gae_carry=torch.zeros(1, dtype=torch.float64, device=device), _dp_eval=None, _gae_published=False,
        loss_acc=torch.zeros(8, dtype=torch.float64, device=device),
        log_sums=torch.zeros(10, dtype=torch.float64, device=device),
        noise=None,            # optional explicit Exp(1) tensor [T][N][A] for the next evaluate() (parity tests)
        tape_stream=torch.cuda.Stream(device=device),   # reset-target tape is drawn one rollout ahead, off the critical path
        _rb_eval=readback.Pending(), _rb_train=readback.Pending(),   # deferred stats / loss readbacks (readback.py)
    )
    return data


def _make_hparams(config, experience):
    return _lib.PpoHparams(float(config.clip_coef), float(config.vf_clip_coef), float(config.vf_coef),
                           float(config.ent_coef), 1 if config.norm_adv else 0, 1 if config.clip_vloss else 0,
                           experience.num_minibatches, int(config.bptt_horizon))


@utils.profile
def evaluate(data):
    config, profile, experience, vecenv = data.config, data.profile, data.experience, data.vecenv
    L = _lib.lib()
    fp = data.flat_params
    if data.host_bridge is not None:      # host vecenv: the reference's recv/forward/store/send loop (hostpath.py)
        from . import hostpath
        return hostpath.evaluate(data)
    T, N = experience.horizon, vecenv.num_agents
    with profile.eval_misc:
        policy = data.policy
        if vecenv.flag != 3:  # RECV: evaluate starts with a recv (clean_pufferl.py:86); the state machine must allow it
            from .exceptions import APIUsageError
            raise APIUsageError('Call reset before stepping')
        noise = data.noise
        if noise is not None:
            noise = noise.to(device=vecenv.device, dtype=torch.float32).contiguous()
            assert tuple(noise.shape) == (T, N, fp.num_actions), noise.shape
        key = _lib.NoiseKey(policy.noise_seed, policy.noise_step)
    wide_view = data.gen_engine.mlp_view if data.gen_engine is not None else None
    if data.gen_engine is not None and not (wide_view is not None and type(vecenv) is Squared):
        # a policy shape outside the fused kernels: policy step + store + send per step
        with profile.eval_forward:
            _rollout_stepwise(data, noise, T, N)
        return _finish_evaluate(data, N, T)
    if isinstance(vecenv, (Memory, Synthetic)) and data.lstm_engine is not None:   # recurrent policy: one persistent kernel
        with profile.env:
            if hasattr(vecenv, 'ensure_tape'):
                vecenv.ensure_tape(T)
        with profile.eval_forward:
            data.lstm_engine.rollout(T, noise, policy.noise_seed, policy.noise_step, vecenv.env_offset)
            vecenv.sends += T
        return _finish_evaluate(data, N, T)
    if isinstance(vecenv, (Memory, Bandit, Multiagent, Spaces, Synthetic)):   # no fused kernel for these (env, policy) pairs: protocol-level pieces, still no host sync per step
        with profile.eval_forward:
            _rollout_stepwise(data, noise, T, N)
        return _finish_evaluate(data, N, T)
    if isinstance(vecenv, Stochastic):    # no reset tape: the env draws no random numbers
        with profile.eval_forward:
            vecenv.fused_rollout_mlp(fp, experience, noise, key, _lib.stream_handle())
        return _finish_evaluate(data, N, T)
    with profile.env:
        main = torch.cuda.current_stream()
        vecenv.ensure_tape(T)                    # waits for the prefetch (vecenv.tape_event); draws nothing when it covered T
        start_point = torch.cuda.Event()
        start_point.record(main)                 # everything before this rollout (incl. the rollout before it) is done
    with profile.eval_forward:           # one persistent kernel for all T steps, either policy
        if data.lstm_engine is not None:
            data.lstm_engine.rollout(T, noise, policy.noise_seed, policy.noise_step, vecenv.env_offset)
        else:
            if noise is None:
                # the whole rollout's Philox action noise in one launch (same numbers the kernel would draw in place): keeps 40
                # quarter-rate integer multiplies per step off the rollout's dependent chain; the kernel prefetches a step ahead.
                # The stream is a pure function of (seed, step, global env index), so the previous evaluate() already drew THIS
                # rollout's numbers on the side stream behind the reset tape (vecenv.tape_event, waited for in ensure_tape above,
                # covers them); anything that moved the stream position in between (policy(obs) calls, a checkpoint load, another
                # seed) fails the tag comparison and the numbers are drawn here, on the compute stream, as before.
                tag = (int(policy.noise_seed), int(policy.noise_step), T, N, int(fp.num_actions), int(vecenv.env_offset))
                pre = getattr(data, '_noise_next', None)
                if pre is not None and pre[0] == tag:
                    noise = pre[1]
                else:
                    noise = _noise_buffer(data, T, N, fp.num_actions, vecenv.device)
                    _lib.check(L.pfa_philox_exp_noise(_lib.ptr(noise), T, N, fp.num_actions, C.byref(key), vecenv.env_offset,
                                                      _lib.stream_handle()), 'philox_exp_noise')
                data._noise_next = None
                data._noise_cur = noise
            if wide_view is not None:     # Default(hidden 64 / 256 / 512): the same persistent kernel, W1 fragments of that width in registers
                _lib.check(L.pfa_rollout_mlp_view_squared(
                    _lib.ptr(vecenv.state), C.byref(vecenv.cfg), C.byref(wide_view), C.byref(experience.c),
                    _lib.ptr(noise), C.byref(key), vecenv.env_offset, _lib.ptr(vecenv.obs_buf), _lib.ptr(vecenv.rewards),
                    _lib.ptr(vecenv.terminals_u8), _lib.ptr(vecenv.truncations_u8), _lib.ptr(vecenv.masks_u8),
                    _lib.stream_handle()), 'rollout')
            else:
                _lib.check(L.pfa_rollout_mlp_squared(
                    _lib.ptr(vecenv.state), C.byref(vecenv.cfg), _lib.ptr(fp.flat), C.byref(fp.dims), C.byref(experience.c),
                    _lib.ptr(noise), C.byref(key), vecenv.env_offset, _lib.ptr(vecenv.obs_buf), _lib.ptr(vecenv.rewards),
                    _lib.ptr(vecenv.terminals_u8), _lib.ptr(vecenv.truncations_u8), _lib.ptr(vecenv.masks_u8),
                    _lib.stream_handle()), 'rollout')
    with profile.env:
        vecenv.sends += T
        # The tape does not depend on actions: draw the NEXT rollout's reset rounds on the side stream while THIS
        # rollout runs (its small workgroup co-resides with the rollout's).  The ring slots it writes belong to the
        # rollout before this one, which `start_point` guarantees has finished.
        if 3 * (vecenv._rounds_needed(T) + 1) <= vecenv.tape_rounds:
            with torch.cuda.stream(data.tape_stream):
                data.tape_stream.wait_event(start_point)
                vecenv.ensure_tape(T)
                if data.noise is None and data.lstm_engine is None and getattr(data, '_noise_cur', None) is not None:
                    # ... and the NEXT rollout's action noise (16.8 MB of writes that depend on nothing but the stream position),
                    # into the buffer the running rollout is not reading
                    nxt = _noise_buffer(data, T, N, fp.num_actions, vecenv.device, other_than=data._noise_cur)
                    nkey = _lib.NoiseKey(policy.noise_seed, policy.noise_step + T)
                    _lib.check(L.pfa_philox_exp_noise(_lib.ptr(nxt), T, N, fp.num_actions, C.byref(nkey), vecenv.env_offset,
                                                      _lib.stream_handle()), 'philox_exp_noise (prefetch)')
                    data._noise_next = ((int(policy.noise_seed), int(policy.noise_step) + T, T, N, int(fp.num_actions),
                                         int(vecenv.env_offset)), nxt)
                ev = torch.cuda.Event()
                ev.record(data.tape_stream)
                vecenv.tape_event = ev           # the vecenv owns it: send()/async_reset() outside evaluate() wait on it too
    return _finish_evaluate(data, N, T)


def _noise_buffer(data, T, N, A, device, other_than=None):
    """One of the two [T][N][A] action-noise buffers of the MLP rollout (the rollout reads one while the side stream fills the other)."""
    bufs = getattr(data, '_noise_bufs', None)
    if bufs is None or tuple(bufs[0].shape) != (T, N, A):
        bufs = data._noise_bufs = [torch.empty(T, N, A, dtype=torch.float32, device=device) for _ in range(2)]
    return bufs[1] if (other_than is not None and other_than.data_ptr() == bufs[0].data_ptr()) else bufs[0]


def _frames_policy(data):
    """True when observations are uint8 frames (models.Convolutional, with or without an LSTM on top): rows are moved as bytes."""
    return data.cnn_engine is not None or (data.gen_engine is not None and data.gen_engine.net.kind == 'cnn')


def _rollout_stepwise(data, noise, T, N):
    """clean_pufferl.evaluate's loop (clean_pufferl.py:84-124) for a device vecenv without a fused rollout kernel: per step
    policy forward + sample (one kernel, MLP or recurrent), Experience.store (one kernel), vecenv.device_send (one kernel)."""
    L = _lib.lib()
    vecenv, policy, fp, exp = data.vecenv, data.policy, data.flat_params, data.experience
    stream = _lib.stream_handle()
    dev = vecenv.device
    eng = data.lstm_engine
    if eng is not None:
        from . import lstm as plstm
        plstm.pack_gates(fp, eng.wpack)
    actions = torch.empty(N, dtype=torch.int64, device=dev)
    logprob = torch.empty(N, device=dev)
    value = torch.empty(N, device=dev)
    # the reset rounds are drawn ahead in chunks the tape ring can hold (a whole rollout where it fits: one launch instead of one
    # per resetting send; a long horizon on few envs — e.g. Spaces with batch_size / num_envs >= 512 — in several)
    tape_chunk = T
    if hasattr(vecenv, 'ensure_tape'):
        tape_chunk = max(1, min(T, int(getattr(vecenv, 'max_sends_per_tape', T))))
    for t in range(T):
        if hasattr(vecenv, 'ensure_tape') and t % tape_chunk == 0:
            vecenv.ensure_tape(min(tape_chunk, T - t))
        key = _lib.NoiseKey(policy.noise_seed, policy.noise_step + t)
        nz = None if noise is None else noise[t]
        if data.gen_engine is not None:
            data.gen_engine.policy_step(vecenv.obs_buf, N, nz, key, vecenv.env_offset, actions, logprob, None, value)
        elif data.cnn_engine is not None:
            data.cnn_engine.policy_step(vecenv.obs_buf, N, nz, key, vecenv.env_offset, actions, logprob, None, value)
        elif eng is None:
            _lib.check(L.pfa_mlp_forward_sample(_lib.ptr(vecenv.obs_buf), N, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(nz),
                                                C.byref(key), vecenv.env_offset, _lib.ptr(actions), _lib.ptr(logprob), None,
                                                _lib.ptr(value), stream), 'forward_sample')
        else:
            _lib.check(L.pfa_lstm_policy_step(_lib.ptr(vecenv.obs_buf), N, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(eng.wpack),
                                              _lib.ptr(eng.lstm_h), _lib.ptr(eng.lstm_c), _lib.ptr(nz), C.byref(key),
                                              vecenv.env_offset, _lib.ptr(actions), _lib.ptr(logprob), None, _lib.ptr(value),
                                              stream), 'lstm_policy_step')
        # (frame rows are bytes: the copy moves them as obs_dim / 4 four-byte words)
        _lib.check(L.pfa_store_step(C.byref(exp.c), t, N, fp.obs_dim // 4 if _frames_policy(data) else fp.obs_stride, _lib.ptr(vecenv.obs_buf), _lib.ptr(vecenv.rewards),
                                    _lib.ptr(vecenv.terminals_u8), _lib.ptr(actions), _lib.ptr(logprob), _lib.ptr(value), stream),
                   'store_step')
        vecenv.device_send(actions)


def _finish_evaluate(data, N, T):
    profile, experience, vecenv, policy = data.profile, data.experience, data.vecenv, data.policy
    L = _lib.lib()
    with profile.eval_misc:
        if data.lstm_engine is not None:
            data.lstm_engine.invalidate_obs_cache()     # new experience rows: per-update caches keyed on them are stale (advisor, round 4)
        policy.noise_step += T
        data.noise = None
        experience.ptr = experience.batch_size
        experience.step = T
        data.global_step += N * T * data.world_size      # sum(mask) per recv (clean_pufferl.py:90), all ranks
        direct_stats = (data.world_size == 1 and type(vecenv) is Squared and readback.direct_ok(vecenv.device))
        if direct_stats:                                # single rank: the five numbers go straight into the readback's pinned buffer
            st = vecenv.stats_with_flag(reset=True, out=data._rb_eval.direct_buffer(5, torch.float64))
        else:
            st = vecenv.stats_with_flag(reset=True)     # 4 sums + the tape underrun flag
        early = _early_gae() and experience.full and data.host_bridge is None
        if data.world_size > 1 and early:
            # data parallel: the episode statistics and what the sharded GAE needs from the other ranks (csrc/gae.hip: the first rows
            # of the shards that follow — all from every rank's own rows, complete once its rollout is) ride ONE all-reduce
            st = _publish_gae(data, st)
        elif data.native_dp:
            _lib.check(L.pfa_dist_all_reduce_f64(_lib.ptr(st), st.numel(), _lib.stream_handle()), 'stats all-reduce')
        elif data.world_size > 1:
            dist, _, _ = _dist()
            dist.all_reduce(st)
        # The sums ride a pinned buffer behind the rollout; the host waits for them here (default) or, with
        # PFA_LAZY_READBACK=1, when the dicts are first read — at the latest at the end of the next evaluate() (readback.py).
        stats, infos = readback.LazyDict(data._rb_eval), readback.LazyDict(data._rb_eval)
        means = vecenv.stats_from_sums

        def finish(host, stats=stats, infos=infos, means=means):
            if host.shape[0] > 4 and host[4] != 0:
                raise RuntimeError('reset-target tape underrun: an env reset before its tape round was drawn (host bookkeeping of '
                                   'ensure_tape / the side-stream prefetch is wrong); the rollout replayed stale targets')
            m = means(host)
            stats.fill(m)
            infos.fill({k: [v] for k, v in m.items()})
        data.stats = stats
        if direct_stats:
            data._rb_eval.submit_direct(finish, vecenv.device, defer=True)
        else:
            data._rb_eval.submit(st, finish, defer=True)
        # The update's first pass — compute_gae + the advantage statistics (train() below) — reads nothing but the rows this rollout
        # just wrote, so it is enqueued HERE, behind the event of the episode statistics: the host waits for the five numbers (and
        # then walks back through the caller into train()) while the device already runs the ~25 us pass instead of idling through
        # that round trip.  train() recognises the pass by its key (_gae_key: storage version of rewards / dones / values,
        # hyper-parameters, partition) and launches it itself when anything changed in between (reward shaping, value
        # re-bootstrapping, another gamma) or when evaluate() was not the previous call.  PFA_EARLY_GAE=0: always in train().
        # Data parallel the pass includes its two exchanges (the halo rows with the episode statistics above, the advantage sums
        # here); a caller that edits the rows between the two calls does so on EVERY rank (train() then repeats both, collectively).
        data._gae_done = None
        if early and data.world_size > 1:
            _finish_gae(data)                   # the scan, this rank's sums, ONE all-reduce of them: train() holds no collective but its optimizer steps'
            data._gae_done = _gae_key(data)
        elif early and _fused_sums_ok(data, data.world_size):
            _launch_gae_sums(data)
            data._gae_done = _gae_key(data)
        if readback.eager():
            data._rb_eval.resolve()
    return stats, infos


def _early_gae():
    return os.environ.get('PFA_EARLY_GAE', '1') != '0'


def _ev_with_gae(data, world):
    """The four explained-variance sums only need advantages and values, so they are taken with the advantage sums (data parallel:
    they ride that all-reduce) — except in the one mode whose y_pred is the host path's arrival-order value buffer."""
    return not (getattr(data, 'arrival_values', None) is not None and _cfg(data.config, 'async_store', 'balanced') == 'reference')


def _fused_sums_ok(data, world=1):
    """GAE, the per-minibatch advantage sums and the explained-variance sums in ONE pass over the rows (pfa_gae_sums_f32 /
    pfa_gae_halo_f32 with sums) where the partition allows it; the separate entry points otherwise."""
    config, experience = data.config, data.experience
    return (_ev_with_gae(data, world)
            and bool(_lib.lib().pfa_gae_sums_supported(experience.batch_size, experience.num_envs, experience.num_minibatches,
                                                       int(config.bptt_horizon))))


def _gae_key(data):
    """What the GAE pass is a function of, as far as the host can see it: the storage (address + version counter: every in-place
    torch write through any view bumps it) of rewards / dones / values, the buffers it writes, the hyper-parameters and the
    partition.  The library's own kernels and writes through numpy / raw-pointer views are invisible to the counter: a pipeline
    that rewrites the rows that way between evaluate() and train() sets ``data._gae_done = None`` (or PFA_EARLY_GAE=0)."""
    config, ex = data.config, data.experience
    rdv = getattr(ex, '_rdv', None)
    ver = rdv._version if rdv is not None else (ex.rewards._version, ex.dones._version, ex.values._version)
    ptrs = tuple(int(getattr(ex, k).data_ptr()) for k in ('rewards', 'dones', 'values', 'advantages', 'returns'))
    return (ver, ptrs, ex.batch_size, float(config.gamma), float(config.gae_lambda), ex.num_envs, ex.num_minibatches,
            int(config.bptt_horizon), bool(config.norm_adv))


def _launch_gae_sums(data):
    """compute_gae over the env-major batch (clean_pufferl.py:163-169) + returns (:482) + the update's advantage statistics, fused form."""
    config, experience = data.config, data.experience
    nmb = experience.num_minibatches
    _lib.check(_lib.lib().pfa_gae_sums_f32(
        _lib.ptr(experience.dones), _lib.ptr(experience.values), _lib.ptr(experience.rewards), _lib.ptr(experience.advantages),
        _lib.ptr(experience.returns), experience.batch_size, float(config.gamma), float(config.gae_lambda), experience.num_envs, nmb,
        int(config.bptt_horizon), _lib.ptr(data.adv_stats), C.c_void_p(data.dp_sums.data_ptr() + 16 * nmb), _lib.ptr(data.loss_acc),
        _lib.ptr(data.workspace), _lib.stream_handle()), 'gae_sums')


def _all_reduce_f64(data, buf, what):
    if data.native_dp:
        _lib.check(_lib.lib().pfa_dist_all_reduce_f64(_lib.ptr(buf), buf.numel(), _lib.stream_handle()), what)
    else:
        _dist()[0].all_reduce(buf)


def _publish_gae(data, extra=None):
    """First half of the data-parallel GAE: what the other ranks need from this rank's rows + `extra` (f64 sums that ride along: the
    episode statistics) -> ONE all-reduce(SUM).  Returns the all-reduced `extra` (a view of the exchange buffer).
    Halo form (csrc/gae.hip gae_halo_*, gamma lambda <= 0.984): the bit patterns of the shard's first rows, so that every rank runs
    the single-rank kernel over its rows + the rows that follow them — the flat scan's own bits.  Otherwise the f64-carry form: six
    numbers per rank (interior map, last value, first row), a few ulps from the flat scan at the shard ends."""
    config, ex = data.config, data.experience
    L, B, stream = _lib.lib(), ex.batch_size, _lib.stream_handle()
    _, rank, world = _dist()
    gamma, lam = float(config.gamma), float(config.gae_lambda)
    n_extra = 0 if extra is None else extra.numel()
    H = int(L.pfa_gae_halo_rows(gamma, lam))
    if getattr(ex, '_rdv', None) is None or ex._rdv.shape[1] < B + H:
        H = 0                          # a caller-built Experience without room behind its arrays
    size = n_extra + (3 * world * min(B, H) if H else 6 * world)
    if data._dp_eval is None or data._dp_eval.numel() != size:
        data._dp_eval = torch.zeros(size, dtype=torch.float64, device=ex.device)
    buf = data._dp_eval
    rows = (_lib.ptr(ex.dones), _lib.ptr(ex.values), _lib.ptr(ex.rewards))
    if H:
        _lib.check(L.pfa_gae_halo_publish(*rows, B, gamma, lam, _lib.ptr(extra), n_extra, _lib.ptr(buf), rank, world, stream), 'gae halo publish')
    else:
        _lib.check(L.pfa_gae_shard_publish(*rows, B, gamma, lam, _lib.ptr(data.gae_ws), _lib.ptr(extra), n_extra, _lib.ptr(buf), rank, world,
                                           stream), 'gae publish')
    _all_reduce_f64(data, buf, 'stats + gae all-reduce')
    data._gae_published = (H, n_extra)
    return buf[:n_extra]


def _finish_gae(data):
    """Second half: compute_gae over the GLOBAL rank-major flat batch (c_gae.pyx:11-32 crosses env boundaries, so it also crosses
    shard boundaries) from what _publish_gae gathered, this rank's share of the update's sums (per-minibatch advantage sums, the
    explained-variance sums) and ONE all-reduce of those — every rank then normalises with the global-minibatch mean / std."""
    config, ex = data.config, data.experience
    L, B, stream = _lib.lib(), ex.batch_size, _lib.stream_handle()
    _, rank, world = _dist()
    gamma, lam = float(config.gamma), float(config.gae_lambda)
    nmb = ex.num_minibatches
    if not data._gae_published:
        _publish_gae(data, None)
    H, n_extra = data._gae_published
    data._gae_published = None
    gathered = data._dp_eval[n_extra:]
    rows = (_lib.ptr(ex.dones), _lib.ptr(ex.values), _lib.ptr(ex.rewards))
    ev4 = C.c_void_p(data.dp_sums.data_ptr() + 16 * nmb)
    fused = bool(H) and _fused_sums_ok(data, world)
    if H:
        halo_len = L.pfa_gae_halo_unpack(_lib.ptr(gathered), rank, world, B, gamma, lam, *rows, stream)
        if halo_len < 0:
            _lib.check(halo_len, 'gae halo unpack')
        _lib.check(L.pfa_gae_halo_f32(*rows, _lib.ptr(ex.advantages), _lib.ptr(ex.returns), B, halo_len, gamma, lam, ex.num_envs, nmb,
                                      int(config.bptt_horizon), _lib.ptr(data.adv_stats) if fused else None, ev4, _lib.ptr(data.loss_acc),
                                      _lib.ptr(data.workspace), stream), 'gae halo')
    else:
        has_next = int(rank < world - 1)
        _lib.check(L.pfa_gae_shard_fold(_lib.ptr(gathered), rank, world, B, gamma, lam, _lib.ptr(data.gae_ws), *rows,
                                        _lib.ptr(data.gae_carry), stream), 'gae fold')
        _lib.check(L.pfa_gae_shard_pass2(*rows, _lib.ptr(ex.advantages), _lib.ptr(ex.returns), B, has_next, gamma, lam,
                                         _lib.ptr(data.gae_ws), _lib.ptr(data.gae_carry) if has_next else None, stream), 'gae pass 2')
    with_ev = _ev_with_gae(data, world)
    if not fused:
        hp = _make_hparams(config, ex)
        if config.norm_adv:
            _lib.check(L.pfa_ppo_adv_stats(C.byref(ex.c), B, C.byref(hp), _lib.ptr(data.adv_stats), _lib.ptr(data.workspace), stream), 'adv_stats')
        else:
            data.adv_stats.zero_()
        if with_ev:
            _lib.check(L.pfa_train_ev_sums(C.byref(ex.c), B, ex.num_envs, ev4, _lib.ptr(data.workspace), stream), 'train_ev_sums')
        data.loss_acc.zero_()
    if with_ev:
        _all_reduce_f64(data, data.dp_sums, 'adv + ev all-reduce')
    elif config.norm_adv:
        _all_reduce_f64(data, data.adv_stats, 'adv all-reduce')


@utils.profile
def train(data):
    config, profile, experience = data.config, data.profile, data.experience
    data.losses = make_losses()
    losses = data.losses
    L = _lib.lib()
    fp, opt = data.flat_params, data.optimizer
    stream = _lib.stream_handle()
    B, nmb = experience.batch_size, experience.num_minibatches
    hp = _make_hparams(config, experience)
    dist, rank, world = _dist()

    with profile.train_misc:
        # compute_gae over the env-major batch (clean_pufferl.py:163-169) + returns (:482)
        # single rank: GAE, the per-minibatch advantage sums and the explained-variance sums in ONE pass over the rows
        # (pfa_gae_sums_f32: 3 launches; the separate entry points below: 6) where the partition allows it
        early_key, data._gae_done = getattr(data, '_gae_done', None), None      # (one use: a second train() on the same rows runs its own pass)
        fresh = early_key is not None and early_key == _gae_key(data)           # evaluate() already ran this very pass
        early_ev = _ev_with_gae(data, world)          # the explained-variance sums exist since the GAE pass (else: taken at the end)
        fused_sums = world == 1 and _fused_sums_ok(data, world)
        if world > 1:
            if not fresh:                             # (host vecenv, PFA_EARLY_GAE=0, rows edited since — on every rank: both exchanges are collectives)
                _finish_gae(data)
        elif fused_sums:
            if not fresh:
                _launch_gae_sums(data)
        else:
            early_ev = False
            _lib.check(L.pfa_gae_f32(_lib.ptr(experience.dones), _lib.ptr(experience.values), _lib.ptr(experience.rewards),
                                     _lib.ptr(experience.advantages), _lib.ptr(experience.returns), B, float(config.gamma),
                                     float(config.gae_lambda), _lib.ptr(data.workspace), stream), 'gae')
            if config.norm_adv:
                _lib.check(L.pfa_ppo_adv_stats(C.byref(experience.c), B, C.byref(hp), _lib.ptr(data.adv_stats),
                                               _lib.ptr(data.workspace), stream), 'adv_stats')
            data.loss_acc.zero_()
        experience.ptr = 0
        experience.step = 0

    global_mb_rows = experience.minibatch_size * world
    loss_scale = 1.0 / (global_mb_rows * nmb)
    epochs_run = 0
    # multi-kernel updates: recurrent (lstm.py) / conv (cnn.py) / any other shape (general.py)
    eng = data.gen_engine if data.gen_engine is not None else (data.lstm_engine if data.lstm_engine is not None else data.cnn_engine)
    if eng is not None:
        eng.update_id = getattr(eng, 'update_id', 0) + 1     # a new batch of experience: per-update caches of the engine are stale
        for epoch in range(config.update_epochs):
            eng.state = None                      # lstm_state = None (clean_pufferl.py:176)
            for mb in range(nmb):
                with profile.train_forward:
                    eng.update(mb, hp, data.adv_stats, global_mb_rows, data.grads, B)
                with profile.learn:
                    if data.native_dp:
                        _lib.check(L.pfa_dist_all_reduce_f32(_lib.ptr(data.grads), data.grads.numel(), stream), 'grad all-reduce')
                    elif world > 1:
                        dist.all_reduce(data.grads)
                    eng.clip_adam(data.grads, opt, config.max_grad_norm, data.loss_acc, loss_scale)
            if config.target_kl is not None:
                if float(data.grads[fp.count + 8:fp.count + 10].double().sum().item()) / global_mb_rows > config.target_kl:
                    break
    native_loop = eng is None and (world == 1 or data.native_dp) and config.target_kl is None
    direct_log, log_out, log_packed = False, None, C.c_int32(0)
    if native_loop:
        # no early exit: the whole epoch x minibatch loop (incl. the per-step RCCL all-reduce when data parallel) is
        # enqueued by one native call on the compute stream
        with profile.learn:
            g = opt.param_groups[0]
            if early_ev:
                # the report (six loss sums + the four explained-variance sums, known since GAE) rides the update's last launch,
                # straight into the readback's pinned buffer where the runtime allows it
                direct_log = readback.direct_ok(experience.device)
                log_out = data._rb_train.direct_buffer(10, torch.float64) if direct_log else data.log_sums
            _lib.check(L.pfa_ppo_mlp_train_logged(
                C.byref(experience.c), B, _lib.ptr(fp.flat), C.byref(fp.dims), C.byref(hp), _lib.ptr(data.adv_stats),
                _lib.ptr(data.grads), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), opt.step_count, float(g['lr']),
                float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(config.max_grad_norm),
                int(config.update_epochs), _lib.ptr(data.loss_acc), _lib.ptr(data.workspace),
                1 if data.native_dp else 0, C.c_void_p(data.dp_sums.data_ptr() + 16 * nmb) if log_out is not None else None,
                _lib.ptr(log_out), C.byref(log_packed), stream), 'ppo_train')
            opt.step_count += config.update_epochs * nmb
    for epoch in range(0 if (native_loop or eng is not None) else config.update_epochs):
        for mb in range(nmb):
            with profile.train_forward:
                _lib.check(L.pfa_ppo_mlp_grad(C.byref(experience.c), B, mb, _lib.ptr(fp.flat), C.byref(fp.dims),
                                              C.byref(hp), _lib.ptr(data.adv_stats), global_mb_rows,
                                              _lib.ptr(data.grads), _lib.ptr(data.workspace), stream), 'ppo_grad')
            with profile.learn:
                if world > 1:
                    dist.all_reduce(data.grads)      # one flat bucket per optimizer step (RCCL over xGMI)
                opt.step_count += 1
                g = opt.param_groups[0]
                _lib.check(L.pfa_adam_clip_step(
                    _lib.ptr(fp.flat), _lib.ptr(data.grads), _lib.ptr(opt.exp_avg), _lib.ptr(opt.exp_avg_sq), fp.count,
                    float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), opt.step_count,
                    float(config.max_grad_norm), 1.0, C.c_void_p(data.grads.data_ptr() + 4 * fp.count),
                    _lib.ptr(data.loss_acc), loss_scale, None, 0, stream), 'adam')
        epochs_run += 1
        if config.target_kl is not None:
            # approx_kl of the LAST minibatch of this epoch (clean_pufferl.py:256-258) — needs a sync
            last_kl = float(data.grads[fp.count + 8:fp.count + 10].double().sum().item()) / global_mb_rows
            if last_kl > config.target_kl:
                break

    with profile.train_misc:
        if config.anneal_lr:
            frac = 1.0 - data.global_step / config.total_timesteps
            opt.param_groups[0]['lr'] = frac * config.learning_rate

        # losses + explained variance exactly as the reference logs them (clean_pufferl.py:249-254,266-270, App. A.8):
        # y_pred = values in STORAGE (step-major) order, y_true = advantages (env-major) + y_pred; one D2H of 10 f64
        if log_packed.value:                                    # the update's last launch already left the ten numbers (pfa_ppo_mlp_train_logged)
            pass
        elif early_ev:                                          # the sums exist since GAE (data parallel: all-reduced with the advantage sums)
            # ... and the ten numbers go straight into the pinned host buffer of the readback (no device-to-host copy launch behind it)
            if log_out is None:
                direct_log = readback.direct_ok(experience.device)
                log_out = data._rb_train.direct_buffer(10, torch.float64) if direct_log else data.log_sums
            _lib.check(L.pfa_train_log_pack(_lib.ptr(data.loss_acc), C.c_void_p(data.dp_sums.data_ptr() + 16 * nmb), _lib.ptr(log_out),
                                            stream), 'train_log_pack')
        else:
            direct_log = False
            _lib.check(L.pfa_train_log_sums(C.byref(experience.c), B, experience.num_envs, _lib.ptr(data.loss_acc),
                                            _lib.ptr(data.log_sums), _lib.ptr(data.workspace), stream), 'train_log_sums')
            if getattr(data, 'arrival_values', None) is not None and _cfg(config, 'async_store', 'balanced') == 'reference':
                # the reference's y_pred is the value buffer in STORAGE order = arrival order here (hostpath.evaluate kept it)
                yp, ad = data.arrival_values.double(), experience.advantages.double()
                yt = ad + yp
                data.log_sums[6:10] = torch.stack([yt.sum(), (yt * yt).sum(), ad.sum(), (ad * ad).sum()])
            if data.native_dp:                                  # explained variance over the GLOBAL batch, like the (global) losses
                _lib.check(L.pfa_dist_all_reduce_f64(C.c_void_p(data.log_sums.data_ptr() + 6 * 8), 4, stream), 'ev all-reduce')
            elif world > 1:
                ev_sums = data.log_sums[6:10].clone()
                dist.all_reduce(ev_sums)
                data.log_sums[6:10] = ev_sums
        # one D2H of 10 f64 (the one sync of train(); with PFA_LAZY_READBACK=1 data.losses fills in when it is first read)
        Bg = B * world

        def finish(acc, losses=losses, Bg=Bg, check_peers=bool(data.native_dp)):
            if check_peers:
                pdist.raise_if_peer_lost()           # the peer all-reduce's bounded waits: a lost rank is an error here, never a stale sum
            if _lib.lib().pfa_ppo_grid_status() != 0:
                _lib.lib().pfa_ppo_grid_reset()      # reported once: the recovery named below can then proceed in this process
                raise RuntimeError('the grid-wide hand-off of the fused reduce + Adam launch timed out (PFA_WAIT_TIMEOUT_MS): the '
                                   'device is shared or CU-masked so that its workgroups were not resident together; the parameters '
                                   'hold NaN.  Set PFA_FUSED_ADAM=0 (two-kernel form) and restore a checkpoint')
            s_y, s_yy, s_a, s_aa = acc[6:10]
            var_y = s_yy / Bg - (s_y / Bg) ** 2
            var_res = s_aa / Bg - (s_a / Bg) ** 2                    # y_true - y_pred = advantages
            ev = float('nan') if var_y == 0 else 1 - var_res / var_y
            losses.fill(policy_loss=float(acc[0]), value_loss=float(acc[1]), entropy=float(acc[2]), old_approx_kl=float(acc[3]),
                        approx_kl=float(acc[4]), clipfrac=float(acc[5]), explained_variance=ev)
        losses.attach(data._rb_train)
        if direct_log:
            data._rb_train.submit_direct(finish, experience.device)
        else:
            data._rb_train.submit(data.log_sums, finish)
        data.epoch += 1

        done_training = data.global_step >= config.total_timesteps
        if profile.update(data) or done_training:
            if _cfg(config, 'dashboard', False) and rank == 0:
                print_dashboard(config.env, data.utilization, data.global_step, data.epoch, profile, data.losses,
                                data.stats, data.msg)
            if data.wandb is not None and data.global_step > 0 and time.time() - data.last_log_time > 3.0:
                data.last_log_time = time.time()
                data.wandb.log({
                    '0verview/SPS': profile.SPS, '0verview/agent_steps': data.global_step,
                    '0verview/epoch': data.epoch, '0verview/learning_rate': opt.param_groups[0]['lr'],
                    **{f'environment/{k}': v for k, v in readback.materialize(data.stats).items()},
                    **{f'losses/{k}': v for k, v in data.losses.items()},
                    **{f'performance/{k}': v for k, v in data.profile},
                })
        interval = _cfg(config, 'checkpoint_interval', 0)
        if rank == 0 and interval and (data.epoch % interval == 0 or done_training):
            save_checkpoint(data)
            data.msg = f'Checkpoint saved at update {data.epoch}'


def close(data):
    data.vecenv.close()
    data.utilization.stop()
    config = data.config
    if data.wandb is not None:
        artifact = data.wandb.Artifact(f'{config.exp_id}_model', type='model')
        artifact.add_file(save_checkpoint(data))
        data.wandb.run.log_artifact(artifact)
        data.wandb.finish()


def save_checkpoint(data):
    """clean_pufferl.py:509-530: whole-module pickle + trainer_state.pt (tmp-then-rename)."""
    config = data.config
    path = os.path.join(config.data_dir, config.exp_id)
    os.makedirs(path, exist_ok=True)
    model_name = f'model_{data.epoch:06d}.pt'
    model_path = os.path.join(path, model_name)
    if getattr(data, '_last_saved_epoch', None) == data.epoch and os.path.exists(model_path):
        return model_path                                # close() right after the final periodic save of THIS process
    torch.save(data.uncompiled_policy, model_path)      # whole module, loadable by the reference's eval path; always overwrites
    state = dict(optimizer_state_dict=data.optimizer.state_dict(), global_step=data.global_step,
                 agent_step=data.global_step, update=data.epoch, model_name=model_name, exp_id=config.exp_id,
                 noise_step=int(data.policy.noise_step))   # Philox action-noise stream position (no reference counterpart)
    state_path = os.path.join(path, 'trainer_state.pt')
    torch.save(state, state_path + '.tmp')
    os.rename(state_path + '.tmp', state_path)
    data._last_saved_epoch = data.epoch
    return model_path


def try_load_checkpoint(data):
    """clean_pufferl.py:532-546"""
    config = data.config
    path = os.path.join(config.data_dir, config.exp_id)
    trainer_path = os.path.join(path, 'trainer_state.pt')
    if not os.path.exists(trainer_path):
        print('No checkpoint found. Assuming new experiment')
        return
    resume = torch.load(trainer_path, weights_only=False)
    sd = torch.load(os.path.join(path, resume['model_name']), map_location=data.flat_params.flat.device,
                    weights_only=False)
    if isinstance(sd, torch.nn.Module):
        sd = sd.state_dict()
    with torch.no_grad():      # copy INTO the views of the flat device buffer the kernels read
        for k, v in data.uncompiled_policy.state_dict().items():
            v.copy_(sd[k])
    if data.cnn_engine is not None:
        data.cnn_engine.version += 1          # the packed weight forms are stale
    if data.gen_engine is not None:
        data.gen_engine.net.version += 1
    data.optimizer.load_state_dict(resume['optimizer_state_dict'])
    data.global_step = resume['global_step']
    data.epoch = resume['update']
    data.policy.noise_step = int(resume.get('noise_step', data.policy.noise_step))
    print(f'Loaded checkpoint {resume["model_name"]}')


def rollout(env_creator, env_kwargs, agent_creator, agent_kwargs, model_path=None, device='cuda', steps=None, frame_sleep=None):
    """clean_pufferl.rollout (clean_pufferl.py:551-594), the eval / viewer loop of ``demo.py --mode eval``: one env behind the
    vecenv API, render, policy forward, step, print the reward.  The env is device-resident when this package hosts its
    creator (pufferlib_amd.demo.device_backend_for), else the reference's own Serial.  `steps` (None = forever, like the
    reference) and `frame_sleep` are additions so that the loop can be driven by tests."""
    from . import demo as _demo
    from . import vector as _vector
    cls = _demo.device_backend_for(env_creator)

    def make_env(kw):
        if cls is not None:
            return _vector.make(env_creator, env_kwargs=kw, backend=cls)
        import pufferlib.vector as ref_vector          # a host env: the reference package must be there anyway
        return ref_vector.make(env_creator, env_kwargs=kw)
    try:
        env = make_env({'render_mode': 'rgb_array', **env_kwargs})
    except Exception:
        env = make_env(env_kwargs)
    if model_path is None:
        agent = agent_creator(env, **agent_kwargs).to(device)
    else:
        agent = torch.load(model_path, map_location=device, weights_only=False)
    ob, info = env.reset()
    driver = env.driver_env
    if steps is None:
        os.system('clear')
    state = None
    n = 0
    rewards = []
    while steps is None or n < steps:
        render = driver.render()
        if driver.render_mode == 'ansi':
            print('\033[0;0H' + render + '\n')
            time.sleep(0.6 if frame_sleep is None else frame_sleep)
        elif driver.render_mode == 'rgb_array':
            import cv2
            render = cv2.cvtColor(render, cv2.COLOR_RGB2BGR)
            cv2.imshow('frame', render)
            cv2.waitKey(1)
            time.sleep(1 / 24 if frame_sleep is None else frame_sleep)
        with torch.no_grad():
            ob = torch.as_tensor(ob).to(device)
            if hasattr(agent, 'lstm'):
                action, _, _, _, state = agent(ob, state)
            else:
                action, _, _, _ = agent(ob)
            action = action.cpu().numpy().reshape(env.action_space.shape)
        ob, reward = env.step(action)[:2]
        reward = float(torch.as_tensor(reward).float().mean())
        rewards.append(reward)
        print(f'Reward: {reward:.4f}')
        n += 1
    return rewards


def count_params(policy):
    """clean_pufferl.py:548-549: trainable parameter count (the dashboard's "Params" row)."""
    return sum(p.numel() for p in policy.parameters() if p.requires_grad)


def print_dashboard(env_name, utilization, global_step, epoch, profile, losses, stats, msg, clear=False):
    """Plain-text stand-in for the rich dashboard (clean_pufferl.py:644-738): same numbers, one block."""
    lines = [f'[pufferlib_amd] env={env_name} steps={global_step} epoch={epoch} SPS={profile.SPS:,.0f} '
             f'uptime={profile.uptime:.1f}s',
             '  eval {:.3f}s train {:.3f}s'.format(profile.eval_time, profile.train_time),
             '  losses ' + ' '.join(f'{k}={v:.5f}' for k, v in losses.items()),
             '  stats ' + ' '.join(f'{k}={v:.4f}' for k, v in stats.items()), f'  {msg}']
    print('\n'.join(lines), flush=True)
