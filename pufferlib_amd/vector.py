"""Vectorised-env boundary: ``make`` and the device-resident Squared backend.

Mirrors pufferlib/vector.py of the reference:
  make(...)                      vector.py:577-637  (same validation, same APIUsageError messages)
  reset / step helpers           vector.py:44-53
  recv_precheck / send_precheck  vector.py:25-42    (RESET -> RECV -> SEND state machine)
  Squared (backend class)        replaces Serial (vector.py:70-166) for ocean ``make_squared`` envs: the N envs
                                 live in HBM and are stepped by HIP kernels (csrc/squared.hip); ``recv`` returns
                                 *device tensors* that alias the backend-owned live buffers, exactly like Serial
                                 returns aliases of its numpy buffers (vector.py:158-162).

Drop-in use with the reference's factory:  pufferlib.vector.make(make_squared, backend=pufferlib_amd.vector.Squared,
num_envs=N, env_kwargs=dict(distance_to_target=3, num_targets=1)).
"""
import ctypes as C

import numpy as np

from . import _lib, spaces
from .exceptions import APIUsageError
from .namespace import Namespace, namespace

# Deliberate interface mirror of pufferlib/vector.py:17-53 (flag constants, recv/send prechecks, reset/step helpers): callers of a
# drop-in backend compare against these constants and match on these messages.  Interface, not hot path.
RESET = 0
STEP = 1
SEND = 2
RECV = 3
CLOSE = 4
MAIN = 5
INFO = 6


def recv_precheck(vecenv):
    if vecenv.flag != RECV:
        raise APIUsageError('Call reset before stepping')
    vecenv.flag = SEND


def send_precheck(vecenv, actions):
    if vecenv.flag != SEND:
        raise APIUsageError('Call (async) reset + recv before sending')
    vecenv.flag = RECV


def reset(vecenv, seed=42):
    vecenv.async_reset(seed)
    obs, rewards, terminals, truncations, infos, env_ids, masks = vecenv.recv()
    return obs, infos


def step(vecenv, actions):
    vecenv.send(actions)
    obs, rewards, terminals, truncations, infos, env_ids, masks = vecenv.recv()
    return obs, rewards, terminals, truncations, infos


def make_seeds(seed, num_envs):
    if isinstance(seed, int):
        return [seed + i for i in range(num_envs)]
    err = f'seed {seed} must be an integer or a list of integers'
    if isinstance(seed, (list, tuple)):
        if len(seed) != num_envs:
            raise APIUsageError(err)
        return seed
    raise APIUsageError(err)


def make_squared(distance_to_target=3, num_targets=1, **kwargs):
    """Env creator token with the signature of ocean.environment.make_squared (ocean/environment.py:28-31).
    The Squared backend never calls it: it only reads its keyword defaults / env_kwargs."""
    return SquaredSpec(distance_to_target, num_targets)


def env_creator(name='squared'):
    """pufferlib.environments.ocean.env_creator (ocean/environment.py:6-26), squared only."""
    if name == 'squared':
        return make_squared
    raise ValueError('Invalid environment name')


def episode_means(st):
    """The means clean_pufferl.evaluate reports (clean_pufferl.py:127-137) from the (all-reduced) ``episode_stats`` sums
    (count, sum episode_return, sum episode_length, sum score)."""
    if st[0] <= 0:
        return {}
    return dict(episode_return=st[1] / st[0], episode_length=st[2] / st[0], score=st[3] / st[0])


class SquaredSpec:
    """What ``driver_env`` exposes to policies (models.py:26-37) and to clean_pufferl (:40-43)."""

    def __init__(self, distance_to_target=3, num_targets=1):
        self.distance_to_target = int(distance_to_target)
        self.num_targets = 4 * self.distance_to_target if num_targets == -1 else int(num_targets)
        g = 2 * self.distance_to_target + 1
        self.grid_size = g
        self.single_observation_space = spaces.Box(low=-1, high=1, shape=(g, g), dtype=np.float32)
        self.single_action_space = spaces.Discrete(8)
        self.observation_space = self.single_observation_space
        self.action_space = self.single_action_space
        self.num_agents = 1
        self.render_mode = 'ansi'
        self.emulated = namespace(observation_dtype=np.dtype(np.float32),
                                  emulated_observation_dtype=np.dtype((np.float32, (g, g))))
        self.done = True
        self._live_obs = None        # the vecenv's live observation buffer (set by the backend): render() draws env 0

    def render(self):
        """ocean.Squared.render (ocean.py:514-527) of env 0: targets blue, the agent red, empty cells grey."""
        if self._live_obs is None:
            return ''
        g = self.grid_size
        grid = self._live_obs[0, :g * g].detach().cpu().numpy().reshape(g, g)
        chars = []
        for row in grid:
            for val in row:
                color = 94 if val == 1 else (91 if val == -1 else 90)
                chars.append(f'\033[{color}m██\033[0m')
            chars.append('\n')
        return ''.join(chars)

    def close(self):
        pass


def _round_up(x, m):
    return (x + m - 1) // m * m


def _squared_kwargs(creator, args, kwargs):
    """Resolve (distance_to_target, num_targets) the way make_squared's signature would."""
    import inspect
    name = getattr(creator, '__name__', '')
    if 'squared' not in name.lower():
        raise APIUsageError(
            f'pufferlib_amd.vector.Squared only hosts ocean make_squared envs on device (got creator {name!r})')
    d, nt = 3, 1
    try:
        sig = inspect.signature(creator)
        if 'distance_to_target' in sig.parameters and sig.parameters['distance_to_target'].default is not inspect._empty:
            d = sig.parameters['distance_to_target'].default
        if 'num_targets' in sig.parameters and sig.parameters['num_targets'].default is not inspect._empty:
            nt = sig.parameters['num_targets'].default
    except (TypeError, ValueError):
        pass
    args = list(args)
    if len(args) > 0:
        d = args[0]
    if len(args) > 1:
        nt = args[1]
    d = kwargs.get('distance_to_target', d)
    nt = kwargs.get('num_targets', nt)
    return int(d), int(nt)


class Squared:
    """Device-resident vecenv of N ocean Squared envs (backend protocol of pufferlib/vector.py).

    State, live buffers and the reset-target tape are torch device tensors (torch is the allocator); every
    method only enqueues HIP kernels on torch's current stream.  ``infos`` follow ``info_mode``:
      'sync'  (default) one tiny D2H per send: exact list of per-episode dicts like Serial/EpisodeStats
      'lazy'  recv() returns [] and episode statistics are read with ``episode_stats()`` (no per-step sync;
              this is what pufferlib_amd.clean_pufferl.evaluate uses)
    """
    reset = reset
    step = step

    @property
    def num_envs(self):
        return self.agents_per_batch

    def __init__(self, env_creators, env_args, env_kwargs, num_envs, obs_stride=None, info_mode='sync',
                 env_offset=0, device=None, **kwargs):
        import torch
        for k in kwargs:
            if k not in ('num_workers', 'batch_size', 'zero_copy', 'backend'):
                raise APIUsageError(f'Invalid argument: {k}')
        if len(env_creators) != num_envs:
            raise APIUsageError('env_creators must be a list of length num_envs')
        specs = {_squared_kwargs(c, a, k) for c, a, k in zip(env_creators, env_args, env_kwargs)}
        if len(specs) != 1:
            raise APIUsageError(f'obs/atn space mismatch: all envs must share one Squared configuration, got {specs}')
        d, nt = specs.pop()
        self.driver_env = SquaredSpec(d, nt)
        d, nt = self.driver_env.distance_to_target, self.driver_env.num_targets
        _lib.require_gpu()
        self.L = _lib.lib()
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.emulated = self.driver_env.emulated
        self.agents_per_env = [1] * num_envs
        self.agents_per_batch = num_envs
        self.num_agents = num_envs
        self.single_observation_space = self.driver_env.single_observation_space
        self.single_action_space = self.driver_env.single_action_space
        self.action_space = spaces.MultiDiscrete([8] * num_envs)
        g = self.driver_env.grid_size
        self.observation_space = spaces.Box(low=-1, high=1, shape=(num_envs, g, g), dtype=np.float32)
        self.agent_ids = np.arange(num_envs)
        self.initialized = False
        self.flag = RESET
        self.info_mode = info_mode
        self.env_offset = int(env_offset)   # global index of local env 0 (multi-GPU sharding)

        self.obs_dim = g * g
        self.obs_stride = int(obs_stride) if obs_stride is not None else max(16, _round_up(self.obs_dim, 16))
        self.episode_len = nt * d + 1        # max_ticks steps + the auto-reset row
        self.tape_rounds = 192       # ring capacity: two rollouts of 4 x 96 sends.. (one being consumed + one prefetched)
        self.cfg = _lib.SquaredConfig(num_envs, d, nt, self.obs_stride, self.tape_rounds)
        nbytes = self.L.pfa_squared_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise APIUsageError(self.L.pfa_last_error().decode())
        dev = self.device
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.obs_buf = torch.zeros(num_envs, self.obs_stride, dtype=torch.float32, device=dev)
        self.observations = self.obs_buf[:, :self.obs_dim].unflatten(1, (g, g))   # view, shape (N, g, g)
        self.driver_env._live_obs = self.obs_buf
        self.rewards = torch.zeros(num_envs, dtype=torch.float32, device=dev)
        self.terminals_u8 = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.truncations_u8 = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.masks_u8 = torch.ones(num_envs, dtype=torch.uint8, device=dev)
        self.terminals = self.terminals_u8.view(torch.bool)
        self.truncations = self.truncations_u8.view(torch.bool)
        self.masks = self.masks_u8.view(torch.bool)
        self._actions = torch.zeros(num_envs, dtype=torch.int64, device=dev)
        self._fin = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self._fin_ret = torch.zeros(num_envs, dtype=torch.float64, device=dev)
        self._fin_len = torch.zeros(num_envs, dtype=torch.int32, device=dev)
        self._fin_score = torch.zeros(num_envs, dtype=torch.float64, device=dev)
        self._stats = torch.zeros(5, dtype=torch.float64, device=dev)   # 4 sums + tape underrun flag (0 for envs without a tape)
        self.infos = []
        self.sends = 0            # sends since async_reset
        self.rounds_filled = 0    # reset rounds drawn into the tape (mirrors the device counter)
        self.tape_event = None    # recorded by whoever filled tape rounds on ANOTHER stream (the trainer's prefetch)

    def _wait_tape(self):
        """Order the current stream behind a tape fill enqueued on a side stream: `rounds_filled` (host mirror) already counts
        those rounds, so every consumer of the tape — send(), a fused rollout, async_reset() — must wait for the fill itself."""
        if self.tape_event is not None:
            import torch
            torch.cuda.current_stream().wait_event(self.tape_event)

    # -- tape bookkeeping (host mirror; the stream position itself lives on device) --------------------
    def _rounds_needed(self, upto_send):
        """Number of reset rounds consumed by sends 1..upto_send (every env resets on sends k*episode_len)."""
        return upto_send // self.episode_len

    @property
    def max_sends_per_tape(self):
        """Sends whose reset rounds one ensure_tape() call may draw ahead (half the ring: the other half may still be in use)."""
        return max(1, (self.tape_rounds // 2) * self.episode_len)

    def ensure_tape(self, extra_sends):
        self._wait_tape()
        need = self._rounds_needed(self.sends + extra_sends)
        consumed = self._rounds_needed(self.sends)
        if need - consumed > self.tape_rounds:
            raise APIUsageError(f'{extra_sends} sends need {need - consumed} reset rounds; tape holds {self.tape_rounds}')
        if need > self.rounds_filled:
            _lib.check(self.L.pfa_squared_fill_tape(_lib.ptr(self.state), C.byref(self.cfg), need - self.rounds_filled,
                                                    _lib.stream_handle()), 'fill_tape')
            self.rounds_filled = need

    def _live(self):
        return (_lib.ptr(self.obs_buf), _lib.ptr(self.rewards), _lib.ptr(self.terminals_u8),
                _lib.ptr(self.truncations_u8), _lib.ptr(self.masks_u8))

    # -- protocol ---------------------------------------------------------------------------------------
    def async_reset(self, seed=42):
        self.flag = RECV
        seeds = make_seeds(seed, self.num_agents)
        if any(s != seeds[0] + i for i, s in enumerate(seeds)):
            raise APIUsageError('pufferlib_amd.vector.Squared needs consecutive seeds (seed + env index)')
        self._wait_tape()
        self.tape_event = None
        _lib.check(self.L.pfa_squared_async_reset(_lib.ptr(self.state), C.byref(self.cfg), int(seeds[0]), *self._live(),
                                                  _lib.stream_handle()), 'async_reset')
        self.sends = 0
        self.rounds_filled = 0
        self.infos = []

    def send(self, actions):
        import torch
        send_precheck(self, actions)
        if not torch.is_tensor(actions):
            a = np.asarray(actions)
            if not self.initialized and not self.action_space.contains(a):
                raise APIUsageError('Actions do not match action space')
            actions = torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64))
        elif not self.initialized:
            if actions.shape != (self.num_agents,) or actions.dtype not in (torch.int64, torch.int32):
                raise APIUsageError('Actions do not match action space')
        self.initialized = True
        self._actions.copy_(actions.reshape(-1), non_blocking=True)
        self.ensure_tape(1)
        _lib.check(self.L.pfa_squared_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._actions), *self._live(),
                                           _lib.stream_handle()), 'send')
        self.sends += 1
        self.infos = self._collect_infos() if self.info_mode == 'sync' else []

    def device_send(self, actions):
        """send() for a device int64 tensor of actions, no protocol bookkeeping and no host sync (rollout loops of policies
        without a fused rollout kernel); the caller has drawn the reset rounds (ensure_tape)."""
        self._wait_tape()
        _lib.check(self.L.pfa_squared_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(actions), *self._live(),
                                           _lib.stream_handle()), 'send')
        self.sends += 1

    def recv(self):
        recv_precheck(self)
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids,
                self.masks)

    def close(self):
        self.flag = CLOSE

    # -- infos -------------------------------------------------------------------------------------------
    def _collect_infos(self):
        # every env of a Squared vecenv finishes on the same sends: skip the readback on the others
        if self.sends % self.episode_len != self.episode_len - 1:
            return []
        _lib.check(self.L.pfa_squared_last_infos(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._fin),
                                                 _lib.ptr(self._fin_ret), _lib.ptr(self._fin_len),
                                                 _lib.ptr(self._fin_score), _lib.stream_handle()), 'last_infos')
        fin = self._fin.cpu().numpy().astype(bool)
        if not fin.any():
            return []
        ret, ln, sc = self._fin_ret.cpu().numpy(), self._fin_len.cpu().numpy(), self._fin_score.cpu().numpy()
        return [dict(episode_return=float(ret[i]), episode_length=int(ln[i]), score=float(sc[i]))
                for i in np.nonzero(fin)[0]]

    def episode_stats(self, reset=True):
        """(count, mean episode_return, mean episode_length, mean score) over episodes finished since the last
        reset of the accumulators — what clean_pufferl.evaluate reports (clean_pufferl.py:144-152)."""
        _lib.check(self.L.pfa_squared_episode_stats(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._stats),
                                                    1 if reset else 0, _lib.stream_handle()), 'episode_stats')
        return self._stats[:4]      # [4] = tape underrun flag: read by the trainer from the same buffer (stats_with_flag)

    stats_from_sums = staticmethod(episode_means)

    def stats_with_flag(self, reset=True, out=None):
        """The four sums + the tape underrun flag; `out`: a 5-element f64 buffer the kernel writes instead of the vecenv's own
        (the trainer's pinned readback buffer: no device-to-host copy launch behind the kernel)."""
        if out is not None:
            _lib.check(self.L.pfa_squared_episode_stats(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(out),
                                                        1 if reset else 0, _lib.stream_handle()), 'episode_stats')
            return out
        self.episode_stats(reset)
        return self._stats

    # -- test introspection -------------------------------------------------------------------------------
    def debug_targets(self):
        import torch
        out = torch.zeros(self.num_agents, self.cfg.num_targets, dtype=torch.int32, device=self.device)
        _lib.check(self.L.pfa_squared_debug_targets(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(out),
                                                    _lib.stream_handle()), 'debug_targets')
        return out.cpu().numpy()

    def debug_stream_pos(self):
        import torch
        out = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.pfa_squared_debug_stream_pos(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(out),
                                                       _lib.stream_handle()), 'debug_stream_pos')
        return int(out.item())


class _SimpleSpec:
    """What ``driver_env`` exposes to policies (models.py:26-37) and clean_pufferl for a one-value-observation ocean env."""

    def __init__(self, low, high, num_actions):
        self.single_observation_space = spaces.Box(low=low, high=high, shape=(1,), dtype=np.float32)
        self.single_action_space = spaces.Discrete(num_actions)
        self.observation_space = self.single_observation_space
        self.action_space = self.single_action_space
        self.num_agents = 1
        self.render_mode = 'ansi'
        self.emulated = namespace(observation_dtype=np.dtype(np.float32),
                                  emulated_observation_dtype=np.dtype((np.float32, (1,))))
        self.done = True

    def render(self):
        return ''

    def close(self):
        pass


def _creator_kwargs(creator, args, kwargs, names, defaults, family):
    """Resolve keyword values the way the reference creator's signature would (defaults < positional < keyword)."""
    import inspect
    name = getattr(creator, '__name__', '')
    if family not in name.lower():
        raise APIUsageError(f'this backend only hosts ocean make_{family} envs on device (got creator {name!r})')
    vals = dict(zip(names, defaults))
    try:
        sig = inspect.signature(creator)
        for n in names:
            if n in sig.parameters and sig.parameters[n].default is not inspect._empty:
                vals[n] = sig.parameters[n].default
    except (TypeError, ValueError):
        pass
    for i, n in enumerate(names):
        if len(args) > i:
            vals[n] = args[i]
        vals[n] = kwargs.get(n, vals[n])
    return tuple(vals[n] for n in names)


class _DeviceVecEnv:
    """Backend protocol (pufferlib/vector.py) over a device-resident env family with one observation value per env: live
    buffers as torch device tensors aliased by recv() (like Serial's numpy buffers, vector.py:158-162), RESET->RECV->SEND
    state machine, ``info_mode`` 'sync' (exact per-episode info dicts, one small D2H on the sends that finish episodes) or
    'lazy' (recv() returns [], statistics through ``episode_stats``).  Subclasses provide the kernels:
    ``_k_reset(seed)``, ``_k_send(actions)``, ``_k_stats(reset)``, ``_k_infos()`` and ``_finishing_send(sends)``."""
    reset = reset
    step = step
    obs_stride = 16
    FAMILY = ''
    NAMES = ()
    DEFAULTS = ()
    SEEDED = False          # whether the env family reads the seeds async_reset is given
    AGENTS_PER_ENV = 1      # agent rows per env, env-major (PettingZoo emulation order, emulation.py:325-345)
    OBS_U8 = False          # live observation buffer of bytes (frame envs) instead of floats

    @property
    def num_envs(self):
        return self.agents_per_batch

    def __init__(self, env_creators, env_args, env_kwargs, num_envs, info_mode='sync', env_offset=0, device=None, **kwargs):
        import torch
        for k in kwargs:
            if k not in ('num_workers', 'batch_size', 'zero_copy', 'backend'):
                raise APIUsageError(f'Invalid argument: {k}')
        if len(env_creators) != num_envs:
            raise APIUsageError('env_creators must be a list of length num_envs')
        specs = {_creator_kwargs(c, a, k, self.NAMES, self.DEFAULTS, self.FAMILY)
                 for c, a, k in zip(env_creators, env_args, env_kwargs)}
        if len(specs) != 1:
            raise APIUsageError(f'obs/atn space mismatch: all envs must share one configuration, got {specs}')
        _lib.require_gpu()
        self.L = _lib.lib()
        self.device = torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')
        self.driver_env = self._spec(*specs.pop())
        self.emulated = self.driver_env.emulated
        self.env_count = num_envs
        self.agents_per_env = [self.AGENTS_PER_ENV] * num_envs
        num_envs = num_envs * self.AGENTS_PER_ENV                      # agent rows from here on
        self.agents_per_batch = self.num_agents = num_envs
        self.single_observation_space = self.driver_env.single_observation_space
        self.single_action_space = self.driver_env.single_action_space
        if hasattr(self.single_action_space, 'n'):
            self.action_space = spaces.MultiDiscrete([self.single_action_space.n] * num_envs)
        else:                                                              # MultiDiscrete per agent: joint_space tiles it (vector.py:55-68)
            self.action_space = spaces.MultiDiscrete(np.tile(np.asarray(self.single_action_space.nvec)[None], (num_envs, 1)))
        self.observation_space = spaces.Box(low=self.single_observation_space.low.min(), high=self.single_observation_space.high.max(),
                                            shape=(num_envs,) + tuple(self.single_observation_space.shape), dtype=self.single_observation_space.dtype)
        self.agent_ids = np.arange(num_envs)
        self.initialized = False
        self.flag = RESET
        self.info_mode = info_mode
        self.env_offset = int(env_offset)
        self.obs_dim = 1
        dev = self.device
        self.obs_buf = torch.zeros(num_envs, self.obs_stride, dtype=torch.uint8 if self.OBS_U8 else torch.float32, device=dev)
        self.observations = self.obs_buf[:, :1]
        self.rewards = torch.zeros(num_envs, dtype=torch.float32, device=dev)
        self.terminals_u8 = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.truncations_u8 = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self.masks_u8 = torch.ones(num_envs, dtype=torch.uint8, device=dev)
        self.terminals = self.terminals_u8.view(torch.bool)
        self.truncations = self.truncations_u8.view(torch.bool)
        self.masks = self.masks_u8.view(torch.bool)
        self._actions = torch.zeros(num_envs, dtype=torch.int64, device=dev)
        self._fin = torch.zeros(num_envs, dtype=torch.uint8, device=dev)
        self._fin_ret = torch.zeros(num_envs, dtype=torch.float64, device=dev)
        self._fin_len = torch.zeros(num_envs, dtype=torch.int32, device=dev)
        self._fin_score = torch.zeros(num_envs, dtype=torch.float64, device=dev)
        self._stats = torch.zeros(5, dtype=torch.float64, device=dev)   # 4 sums + tape underrun flag (0 for envs without a tape)
        self.infos = []
        self.sends = 0
        self._alloc_state()

    def _live(self):
        return (_lib.ptr(self.obs_buf), _lib.ptr(self.rewards), _lib.ptr(self.terminals_u8), _lib.ptr(self.truncations_u8),
                _lib.ptr(self.masks_u8))

    def _fin_ptrs(self):
        return _lib.ptr(self._fin), _lib.ptr(self._fin_ret), _lib.ptr(self._fin_len), _lib.ptr(self._fin_score)

    def async_reset(self, seed=42):
        self.flag = RECV
        seeds = make_seeds(seed, self.env_count)
        if self.SEEDED and any(s != seeds[0] + i for i, s in enumerate(seeds)):
            raise APIUsageError(f'pufferlib_amd.vector.{type(self).__name__} needs consecutive seeds (seed + env index)')
        self._k_reset(int(seeds[0]))
        self.sends = 0
        self.infos = []

    def device_send(self, actions):
        """send() for a device int64 tensor of actions, no protocol bookkeeping and no host sync (rollout loops)."""
        self._k_send(actions)
        self.sends += 1

    def send(self, actions):
        import torch
        send_precheck(self, actions)
        if not torch.is_tensor(actions):
            a = np.asarray(actions)
            if not self.initialized and not self.action_space.contains(a):
                raise APIUsageError('Actions do not match action space')
            actions = torch.as_tensor(np.ascontiguousarray(a, dtype=np.int64))
        elif not self.initialized:
            if actions.shape != (self.num_agents,) or actions.dtype not in (torch.int64, torch.int32):
                raise APIUsageError('Actions do not match action space')
        self.initialized = True
        self._actions.copy_(actions.reshape(-1), non_blocking=True)
        self.device_send(self._actions)
        self.infos = self._collect_infos() if self.info_mode == 'sync' else []

    def recv(self):
        recv_precheck(self)
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)

    def close(self):
        self.flag = CLOSE

    def _collect_infos(self):
        if not self._finishing_send(self.sends):     # every env of these families finishes on the same sends
            return []
        self._k_infos()
        fin = self._fin.cpu().numpy().astype(bool)
        ret, ln, sc = self._fin_ret.cpu().numpy(), self._fin_len.cpu().numpy(), self._fin_score.cpu().numpy()
        return [dict(episode_return=float(ret[i]), episode_length=int(ln[i]), score=float(sc[i])) for i in np.nonzero(fin)[0]]

    def episode_stats(self, reset=True):
        """(count, sum episode_return, sum episode_length, sum score) of the episodes finished since the last reset of the
        accumulators, as a device f64 tensor — what clean_pufferl.evaluate averages (clean_pufferl.py:127-137)."""
        self._k_stats(1 if reset else 0)
        return self._stats[:4]      # [4] = tape underrun flag: read by the trainer from the same buffer (stats_with_flag)

    stats_from_sums = staticmethod(episode_means)

    def stats_with_flag(self, reset=True):
        self.episode_stats(reset)
        return self._stats


def make_stochastic(p=0.7, horizon=100, **kwargs):
    """Env creator token with the signature of ocean.environment.make_stochastic (ocean/environment.py:61-64).  The
    reference's creator ignores its ``horizon`` argument and always builds Stochastic(horizon=100); so does this one."""
    return StochasticSpec(p)


class StochasticSpec(_SimpleSpec):
    """ocean.Stochastic (ocean.py:543-549)."""

    def __init__(self, p=0.7):
        super().__init__(0, 1, 2)
        self.p = float(p)
        self.horizon = 100


class Stochastic(_DeviceVecEnv):
    """Device-resident vecenv of N ocean Stochastic envs (csrc/stochastic.hip).  The env draws no random numbers, so there
    is no reset tape and seeds only matter to the policy's noise rows; it has a fused rollout kernel for the MLP policy."""
    FAMILY, NAMES, DEFAULTS = 'stochastic', ('p',), (0.7,)

    def _spec(self, p):
        return StochasticSpec(p)

    def _alloc_state(self):
        import torch
        self.p, self.horizon = self.driver_env.p, self.driver_env.horizon
        self.episode_len = self.horizon + 1
        self.state = torch.zeros(self.L.pfa_stochastic_state_bytes(self.num_agents), dtype=torch.uint8, device=self.device)

    def _finishing_send(self, sends):
        return sends % self.episode_len == self.horizon

    def _k_reset(self, seed):
        _lib.check(self.L.pfa_stochastic_async_reset(_lib.ptr(self.state), self.num_agents, *self._live(), _lib.stream_handle()),
                   'async_reset')

    def _k_send(self, actions):
        _lib.check(self.L.pfa_stochastic_send(_lib.ptr(self.state), self.num_agents, self.p, self.horizon, _lib.ptr(actions),
                                              *self._live(), _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_stochastic_episode_stats(_lib.ptr(self.state), self.num_agents, _lib.ptr(self._stats), reset,
                                                       _lib.stream_handle()), 'episode_stats')

    def _k_infos(self):
        _lib.check(self.L.pfa_stochastic_last_infos(_lib.ptr(self.state), self.num_agents, *self._fin_ptrs(), _lib.stream_handle()),
                   'last_infos')

    def fused_rollout_mlp(self, fp, experience, noise, key, stream):
        """clean_pufferl.evaluate's T-step loop as one persistent kernel (csrc/stochastic.hip)."""
        _lib.check(self.L.pfa_rollout_mlp_stochastic(_lib.ptr(self.state), self.num_agents, self.p, self.horizon, _lib.ptr(fp.flat),
                                                     C.byref(fp.dims), C.byref(experience.c), _lib.ptr(noise), C.byref(key),
                                                     self.env_offset, *self._live(), stream), 'rollout_stochastic')
        self.sends += experience.horizon


def make_memory(mem_length=2, mem_delay=2, **kwargs):
    """Env creator token with the signature of ocean.environment.make_memory (ocean/environment.py:41-44)."""
    return MemorySpec(mem_length, mem_delay)


class MemorySpec(_SimpleSpec):
    """ocean.Memory (ocean.py:80-88)."""

    def __init__(self, mem_length=2, mem_delay=2):
        super().__init__(-1, 1, 2)
        self.mem_length, self.mem_delay = int(mem_length), int(mem_delay)
        self.horizon = 2 * self.mem_length + self.mem_delay


class Memory(_DeviceVecEnv):
    """Device-resident vecenv of N ocean Memory envs (csrc/memory.hip) — the env family that needs the recurrent policy.
    The solutions of future episodes come from the tape the backend keeps ahead of the sends (numpy's process-global legacy
    stream, which does not depend on actions).  There is no fused rollout kernel for this env: clean_pufferl.evaluate steps
    it through ``device_send`` (no host sync per step)."""
    FAMILY, NAMES, DEFAULTS = 'memory', ('mem_length', 'mem_delay'), (2, 2)
    SEEDED = True

    def _spec(self, mem_length, mem_delay):
        return MemorySpec(mem_length, mem_delay)

    def _alloc_state(self):
        import torch
        self.horizon = self.driver_env.horizon
        self.episode_len = self.horizon          # H - 1 steps + the auto-reset row
        self.tape_rounds = 256
        self.cfg = _lib.MemoryConfig(self.num_agents, self.driver_env.mem_length, self.driver_env.mem_delay, self.tape_rounds)
        nbytes = self.L.pfa_memory_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise APIUsageError(self.L.pfa_last_error().decode())
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.rounds_filled = 0

    def _rounds_needed(self, upto_send):
        """Reset rounds consumed by sends 1..upto_send: every env resets on sends k*horizon."""
        return upto_send // self.episode_len

    @property
    def max_sends_per_tape(self):
        return max(1, (self.tape_rounds // 2) * self.episode_len)

    def ensure_tape(self, extra_sends):
        need = self._rounds_needed(self.sends + extra_sends)
        if need - self._rounds_needed(self.sends) > self.tape_rounds:
            raise APIUsageError(f'{extra_sends} sends need more reset rounds than the tape holds ({self.tape_rounds})')
        if need > self.rounds_filled:
            _lib.check(self.L.pfa_memory_fill_tape(_lib.ptr(self.state), C.byref(self.cfg), need - self.rounds_filled,
                                                   _lib.stream_handle()), 'fill_tape')
            self.rounds_filled = need

    def _finishing_send(self, sends):
        return sends % self.episode_len == self.episode_len - 1

    def _k_reset(self, seed):
        _lib.check(self.L.pfa_memory_async_reset(_lib.ptr(self.state), C.byref(self.cfg), seed, *self._live(), _lib.stream_handle()),
                   'async_reset')
        self.sends = 0
        self.rounds_filled = 0

    def _k_send(self, actions):
        self.ensure_tape(1)
        _lib.check(self.L.pfa_memory_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(actions), *self._live(),
                                          _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_memory_episode_stats(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._stats), reset,
                                                   _lib.stream_handle()), 'episode_stats')

    def _k_infos(self):
        _lib.check(self.L.pfa_memory_last_infos(_lib.ptr(self.state), C.byref(self.cfg), *self._fin_ptrs(), _lib.stream_handle()),
                   'last_infos')

    def debug_solutions(self):
        """(solution digits per env as a bit mask, tape underrun flag) — test introspection."""
        import torch
        bits = torch.zeros(self.num_agents, dtype=torch.int32, device=self.device)
        under = torch.zeros(1, dtype=torch.int32, device=self.device)
        _lib.check(self.L.pfa_memory_debug_solutions(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(bits), _lib.ptr(under),
                                                     _lib.stream_handle()), 'debug_solutions')
        return bits.cpu().numpy(), int(under.item())


def make_spaces(**kwargs):
    """Env creator token with the signature of ocean.environment.make_spaces (ocean/environment.py:66-69)."""
    return SpacesSpec()


class SpacesSpec:
    """What ``driver_env`` exposes for ocean.Spaces behind GymnasiumPufferEnv: the EMULATED spaces (emulation.py:96-121) — the
    Dict observation {flat: int8[5], image: f32[5,5]} as 108 bytes (flat @0, image @8, numpy align=True), the Dict action
    {flat, image} as MultiDiscrete([2, 2]) in sorted-key order."""
    ROW_BYTES = 108

    def __init__(self):
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(self.ROW_BYTES,), dtype=np.uint8)
        self.single_action_space = spaces.MultiDiscrete([2, 2])
        self.observation_space = self.single_observation_space
        self.action_space = self.single_action_space
        self.num_agents = 1
        self.render_mode = 'ansi'
        self.emulated = namespace(
            observation_dtype=np.dtype(np.uint8),
            emulated_observation_dtype=np.dtype([('flat', np.int8, (5,)), ('image', np.float32, (5, 5))], align=True))
        self.done = True

    def render(self):
        return ''

    def close(self):
        pass


class Spaces(_DeviceVecEnv):
    """Device-resident vecenv of N ocean Spaces envs (csrc/spaces.hip): structured observation (108-byte emulated rows, kept
    on device as rows of 128 floats holding the byte values — what models.Default's ``observations.float()`` computes) and a
    MultiDiscrete([2, 2]) action.  Observations come from numpy's process-global legacy generator, whose data-dependent
    stream positions the tape kernel resolves ahead of the sends; ``async_reset(seed)`` = ``np.random.seed(seed)`` (what
    clean_pufferl.seed_everything does before it) followed by the initial resets.  clean_pufferl.evaluate steps it through
    ``device_send`` with packed action words (head h in bits 4h..4h+3), no host sync per step."""
    FAMILY, NAMES, DEFAULTS = 'spaces', (), ()
    SEEDED = False
    obs_stride = 128

    def _spec(self):
        return SpacesSpec()

    def _alloc_state(self):
        import torch
        self.obs_dim = SpacesSpec.ROW_BYTES
        self.observations = self.obs_buf[:, :self.obs_dim]       # float byte values; recv() hands out the uint8 rows
        self.episode_len = 2                                      # one (terminal) step + the auto-reset row
        self.tape_rounds = 256
        self.cfg = _lib.SpacesConfig(self.num_agents, self.tape_rounds)
        nbytes = self.L.pfa_spaces_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise APIUsageError(self.L.pfa_last_error().decode())
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.rounds_filled = 1

    def _rounds_needed(self, upto_send):
        """Tape rounds consumed by async_reset (round 0) and sends 1..upto_send (every second send is a reset row)."""
        return 1 + upto_send // 2

    @property
    def max_sends_per_tape(self):
        return max(2, self.tape_rounds)          # a reset round every second send: half the ring

    def ensure_tape(self, extra_sends):
        need = self._rounds_needed(self.sends + extra_sends)
        if need - self._rounds_needed(self.sends) >= self.tape_rounds:
            raise APIUsageError(f'{extra_sends} sends need more reset rounds than the tape holds ({self.tape_rounds})')
        while need > self.rounds_filled:
            n = min(need - self.rounds_filled, self.tape_rounds // 2)
            _lib.check(self.L.pfa_spaces_fill_tape(_lib.ptr(self.state), C.byref(self.cfg), n, _lib.stream_handle()), 'fill_tape')
            self.rounds_filled += n

    def _finishing_send(self, sends):
        return sends % 2 == 1

    def _k_reset(self, seed):
        _lib.check(self.L.pfa_spaces_async_reset(_lib.ptr(self.state), C.byref(self.cfg), seed, *self._live(), _lib.stream_handle()),
                   'async_reset')
        self.sends = 0
        self.rounds_filled = 1

    def _k_send(self, actions):
        self.ensure_tape(1)
        _lib.check(self.L.pfa_spaces_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(actions), *self._live(),
                                          _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_spaces_episode_stats(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._stats), reset,
                                                   _lib.stream_handle()), 'episode_stats')

    def _k_infos(self):
        _lib.check(self.L.pfa_spaces_last_infos(_lib.ptr(self.state), C.byref(self.cfg), *self._fin_ptrs(), _lib.stream_handle()),
                   'last_infos')

    def send(self, actions):
        """Actions [N, 2] = (flat, image) per env, as the reference's emulation hands them on (emulation.py:111-121)."""
        import torch
        send_precheck(self, actions)
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.ascontiguousarray(np.asarray(actions), dtype=np.int64))
        if tuple(a.shape) != (self.num_agents, 2) or (not self.initialized and (int(a.min()) < 0 or int(a.max()) > 1)):
            raise APIUsageError('Actions do not match action space')
        self.initialized = True
        a = a.to(device=self.device, dtype=torch.int64)
        self._actions.copy_(a[:, 0] | (a[:, 1] << 4))
        self.device_send(self._actions)
        self.infos = self._collect_infos() if self.info_mode == 'sync' else []

    def recv(self):
        import torch
        recv_precheck(self)
        return (self.observations.to(torch.uint8), self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids,
                self.masks)


def make_synthetic(obs_values=160, num_actions=7, episode_length=100, obs_high=10, **kwargs):
    """Env creator token of the synthetic byte-row env (BASELINE configs[2]'s workload shape; defaults = MiniGrid's emulated row:
    160 bytes, 7 actions, episodes cut at 100 steps, minigrid/environment.py:14-48)."""
    return SyntheticSpec(obs_values, num_actions, episode_length, obs_high)


class SyntheticSpec(_SimpleSpec):
    def __init__(self, obs_values=160, num_actions=7, episode_length=100, obs_high=10):
        super().__init__(0, int(obs_high), int(num_actions))
        self.obs_values, self.num_actions = int(obs_values), int(num_actions)
        self.episode_length, self.obs_high = int(episode_length), int(obs_high)
        self.single_observation_space = spaces.Box(low=0, high=int(obs_high), shape=(self.obs_values,), dtype=np.uint8)
        self.observation_space = self.single_observation_space
        self.emulated = namespace(observation_dtype=np.dtype(np.uint8),
                                  emulated_observation_dtype=np.dtype((np.uint8, (self.obs_values,))))


class Synthetic(_DeviceVecEnv):
    """Device-resident synthetic byte-row vecenv (csrc/synth_env.hpp): observations from a counter-based generator keyed by
    (seed, global env, episode, tick), reward 1 when the action equals byte 0 of the shown row modulo the action count.  It is
    the workload of BASELINE configs[2] (MiniGrid-shaped rows + LSTM policy) — the third-party simulator itself is not part
    of the reference tree, so there is nothing to be bit-exact with on the env side."""
    FAMILY, NAMES, DEFAULTS = 'synthetic', ('obs_values', 'num_actions', 'episode_length', 'obs_high'), (160, 7, 100, 10)
    SEEDED = True

    def _spec(self, obs_values, num_actions, episode_length, obs_high):
        from .cleanrl import obs_stride_for
        self.obs_stride = obs_stride_for(int(obs_values), recurrent=True)     # 16 .. 160 floats per row
        return SyntheticSpec(obs_values, num_actions, episode_length, obs_high)

    def _alloc_state(self):
        import torch
        sp = self.driver_env
        self.obs_dim = sp.obs_values
        self.observations = self.obs_buf[:, :self.obs_dim]
        self.episode_len = sp.episode_length + 1
        self.cfg = _lib.SynthConfig(self.num_agents, sp.obs_values, self.obs_stride, sp.num_actions, sp.episode_length, sp.obs_high, 0,
                                    self.env_offset)
        nbytes = self.L.pfa_synth_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise APIUsageError(self.L.pfa_last_error().decode())
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    def _finishing_send(self, sends):
        return sends % self.episode_len == self.episode_len - 1

    def _k_reset(self, seed):
        self.cfg.seed = int(seed)
        self.cfg.env_offset = int(self.env_offset)
        _lib.check(self.L.pfa_synth_async_reset(_lib.ptr(self.state), C.byref(self.cfg), *self._live(), _lib.stream_handle()),
                   'async_reset')

    def _k_send(self, actions):
        _lib.check(self.L.pfa_synth_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(actions), *self._live(),
                                         _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_synth_episode_stats(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(self._stats), reset,
                                                  _lib.stream_handle()), 'episode_stats')

    def _k_infos(self):
        _lib.check(self.L.pfa_synth_last_infos(_lib.ptr(self.state), C.byref(self.cfg), *self._fin_ptrs(), _lib.stream_handle()),
                   'last_infos')


def make_frames(framestack=4, num_actions=4, episode_length=100, **kwargs):
    """Env creator token of the synthetic frame env (BASELINE configs[3]'s workload shape, SURVEY config C4: uint8
    (framestack, 84, 84) observations uniform in 0..255 as the Atari wrappers hand them to the NatureCNN, atari/environment.py:14-41,
    4 actions like Breakout)."""
    return FramesSpec(framestack, num_actions, episode_length)


class FramesSpec(_SimpleSpec):
    def __init__(self, framestack=4, num_actions=4, episode_length=100):
        super().__init__(0, 255, int(num_actions))
        self.framestack, self.num_actions, self.episode_length = int(framestack), int(num_actions), int(episode_length)
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(self.framestack, 84, 84), dtype=np.uint8)
        self.observation_space = self.single_observation_space
        self.emulated = namespace(observation_dtype=np.dtype(np.uint8),
                                  emulated_observation_dtype=np.dtype((np.uint8, (self.framestack, 84, 84))))


class Frames(Synthetic):
    """Device-resident synthetic frame vecenv: the byte generator of ``Synthetic`` writing uint8 (framestack, 84, 84) rows, reward 1
    when the action equals byte 0 of the shown frame modulo the action count.  Atari itself is a third-party emulator outside the
    reference tree (parity unpinned), so the workload keeps its observation / action shapes."""
    FAMILY, NAMES, DEFAULTS = 'frames', ('framestack', 'num_actions', 'episode_length'), (4, 4, 100)
    OBS_U8 = True

    def _spec(self, framestack, num_actions, episode_length):
        self.obs_stride = int(framestack) * 84 * 84            # bytes per row
        return FramesSpec(framestack, num_actions, episode_length)

    def _alloc_state(self):
        import torch
        sp = self.driver_env
        self.obs_dim = self.obs_stride
        self.observations = self.obs_buf.view(self.num_agents, sp.framestack, 84, 84)
        self.episode_len = sp.episode_length + 1
        self.cfg = _lib.SynthConfig(self.num_agents, self.obs_stride, self.obs_stride, sp.num_actions, sp.episode_length, 255, 0, self.env_offset)
        nbytes = self.L.pfa_synth_state_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise APIUsageError(self.L.pfa_last_error().decode())
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    def _k_reset(self, seed):
        self.cfg.seed = int(seed)
        self.cfg.env_offset = int(self.env_offset)
        _lib.check(self.L.pfa_frames_async_reset(_lib.ptr(self.state), C.byref(self.cfg), *self._live(), _lib.stream_handle()), 'async_reset')

    def _k_send(self, actions):
        _lib.check(self.L.pfa_frames_send(_lib.ptr(self.state), C.byref(self.cfg), _lib.ptr(actions), *self._live(), _lib.stream_handle()),
                   'send')


def make_bandit(num_actions=10, reward_scale=1, reward_noise=1, **kwargs):
    """Env creator token with the signature of ocean.environment.make_bandit (ocean/environment.py:33-37)."""
    return BanditSpec(num_actions, reward_scale, reward_noise)


class BanditSpec(_SimpleSpec):
    """ocean.Bandit (ocean.py:22-31)."""

    def __init__(self, num_actions=10, reward_scale=1, reward_noise=1):
        super().__init__(-1, 1, int(num_actions))
        self.num_actions, self.reward_scale, self.reward_noise = int(num_actions), reward_scale, reward_noise
        self.hard_fixed_seed = 42


class Bandit(_DeviceVecEnv):
    """Device-resident vecenv of N ocean Bandit envs (csrc/bandit.hip).  Every reset reseeds numpy's global generator with the
    hard fixed seed, so the solution and the reward noise env i sees are the same in every episode: they are drawn here,
    once, with numpy's own legacy RandomState — the generator the reference calls — and the kernels keep the state machine."""
    FAMILY, NAMES, DEFAULTS = 'bandit', ('num_actions', 'reward_scale', 'reward_noise'), (10, 1, 1)

    def _spec(self, num_actions, reward_scale, reward_noise):
        return BanditSpec(num_actions, reward_scale, reward_noise)

    def _alloc_state(self):
        import torch
        spec = self.driver_env
        if not 2 <= spec.num_actions <= 15:
            raise APIUsageError('the policy kernels take 1..15 actions')
        self.episode_len = 2                      # one step + the auto-reset row
        rs = np.random.RandomState(spec.hard_fixed_seed)              # ocean.py:36-42: seed(42) then randint for the solution
        self.solution = int(rs.randint(0, spec.num_actions))
        self.scale = float(spec.reward_scale)
        self.noise = None
        if spec.reward_noise != 0:                                     # ocean.py:55-57: randn() * reward_scale, env order
            table = np.array([rs.randn() * spec.reward_scale for _ in range(self.num_agents)], np.float64)
            self.noise = torch.as_tensor(table).to(self.device)
        self.state = torch.zeros(self.L.pfa_bandit_state_bytes(self.num_agents), dtype=torch.uint8, device=self.device)

    def _finishing_send(self, sends):
        return sends % 2 == 1

    def _k_reset(self, seed):
        _lib.check(self.L.pfa_bandit_async_reset(_lib.ptr(self.state), self.num_agents, *self._live(), _lib.stream_handle()),
                   'async_reset')

    def _k_send(self, actions):
        _lib.check(self.L.pfa_bandit_send(_lib.ptr(self.state), self.num_agents, self.solution, self.scale, _lib.ptr(self.noise),
                                          _lib.ptr(actions), *self._live(), _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_bandit_episode_stats(_lib.ptr(self.state), self.num_agents, _lib.ptr(self._stats), reset,
                                                   _lib.stream_handle()), 'episode_stats')

    def _k_infos(self):
        _lib.check(self.L.pfa_bandit_last_infos(_lib.ptr(self.state), self.num_agents, *self._fin_ptrs(), _lib.stream_handle()),
                   'last_infos')


def make_multiagent(**kwargs):
    """Env creator token with the signature of ocean.environment.make_multiagent (ocean/environment.py:76-79)."""
    return MultiagentSpec()


class MultiagentSpec(_SimpleSpec):
    """ocean.Multiagent behind PettingZooPufferEnv (ocean.py:148-174, emulation.py:236-306)."""

    def __init__(self):
        super().__init__(0, 1, 2)
        self.num_agents = 2
        self.possible_agents = [1, 2]


class Multiagent(_DeviceVecEnv):
    """Device-resident vecenv of N ocean Multiagent envs (csrc/multiagent.hip): two agent rows per env, env-major, so
    ``num_agents == 2 * len(env_creators)`` like the reference's Serial over PettingZooPufferEnv (vector.py:86-93).  Infos are
    the env's own ``{1: {'score': r}, 2: {'score': r}}`` per env; ``episode_stats`` holds the per-slot sums behind the
    ``1/score`` / ``2/score`` means clean_pufferl.evaluate reports."""
    FAMILY, NAMES, DEFAULTS = 'multiagent', (), ()
    AGENTS_PER_ENV = 2

    def _spec(self):
        return MultiagentSpec()

    def _alloc_state(self):
        import torch
        self.episode_len = 2
        self.state = torch.zeros(self.L.pfa_multiagent_state_bytes(self.env_count), dtype=torch.uint8, device=self.device)

    def _finishing_send(self, sends):
        return sends % 2 == 1

    def _k_reset(self, seed):
        _lib.check(self.L.pfa_multiagent_async_reset(_lib.ptr(self.state), self.env_count, *self._live(), _lib.stream_handle()),
                   'async_reset')

    def _k_send(self, actions):
        _lib.check(self.L.pfa_multiagent_send(_lib.ptr(self.state), self.env_count, _lib.ptr(actions), *self._live(),
                                              _lib.stream_handle()), 'send')

    def _k_stats(self, reset):
        _lib.check(self.L.pfa_multiagent_episode_stats(_lib.ptr(self.state), self.env_count, _lib.ptr(self._stats), reset,
                                                       _lib.stream_handle()), 'episode_stats')

    def _collect_infos(self):
        if not self._finishing_send(self.sends):
            return []
        r = self.rewards.cpu().numpy().astype(np.int64).reshape(-1, 2)      # ocean.py:207-210: score = the int reward
        return [{1: {'score': int(a)}, 2: {'score': int(b)}} for a, b in r]

    @staticmethod
    def stats_from_sums(st):
        out = {}
        if st[0] > 0:
            out['1/score'] = st[1] / st[0]
        if st[2] > 0:
            out['2/score'] = st[3] / st[2]
        return out


def make(env_creator_or_creators, env_args=None, env_kwargs=None, backend=Squared, num_envs=1, **kwargs):
    """pufferlib.vector.make (vector.py:577-637): same argument validation and error messages.

    Deliberate interface mirror of vector.py:579-624 — the reference's tests (tests/test_api.py:15-131) and user code match on these
    APIUsageError messages, so the checks keep the reference's order and wording.  Interface, not hot path."""
    if num_envs < 1:
        raise APIUsageError('num_envs must be at least 1')
    if num_envs != int(num_envs):
        raise APIUsageError('num_envs must be an integer')

    if 'num_workers' in kwargs:
        num_workers = kwargs['num_workers']
        envs_per_worker = num_envs / num_workers
        if envs_per_worker != int(envs_per_worker):
            raise APIUsageError('num_envs must be divisible by num_workers')
        if 'batch_size' in kwargs:
            batch_size = kwargs['batch_size']
            if batch_size is None:
                batch_size = num_envs
            if batch_size % envs_per_worker != 0:
                raise APIUsageError('batch_size must be divisible by (num_envs / num_workers)')

    if env_args is None:
        env_args = []
    if env_kwargs is None:
        env_kwargs = {}

    if not isinstance(env_creator_or_creators, (list, tuple)):
        env_creators = [env_creator_or_creators] * num_envs
        env_args = [env_args] * num_envs
        env_kwargs = [env_kwargs] * num_envs
    else:
        env_creators = env_creator_or_creators

    if len(env_creators) != num_envs:
        raise APIUsageError('env_creators must be a list of length num_envs')
    if len(env_args) != num_envs:
        raise APIUsageError('env_args must be a list of length num_envs')
    if len(env_kwargs) != num_envs:
        raise APIUsageError('env_kwargs must be a list of length num_envs')

    for i in range(num_envs):
        if not callable(env_creators[i]):
            raise APIUsageError('env_creators must be a list of callables')
        if not isinstance(env_args[i], (list, tuple)):
            raise APIUsageError('env_args must be a list of lists or tuples')
        if not isinstance(env_kwargs[i], (dict, Namespace)):
            raise APIUsageError('env_kwargs must be a list of dictionaries')

    for k in kwargs:
        if k not in ['num_workers', 'batch_size', 'zero_copy', 'backend', 'obs_stride', 'info_mode', 'env_offset', 'device']:
            raise APIUsageError(f'Invalid argument: {k}')

    return backend(env_creators, env_args, env_kwargs, num_envs, **kwargs)
