"""ctypes bindings of oracle/puffer_oracle.c (CPU ORACLE — test infrastructure only)."""
import ctypes as C

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build_oracle()      # no-op when oracle/_build/libpuffer_oracle.so is newer than its sources
        L = C.CDLL(path)
        L.po_squared_create.restype = C.c_void_p
        L.po_squared_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.po_squared_free.argtypes = [C.c_void_p]
        L.po_squared_async_reset.argtypes = [C.c_void_p, C.c_int64]
        L.po_squared_send.argtypes = [C.c_void_p, C.c_void_p]
        L.po_squared_obs_size.argtypes = [C.c_void_p]
        for name, rt in [('observations', C.c_float), ('rewards', C.c_float), ('terminals', C.c_uint8),
                         ('truncations', C.c_uint8), ('masks', C.c_uint8), ('info_env', C.c_int32),
                         ('info_return', C.c_double), ('info_length', C.c_int32), ('info_score', C.c_double)]:
            f = getattr(L, 'po_squared_' + name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.po_squared_num_infos.argtypes = [C.c_void_p]
        L.po_squared_targets.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.po_squared_stream_pos.restype = C.c_uint64
        L.po_squared_stream_pos.argtypes = [C.c_void_p]
        L.po_mt_seed_numpy.argtypes = [C.c_void_p, C.c_uint32]
        L.po_np_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.po_np_randint.restype = C.c_uint32
        L.po_np_randint.argtypes = [C.c_void_p, C.c_uint32]
        L.po_np_randn.restype = C.c_double
        L.po_np_randn.argtypes = [C.c_void_p]
        L.po_np_randint_i8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.po_np_sum_f32.restype = C.c_float
        L.po_np_sum_f32.argtypes = [C.c_void_p, C.c_int]
        L.po_spaces_create.restype = C.c_void_p
        L.po_spaces_create.argtypes = [C.c_int]
        L.po_spaces_free.argtypes = [C.c_void_p]
        L.po_spaces_seed_global.argtypes = [C.c_void_p, C.c_uint32]
        L.po_spaces_async_reset.argtypes = [C.c_void_p]
        L.po_spaces_send.argtypes = [C.c_void_p, C.c_void_p]
        for name, rt in (('observations', C.c_uint8), ('rewards', C.c_float), ('terminals', C.c_uint8), ('info_score', C.c_double)):
            f = getattr(L, 'po_spaces_' + name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.po_spaces_num_infos.argtypes = [C.c_void_p]
        L.po_bandit_create.restype = C.c_void_p
        L.po_bandit_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.po_bandit_free.argtypes = [C.c_void_p]
        L.po_bandit_async_reset.argtypes = [C.c_void_p, C.c_int64]
        L.po_bandit_send.argtypes = [C.c_void_p, C.c_void_p]
        for name, rt in (('observations', C.c_float), ('rewards', C.c_float), ('terminals', C.c_uint8), ('masks', C.c_uint8),
                         ('info_return', C.c_double), ('info_score', C.c_double)):
            f = getattr(L, 'po_bandit_' + name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.po_bandit_num_infos.argtypes = [C.c_void_p]
        L.po_bandit_solution.argtypes = [C.c_void_p]
        L.po_memory_create.restype = C.c_void_p
        L.po_memory_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.po_memory_free.argtypes = [C.c_void_p]
        L.po_memory_async_reset.argtypes = [C.c_void_p, C.c_int64]
        L.po_memory_send.argtypes = [C.c_void_p, C.c_void_p]
        L.po_memory_solution.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for name, rt in (('observations', C.c_float), ('rewards', C.c_float), ('terminals', C.c_uint8), ('truncations', C.c_uint8),
                         ('masks', C.c_uint8), ('info_env', C.c_int32), ('info_return', C.c_double), ('info_length', C.c_int32),
                         ('info_score', C.c_double)):
            f = getattr(L, 'po_memory_' + name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.po_memory_num_infos.argtypes = [C.c_void_p]
        L.po_stochastic_create.restype = C.c_void_p
        L.po_stochastic_create.argtypes = [C.c_int, C.c_double, C.c_int]
        L.po_stochastic_free.argtypes = [C.c_void_p]
        L.po_stochastic_async_reset.argtypes = [C.c_void_p, C.c_int64]
        L.po_stochastic_send.argtypes = [C.c_void_p, C.c_void_p]
        for name, rt in (('observations', C.c_float), ('rewards', C.c_float), ('terminals', C.c_uint8), ('truncations', C.c_uint8),
                         ('masks', C.c_uint8), ('info_env', C.c_int32), ('info_return', C.c_double), ('info_length', C.c_int32),
                         ('info_score', C.c_double)):
            f = getattr(L, 'po_stochastic_' + name)
            f.restype = C.POINTER(rt)
            f.argtypes = [C.c_void_p]
        L.po_stochastic_num_infos.argtypes = [C.c_void_p]
        L.po_stochastic_reward.restype = C.c_double
        L.po_stochastic_reward.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.po_compute_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.po_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.po_mt_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.po_mt_u32.restype = C.c_uint32
        L.po_mt_u32.argtypes = [C.c_void_p]
        L.po_mt_randbelow.restype = C.c_uint32
        L.po_mt_randbelow.argtypes = [C.c_void_p, C.c_uint32]
        L.po_mt_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class MT(C.Structure):
    """CPython ``random.Random`` restated (seed/getrandbits/_randbelow/sample)."""
    _fields_ = [('mt', C.c_uint32 * 624), ('idx', C.c_int32), ('count', C.c_uint64)]

    def seed(self, s):
        lib().po_mt_seed(C.byref(self), abs(int(s)))
        return self

    def u32(self):
        return int(lib().po_mt_u32(C.byref(self)))

    def randbelow(self, n):
        return int(lib().po_mt_randbelow(C.byref(self), n))

    def sample(self, n, k):
        out = (C.c_int * max(k, 1))()
        lib().po_mt_sample(C.byref(self), n, k, out)
        return list(out[:k])

    def state_words(self):
        return np.ctypeslib.as_array(self.mt).copy(), int(self.idx)


class SquaredSerial:
    """``pufferlib.vector.Serial`` over ``make_squared`` envs, restated in C.

    Same call protocol as the reference backend (vector.py:112-162): async_reset / recv / send.
    ``recv`` returns live views of the C buffers, like Serial returns its numpy buffers.
    """

    def __init__(self, num_envs, distance_to_target=3, num_targets=1):
        self.L = lib()
        self.num_envs = num_envs
        self.d = distance_to_target
        self.nt = 4 * distance_to_target if num_targets == -1 else num_targets
        self.g = 2 * distance_to_target + 1
        self.h = self.L.po_squared_create(num_envs, distance_to_target, num_targets)
        n, s = num_envs, self.L.po_squared_obs_size(self.h)
        as_arr = np.ctypeslib.as_array
        self.observations = as_arr(self.L.po_squared_observations(self.h), (n, s)).reshape(n, self.g, self.g)
        self.rewards = as_arr(self.L.po_squared_rewards(self.h), (n,))
        self.terminals = as_arr(self.L.po_squared_terminals(self.h), (n,)).view(bool)
        self.truncations = as_arr(self.L.po_squared_truncations(self.h), (n,)).view(bool)
        self.masks = as_arr(self.L.po_squared_masks(self.h), (n,)).view(bool)
        self.agent_ids = np.arange(n)

    def __del__(self):
        try:
            self.L.po_squared_free(self.h)
        except Exception:
            pass

    def _infos(self):
        k = self.L.po_squared_num_infos(self.h)
        if k == 0:
            return []
        env = np.ctypeslib.as_array(self.L.po_squared_info_env(self.h), (k,))
        ret = np.ctypeslib.as_array(self.L.po_squared_info_return(self.h), (k,))
        ln = np.ctypeslib.as_array(self.L.po_squared_info_length(self.h), (k,))
        sc = np.ctypeslib.as_array(self.L.po_squared_info_score(self.h), (k,))
        return [dict(episode_return=float(ret[i]), episode_length=int(ln[i]), score=float(sc[i]), _env=int(env[i]))
                for i in range(k)]

    def async_reset(self, seed=42):
        self.L.po_squared_async_reset(self.h, int(seed))
        self.infos = self._infos()

    def send(self, actions):
        a = np.ascontiguousarray(np.asarray(actions), dtype=np.int64)
        assert a.shape == (self.num_envs,)
        self.L.po_squared_send(self.h, a.ctypes.data)
        self.infos = self._infos()

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos,
                self.agent_ids, self.masks)

    def targets(self, env):
        out = (C.c_int * max(self.nt, 1))()
        self.L.po_squared_targets(self.h, env, out)
        return [(c // self.g, c % self.g) for c in out[:self.nt] if c >= 0]

    def stream_pos(self):
        return int(self.L.po_squared_stream_pos(self.h))


def compute_gae(dones, values, rewards, gamma, gae_lambda):
    """c_gae.compute_gae restated (c_gae.pyx:11-32)."""
    dones = np.ascontiguousarray(dones, dtype=np.float32)
    values = np.ascontiguousarray(values, dtype=np.float32)
    rewards = np.ascontiguousarray(rewards, dtype=np.float32)
    adv = np.empty(len(rewards), dtype=np.float32)
    lib().po_compute_gae(dones.ctypes.data, values.ctypes.data, rewards.ctypes.data, adv.ctypes.data,
                         len(rewards), gamma, gae_lambda)
    return adv


def philox4x32_10(ctr, key):
    c = np.ascontiguousarray(ctr, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().po_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def philox4x32_10_bulk(c0, c1, c2, c3, k0, k1):
    """The same 10 rounds on uint32 arrays (numpy, vectorised): returns the four output word arrays.  Checked word for word
    against po_philox4x32_10 in tests/test_oracle_golden.py."""
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) & 0xFFFFFFFF for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def philox_exp_noise(seed, step, rows, cols, row_offset=0):
    """Exp(1) action noise [rows][cols] of rollout step `step` as pufferlib_amd/csrc/philox.hpp defines the stream:
    key = (seed lo, seed hi), counter = (global row, column / 4, step lo, step hi), u = ((w >> 8) + 0.5) * 2^-24, q = -log(u)."""
    nj = (cols + 3) // 4
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row_offset))[:, None]
    j = np.arange(nj, dtype=np.uint64)[None, :]
    w = philox4x32_10_bulk(r, j, np.uint64(step & 0xFFFFFFFF), np.uint64(step >> 32), seed & 0xFFFFFFFF, seed >> 32)
    words = np.stack(w, axis=-1).reshape(rows, nj * 4)[:, :cols]
    u = ((words >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return -np.log(u)


class StochasticSerial:
    """``pufferlib.vector.Serial`` over ``make_stochastic`` envs (ocean/environment.py:61-64: horizon 100), restated in C.
    Same protocol and buffer aliasing as SquaredSerial."""

    def __init__(self, num_envs, p=0.7, horizon=100):
        self.L = lib()
        self.num_envs = num_envs
        self.p, self.horizon = p, horizon
        self.h = self.L.po_stochastic_create(num_envs, p, horizon)
        n = num_envs
        as_arr = np.ctypeslib.as_array
        self.observations = as_arr(self.L.po_stochastic_observations(self.h), (n, 1))
        self.rewards = as_arr(self.L.po_stochastic_rewards(self.h), (n,))
        self.terminals = as_arr(self.L.po_stochastic_terminals(self.h), (n,)).view(bool)
        self.truncations = as_arr(self.L.po_stochastic_truncations(self.h), (n,)).view(bool)
        self.masks = as_arr(self.L.po_stochastic_masks(self.h), (n,)).view(bool)
        self.agent_ids = np.arange(n)
        self.infos = []

    def __del__(self):
        try:
            self.L.po_stochastic_free(self.h)
        except Exception:
            pass

    def _infos(self):
        k = self.L.po_stochastic_num_infos(self.h)
        if k == 0:
            return []
        env = np.ctypeslib.as_array(self.L.po_stochastic_info_env(self.h), (k,))
        ret = np.ctypeslib.as_array(self.L.po_stochastic_info_return(self.h), (k,))
        ln = np.ctypeslib.as_array(self.L.po_stochastic_info_length(self.h), (k,))
        sc = np.ctypeslib.as_array(self.L.po_stochastic_info_score(self.h), (k,))
        return [dict(episode_return=float(ret[i]), episode_length=int(ln[i]), score=float(sc[i]), _env=int(env[i]))
                for i in range(k)]

    def async_reset(self, seed=42):
        self.L.po_stochastic_async_reset(self.h, int(seed))
        self.infos = self._infos()

    def send(self, actions):
        a = np.ascontiguousarray(np.asarray(actions), dtype=np.int64)
        assert a.shape == (self.num_envs,)
        self.L.po_stochastic_send(self.h, a.ctypes.data)
        self.infos = self._infos()

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)


def stochastic_reward(p, tick, count, action):
    """(reward, proximity) of ocean.Stochastic.step (ocean.py:566-580) in the C restatement."""
    prox = C.c_double(0.0)
    r = lib().po_stochastic_reward(float(p), int(tick), int(count), int(action), C.byref(prox))
    return r, prox.value


class MemorySerial:
    """``pufferlib.vector.Serial`` over ``make_memory`` envs (ocean/environment.py:41-44), restated in C (numpy's global legacy
    stream included).  Same protocol and buffer aliasing as SquaredSerial."""

    def __init__(self, num_envs, mem_length=2, mem_delay=2):
        self.L = lib()
        self.num_envs = num_envs
        self.mem_length, self.mem_delay, self.horizon = mem_length, mem_delay, 2 * mem_length + mem_delay
        self.h = self.L.po_memory_create(num_envs, mem_length, mem_delay)
        assert self.h, 'horizon out of range'
        n = num_envs
        as_arr = np.ctypeslib.as_array
        self.observations = as_arr(self.L.po_memory_observations(self.h), (n, 1))
        self.rewards = as_arr(self.L.po_memory_rewards(self.h), (n,))
        self.terminals = as_arr(self.L.po_memory_terminals(self.h), (n,)).view(bool)
        self.truncations = as_arr(self.L.po_memory_truncations(self.h), (n,)).view(bool)
        self.masks = as_arr(self.L.po_memory_masks(self.h), (n,)).view(bool)
        self.agent_ids = np.arange(n)
        self.infos = []

    def __del__(self):
        try:
            self.L.po_memory_free(self.h)
        except Exception:
            pass

    def _infos(self):
        k = self.L.po_memory_num_infos(self.h)
        if k == 0:
            return []
        env = np.ctypeslib.as_array(self.L.po_memory_info_env(self.h), (k,))
        ret = np.ctypeslib.as_array(self.L.po_memory_info_return(self.h), (k,))
        ln = np.ctypeslib.as_array(self.L.po_memory_info_length(self.h), (k,))
        sc = np.ctypeslib.as_array(self.L.po_memory_info_score(self.h), (k,))
        return [dict(episode_return=float(ret[i]), episode_length=int(ln[i]), score=float(sc[i]), _env=int(env[i]))
                for i in range(k)]

    def async_reset(self, seed=42):
        self.L.po_memory_async_reset(self.h, int(seed))
        self.infos = self._infos()

    def send(self, actions):
        a = np.ascontiguousarray(np.asarray(actions), dtype=np.int64)
        assert a.shape == (self.num_envs,)
        self.L.po_memory_send(self.h, a.ctypes.data)
        self.infos = self._infos()

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)

    def solution(self, env):
        out = (C.c_float * self.horizon)()
        self.L.po_memory_solution(self.h, env, out)
        return np.array(out[:], np.float32)


class BanditSerial:
    """``pufferlib.vector.Serial`` over ``make_bandit`` envs (ocean/environment.py:33-37), restated in C incl. numpy's legacy
    randint / gauss on the process-global generator."""

    def __init__(self, num_envs, num_actions=10, reward_scale=1, reward_noise=1):
        self.L = lib()
        self.num_envs = num_envs
        self.h = self.L.po_bandit_create(num_envs, num_actions, float(reward_scale), float(reward_noise))
        n = num_envs
        as_arr = np.ctypeslib.as_array
        self.observations = as_arr(self.L.po_bandit_observations(self.h), (n, 1))
        self.rewards = as_arr(self.L.po_bandit_rewards(self.h), (n,))
        self.terminals = as_arr(self.L.po_bandit_terminals(self.h), (n,)).view(bool)
        self.truncations = np.zeros(n, bool)
        self.masks = as_arr(self.L.po_bandit_masks(self.h), (n,)).view(bool)
        self.agent_ids = np.arange(n)
        self.infos = []

    def __del__(self):
        try:
            self.L.po_bandit_free(self.h)
        except Exception:
            pass

    def _infos(self):
        k = self.L.po_bandit_num_infos(self.h)
        if k == 0:
            return []
        ret = np.ctypeslib.as_array(self.L.po_bandit_info_return(self.h), (k,))
        sc = np.ctypeslib.as_array(self.L.po_bandit_info_score(self.h), (k,))
        return [dict(episode_return=float(ret[i]), episode_length=1, score=float(sc[i])) for i in range(k)]

    @property
    def solution(self):
        return self.L.po_bandit_solution(self.h)

    def async_reset(self, seed=42):
        self.L.po_bandit_async_reset(self.h, int(seed))
        self.infos = []

    def send(self, actions):
        a = np.ascontiguousarray(np.asarray(actions), dtype=np.int64)
        assert a.shape == (self.num_envs,)
        self.L.po_bandit_send(self.h, a.ctypes.data)
        self.infos = self._infos()

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)


class MultiagentSerial:
    """``pufferlib.vector.Serial`` over ``make_multiagent`` envs (ocean/environment.py:76-79), restated in numpy: two agent rows
    per env in env-major order (emulation.py:325-345), row 2e observes 0 and scores with action 0, row 2e+1 observes 1 and
    scores with action 1 (ocean.py:187-205); every step is terminal for both (ocean.py:160-163), so the next send is the
    env's reset row (emulation.py:283-284, vector.py:147-149).  Infos: the env's own per-agent dicts (ocean.py:207-210;
    MultiagentEpisodeStats leaves them untouched, postprocess.py:159-177)."""

    def __init__(self, num_envs):
        self.num_envs = num_envs
        n = 2 * num_envs
        self.observations = np.zeros((n, 1), np.float32)
        self.rewards = np.zeros(n, np.float32)
        self.terminals = np.zeros(n, bool)
        self.truncations = np.zeros(n, bool)
        self.masks = np.ones(n, bool)
        self.agent_ids = np.arange(n)
        self.done = np.zeros(num_envs, bool)
        self.infos = []

    def async_reset(self, seed=42):
        self.observations[:, 0] = np.arange(2 * self.num_envs) % 2
        self.rewards[:] = 0
        self.terminals[:] = False
        self.done[:] = False
        self.infos = []

    def send(self, actions):
        a = np.asarray(actions).reshape(self.num_envs, 2)
        stepped = ~self.done
        score = (a == np.array([0, 1])) & stepped[:, None]
        self.rewards[:] = score.reshape(-1)
        self.terminals[:] = np.repeat(stepped, 2)
        self.infos = [{1: {'score': int(s[0])}, 2: {'score': int(s[1])}} for s, st in zip(score, stepped) if st]
        self.done = stepped.copy()

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)


class SpacesSerial:
    """``pufferlib.vector.Serial`` over ``make_spaces`` envs (ocean/environment.py:66-69) restated in C: Dict observation emulated
    to 108-byte rows, Dict action emulated to MultiDiscrete([2, 2]); every reset draws ``randn(5, 5)`` and
    ``randint(-1, 2, (5,), dtype=int8)`` from numpy's process-global legacy generator, which the env never seeds (ocean.py:380) —
    ``global_seed`` is the ``np.random.seed`` the trainer issued before (clean_pufferl.py:596-600)."""

    def __init__(self, num_envs, global_seed=1):
        self.L = lib()
        self.num_envs = num_envs
        self.h = self.L.po_spaces_create(num_envs)
        self.L.po_spaces_seed_global(self.h, int(global_seed))
        n = num_envs
        as_arr = np.ctypeslib.as_array
        self.observations = as_arr(self.L.po_spaces_observations(self.h), (n, 108))
        self.rewards = as_arr(self.L.po_spaces_rewards(self.h), (n,))
        self.terminals = as_arr(self.L.po_spaces_terminals(self.h), (n,)).view(bool)
        self.truncations = np.zeros(n, bool)
        self.masks = np.ones(n, bool)
        self.agent_ids = np.arange(n)
        self.infos = []

    def __del__(self):
        try:
            self.L.po_spaces_free(self.h)
        except Exception:
            pass

    def async_reset(self, seed=42):
        self.L.po_spaces_async_reset(self.h)      # the seed never reaches the env's generator
        self.infos = []

    def send(self, actions):
        a = np.ascontiguousarray(np.asarray(actions), dtype=np.int64)
        assert a.shape == (self.num_envs, 2)
        self.L.po_spaces_send(self.h, a.ctypes.data)
        k = self.L.po_spaces_num_infos(self.h)
        sc = np.ctypeslib.as_array(self.L.po_spaces_info_score(self.h), (max(k, 1),))[:k]
        self.infos = [dict(score=float(s), episode_return=float(s), episode_length=1) for s in sc]

    def recv(self):
        return (self.observations, self.rewards, self.terminals, self.truncations, self.infos, self.agent_ids, self.masks)
