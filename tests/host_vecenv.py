"""A HOST vecenv for the tests: the reference's backend protocol (vector.py:112-162) over the C oracle's SquaredSerial, plus
the attributes policies and create() read.  ``order`` permutes the rows every recv() hands out (and expects back in send()),
standing in for backends whose batches arrive out of env order."""
import numpy as np


class HostSquared:
    def __init__(self, num_envs, distance_to_target=3, num_targets=1, order='natural', seed_perm=0):
        from oracle import c_oracle
        from pufferlib_amd import vector
        self.inner = c_oracle.SquaredSerial(num_envs, distance_to_target, num_targets)
        self.driver_env = vector.SquaredSpec(distance_to_target, num_targets)
        self.single_observation_space = self.driver_env.single_observation_space
        self.single_action_space = self.driver_env.single_action_space
        self.num_envs = self.num_agents = self.agents_per_batch = num_envs
        self.agent_ids = np.arange(num_envs)
        self.emulated = True
        self.order = order
        self.rng = np.random.RandomState(seed_perm)
        self.perm = np.arange(num_envs)
        self.sends = 0

    def async_reset(self, seed=42):
        self.inner.async_reset(seed)

    def recv(self):
        o, r, d, t, infos, ids, mask = self.inner.recv()
        if self.order == 'reversed':
            self.perm = np.arange(self.num_envs)[::-1].copy()
        elif self.order == 'shuffled':
            self.perm = self.rng.permutation(self.num_envs)
        infos = [{k: v for k, v in i.items() if not k.startswith('_')} for i in infos]   # oracle bookkeeping keys
        p = self.perm
        return o[p].copy(), r[p].copy(), d[p].copy(), t[p].copy(), infos, ids[p].copy(), mask[p].copy()

    def send(self, actions):
        a = np.empty(self.num_envs, np.int64)
        a[self.perm] = np.asarray(actions)
        self.inner.send(a)
        self.sends += 1

    def close(self):
        pass


class HostSquaredPool:
    """EnvPool-style host vecenv (vector.py:218-447 semantics in miniature): `workers` independent groups of envs, each
    recv() hands out ONE group's rows (agents_per_batch = num_envs / workers < num_agents), send() steps that group only,
    groups take turns.  Each group is its own C-oracle SquaredSerial, i.e. has its own `random` stream like a worker process."""

    def __init__(self, num_envs, workers=2, distance_to_target=3, num_targets=1, schedule=None):
        from oracle import c_oracle
        from pufferlib_amd import vector
        assert num_envs % workers == 0
        self.schedule = list(schedule) if schedule is not None else list(range(workers))   # which group answers recv() k (cyclic)
        self.per = num_envs // workers
        self.groups = [c_oracle.SquaredSerial(self.per, distance_to_target, num_targets) for _ in range(workers)]
        self.driver_env = vector.SquaredSpec(distance_to_target, num_targets)
        self.single_observation_space = self.driver_env.single_observation_space
        self.single_action_space = self.driver_env.single_action_space
        self.num_envs = self.num_agents = num_envs
        self.agents_per_batch = self.per
        self.emulated = True
        self.turn = 0
        self.seed = None

    def async_reset(self, seed=42):
        self.seed = seed
        for w, g in enumerate(self.groups):
            g.async_reset(seed + w * self.per)          # make_seeds: env i gets seed + i (vector.py:639-641)
        self.turn = 0

    def recv(self):
        w = self.schedule[self.turn % len(self.schedule)]
        o, r, d, t, infos, ids, mask = self.groups[w].recv()
        infos = [{k: v for k, v in i.items() if not k.startswith('_')} for i in infos]
        return o.copy(), r.copy(), d.copy(), t.copy(), infos, ids + w * self.per, mask.copy()

    def send(self, actions):
        self.groups[self.schedule[self.turn % len(self.schedule)]].send(np.asarray(actions, np.int64))
        self.turn += 1

    def close(self):
        pass


class SpacesReplay:
    """``pufferlib.vector.Serial`` over ``make_spaces`` envs (ocean.py:356-404 behind GymnasiumPufferEnv + EpisodeStats) with the
    OBSERVATIONS played back from a recording of the unmodified reference (tests/golden/ppo_spaces.npz): the env draws them from
    numpy's global generator without looking at the actions, so only the reward rule is computed here.  Rows are the emulated
    108-byte structs {flat: int8[5] @0, image: f32[5,5] @8}; actions are MultiDiscrete([2, 2]) = (flat, image) (Dict keys in
    sorted order, emulation.py:111-121).  Every step is terminal, the next send is the reset row."""

    def __init__(self, obs_rounds):
        from pufferlib_amd import spaces
        self.obs_rounds = np.asarray(obs_rounds, np.uint8)            # [recvs][N][108]
        n = self.obs_rounds.shape[1]
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(108,), dtype=np.uint8)
        self.single_action_space = spaces.MultiDiscrete([2, 2])
        self.driver_env = self
        self.num_envs = self.num_agents = self.agents_per_batch = n
        self.agent_ids = np.arange(n)
        self.observations = np.zeros((n, 108), np.uint8)            # what ppo_torch.Trainer sizes its buffers from
        self.emulated = True
        self.k = 0
        self.rewards = np.zeros(n, np.float32)
        self.terminals = np.zeros(n, bool)
        self.infos = []

    def async_reset(self, seed=42):
        self.k = 0
        self.rewards[:] = 0
        self.terminals[:] = False
        self.infos = []

    def recv(self):
        n = self.num_agents
        return (self.obs_rounds[self.k].copy(), self.rewards.copy(), self.terminals.copy(), np.zeros(n, bool), self.infos,
                self.agent_ids, np.ones(n, bool))

    def send(self, actions):
        a = np.asarray(actions).reshape(self.num_agents, 2)
        rows = self.obs_rounds[self.k]
        self.k += 1
        if self.terminals.any():                                     # vector.py:147-149: reset row, action ignored
            self.rewards[:] = 0
            self.terminals[:] = False
            self.infos = []
            return
        self.infos = []
        for e in range(self.num_agents):
            flat = rows[e, 0:5].view(np.int8)
            image = rows[e, 8:108].view(np.float32).reshape(5, 5)
            reward = 0                                               # ocean.py:392-399
            if (np.sum(image) > 0) == a[e, 1]:
                reward += 0.5
            if (np.sum(flat) > 0) == a[e, 0]:
                reward += 0.5
            self.rewards[e] = reward
            self.infos.append(dict(score=reward, episode_return=reward, episode_length=1))
        self.terminals[:] = True

    def close(self):
        pass


class HostMultiHead:
    """A small deterministic host vecenv with a MultiDiscrete action space (test stand-in for the reference's emulated Dict /
    Tuple action envs): observations are seeded N(0, 1) noise, head h is rewarded for naming the equal-probability bin obs[h]
    falls in (nvec[h] bins), an env terminates every ``period`` steps and the following send is its reset row (vector.py:147-149)."""

    def __init__(self, num_envs, nvec, obs_dim=20, period=5, seed=0):
        from pufferlib_amd import spaces
        self.nvec = [int(x) for x in nvec]
        self.single_observation_space = spaces.Box(low=-10, high=10, shape=(obs_dim,), dtype=np.float32)
        self.single_action_space = spaces.MultiDiscrete(self.nvec)
        self.driver_env = self
        self.num_envs = self.num_agents = self.agents_per_batch = num_envs
        self.agent_ids = np.arange(num_envs)
        self.emulated = True
        self.period, self.seed = period, seed
        from statistics import NormalDist
        self.edges = [np.array([NormalDist().inv_cdf(i / k) for i in range(1, k)]) for k in self.nvec]
        self.observations = np.zeros((num_envs, obs_dim), np.float32)
        self.async_reset()

    def _draw(self):
        self.observations[:] = self.rng.randn(*self.observations.shape).astype(np.float32)

    def async_reset(self, seed=42):
        n = self.num_agents
        self.rng = np.random.RandomState(self.seed)
        self.tick = np.arange(n) % self.period                       # staggered episode ends
        self.done = np.zeros(n, bool)
        self.rewards = np.zeros(n, np.float32)
        self.terminals = np.zeros(n, bool)
        self.ep_return = np.zeros(n, np.float64)
        self.infos = []
        self._draw()

    def recv(self):
        n = self.num_agents
        return (self.observations.copy(), self.rewards.copy(), self.terminals.copy(), np.zeros(n, bool), self.infos, self.agent_ids,
                np.ones(n, bool))

    def send(self, actions):
        a = np.asarray(actions).reshape(self.num_agents, len(self.nvec))
        target = np.stack([np.digitize(self.observations[:, h], self.edges[h]) for h in range(len(self.nvec))], 1)
        self.infos = []
        old_done = self.done.copy()
        hit = (a == target).mean(1)
        self._draw()
        for e in range(self.num_agents):
            if old_done[e]:
                self.rewards[e], self.terminals[e], self.done[e], self.tick[e], self.ep_return[e] = 0, False, False, 0, 0
                continue
            self.rewards[e] = hit[e]
            self.ep_return[e] += float(np.float32(hit[e]))
            self.tick[e] += 1
            self.terminals[e] = self.done[e] = self.tick[e] >= self.period
            if self.done[e]:
                self.infos.append(dict(episode_return=self.ep_return[e], episode_length=int(self.tick[e]),
                                       score=self.ep_return[e] / self.tick[e]))

    def close(self):
        pass


class HostByteRows:
    """Host vecenv with MiniGrid-shaped observations (SURVEY config C3: the aligned struct {direction: int64, image:
    uint8[7,7,3]} = 155 -> 160 bytes per row, which models.Default reads as 160 floats, models.py:50) and Discrete(7) actions.
    Bytes are seeded noise in [0, 10]; the rewarded action is byte 8 modulo 7; episodes of ``period`` steps with the reset row
    after the terminal (vector.py:147-149)."""

    def __init__(self, num_envs, row_bytes=160, num_actions=7, period=9, seed=0):
        from pufferlib_amd import spaces
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(row_bytes,), dtype=np.uint8)
        self.single_action_space = spaces.Discrete(num_actions)
        self.driver_env = self
        self.num_envs = self.num_agents = self.agents_per_batch = num_envs
        self.agent_ids = np.arange(num_envs)
        self.emulated = True
        self.num_actions, self.period, self.seed = num_actions, period, seed
        self.observations = np.zeros((num_envs, row_bytes), np.uint8)
        self.async_reset()

    def async_reset(self, seed=42):
        n = self.num_agents
        self.rng = np.random.RandomState(self.seed)
        self.tick = np.arange(n) % self.period
        self.done = np.zeros(n, bool)
        self.rewards = np.zeros(n, np.float32)
        self.terminals = np.zeros(n, bool)
        self.ep_return = np.zeros(n, np.float64)
        self.infos = []
        self.observations[:] = self.rng.randint(0, 11, self.observations.shape)

    def recv(self):
        n = self.num_agents
        return (self.observations.copy(), self.rewards.copy(), self.terminals.copy(), np.zeros(n, bool), self.infos, self.agent_ids,
                np.ones(n, bool))

    def send(self, actions):
        a = np.asarray(actions).reshape(self.num_agents)
        hit = (a == self.observations[:, 8] % self.num_actions).astype(np.float32)
        self.observations[:] = self.rng.randint(0, 11, self.observations.shape)
        self.infos = []
        for e in range(self.num_agents):
            if self.done[e]:
                self.rewards[e], self.terminals[e], self.done[e], self.tick[e], self.ep_return[e] = 0, False, False, 0, 0
                continue
            self.rewards[e] = hit[e]
            self.ep_return[e] += float(hit[e])
            self.tick[e] += 1
            self.terminals[e] = self.done[e] = self.tick[e] >= self.period
            if self.done[e]:
                self.infos.append(dict(episode_return=self.ep_return[e], episode_length=int(self.tick[e]),
                                       score=self.ep_return[e] / self.tick[e]))

    def close(self):
        pass


class HostFrames:
    """Host vecenv with Atari-shaped observations (uint8 (4, 84, 84), 4 actions): numpy frames, reward 1 when the action equals
    byte 0 of the shown frame modulo 4, episodes of `episode_length` steps, Serial's protocol (the send after a terminal row is
    the reset row; finished episodes report an info dict)."""

    def __init__(self, num_envs, episode_length=5, seed=0):
        from pufferlib_amd import spaces
        self.single_observation_space = spaces.Box(low=0, high=255, shape=(4, 84, 84), dtype=np.uint8)
        self.single_action_space = spaces.Discrete(4)
        self.driver_env = self
        self.num_envs = self.num_agents = self.agents_per_batch = num_envs
        self.emulated = True
        self.episode_length = episode_length
        self.rng = np.random.RandomState(seed)
        self.obs = np.zeros((num_envs, 4, 84, 84), np.uint8)
        self.rew = np.zeros(num_envs, np.float32)
        self.term = np.zeros(num_envs, bool)
        self.tick = np.zeros(num_envs, int)
        self.ret = np.zeros(num_envs)
        self.infos = []

    def _draw(self, idx):
        self.obs[idx] = self.rng.randint(0, 256, (len(idx), 4, 84, 84)).astype(np.uint8)

    def async_reset(self, seed=42):
        self._draw(np.arange(self.num_envs))
        self.rew[:] = 0
        self.term[:] = False
        self.tick[:] = 0
        self.ret[:] = 0
        self.infos = []

    def recv(self):
        n = self.num_envs
        return self.obs.copy(), self.rew.copy(), self.term.copy(), np.zeros(n, bool), self.infos, np.arange(n), np.ones(n, bool)

    def send(self, actions):
        a = np.asarray(actions).reshape(-1)
        done = self.term.copy()
        hit = (a == self.obs[:, 0, 0, 0].astype(int) % 4) & ~done
        self.rew = np.where(done, 0.0, hit).astype(np.float32)
        self.ret = np.where(done, 0.0, self.ret + hit)
        self.tick = np.where(done, 0, self.tick + 1)
        self.term = (self.tick >= self.episode_length) & ~done
        self.infos = [dict(episode_return=float(self.ret[i]), episode_length=int(self.tick[i]), score=float(self.ret[i] / self.tick[i]))
                      for i in np.nonzero(self.term)[0]]
        self._draw(np.arange(self.num_envs))

    def close(self):
        pass
