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

    def __init__(self, num_envs, workers=2, distance_to_target=3, num_targets=1):
        from oracle import c_oracle
        from pufferlib_amd import vector
        assert num_envs % workers == 0
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
        w = self.turn
        o, r, d, t, infos, ids, mask = self.groups[w].recv()
        infos = [{k: v for k, v in i.items() if not k.startswith('_')} for i in infos]
        return o.copy(), r.copy(), d.copy(), t.copy(), infos, ids + w * self.per, mask.copy()

    def send(self, actions):
        self.groups[self.turn].send(np.asarray(actions, np.int64))
        self.turn = (self.turn + 1) % len(self.groups)

    def close(self):
        pass
