"""TEST INFRASTRUCTURE — the two forms of the ocean Squared env step side by side in plain Python: the general form (a line-by-
line restatement of ocean.py:448-513, what csrc/squared_env.hpp's squared_reset / squared_step implement) and the single-target
form the fused rollout uses when num_targets == 1 (squared_reset_nt1 / squared_step_nt1: target coordinates kept, reward from a
table of the possible distances built with the same float division, a reset that clears only the agent's and the target's cell).
tests/test_oracle_golden.py drives both with the same actions and targets and requires identical grids, rewards (as float32 bit
patterns), dones and scores — in particular the invariant the two-cell clear rests on: at a reset the only non-zero cells of a
finished episode's grid are the agent's and the target marker's."""
import numpy as np

MOVES = [(0, -1), (0, 1), (-1, 0), (1, 0), (1, -1), (-1, -1), (1, 1), (-1, 1)]     # ocean.py:424


class General:
    def __init__(self, d):
        self.d, self.g = d, 2 * d + 1
        self.grid = np.zeros((self.g, self.g), np.float32)

    def reset(self, target):
        d = self.d
        self.grid[:] = 0
        self.grid[d, d] = -1
        self.pos = (d, d)
        self.tick = 0
        self.targets = [target]
        self.grid[target] = 1

    def step(self, action):
        d = self.d
        x, y = self.pos
        self.grid[x, y] = 0
        dx, dy = MOVES[action]
        x += dx
        y += dy
        min_dist = min(max(abs(x - tx), abs(y - ty)) for tx, ty in self.targets)
        reward = 1 - min_dist / d
        if (x, y) in self.targets:
            self.targets.remove((x, y))
        if max(abs(x - d), abs(y - d)) >= d:
            self.pos = (d, d)
        else:
            self.pos = (x, y)
        self.grid[self.pos] = -1
        self.tick += 1
        done = self.tick >= d            # max_ticks = num_targets * distance_to_target
        score = (1 - len(self.targets)) / 1
        return np.float32(reward), done, score


class SingleTarget:
    def __init__(self, d):
        self.d, self.g = d, 2 * d + 1
        self.grid = np.zeros((self.g, self.g), np.float32)
        self.rd = [1.0 - k / d for k in range(32)]                  # RewardTable: the same float division, once
        self.rf = [np.float32(r) for r in self.rd]
        self.pos, self.target = (d, d), (0, 0)

    def reset(self, target):
        d = self.d
        self.grid[self.pos] = 0                                     # the only cells a finished episode leaves non-zero
        self.grid[self.target] = 0
        self.grid[d, d] = -1
        self.pos = (d, d)
        self.tick = 0
        self.target = target
        self.grid[target] = 1
        self.rem = 1

    def step(self, action):
        d = self.d
        x, y = self.pos
        self.grid[x, y] = 0
        dx, dy = MOVES[action]
        x += dx
        y += dy
        dist = max(abs(x - self.target[0]), abs(y - self.target[1])) if self.rem else 1 << 30
        reward = self.rf[dist] if dist < 32 else np.float32(1.0 - dist / d)
        if dist == 0:
            self.rem = 0
        if max(abs(x - d), abs(y - d)) >= d:
            x, y = d, d
        self.pos = (x, y)
        self.grid[x, y] = -1
        self.tick += 1
        done = self.tick >= d
        return reward, done, 0.0 if self.rem else 1.0
