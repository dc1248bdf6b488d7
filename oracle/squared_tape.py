"""TEST INFRASTRUCTURE — a numpy restatement of the PARALLEL form of the Squared reset-target tape for one target per
episode (csrc/squared.hip: squared_tape_words / _count / _select), checked against CPython's own `random` module
(tests/test_oracle_golden.py).  Nothing here is product code; it pins the algorithm the three kernels implement.

The reference resets every env with ``random.sample(possible_targets, num_targets)`` on the process-global generator, in env
order (ocean.py:449-459).  For one target that is a single ``_randbelow(n)`` (Lib/random.py: ``getrandbits(n.bit_length())``
until the value is below n), i.e. MT19937 word w yields a draw iff ``(w >> (32 - bits)) < n`` — independently of every
other word.  The device therefore splits the work into

  words   the only sequential part: raw MT19937 words.  The recurrence x[k+624] = twist(x[k], x[k+1], x[k+397]) lets 227
          consecutive words be computed at once; the far operand of a 227-word step is what the same slot produced one
          step earlier (k + 397 - 624 = k - 227).  The stream slides through a LINEAR window that moves back to the front
          when it reaches the end (``LIN`` words of LDS on the device).
  select  everything else in parallel over all words: temper, test, exclusive prefix count -> draw number -> (round, env)
          tape slot; the position of the word that produced the LAST draw fixes the generator state handed back:
          (block that holds the next word, index inside it), an index of 624 staying on the old block as CPython leaves it.

``blocks_for`` is the number of 624-word blocks produced per fill: the expectation + 2 % + 6 blocks.
"""
import numpy as np

U32 = np.uint32
N, M, STEP, LIN = 624, 397, 227, 12288


def twist(cur, nxt, far):
    y = (cur & U32(0x80000000)) | (nxt & U32(0x7fffffff))
    return far ^ (y >> U32(1)) ^ np.where(y & U32(1), U32(0x9908b0df), U32(0))


def temper(y):
    y = y ^ (y >> U32(11))
    y = y ^ ((y << U32(7)) & U32(0x9d2c5680))
    y = y ^ ((y << U32(15)) & U32(0xefc60000))
    return y ^ (y >> U32(18))


def blocks_for(need, n_pop):
    bits = int(n_pop).bit_length()
    words = need * float(1 << bits) / n_pop * 1.02 + 6.0 * N
    return int(words / N) + 1


def raw_words(state, blocks):
    """squared_tape_words_kernel: raw[0:624] = the current block, raw[624 j : 624 (j+1)] = its j-th regeneration (plus up to one
    step of slack).  One 64-lane wavefront; lane l owns words 64 r + l (r < 4; r = 3: l < 35) of every 227-word step."""
    lin = np.zeros(LIN, dtype=U32)
    lin[:N] = state
    raw = np.zeros((blocks + 1) * N + STEP, dtype=U32)
    raw[:N] = state
    lane = np.arange(64)
    far = [lin[lane + 64 * r + M].copy() for r in range(4)]
    cur = [lin[lane + 64 * r].copy() for r in range(4)]
    nxt = [lin[lane + 64 * r + 1].copy() for r in range(4)]
    w, out = 0, N
    for _ in range((blocks * N + STEP - 1) // STEP):
        cur1 = [lin[w + lane + STEP + 64 * r].copy() for r in range(4)]          # operands of the NEXT step, fetched ahead
        nxt1 = [lin[w + lane + STEP + 64 * r + 1].copy() for r in range(4)]
        for r in range(4):
            x = twist(cur[r], nxt[r], far[r])
            own = np.ones(64, bool) if r < 3 else lane < STEP - 192
            lin[w + N + lane[own] + 64 * r] = x[own]
            raw[out + lane[own] + 64 * r] = x[own]
            far[r], cur[r], nxt[r] = x, cur1[r], nxt1[r]
        w += STEP
        out += STEP
        if w + N + 2 * STEP + 64 > LIN:
            lin[:N] = lin[w:w + N].copy()
            w = 0
    return raw


def fill(state, idx0, need, n_pop):
    """One tape fill of ``need`` draws from the generator (state[624], idx0).  Returns (draws as population indices in draw
    order, new state block, new index, words consumed)."""
    bits = int(n_pop).bit_length()
    blocks = blocks_for(need, n_pop)
    raw = raw_words(np.asarray(state, dtype=U32), blocks)
    avail = (blocks + 1) * N - idx0
    p = np.arange(avail)
    cand = temper(raw[idx0 + p]) >> U32(32 - bits)
    accepted = cand < n_pop
    pos = np.cumsum(accepted) - accepted                 # exclusive prefix = draw number of an accepted word
    chosen = accepted & (pos < need)
    if int(chosen.sum()) < need:
        raise RuntimeError('tape underrun: the margin of extra words did not cover the rejections')
    last = int(p[chosen][-1])
    consumed = last + 1
    blk, idx = divmod(idx0 + consumed, N)
    if idx == 0 and blk > 0:
        blk, idx = blk - 1, N
    return cand[chosen].astype(np.int64), raw[blk * N:(blk + 1) * N].copy(), idx, consumed
