"""TEST INFRASTRUCTURE — a numpy prototype of the PARALLEL form of ocean.Spaces' observation stream ("what comes
next" item 4), checked against the sequential C restatement (oracle/puffer_oracle.c: po_spaces_*, itself pinned against the
reference).  Nothing here is product code; it fixes the algorithm the device tape kernel has to implement.

The reference draws, per env reset and in env order, from numpy's process-global legacy generator (ocean.py:380-389):
25 gaussians (legacy polar method: an attempt takes two 53-bit doubles = 4 MT19937 words and is rejected when r2 >= 1 or
r2 == 0; an accepted attempt yields TWO values, the second one is cached and returned by the next call) and 5 int8 values
(randint(-1, 2, dtype=int8): bytes of buffered 32-bit words, low byte first, a byte b is accepted when b & 3 <= 2, the buffer
is dropped at the end of the call).  The number of words a reset consumes therefore depends on the data, and whether a reset
starts with a cached gaussian alternates (25 is odd).

Parallel form:
  1. accepted[p]  for every word position p: would a polar attempt starting at p be accepted?            (independent per p)
  2. length[p][c] for every p and entry parity c (c = 1: a cached gaussian is waiting): words consumed by one reset (walk the
     marks for 13 - c accepted attempts, then the byte loop)                                                (independent per p)
  3. start of reset k = next^k(0, 0) with next(p, c) = (p + length[p][c], 1 - c): pointer doubling, log2(K) rounds
  4. every reset fills its row from its own start (an odd reset re-reads the last accepted attempt of its predecessor for the
     cached value)                                                                                          (independent per reset)
"""
import numpy as np


def _mt_words(seed, count):
    """The first ``count`` 32-bit outputs of numpy's legacy MT19937 after np.random.seed(seed)."""
    rs = np.random.RandomState(seed)
    return rs.randint(0, 2 ** 32, size=count, dtype=np.uint64).astype(np.uint32)     # one word per draw on this path


def _double(w, p):
    a, b = int(w[p]) >> 5, int(w[p + 1]) >> 6
    return (a * 67108864.0 + b) / 9007199254740992.0


def _attempt(w, p):
    """(accepted, first_returned, cached) of a polar attempt on words p..p+3 (legacy_gauss)."""
    x1 = 2.0 * _double(w, p) - 1.0
    x2 = 2.0 * _double(w, p + 2) - 1.0
    r2 = x1 * x1 + x2 * x2
    if r2 >= 1.0 or r2 == 0.0:
        return False, 0.0, 0.0
    f = np.sqrt(-2.0 * np.log(r2) / r2)
    return True, f * x2, f * x1


def _bytes_part(w, q):
    """5 accepted bytes starting with word q -> (values int8[5], words consumed)."""
    vals, examined = [], 0
    while len(vals) < 5:
        b = (int(w[q + examined // 4]) >> (8 * (examined % 4))) & 0xFF
        examined += 1
        if b & 3 <= 2:
            vals.append(np.int8(np.uint8((255 + (b & 3)) & 0xFF)))      # off = uint8(-1) = 255, wraps like the C cast
    return np.array(vals, np.int8), (examined + 3) // 4


def parallel_rows(seed, resets, window):
    """Rows [resets][108] of ``resets`` consecutive env resets after np.random.seed(seed), computed by the parallel form."""
    w = _mt_words(seed, window + 8)
    # 1. acceptance marks
    accepted = np.array([_attempt(w, p)[0] for p in range(window)])
    # 2. words per reset for both entry parities (and where the walk ends: the last accepted attempt, the byte start)
    length = np.zeros((window, 2), np.int64)
    last_pair = np.zeros((window, 2), np.int64)
    valid = np.zeros((window, 2), bool)
    for p in range(window):
        for c in (0, 1):
            q, k, lp = p, 0, -1
            while k < 13 - c and q < window:
                if accepted[q]:
                    k, lp = k + 1, q
                q += 4
            if k < 13 - c or q + 8 >= window:
                continue
            _, nb = _bytes_part(w, q)
            length[p, c], last_pair[p, c], valid[p, c] = (q - p) + nb, lp, True
    # 3. pointer doubling over states s = 2 p + c
    nstate = 2 * window
    nxt = np.full(nstate, -1, np.int64)
    for p in range(window):
        for c in (0, 1):
            if valid[p, c] and p + length[p, c] < window:
                nxt[2 * p + c] = 2 * (p + length[p, c]) + (1 - c)
    jumps = [nxt]
    while (1 << len(jumps)) < resets:
        j = jumps[-1]
        jumps.append(np.where(j >= 0, j[np.maximum(j, 0)], -1))
    starts = np.zeros(resets, np.int64)
    for k in range(resets):                       # independent per reset: binary decomposition of k
        s, bit = 0, 0
        while (k >> bit) and s >= 0:
            if (k >> bit) & 1:
                s = jumps[bit][s]
            bit += 1
        assert s >= 0, 'window too small'
        starts[k] = s
    # 4. fill rows
    rows = np.zeros((resets, 108), np.uint8)
    for k in range(resets):                       # independent per reset
        p, c = int(starts[k]) // 2, int(starts[k]) % 2
        vals = []
        if c:                                     # the predecessor's 13th attempt left its second value behind
            pp, pc = int(starts[k - 1]) // 2, int(starts[k - 1]) % 2
            vals.append(_attempt(w, int(last_pair[pp, pc]))[2])
        q = p
        while len(vals) < 25:
            ok, first, second = _attempt(w, q)
            q += 4
            if ok:
                vals.append(first)
                if len(vals) < 25:
                    vals.append(second)
        flat, _ = _bytes_part(w, q)
        rows[k, 0:5] = flat.view(np.uint8)
        rows[k, 8:108] = np.array(vals, np.float64).astype(np.float32).view(np.uint8)
    return rows
