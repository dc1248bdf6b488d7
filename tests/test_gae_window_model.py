"""The claim csrc/gae.hip's self-starting form (gae_exact_kernel<.., SELF>) rests on, checked on the CPU against the oracle's
sequential compute_gae (c_gae.pyx:11-32 restated): a walker that starts from x = 0 `warm` elements behind its 8 items and runs the
reference's own fp32 statement (product and sum rounded separately) arrives ON the reference's rounded sequence — bit-identical
advantages — once (gamma lambda)^warm <= 1e-7 * 2^-24 (gae_warm_self).  numpy model of the kernel's arithmetic, vectorised over
the 8-element groups; the GPU tests (tests/test_gpu_gae.py) assert the same for the kernel itself."""
import math

import numpy as np
import pytest

from oracle import c_oracle

f32 = np.float32


def warm_self(gamma, lam):
    gl = float(f32(gamma) * f32(lam))
    return int(math.ceil(math.log(1e-7 * 2.0 ** -24) / math.log(gl) / 8.0) * 8)


def window_gae(r, v, d, gamma, lam, warm):
    n = len(r)
    g, gl = f32(gamma), f32(gamma) * f32(lam)
    nnt = (f32(1) - d[1:]).astype(f32)
    delta = ((r[1:] + ((g * v[1:]).astype(f32) * nnt).astype(f32)).astype(f32) - v[:-1]).astype(f32)
    coef = (gl * nnt).astype(f32)
    pad = np.zeros(warm + 17, f32)                       # element n - 1 and everything behind the array: (0, 0)
    coef, delta = np.concatenate([coef, pad]), np.concatenate([delta, pad])
    starts = np.arange((n + 7) // 8) * 8
    x = np.zeros(len(starts), f32)
    adv = np.zeros(len(starts) * 8, f32)
    for k in range(warm + 7, -1, -1):
        idx = starts + k
        x = (delta[idx] + (coef[idx] * x).astype(f32)).astype(f32)
        if k < 8:
            adv[idx] = x
    return adv[:n]


@pytest.mark.parametrize('gamma,lam', [(0.99, 0.95), (0.995, 0.97), (0.9, 0.8), (0.995, 0.985)])
@pytest.mark.parametrize('p_done,scale', [(0.0, 1.0), (0.001, 1.0), (0.15, 1.0), (0.0005, 1e4)])
def test_window_started_from_zero_lands_on_the_reference_sequence(gamma, lam, p_done, scale):
    rng = np.random.default_rng(int(gamma * 1000) + int(p_done * 10000))
    n = 20000
    r = (rng.standard_normal(n) * scale * 0.1).astype(f32)
    v = (rng.standard_normal(n) * scale).astype(f32)
    d = (rng.random(n) < p_done).astype(f32)
    warm = warm_self(gamma, lam)
    assert warm % 8 == 0 and warm <= 2048                 # fits the kernel's 1024- (gamma lambda <= 0.968) or 2048-element window
    want = c_oracle.compute_gae(d, v, r, gamma, lam)
    got = window_gae(r, v, d, gamma, lam, warm)
    assert np.array_equal(got.view(np.uint32), np.asarray(want, f32).view(np.uint32))


def test_a_window_that_is_too_short_is_visible():
    """The margin is not decoration: with a quarter of the warm-up the model no longer reproduces every bit on a done-free batch."""
    rng = np.random.default_rng(5)
    n = 20000
    r, v, d = rng.standard_normal(n).astype(f32), rng.standard_normal(n).astype(f32), np.zeros(n, f32)
    want = np.asarray(c_oracle.compute_gae(d, v, r, 0.99, 0.95), f32)
    short = window_gae(r, v, d, 0.99, 0.95, warm_self(0.99, 0.95) // 4 // 8 * 8)
    assert not np.array_equal(short.view(np.uint32), want.view(np.uint32))
