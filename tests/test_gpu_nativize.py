"""Structured-observation unpack on the GPU (csrc/nativize.hip behind pufferlib_amd.pytorch.nativize_tensor) against leaf
values produced by the unmodified reference (tests/golden/nativize.npz) and, at sizes that fill the chip, the numpy oracle:
raw leaves bit for bit, the fused .float() and the fused torch.cat(...).float() matrix."""
import ast
import os

import numpy as np
import pytest
import torch


def _dtype_from_spec(spec):
    """Aligned structured dtype from the fixture's nested literal (tests/golden/make_golden.py: dtype_spec)."""
    if isinstance(spec, list):
        return np.dtype([(name, _dtype_from_spec(sub)) for name, sub in spec], align=True)
    return np.dtype((spec[0], tuple(spec[1])), align=True)

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'nativize.npz'))
    for name in g['cases']:
        name = str(name)
        yield (g, name, np.dtype(str(g[f'{name}:sample'])), _dtype_from_spec(ast.literal_eval(str(g[f'{name}:descr']))),
               ast.literal_eval(str(g[f'{name}:table'])))


def _native(sample, structured):
    from pufferlib_amd import namespace, pytorch as ppt
    return ppt.nativize_dtype(namespace(observation_dtype=sample, emulated_observation_dtype=structured))


def _to_numpy(t):
    return t.cpu().contiguous().view(torch.uint8).numpy()


def test_leaves_match_the_reference_bit_for_bit(golden_dir):
    from pufferlib_amd import pytorch as ppt
    for g, name, sample, structured, table in _golden(golden_dir):
        native = _native(sample, structured)
        rows = torch.from_numpy(g[f'{name}:rows'].view(sample).copy()).cuda()
        leaves = ppt.nativize_tensor(rows, native)
        for path, (dt, shape, _, _) in ppt._leaves(native):
            leaf = leaves
            for k in path:
                leaf = leaf[k]
            assert leaf.dtype == dt and tuple(leaf.shape) == (rows.shape[0],) + tuple(shape), (name, path)
            want = g[f'{name}:leaf:' + '/'.join(str(k) for k in path)]
            assert np.array_equal(_to_numpy(leaf).reshape(want.shape), want), (name, path)


@pytest.mark.parametrize('n', [1, 15, 16, 17, 1000, 70001])
def test_all_modes_match_the_oracle_across_tile_boundaries(golden_dir, n):
    from oracle import nativize as onat
    from pufferlib_amd import pytorch as ppt
    rng = np.random.default_rng(n)
    for g, name, sample, structured, table in _golden(golden_dir):
        if name == 'wide' and n > 1000:
            n = 1000
        native = _native(sample, structured)
        row_bytes = g[f'{name}:rows'].shape[1]
        raw = rng.integers(0, 256, size=(n, row_bytes), dtype=np.uint8)
        rows = raw.view(sample)
        want = onat.nativize_rows(rows, onat.leaf_table(sample, structured))
        dev = torch.from_numpy(rows.copy()).cuda()
        plan = ppt.NativizePlan(native, sample.itemsize)
        got, gotf, cat = plan(dev), plan(dev, to_float=True), plan.concat(dev)
        cols = []
        for path, _ in ppt._leaves(native):
            a, b = got, gotf
            for k in path:
                a, b = a[k], b[k]
            w = want[path]
            assert np.array_equal(_to_numpy(a).reshape(-1), w.view(np.uint8).reshape(-1)), (name, path, 'raw')
            with np.errstate(all='ignore'):
                wf = w.astype(np.float32)
            assert np.array_equal(b.cpu().numpy(), wf, equal_nan=True), (name, path, 'float')
            cols.append(wf.reshape(n, -1))
        assert np.array_equal(cat.cpu().numpy(), np.concatenate(cols, 1), equal_nan=True), (name, 'concat')


def test_bad_arguments_fail_loudly():
    from pufferlib_amd import pytorch as ppt
    native = {'a': (torch.float32, (4,), 0, 16), 'b': (torch.uint8, (3,), 16, 3)}
    with pytest.raises(RuntimeError):       # leaf b ends at byte 19 of 16-byte rows
        ppt.nativize_tensor(torch.zeros(8, 16, dtype=torch.uint8, device='cuda'), native)
    with pytest.raises(ValueError):         # rows of the wrong element size
        ppt.NativizePlan(native, 1)(torch.zeros(8, 5, device='cuda'))
    out = ppt.nativize_tensor(torch.zeros(0, 24, dtype=torch.uint8, device='cuda'), native)
    assert out['a'].shape == (0, 4) and out['b'].shape == (0, 3)


def test_unpack_runs_near_the_hbm_roofline(golden_dir):
    """ocean.Spaces-shaped rows (108 bytes: 5x5 f32 image + 5 int8) at 4M rows: the launch moves 432 MB in + 420 MB out.  The
    bar is loose (a guard against a gather-per-field regression, not the measurement: that is profiles/ + DESIGN.md)."""
    from pufferlib_amd import pytorch as ppt
    g, name, sample, structured, table = next(c for c in _golden(golden_dir) if c[1] == 'spaces_env')
    native = _native(sample, structured)
    n = 4 << 20
    dev = torch.randint(0, 256, (n, 108), dtype=torch.uint8, device='cuda')
    plan = ppt.NativizePlan(native, 1)
    import ctypes as C
    from pufferlib_amd import _lib
    L = _lib.lib()
    L.pfa_timing_select(b'nativize')
    L.pfa_timing_enable(1)
    try:
        for mode in ('raw', 'concat'):
            fn = (lambda: plan(dev)) if mode == 'raw' else (lambda: plan.concat(dev))
            fn()
            torch.cuda.synchronize()
            best = 0.0
            for attempt in range(4):       # best of four rounds: the pool's boxes are shared, a round can land next to somebody's job
                L.pfa_timing_reset()
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                launches, total = C.c_int64(0), C.c_double(0)
                L.pfa_timing_read(b'nativize', C.byref(launches), C.byref(total))   # HIP events around the kernel: no allocator time
                assert launches.value == 10
                ms = total.value / launches.value
                out_bytes = n * (105 if mode == 'raw' else 120)
                gbs = (n * 108 + out_bytes) / ms / 1e6
                best = max(best, gbs)
                if best > 1000:
                    break
            print(f'nativize {mode}: {ms:.3f} ms, best {best:.0f} GB/s algorithmic')
            assert best > 250, (mode, ms, best)     # a one-launch streaming unpack; ~1.7 TB/s on an idle box (profiles/)
    finally:
        L.pfa_timing_enable(0)
        L.pfa_timing_select(b'ppo_mlp_grad')
