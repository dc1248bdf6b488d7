"""TEST INFRASTRUCTURE — CPU restatement of pufferlib.pytorch.nativize_dtype / nativize_tensor (pufferlib/pytorch.py:48-145) in
numpy.  Only tests/ may import this; the product path is csrc/nativize.hip behind pufferlib_amd/pytorch.py.

Pinned against tests/golden/nativize.npz: leaf tables and leaf values produced by the unmodified reference
(tests/golden/make_golden.py: gen_nativize) for Dict / Tuple / nested observation spaces."""
import numpy as np


def leaf_table(sample_dtype, structured_dtype):
    """[(path, numpy dtype, shape, offset, delta)] in field order (pytorch.py:63-94).  Offsets and deltas count sample elements;
    for byte samples every leaf is rounded up to its own alignment and nothing else is (pytorch.py:73-75)."""
    sample_dtype, out = np.dtype(sample_dtype), []

    def walk(dt, offset, path):
        if dt.fields is None:
            leaf, shape = dt.subdtype if dt.subdtype is not None else (dt, (1,))
            delta = int(np.prod(shape))
            if sample_dtype.base.itemsize == 1:
                offset = int(leaf.alignment * np.ceil(offset / leaf.alignment))
                delta *= leaf.itemsize
            else:
                assert leaf.itemsize == sample_dtype.base.itemsize
            out.append((path, leaf, tuple(shape), offset, delta))
            return offset, delta
        start, total = offset, 0
        for name, (sub, _) in dt.fields.items():
            offset, delta = walk(sub, offset, path + (name,))
            offset += delta
            total += delta
        return start, total

    walk(np.dtype(structured_dtype), 0, ())
    return out


def nativize_rows(rows, table):
    """{path: array [N, *shape]} — narrow(1, offset, delta).view(dtype).view(N, *shape) (pytorch.py:128-141)."""
    rows = np.ascontiguousarray(rows)
    n = rows.shape[0]
    return {path: np.ascontiguousarray(rows[:, off:off + delta]).view(leaf).reshape((n,) + shape)
            for path, leaf, shape, off, delta in table}
