"""Structured-observation unpack, CPU side: the numpy oracle (oracle/nativize.py) and the product's host logic
(pufferlib_amd.pytorch.nativize_dtype) against leaf tables and leaf values produced by the unmodified reference
(tests/golden/nativize.npz <- pufferlib.pytorch.nativize_dtype / nativize_tensor on the reference's own emulated dtypes)."""
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


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'nativize.npz'))
    for name in g['cases']:
        name = str(name)
        structured = _dtype_from_spec(ast.literal_eval(str(g[f'{name}:descr'])))
        table = ast.literal_eval(str(g[f'{name}:table']))
        yield g, name, np.dtype(str(g[f'{name}:sample'])), structured, table


def test_oracle_leaf_tables_and_values_match_the_reference(golden_dir):
    from oracle import nativize as onat
    seen = 0
    for g, name, sample, structured, table in _cases(golden_dir):
        mine = onat.leaf_table(sample, structured)
        assert [('/'.join(p), str(dt), list(sh), off, d) for p, dt, sh, off, d in mine] == \
               [(p, dt, sh, off, d) for p, dt, sh, off, d in table], name
        rows = g[f'{name}:rows'].view(sample)
        for path, arr in onat.nativize_rows(rows, mine).items():
            want = g[f'{name}:leaf:' + '/'.join(path)]
            assert np.array_equal(arr.view(np.uint8).reshape(want.shape), want), (name, path)
            seen += 1
    assert seen == 21


def test_product_nativize_dtype_matches_the_reference_tables(golden_dir):
    from pufferlib_amd import namespace, pytorch as ppt
    for g, name, sample, structured, table in _cases(golden_dir):
        native = ppt.nativize_dtype(namespace(observation_dtype=sample, emulated_observation_dtype=structured))
        flat = [('/'.join(str(k) for k in p), str(dt).replace('torch.', ''), list(sh), off, d)
                for p, (dt, sh, off, d) in ppt._leaves(native)]
        assert flat == [tuple(t) for t in table], name
        assert ppt.flattened_tensor_size(native) == sum(int(np.prod(t[2])) for t in table)


def test_a_plain_box_is_a_single_leaf_and_cpu_tensors_are_refused():
    from pufferlib_amd import namespace, pytorch as ppt
    native = ppt.nativize_dtype(namespace(observation_dtype=np.dtype(np.float32),
                                          emulated_observation_dtype=np.dtype((np.float32, (7, 7)))))
    assert native == (torch.float32, (7, 7), 0, 49)
    with pytest.raises(RuntimeError):
        ppt.nativize_tensor(torch.zeros(4, 49), native)
