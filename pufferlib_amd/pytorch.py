"""Structured observations on the device — the mirror of ``pufferlib.pytorch`` 's nativize functions (pufferlib/pytorch.py:48-145,
SURVEY.md §8f rank 3).

The reference's emulation layer packs Dict / Tuple observations into one flat row per agent (bytes when leaf dtypes differ,
emulation.py:68-110); policies undo that with ``nativize_tensor``, which reinterprets column ranges of the [N, D] batch as typed
views.  Same names, arguments and return structure here; what differs is the mechanism: one HIP launch
(``pfa_nativize_rows``, csrc/nativize.hip) writes every leaf as its own dense, aligned tensor (values identical to the
reference's views, no aliasing of the input), optionally already converted to f32 or laid out as column ranges of one
[N, total] f32 matrix, which is what encoders compute next (``.float()``, ``torch.cat`` of the flattened leaves).

``nativize_dtype`` follows the reference's own offset rule, including where it departs from numpy's struct layout: leaves are
placed one after the other, each rounded up to ITS OWN alignment (pytorch.py:73-75) — nested structs are not aligned as a
whole.  For the emulated dtypes the reference builds (``np.dtype(..., align=True)`` of leaf arrays) both agree unless a
nested struct begins with a member of smaller alignment than its largest one.
"""

import numpy as np
import torch

from . import _lib
from .models import layer_init  # noqa: F401  (pufferlib.pytorch.layer_init, pytorch.py:193-197)

numpy_to_torch_dtype_dict = {
    np.dtype('float64'): torch.float64, np.dtype('float32'): torch.float32, np.dtype('float16'): torch.float16,
    np.dtype('uint64'): torch.uint64, np.dtype('uint32'): torch.uint32, np.dtype('uint16'): torch.uint16,
    np.dtype('uint8'): torch.uint8, np.dtype('int64'): torch.int64, np.dtype('int32'): torch.int32,
    np.dtype('int16'): torch.int16, np.dtype('int8'): torch.int8,
}

_CODES = {torch.uint8: 0, torch.int8: 1, torch.uint16: 2, torch.int16: 3, torch.uint32: 4, torch.int32: 5, torch.uint64: 6,
          torch.int64: 7, torch.float16: 8, torch.float32: 9, torch.float64: 10}


def _place(sample_dtype, structured, offset):
    """(tree, offset, delta) of ``structured`` placed at running position ``offset`` (pytorch.py:63-94).  Positions count
    sample elements: bytes for byte rows, elements of the common dtype otherwise."""
    if structured.fields is None:
        leaf, shape = structured.subdtype if structured.subdtype is not None else (structured, (1,))
        delta = int(np.prod(shape))
        if sample_dtype.base.itemsize == 1:
            offset = -(-offset // leaf.alignment) * leaf.alignment
            delta *= leaf.itemsize
        elif leaf.itemsize != sample_dtype.base.itemsize:
            raise ValueError(f'leaf dtype {leaf} does not match the sample dtype {sample_dtype}')
        return (numpy_to_torch_dtype_dict[leaf], tuple(int(x) for x in shape), offset, delta), offset, delta
    tree, start, total = {}, offset, 0
    for name, (sub, _) in structured.fields.items():
        node, offset, delta = _place(sample_dtype, sub, offset)
        tree[name] = node
        offset += delta
        total += delta
    return tree, start, total


def nativize_dtype(emulated):
    """pufferlib.pytorch.nativize_dtype (pytorch.py:48-60): the leaf table of ``emulated.emulated_observation_dtype`` as seen in
    rows of ``emulated.observation_dtype`` — a (torch dtype, shape, offset, delta) tuple, or a (nested) dict of them."""
    tree, _, _ = _place(np.dtype(emulated.observation_dtype), np.dtype(emulated.emulated_observation_dtype), 0)
    return tree


def _leaves(native_dtype, prefix=()):
    if isinstance(native_dtype, tuple):
        yield prefix, native_dtype
    else:
        for k, v in native_dtype.items():
            yield from _leaves(v, prefix + (k,))


def _flattened_tensor_size(native_dtype):
    return int(sum(np.prod(leaf[1]) for _, leaf in _leaves(native_dtype)))


def flattened_tensor_size(native_dtype):
    """pufferlib.pytorch.flattened_tensor_size (pytorch.py:157-171)."""
    return _flattened_tensor_size(native_dtype)


class NativizePlan:
    """The field table of one (native_dtype, row dtype) pair, built once: every ``nativize_tensor`` call is one launch."""

    def __init__(self, native_dtype, sample_itemsize):
        self.native_dtype = native_dtype
        self.paths, self.leaves = zip(*_leaves(native_dtype))
        if len(self.leaves) > _lib.NAT_MAX_FIELDS:
            raise NotImplementedError(f'{len(self.leaves)} leaves: pfa_nativize_rows takes up to {_lib.NAT_MAX_FIELDS} per launch')
        self.sample_itemsize = int(sample_itemsize)
        self.total = _flattened_tensor_size(native_dtype)
        self.fields = (_lib.NatField * len(self.leaves))()
        for f, (dt, shape, off, delta) in zip(self.fields, self.leaves):
            f.offset = off * self.sample_itemsize
            f.count = int(np.prod(shape))
            f.dtype = _CODES[dt]

    def _rows(self, observation):
        if not observation.is_cuda:
            raise RuntimeError('pufferlib_amd.pytorch.nativize_tensor runs on the GPU (csrc/nativize.hip); got a CPU tensor')
        if observation.dim() != 2 or observation.element_size() != self.sample_itemsize:
            raise ValueError(f'expected [N, D] rows of {self.sample_itemsize}-byte elements, got {tuple(observation.shape)} {observation.dtype}')
        obs = observation.contiguous()
        if obs.data_ptr() % 16:
            obs = obs.clone()
        return obs, obs.shape[0], obs.shape[1] * self.sample_itemsize

    def _launch(self, obs, n, row_bytes):
        if n == 0:
            return
        _lib.check(_lib.lib().pfa_nativize_rows(_lib.ptr(obs), n, row_bytes, self.fields, len(self.leaves), _lib.stream_handle()),
                   'nativize_rows')

    def _rebuild(self, flat):
        it = iter(flat)

        def build(node):
            return next(it) if isinstance(node, tuple) else {k: build(v) for k, v in node.items()}
        return build(self.native_dtype)

    def __call__(self, observation, to_float=False):
        """Leaf tensors [N, *shape] in the structure of ``native_dtype``; ``to_float`` yields them as f32."""
        obs, n, row_bytes = self._rows(observation)
        outs = []
        for f, (dt, shape, _, _) in zip(self.fields, self.leaves):
            out = torch.empty((n,) + tuple(shape), dtype=torch.float32 if to_float else dt, device=obs.device)
            f.out = out.data_ptr()
            f.to_f32 = int(to_float)
            f.out_stride = f.count
            outs.append(out)
        self._launch(obs, n, row_bytes)
        return self._rebuild(outs)

    def concat(self, observation, out=None):
        """``torch.cat([leaf.view(N, -1).float() for leaf in leaves], dim=1)`` in the same launch: [N, total] f32."""
        obs, n, row_bytes = self._rows(observation)
        if out is None:
            out = torch.empty(n, self.total, dtype=torch.float32, device=obs.device)
        elif out.shape != (n, self.total) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError(f'out must be a contiguous f32 [{n}, {self.total}] tensor')
        col = 0
        for f in self.fields:
            f.out = out.data_ptr() + 4 * col
            f.to_f32 = 1
            f.out_stride = self.total
            col += f.count
        self._launch(obs, n, row_bytes)
        return out


_plans = {}


def _plan(native_dtype, sample_itemsize):
    key = (repr(native_dtype), sample_itemsize)
    if key not in _plans:
        _plans[key] = NativizePlan(native_dtype, sample_itemsize)
    return _plans[key]


def nativize_tensor(observation, native_dtype):
    """pufferlib.pytorch.nativize_tensor (pytorch.py:96-145): [N, D] emulated rows -> tensor or (nested) dict of tensors
    [N, *shape] of each leaf's own dtype."""
    return _plan(native_dtype, observation.element_size())(observation)


def nativize_observation(observation, emulated):
    """pufferlib.pytorch.nativize_observation (pytorch.py:148-155; the reference passes its two dtypes to the two-argument
    nativize_tensor and raises — this does what its docstring means)."""
    return nativize_tensor(observation, nativize_dtype(emulated))
