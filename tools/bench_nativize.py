"""Micro-benchmark of the structured-observation unpack (csrc/nativize.hip): algorithmic GB/s per mode, kernel time from HIP
events on the launch stream (pfa_timing_*).  `python tools/bench_nativize.py` on an MI355X; one JSON line per case."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pufferlib_amd import _lib, namespace, pytorch as ppt  # noqa: E402

HBM_PEAK_GBS = 8000.0

CASES = {
    'spaces 5x5 f32 + 5 i8 (108 B rows)': (np.dtype([('flat', np.int8, (5,)), ('image', np.float32, (5, 5))], align=True), 4 << 20),
    'six mixed leaves (112 B rows)': (np.dtype([('u8', np.uint8, (3,)), ('f64', np.float64, (2,)), ('i16', np.int16, (5,)),
                                                ('f32', np.float32, (4, 3)), ('i64', np.int64, (1,)), ('f16', np.float16, (7,))], align=True), 4 << 20),
    'map 30x30 u8 + 31 f32 (1024 B rows)': (np.dtype([('map', np.uint8, (30, 30)), ('vec', np.float32, (31,))], align=True), 1 << 19),
    'map 40x40 u8 + 700 f32 + 3 i32 (4412 B rows)': (np.dtype([('map', np.uint8, (40, 40)), ('vec', np.float32, (700,)),
                                                               ('id', np.int32, (3,))], align=True), 1 << 17),
}


def kernel_ms(L, fn, iters=20):
    fn()
    torch.cuda.synchronize()
    L.pfa_timing_reset()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    n, ms = C.c_int64(0), C.c_double(0)
    L.pfa_timing_read(b'nativize', C.byref(n), C.byref(ms))
    return ms.value / max(n.value, 1)


def main():
    L = _lib.lib()
    L.pfa_timing_select(b'nativize')
    L.pfa_timing_enable(1)
    for name, (dt, n) in CASES.items():
        native = ppt.nativize_dtype(namespace(observation_dtype=np.dtype(np.uint8), emulated_observation_dtype=dt))
        D = dt.itemsize
        dev = torch.randint(0, 256, (n, D), dtype=torch.uint8, device='cuda')
        plan = ppt.NativizePlan(native, 1)
        leaf_bytes = sum(int(np.prod(leaf[1])) * leaf[0].itemsize for leaf in plan.leaves)
        out = dict(case=name, rows=n, row_bytes=D)
        for mode, fn, out_bytes in (('leaves', lambda: plan(dev), n * leaf_bytes),
                                    ('leaves_f32', lambda: plan(dev, to_float=True), n * plan.total * 4),
                                    ('matrix_f32', lambda: plan.concat(dev), n * plan.total * 4)):
            ms = kernel_ms(L, fn)
            gbs = (n * D + out_bytes) / ms / 1e6
            out[mode] = dict(ms=round(ms, 4), algorithmic_GBs=round(gbs), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 3))
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
