"""The LDS layouts chosen with the banking model of MI355X_MICROARCH.md (tools/lds_bank_model.py): the address expressions of the
frequent LDS accesses of csrc/ppo_bf16.hpp and of the rollout kernels' partial tiles, restated here, must stay conflict-free in the
model (the device's SQ_LDS_BANK_CONFLICT agreed when they were chosen: docs/lab-notebook.md).  A stride changed in the kernel without
a look at the banks shows up here — the constants are read from the sources."""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
from lds_bank_model import cycles, lane_cg  # noqa: E402

CSRC = os.path.join(REPO, 'pufferlib_amd', 'csrc')


def _const(path, name):
    m = re.search(r'constexpr int ' + name + r'\s*=\s*(\d+)\s*;', open(os.path.join(CSRC, path)).read())
    assert m, (path, name)
    return int(m.group(1))


def test_bf16_gradient_kernel_operand_reads_are_conflict_free():
    XRS, DRS, HRS = (_const('ppo_bf16.hpp', n) for n in ('XRS', 'DRS', 'HRS'))
    free = lambda addr, nbytes, kind: cycles(addr, nbytes, kind)[0] == cycles(addr, nbytes, kind)[1]   # noqa: E731

    def tr(stride, col0, second=0):   # bf_tr8: lane group g reads rows 4g + (c >> 2) (+ 16), 8 bytes at columns col0 + 4 (c & 3)
        return lambda l: (4 * lane_cg(l)[1] + (lane_cg(l)[0] >> 2) + 16 * second) * stride + (col0 + 4 * (lane_cg(l)[0] & 3)) * 2
    for ks in (0, 1):      # forward: B fragments of X, row c, 16 bytes at k = 32 ks + 8 g
        assert free(lambda l: lane_cg(l)[0] * XRS + (32 * ks + 8 * lane_cg(l)[1]) * 2, 16, 'read_b128')
    for kt in range(4):    # dW1: X^T fragments by transposed reads
        for second in (0, 1):
            assert free(tr(XRS, 16 * kt, second), 8, 'read_tr_b64')
    assert free(tr(DRS, 0), 8, 'read_tr_b64')                                                        # dW2v: dout fragments
    assert free(lambda l: lane_cg(l)[0] * DRS + 16 * (lane_cg(l)[1] & 1), 16, 'read_b128')           # dh: dout rows as the A operand
    for q in range(4):     # staging: a wave pair writes 8 bytes per piece and chunk, 16 lanes per row
        assert free(lambda l: ((l + 128 * q) >> 4) * XRS + ((l + 128 * q) & 15) * 8, 8, 'write_b64')
    assert free(lambda l: (l >> 4) * 256 + (l & 15) * 16, 16, 'read_b128')                           # ... from the landing buffer
    for i in (0, 1):       # the hidden patch: its b64 stores are conflict-free, its transposed reads 2-way (no stride serves both)
        assert free(lambda l: lane_cg(l)[0] * HRS + (16 * i + 4 * lane_cg(l)[1]) * 2, 8, 'write_b64')
        assert cycles(tr(HRS, 16 * i), 8, 'read_tr_b64') == (4, 2)


def test_rollout_partial_tiles_are_read_without_conflicts_and_written_two_way():
    S = _const('mlp_tile.hpp', 'kPartStride')
    # sampling thread (row le = lane >> 4 (+ 4 per wave), output lo = lane & 15) reads part[w][lo * S + le]
    assert cycles(lambda l: ((l & 15) * S + (l >> 4)) * 4, 4, 'read_b32') == (2, 2)
    assert cycles(lambda l: ((l & 15) * 16 + (l >> 4)) * 4, 4, 'read_b32')[0] == 16      # what rows of 16 floats cost: 8-way
    # fragment-order stores part[wv][(4g + r) * S + c]: 2-way at most, which a ds_write_b32 absorbs (its transfer, not the array, sets its time)
    for r in range(4):
        assert cycles(lambda l: ((4 * lane_cg(l)[1] + r) * S + lane_cg(l)[0]) * 4, 4, 'write_b32')[0] <= 4
