"""LDS layouts against the banking model of MI355X_MICROARCH.md (tools/lds_bank_model.py): the address expressions of the frequent LDS
accesses of csrc/ppo_bf16.hpp and of the rollout kernels' partial tiles, restated here with the strides read from the sources (the
device's SQ_LDS_BANK_CONFLICT agrees with the model: docs/lab-notebook.md).  A stride changed in a kernel without a look at the banks
shows up here."""
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


def test_bf16_gradient_kernel_layouts_in_the_banking_model():
    """The kept layout of csrc/ppo_bf16.hpp (X rows 144 B, dout rows 80 B, patch rows 72 B, contraction index k = row 8g + e): its
    operand reads are 2-way in the model; the variant with conflict-free strides (X 160 B, dout 32 B, k = row 16 (e >> 2) + 4g + (e & 3);
    tools/experiments/ppo_bf16_glds_variant.hpp) was built and measured slower as a whole (docs/lab-notebook.md).  Both are pinned here
    so that the model and the sources stay in step."""
    XRS, DRS, HRS = (_const('ppo_bf16.hpp', n) for n in ('XRS', 'DRS', 'HRS'))
    assert (XRS, DRS, HRS) == (144, 80, 72)

    def tr(stride, col0, rows):       # transposed read: lane group g reads rows rows(g) + (c >> 2), 8 bytes at columns col0 + 4 (c & 3)
        return lambda l: (rows(lane_cg(l)[1]) + (lane_cg(l)[0] >> 2)) * stride + (col0 + 4 * (lane_cg(l)[0] & 3)) * 2
    kept, variant = (lambda g: 8 * g), (lambda g: 4 * g)
    # kept layout: forward B fragments, X^T and dout by transposed reads, the dout A operand with permuted rows
    assert cycles(lambda l: lane_cg(l)[0] * XRS + 8 * lane_cg(l)[1] * 2, 16, 'read_b128') == (8, 4)
    assert cycles(tr(XRS, 0, kept), 8, 'read_tr_b64') == (4, 2)
    assert cycles(tr(DRS, 0, kept), 8, 'read_tr_b64') == (4, 2)
    assert cycles(tr(HRS, 0, kept), 8, 'read_tr_b64') == (4, 2)
    for i in (0, 1):                  # ... and its b64 stores of the hidden patch and of the staged X are conflict-free
        assert cycles(lambda l: lane_cg(l)[0] * HRS + (16 * i + 4 * lane_cg(l)[1]) * 2, 8, 'write_b64') == (4, 4)
    for q in range(2):
        assert cycles(lambda l: ((l + 128 * q) >> 4) * XRS + ((l + 128 * q) & 15) * 8, 8, 'write_b64') == (4, 4)
    # the variant's strides: conflict-free reads
    assert cycles(lambda l: lane_cg(l)[0] * 160 + 8 * lane_cg(l)[1] * 2, 16, 'read_b128') == (4, 4)
    assert cycles(tr(160, 0, variant), 8, 'read_tr_b64') == (2, 2)
    assert cycles(tr(32, 0, variant), 8, 'read_tr_b64') == (2, 2)
    assert cycles(lambda l: lane_cg(l)[0] * 32 + 16 * (lane_cg(l)[1] & 1), 16, 'read_b128') == (4, 4)


def test_rollout_partial_tiles_are_read_without_conflicts_and_written_two_way():
    S = _const('mlp_tile.hpp', 'kPartStride')
    # sampling thread (row le = lane >> 4 (+ 4 per wave), output lo = lane & 15) reads part[w][lo * S + le]
    assert cycles(lambda l: ((l & 15) * S + (l >> 4)) * 4, 4, 'read_b32') == (2, 2)
    assert cycles(lambda l: ((l & 15) * 16 + (l >> 4)) * 4, 4, 'read_b32')[0] == 16      # what rows of 16 floats cost: 8-way
    # fragment-order stores part[wv][(4g + r) * S + c]: 2-way at most, which a ds_write_b32 absorbs (its transfer, not the array, sets its time)
    for r in range(4):
        assert cycles(lambda l: ((4 * lane_cg(l)[1] + r) * S + lane_cg(l)[0]) * 4, 4, 'write_b32')[0] <= 4
