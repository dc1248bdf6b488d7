"""CPU-side checks of the C ABI: the in-tree library loads and exports every symbol include/pufferlib_amd.h
declares (no compute calls — there is no GPU here), and size helpers behave."""
import ctypes as C
import os
import re

import pytest

from pufferlib_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(REPO, 'include', 'pufferlib_amd.h')).read()
    declared = set(re.findall(r'\b(pfa_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib._SIGNATURES), declared ^ set(_lib._SIGNATURES)
    for name in declared:
        assert hasattr(L, name), name


def test_size_helpers(L):
    assert L.pfa_version() >= 1
    cfg = _lib.SquaredConfig(4096, 3, 1, 64, 64)
    assert L.pfa_squared_state_bytes(C.byref(cfg)) > 4096 * 624 * 4
    bad = _lib.SquaredConfig(0, 3, 1, 64, 64)
    assert L.pfa_squared_state_bytes(C.byref(bad)) == 0
    assert b'num_envs' in L.pfa_last_error()
    dims = _lib.MlpDims(49, 64, 128, 8)
    assert L.pfa_mlp_param_count(C.byref(dims)) == 128 * 64 + 128 + 8 * 128 + 8 + 128 + 1
    assert L.pfa_gae_workspace_bytes(524288) == 512 * 16      # one f64 affine map per 1024-element chunk


def test_weight_gradient_workspace_never_shrinks_with_more_rows(L):
    """cnn.Engine / general.Engine size the dW workspace once for their largest chunk and reuse it for shorter ones: the launch plan
    of csrc/igemm.hip (row splits x partial tiles) must therefore never need MORE room for FEWER rows — also where a small gradient
    is split finer (>= 256 rows per split until ~1024 workgroups exist)."""
    shapes = [(64, 256), (256, 16), (512, 16), (256, 32), (512, 64), (576, 64), (3136, 512), (160, 128), (48, 64), (1024, 512)]
    for K, N in shapes:
        prev = 0
        for M in (1, 16, 255, 256, 257, 4096, 8192, 65536, 131072, 8192 * 49, 8192 * 81, 8192 * 400):
            b = L.pfa_igemm_weights_workspace_bytes(M, K, N)
            assert b >= prev > -1, (K, N, M, b, prev)
            assert b >= K * N * 4, (K, N, M)
            prev = b
    assert L.pfa_igemm_weights_workspace_bytes(0, 64, 64) == 0 and L.pfa_igemm_weights_workspace_bytes(100, 64, 24) == 0
    # gemm_tn (csrc/gemm.hip): the 160-float encoder gradient has its own strip
    assert L.pfa_gemm_tn_workspace_bytes(128, 160, 131072) > 0 and L.pfa_gemm_tn_workspace_bytes(128, 24, 4096) == 0


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU."""
    import torch
    from pufferlib_amd import vector
    from pufferlib_amd.exceptions import ExtensionError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(ExtensionError):
        vector.make(vector.make_squared, num_envs=4)


def _kernel_metadata(obj):
    """{kernel symbol: (vgpr_spill_count, private_segment_fixed_size)} of the gfx950 code object inside a hipcc object file."""
    import re
    import subprocess
    import tempfile
    llvm = '/opt/rocm/lib/llvm/bin'
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'dev.co')
        subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat])
        subprocess.check_call([llvm + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
        notes = subprocess.check_output([llvm + '/llvm-readelf', '--notes', co]).decode()
    out = {}
    for blk in notes.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk)
        if name:
            out[name.group(1)] = tuple(int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)) for k in ('vgpr_spill_count', 'private_segment_fixed_size'))
    return out


def test_no_kernel_spills_registers_except_the_known_gradient_instantiations():
    """Register budget of every kernel in the library, read from the code objects' metadata (what `hipcc -S` prints as
    vgpr_spill_count / private_segment_fixed_size).  Nothing spills except four instantiations of the 128-wide fused gradient kernel
    (and the two of its opt-in bf16-path form, csrc/ppo_bf16.hpp), which sit on the 256-register line of their two-waves-per-SIMD budget: a handful of loop-invariant dwords (DESIGN 3.4 — freeing the
    producer's prefetch registers does not change it, the allocation is set by the consumer branch).  Two seeding / tape kernels keep a
    small per-thread array in scratch by design (no spill).  A change that makes any hot kernel spill shows up here, without a GPU."""
    import shutil
    from pufferlib_amd import _lib
    if not (shutil.which('objcopy') and os.path.exists('/opt/rocm/lib/llvm/bin/clang-offload-bundler')):
        pytest.skip('binutils / ROCm LLVM tools not available')
    _lib.build()
    spills, scratch_only, kernels = {}, {}, 0
    for src in _lib.SOURCES:
        if not src.endswith('.hip'):
            continue
        for name, (spill, scratch) in _kernel_metadata(os.path.join(_lib.LIB_DIR, os.path.splitext(src)[0] + '.o')).items():
            kernels += 1
            if spill:
                spills[name] = spill
            elif scratch:
                scratch_only[name] = scratch
    assert kernels > 250
    allowed = ('ppo_mlp_grad_kernel', 'ppo_mlp_grad_bf16_kernel',
               'ppo_wide_grad_kernelILi64ELi13ELi3ELb1ELi4ELi8E')   # hidden 512 as eight waves on a 256-register budget: 2 dwords
    assert all(any(a in k for a in allowed) for k in spills), spills
    assert len(spills) <= 7 and max(spills.values(), default=0) <= 6, spills
    assert all(any(t in k for t in ('squared_seed_kernel', 'spaces_tape_kernel')) for k in scratch_only), scratch_only
    # the round-4 kernels in particular: the width-templated rollout (up to 324 registers), the hidden-split gradient kernel (up to 498),
    # the one-launch reduce + Adam
    for frag in ('rollout_mlp_squared_kernel', 'ppo_wide_grad_kernel', 'ppo_reduce_adam_kernel', 'mlp_forward_sample_kernel', 'lstm_seq_bwd_kernel'):
        assert not any(frag in k and not any(a in k for a in allowed) for k in spills)


def test_matrix_products_switch_is_host_state_and_validates_its_argument():
    """pufferlib_amd.set_matrix_products / PFA_MATRIX_PRODUCTS: the opt-in product form is a process-wide flag of the library (no
    GPU needed to flip it); anything but the two names raises."""
    import pufferlib_amd
    from pufferlib_amd import _lib
    assert pufferlib_amd.get_matrix_products() == 'fp32'
    try:
        pufferlib_amd.set_matrix_products('bf16x6')
        assert pufferlib_amd.get_matrix_products() == 'bf16x6' and _lib.lib().pfa_igemm_get_products() == 1
        dims = _lib.MlpDims(49, 64, 128, 8, 0)
        import ctypes as C
        assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(dims), 131072) == 1       # the bench shape takes the bf16-path kernel ...
        assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(dims), 131072 + 16) == 0  # ... minibatches that are not whole 32-row tiles do not
        assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(_lib.MlpDims(64, 64, 128, 8, 0)), 131072) == 0   # nor other row widths
    finally:
        pufferlib_amd.set_matrix_products('fp32')
    dims = _lib.MlpDims(49, 64, 128, 8, 0)
    assert _lib.lib().pfa_ppo_mlp_grad_path(C.byref(dims), 131072) == 0
    with pytest.raises(ValueError):
        pufferlib_amd.set_matrix_products('tf32')


def test_every_header_a_source_includes_is_a_build_dependency():
    """_lib.HEADERS drives both the rebuild check and source_hash (the stamp of profiles/pmc_summary.json): a header a kernel source
    includes but the list misses means a library that silently is not rebuilt, and a stamp that does not see the change."""
    import re
    from pufferlib_amd import _lib
    listed = {os.path.basename(h) for h in _lib.HEADERS}
    for f in _lib.SOURCES + [h for h in _lib.HEADERS if not h.startswith('..')]:
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(_lib.CSRC, f)).read()):
            if inc.endswith(('.hpp', '.h')) and not inc.startswith('hip/'):
                assert os.path.basename(inc) in listed, (f, inc)
