"""ctypes binding of libpufferlib_amd.so (the C ABI in include/pufferlib_amd.h) + the hipcc build recipe.

The library is built IN-TREE (pufferlib_amd/_lib/libpufferlib_amd.so) so that it travels with the source
snapshot.  ``lib()`` raises ExtensionError when it is missing — there is deliberately no CPU fallback.
``import torch`` must happen before the library is loaded so that the HIP runtime already mapped by PyTorch
(same SONAME libamdhip64.so.7) is the one our kernels launch on; stream handles are then interchangeable.
"""
import ctypes as C
import os
import subprocess

from .exceptions import ExtensionError

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, '_lib')
LIB_PATH = os.path.join(LIB_DIR, 'libpufferlib_amd.so')
SOURCES = ['common.cpp', 'dist.cpp', 'p2p.hip', 'gae.hip', 'squared.hip', 'rollout.hip', 'ppo_update.hip', 'lstm.hip', 'gemm.hip', 'lstm_fused.hip', 'lstm_seq.hip', 'stochastic.hip', 'memory.hip', 'bandit.hip', 'multiagent.hip', 'spaces.hip', 'synthetic.hip', 'nativize.hip', 'igemm.hip', 'cnn_heads.hip', 'general.hip', 'ppo_wide.hip']
# every header under csrc/ (sorted: the order is part of source_hash) + the public C header: a header missing from a hand-kept list is a
# library that silently is not rebuilt when only that header changes
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.hpp')) + [os.path.join('..', '..', 'include', 'pufferlib_amd.h')]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def source_hash():
    """16 hex digits over every kernel source and header of the library, in build order: the identity profiles/pmc_summary.json is
    stamped with when the counters are collected and bench.py compares before it quotes them."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU.  One object per source, then link."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    import shlex
    extra = shlex.split(os.environ.get('PFA_HIPCC_FLAGS', ''))     # developer A/B builds (-DPFA_GAE_FMA=0, -DPFA_PROBES ...)
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + extra + ['-x', 'hip', '-c',
               os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise ExtensionError(f'hipcc failed on {src}:\n{out.decode()}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs + ['-ldl']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise ExtensionError(f'link failed:\n{r.stdout.decode()}')
    return LIB_PATH


class SquaredConfig(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('distance_to_target', C.c_int32), ('num_targets', C.c_int32),
                ('obs_stride', C.c_int32), ('tape_rounds', C.c_int32)]


NAT_MAX_FIELDS = 32


class NatField(C.Structure):
    _fields_ = [('out', C.c_void_p), ('offset', C.c_int32), ('count', C.c_int32), ('dtype', C.c_int32), ('to_f32', C.c_int32),
                ('out_stride', C.c_int32), ('reserved', C.c_int32)]


class MemoryConfig(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('mem_length', C.c_int32), ('mem_delay', C.c_int32), ('tape_rounds', C.c_int32)]


class SpacesConfig(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('tape_rounds', C.c_int32)]


class SynthConfig(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('obs_values', C.c_int32), ('obs_stride', C.c_int32), ('num_actions', C.c_int32),
                ('episode_length', C.c_int32), ('obs_high', C.c_int32), ('seed', C.c_uint64), ('env_offset', C.c_int64)]


class IgemmOperand(C.Structure):
    _fields_ = [('mode', C.c_int32), ('reserved', C.c_int32), ('ptr', C.c_void_p), ('lda', C.c_int64), ('IC', C.c_int32), ('IH', C.c_int32),
                ('IW', C.c_int32), ('OC', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32), ('S', C.c_int32)]


class MlpDims(C.Structure):
    _fields_ = [('obs_dim', C.c_int32), ('obs_stride', C.c_int32), ('hidden', C.c_int32), ('num_actions', C.c_int32),
                ('heads', C.c_uint32)]       # MultiDiscrete head sizes, 4 bits each (0 = one Discrete head)


class MlpView(C.Structure):
    _fields_ = [('w1', C.c_void_p), ('ldw1', C.c_int32), ('obs_dim', C.c_int32), ('obs_stride', C.c_int32), ('hidden', C.c_int32),
                ('num_actions', C.c_int32), ('reserved', C.c_int32), ('b1', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p),
                ('wv', C.c_void_p), ('bv', C.c_void_p)]


class NoiseKey(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('step', C.c_uint64)]


class Experience(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('actions', C.c_void_p), ('logprobs', C.c_void_p), ('values', C.c_void_p),
                ('rewards', C.c_void_p), ('dones', C.c_void_p), ('advantages', C.c_void_p), ('returns', C.c_void_p),
                ('horizon_T', C.c_int32)]


class ReduceJob(C.Structure):
    _fields_ = [('kind', C.c_int32), ('splits', C.c_int32), ('mo', C.c_int32), ('no', C.c_int32), ('partial', C.c_void_p), ('c', C.c_void_p),
                ('ldc', C.c_int64), ('c2', C.c_void_p), ('ldc2', C.c_int64), ('nb', C.c_int32), ('reserved', C.c_int32)]


class PpoHparams(C.Structure):
    _fields_ = [('clip_coef', C.c_float), ('vf_clip_coef', C.c_float), ('vf_coef', C.c_float), ('ent_coef', C.c_float),
                ('norm_adv', C.c_int32), ('clip_vloss', C.c_int32), ('num_minibatches', C.c_int32),
                ('bptt_horizon', C.c_int32)]


P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes) — must list every symbol include/pufferlib_amd.h declares
    'pfa_version': (C.c_int, []),
    'pfa_last_error': (C.c_char_p, []),
    'pfa_timing_enable': (C.c_int, [C.c_int]),
    'pfa_timing_select': (C.c_int, [C.c_char_p]),
    'pfa_timing_stride': (C.c_int, [C.c_int]),
    'pfa_timing_reset': (C.c_int, []),
    'pfa_timing_read': (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    'pfa_gae_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'pfa_gae_f32': (C.c_int, [P, P, P, P, P, C.c_int64, C.c_float, C.c_float, P, P]),
    'pfa_gae_sums_supported': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    'pfa_gae_sums_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int32]),
    'pfa_gae_sums_f32': (C.c_int, [P, P, P, P, P, C.c_int64, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P]),
    'pfa_gae_halo_rows': (C.c_int32, [C.c_float, C.c_float]),
    'pfa_gae_halo_publish': (C.c_int, [P, P, P, C.c_int64, C.c_float, C.c_float, P, C.c_int32, P, C.c_int32, C.c_int32, P]),
    'pfa_gae_halo_unpack': (C.c_int, [P, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, P, P, P, P]),
    'pfa_gae_halo_f32': (C.c_int, [P, P, P, P, P, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P]),
    'pfa_gae_shard_pass2': (C.c_int, [P, P, P, P, P, C.c_int64, C.c_int, C.c_float, C.c_float, P, P, P]),
    'pfa_gae_shard_publish': (C.c_int, [P, P, P, C.c_int64, C.c_float, C.c_float, P, P, C.c_int32, P, C.c_int32, C.c_int32, P]),
    'pfa_gae_shard_fold': (C.c_int, [P, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, P, P, P, P, P, P]),
    'pfa_train_ev_sums': (C.c_int, [C.POINTER(Experience), C.c_int64, C.c_int32, P, P, P]),
    'pfa_train_log_pack': (C.c_int, [P, P, P, P]),
    'pfa_squared_state_bytes': (C.c_size_t, [C.POINTER(SquaredConfig)]),
    'pfa_squared_async_reset': (C.c_int, [P, C.POINTER(SquaredConfig), C.c_int64, P, P, P, P, P, P]),
    'pfa_squared_fill_tape': (C.c_int, [P, C.POINTER(SquaredConfig), C.c_int32, P]),
    'pfa_squared_send': (C.c_int, [P, C.POINTER(SquaredConfig), P, P, P, P, P, P, P]),
    'pfa_squared_episode_stats': (C.c_int, [P, C.POINTER(SquaredConfig), P, C.c_int32, P]),
    'pfa_squared_last_infos': (C.c_int, [P, C.POINTER(SquaredConfig), P, P, P, P, P]),
    'pfa_squared_debug_targets': (C.c_int, [P, C.POINTER(SquaredConfig), P, P]),
    'pfa_squared_debug_stream_pos': (C.c_int, [P, C.POINTER(SquaredConfig), P, P]),
    'pfa_mlp_param_count': (C.c_int64, [C.POINTER(MlpDims)]),
    'pfa_mlp_forward_sample': (C.c_int, [P, C.c_int64, P, C.POINTER(MlpDims), P, C.POINTER(NoiseKey), C.c_int64,
                                         P, P, P, P, P]),
    'pfa_philox_exp_noise': (C.c_int, [P, C.c_int64, C.c_int64, C.c_int32, C.POINTER(NoiseKey), C.c_int64, P]),
    'pfa_rollout_mlp_squared': (C.c_int, [P, C.POINTER(SquaredConfig), P, C.POINTER(MlpDims), C.POINTER(Experience),
                                          P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P, P]),
    'pfa_mlp_view_supported': (C.c_int, [C.POINTER(MlpView)]),
    'pfa_ppo_wide_supported': (C.c_int, [C.POINTER(MlpView)]),
    'pfa_ppo_wide_workspace_bytes': (C.c_size_t, [C.POINTER(MlpView)]),
    'pfa_ppo_wide_grad': (C.c_int, [C.POINTER(Experience), C.c_int64, C.c_int32, C.POINTER(MlpView), C.POINTER(MlpView), P, C.POINTER(PpoHparams), P,
                                    C.c_int64, P, P]),
    'pfa_mlp_view_forward_sample': (C.c_int, [P, C.c_int64, C.POINTER(MlpView), P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P]),
    'pfa_rollout_mlp_view_squared': (C.c_int, [P, C.POINTER(SquaredConfig), C.POINTER(MlpView), C.POINTER(Experience), P, C.POINTER(NoiseKey),
                                               C.c_int64, P, P, P, P, P, P]),
    'pfa_stochastic_state_bytes': (C.c_size_t, [C.c_int32]),
    'pfa_stochastic_async_reset': (C.c_int, [P, C.c_int32, P, P, P, P, P, P]),
    'pfa_stochastic_send': (C.c_int, [P, C.c_int32, C.c_double, C.c_int32, P, P, P, P, P, P, P]),
    'pfa_stochastic_episode_stats': (C.c_int, [P, C.c_int32, P, C.c_int32, P]),
    'pfa_stochastic_last_infos': (C.c_int, [P, C.c_int32, P, P, P, P, P]),
    'pfa_rollout_mlp_stochastic': (C.c_int, [P, C.c_int32, C.c_double, C.c_int32, P, C.POINTER(MlpDims), C.POINTER(Experience),
                                             P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P, P]),
    'pfa_memory_state_bytes': (C.c_size_t, [C.POINTER(MemoryConfig)]),
    'pfa_memory_async_reset': (C.c_int, [P, C.POINTER(MemoryConfig), C.c_int64, P, P, P, P, P, P]),
    'pfa_memory_fill_tape': (C.c_int, [P, C.POINTER(MemoryConfig), C.c_int32, P]),
    'pfa_memory_send': (C.c_int, [P, C.POINTER(MemoryConfig), P, P, P, P, P, P, P]),
    'pfa_memory_episode_stats': (C.c_int, [P, C.POINTER(MemoryConfig), P, C.c_int32, P]),
    'pfa_memory_last_infos': (C.c_int, [P, C.POINTER(MemoryConfig), P, P, P, P, P]),
    'pfa_memory_debug_solutions': (C.c_int, [P, C.POINTER(MemoryConfig), P, P, P]),
    'pfa_bandit_state_bytes': (C.c_size_t, [C.c_int32]),
    'pfa_bandit_async_reset': (C.c_int, [P, C.c_int32, P, P, P, P, P, P]),
    'pfa_bandit_send': (C.c_int, [P, C.c_int32, C.c_int32, C.c_double, P, P, P, P, P, P, P, P]),
    'pfa_bandit_episode_stats': (C.c_int, [P, C.c_int32, P, C.c_int32, P]),
    'pfa_bandit_last_infos': (C.c_int, [P, C.c_int32, P, P, P, P, P]),
    'pfa_multiagent_state_bytes': (C.c_size_t, [C.c_int32]),
    'pfa_multiagent_async_reset': (C.c_int, [P, C.c_int32, P, P, P, P, P, P]),
    'pfa_multiagent_send': (C.c_int, [P, C.c_int32, P, P, P, P, P, P, P]),
    'pfa_spaces_state_bytes': (C.c_size_t, [C.POINTER(SpacesConfig)]),
    'pfa_spaces_async_reset': (C.c_int, [P, C.POINTER(SpacesConfig), C.c_int64, P, P, P, P, P, P]),
    'pfa_spaces_fill_tape': (C.c_int, [P, C.POINTER(SpacesConfig), C.c_int32, P]),
    'pfa_spaces_send': (C.c_int, [P, C.POINTER(SpacesConfig), P, P, P, P, P, P, P]),
    'pfa_spaces_episode_stats': (C.c_int, [P, C.POINTER(SpacesConfig), P, C.c_int32, P]),
    'pfa_spaces_last_infos': (C.c_int, [P, C.POINTER(SpacesConfig), P, P, P, P, P]),
    'pfa_synth_state_bytes': (C.c_size_t, [C.POINTER(SynthConfig)]),
    'pfa_synth_async_reset': (C.c_int, [P, C.POINTER(SynthConfig), P, P, P, P, P, P]),
    'pfa_synth_send': (C.c_int, [P, C.POINTER(SynthConfig), P, P, P, P, P, P, P]),
    'pfa_frames_async_reset': (C.c_int, [P, C.POINTER(SynthConfig), P, P, P, P, P, P]),
    'pfa_frames_send': (C.c_int, [P, C.POINTER(SynthConfig), P, P, P, P, P, P, P]),
    'pfa_synth_episode_stats': (C.c_int, [P, C.POINTER(SynthConfig), P, C.c_int32, P]),
    'pfa_synth_last_infos': (C.c_int, [P, C.POINTER(SynthConfig), P, P, P, P, P]),
    'pfa_multiagent_episode_stats': (C.c_int, [P, C.c_int32, P, C.c_int32, P]),
    'pfa_nativize_rows': (C.c_int, [P, C.c_int64, C.c_int32, P, C.c_int32, P]),
    'pfa_ppo_workspace_bytes': (C.c_size_t, [C.POINTER(MlpDims), C.c_int64, C.POINTER(PpoHparams)]),
    'pfa_ppo_adv_stats': (C.c_int, [C.POINTER(Experience), C.c_int64, C.POINTER(PpoHparams), P, P, P]),
    'pfa_ppo_mlp_grad': (C.c_int, [C.POINTER(Experience), C.c_int64, C.c_int32, P, C.POINTER(MlpDims),
                                   C.POINTER(PpoHparams), P, C.c_int64, P, P, P]),
    'pfa_ppo_mlp_train': (C.c_int, [C.POINTER(Experience), C.c_int64, P, C.POINTER(MlpDims), C.POINTER(PpoHparams), P, P, P, P,
                                    C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, P, P,
                                    C.c_int32, P]),
    'pfa_ppo_mlp_train_logged': (C.c_int, [C.POINTER(Experience), C.c_int64, P, C.POINTER(MlpDims), C.POINTER(PpoHparams), P, P, P, P,
                                           C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, P, P,
                                           C.c_int32, P, P, C.POINTER(C.c_int32), P]),
    'pfa_store_step': (C.c_int, [C.POINTER(Experience), C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P, P, P]),
    'pfa_store_rows': (C.c_int, [C.POINTER(Experience), C.c_int32, C.c_int32, C.c_int32, P, P, P, P, P, P, P, P, P, P, P]),
    'pfa_gather_obs_time_major': (C.c_int, [C.POINTER(Experience), C.c_int64, C.c_int32, C.POINTER(PpoHparams), C.c_int32, P, P]),
    'pfa_lstm_heads_loss_workspace_bytes': (C.c_size_t, []),
    'pfa_lstm_heads_loss': (C.c_int, [P, C.POINTER(Experience), C.c_int64, C.c_int32, P, C.POINTER(MlpDims),
                                      C.POINTER(PpoHparams), P, C.c_int64, P, P, P, P, P, P]),
    'pfa_lstm_param_count': (C.c_int64, [C.POINTER(MlpDims)]),
    'pfa_lstm_pack_bytes': (C.c_size_t, []),
    'pfa_lstm_pack': (C.c_int, [P, C.POINTER(MlpDims), P, P]),
    'pfa_lstm_policy_step': (C.c_int, [P, C.c_int64, P, C.POINTER(MlpDims), P, P, P, P, C.POINTER(NoiseKey), C.c_int64,
                                       P, P, P, P, P]),
    'pfa_rollout_lstm_squared': (C.c_int, [P, C.POINTER(SquaredConfig), P, C.POINTER(MlpDims), P, P, P, C.POINTER(Experience),
                                           P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P, P]),
    'pfa_rollout_lstm_memory': (C.c_int, [P, C.POINTER(MemoryConfig), P, C.POINTER(MlpDims), P, P, P, C.POINTER(Experience),
                                           P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P, P]),
    'pfa_rollout_lstm_synth': (C.c_int, [P, C.POINTER(SynthConfig), P, C.POINTER(MlpDims), P, P, P, C.POINTER(Experience),
                                           P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P, P]),
    'pfa_lstm_pack_bwd': (C.c_int, [P, C.POINTER(MlpDims), P, P]),
    'pfa_lstm_seq_forward': (C.c_int, [P, C.c_int64, C.c_int32, P, C.POINTER(MlpDims), P, P, P, P, P, C.c_int32, P]),
    'pfa_lstm_pack_both': (C.c_int, [P, C.POINTER(MlpDims), P, P, P]),
    'pfa_lstm_seq_backward_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'pfa_lstm_seq_backward': (C.c_int, [P, P, P, P, C.c_int64, C.c_int32, P, P, P, P, P, P, P]),
    'pfa_gemm_tn_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int64]),
    'pfa_gemm_tn_f32': (C.c_int, [P, C.c_int64, P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, P, P]),
    'pfa_gemm_tn2_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int64]),
    'pfa_gemm_tn_partial_f32': (C.c_int, [P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int32, C.c_int64, P, C.POINTER(ReduceJob), P]),
    'pfa_gemm_tn2_partial_f32': (C.c_int, [P, C.c_int64, P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int64, P, C.POINTER(ReduceJob), P]),
    'pfa_reduce_multi': (C.c_int, [C.POINTER(ReduceJob), C.c_int32, P]),
    'pfa_gemm_tn2_f32': (C.c_int, [P, C.c_int64, P, C.c_int64, P, C.c_int64, P, C.c_int64, P, C.c_int64, C.c_int32, C.c_int64, P, P]),
    'pfa_lstm_finish_grads': (C.c_int, [P, C.POINTER(MlpDims), P, P, P]),
    'pfa_sumsq_partials': (C.c_int, [P, C.c_int64, P, C.c_int32, P]),
    'pfa_dist_unique_id': (C.c_int, [P]),
    'pfa_dist_init': (C.c_int, [P, C.c_int32, C.c_int32]),
    'pfa_dist_finalize': (C.c_int, []),
    'pfa_dist_all_reduce_f32': (C.c_int, [P, C.c_int64, P]),
    'pfa_dist_all_reduce_f64': (C.c_int, [P, C.c_int64, P]),
    'pfa_dist_info': (C.c_int, [P]),
    'pfa_ppo_mlp_grad_mfma_per_tile': (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    'pfa_ppo_mlp_grad_path': (C.c_int, [C.POINTER(MlpDims), C.c_int64]),
    'pfa_heads_rows_sample': (C.c_int, [P, C.c_int32, C.c_int64, C.c_int32, C.c_uint32, P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P]),
    'pfa_heads_rows_eval': (C.c_int, [P, C.c_int32, C.c_int64, C.c_int32, C.c_uint32, P, P, P, P, P]),
    'pfa_heads_rows_loss_workspace_bytes': (C.c_size_t, [C.c_int64]),
    'pfa_heads_rows_loss': (C.c_int, [P, C.c_int32, C.POINTER(Experience), C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_uint32, C.POINTER(PpoHparams), P, C.c_int64, P, C.c_int32, C.c_int32, P, C.c_int32, P, P]),
    'pfa_lstm_cell_forward': (C.c_int, [P, P, P, P, C.c_int32, P, C.c_int32, C.c_int64, C.c_int32, P]),
    'pfa_lstm_cell_backward': (C.c_int, [P, C.c_int32, P, C.c_int32, P, P, P, P, P, C.c_int64, C.c_int32, P]),
    'pfa_rows_perm': (C.c_int, [P, C.c_int32, P, C.c_int32, P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P]),
    'pfa_igemm_set_products': (C.c_int, [C.c_int32]),
    'pfa_igemm_get_products': (C.c_int, []),
    'pfa_igemm_rows': (C.c_int, [C.POINTER(IgemmOperand), C.c_int64, C.c_int32, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, P, P, C.c_int32, P]),
    'pfa_igemm_weights_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    'pfa_igemm_weights': (C.c_int, [C.POINTER(IgemmOperand), C.c_int64, C.c_int32, P, C.c_int32, C.c_int32, P, C.c_int32, C.c_int32, P, P, P]),
    'pfa_colsum_workspace_bytes': (C.c_size_t, [C.c_int32]),
    'pfa_colsum': (C.c_int, [P, C.c_int64, C.c_int32, C.c_int32, P, C.c_int32, P, P]),
    'pfa_cnn_pack_conv': (C.c_int, [P, C.POINTER(IgemmOperand), C.c_int32, P, P, P]),
    'pfa_cnn_transpose': (C.c_int, [P, C.c_int32, C.c_int32, P, P]),
    'pfa_cnn_pack_fc': (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, P, P, P]),
    'pfa_cnn_heads_sample': (C.c_int, [P, C.c_int64, P, P, P, P, C.c_int32, P, C.POINTER(NoiseKey), C.c_int64, P, P, P, P, P]),
    'pfa_cnn_heads_loss_workspace_bytes': (C.c_size_t, []),
    'pfa_cnn_heads_loss': (C.c_int, [P, C.POINTER(Experience), C.c_int64, C.c_int32, C.c_int64, C.c_int64, P, P, P, P, C.c_int32,
                                     C.POINTER(PpoHparams), P, C.c_int64, P, P, P, C.c_int32, P, P]),
    'pfa_cnn_gather_frames': (C.c_int, [P, C.c_int64, C.c_int64, C.c_int32, C.POINTER(PpoHparams), C.c_int64, C.c_int64, P, P]),
    'pfa_p2p_alloc': (C.c_int, [C.c_int64, C.c_int32, P]),
    'pfa_p2p_open': (C.c_int, [P, C.c_int32, C.c_int32]),
    'pfa_p2p_close': (C.c_int, []),
    'pfa_p2p_status': (C.c_int, []),
    'pfa_ppo_grid_status': (C.c_int, []),
    'pfa_ppo_grid_reset': (C.c_int, []),
    'pfa_p2p_seq': (C.c_int64, []),
    'pfa_p2p_reset': (C.c_int, [C.c_int64]),
    'pfa_p2p_debug_set_status': (C.c_int, [C.c_int]),
    'pfa_p2p_all_reduce_f32': (C.c_int, [P, C.c_int64, P]),
    'pfa_p2p_all_reduce_f64': (C.c_int, [P, C.c_int64, P]),
    'pfa_p2p_ll_all_reduce_f32': (C.c_int, [P, C.c_int64, P]),
    'pfa_p2p_ll_calls': (C.c_int64, []),
    'pfa_p2p_wait_stats': (C.c_int, [P, C.c_int]),
    'pfa_p2p_enable': (C.c_int, [C.c_int]),
    'pfa_train_log_sums': (C.c_int, [C.POINTER(Experience), C.c_int64, C.c_int32, P, P, P, P]),
    'pfa_adam_clip_step': (C.c_int, [P, P, P, P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64,
                                     C.c_float, C.c_float, P, P, C.c_double, P, C.c_int32, P]),
}

_lib = None


def lib():
    """Load the shared library (after torch, see module docstring) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionError(
            f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  pufferlib_amd has no CPU fallback.')
    import torch  # noqa: F401  (maps libamdhip64.so.7 first)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ExtensionError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            f = getattr(L, name)
        except AttributeError as e:
            raise ExtensionError(f'{LIB_PATH} does not export {name}') from e
        f.restype = res
        f.argtypes = args
    _lib = L
    form = os.environ.get('PFA_MATRIX_PRODUCTS', 'fp32')
    if form not in ('fp32', 'bf16x6'):
        raise ExtensionError(f"PFA_MATRIX_PRODUCTS={form!r}: expected 'fp32' or 'bf16x6'")
    if form == 'bf16x6':
        check(L.pfa_igemm_set_products(1), 'PFA_MATRIX_PRODUCTS')
    return L


def check(rc, what=''):
    if rc != 0:
        msg = lib().pfa_last_error().decode(errors='replace')
        raise ExtensionError(f'{what} failed ({rc}): {msg}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_handle():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise ExtensionError('pufferlib_amd needs a ROCm GPU (torch.cuda.is_available() is False); '
                             'there is no CPU fallback')
