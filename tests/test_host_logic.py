"""Host-side logic that needs no GPU: flat parameter layout vs the native offsets, obs stride selection, experience
shape checks, the rollout noise-row contract of the host path."""
import ctypes as C

import numpy as np
import pytest
import torch


def _policy(recurrent):
    from pufferlib_amd import cleanrl, models, vector
    env = vector.make_squared()
    base = models.Default(env)
    return cleanrl.RecurrentPolicy(models.LSTMWrapper(env, base)) if recurrent else cleanrl.Policy(base)


@pytest.mark.parametrize('recurrent', [False, True])
def test_flat_params_layout_matches_native_offsets(recurrent):
    from pufferlib_amd import _lib
    from pufferlib_amd.models import FlatParams
    pol = _policy(recurrent)
    before = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    fp = FlatParams(pol.policy, 64, 'cpu')
    L = _lib.lib()
    n_native = L.pfa_lstm_param_count(C.byref(fp.dims)) if recurrent else L.pfa_mlp_param_count(C.byref(fp.dims))
    assert fp.count == n_native == fp.flat.numel()
    # the module's parameters are now views of the flat buffer, values unchanged, pad columns zero
    for k, v in pol.state_dict().items():
        assert torch.equal(v, before[k]), k
        assert v.untyped_storage().data_ptr() == fp.flat.untyped_storage().data_ptr(), k
    w1p = fp.encoder_weight_padded()
    assert w1p.shape == (128, 64) and torch.count_nonzero(w1p[:, 49:]) == 0
    # writing through the flat buffer is visible through the module (what the Adam kernel relies on)
    fp.flat.add_(1.0)
    for k, v in pol.state_dict().items():
        assert torch.allclose(v, before[k] + 1.0), k
    # split() of another flat vector has the same shapes
    g = fp.split(torch.zeros_like(fp.flat))
    assert {k: tuple(v.shape) for k, v in g.items()} == {k.split('policy.')[-1] if not recurrent else
                                                         (k[len('policy.'):] if 'recurrent' in k else k[len('policy.policy.'):]): tuple(v.shape)
                                                         for k, v in before.items()}


def test_obs_stride_selection():
    from pufferlib_amd import hostpath
    assert [hostpath.obs_stride_for(d) for d in (1, 16, 17, 49, 64, 65, 96, 97, 128)] == [16, 16, 32, 64, 64, 96, 96, 128, 128]
    # wider rows run in the GEMM path (general.py): the next multiple of 16
    assert [hostpath.obs_stride_for(d) for d in (129, 300, 1024)] == [144, 304, 1024]
    # the recurrent path also takes MiniGrid-shaped 160-byte rows (BASELINE configs[2])
    assert [hostpath.obs_stride_for(d, True) for d in (49, 128, 129, 155, 160, 161)] == [64, 128, 160, 160, 160, 176]


def test_which_policy_shapes_take_the_gemm_path():
    """cleanrl.needs_general: the fused kernels cover Default(128) on rows of up to 128 (160 recurrent) floats with up to 15 logits
    and LSTMWrapper(128, 128); every other shape of the reference's models goes through general.py."""
    from pufferlib_amd import cleanrl, models, namespace, spaces

    def env(obs, act):
        return namespace(single_observation_space=spaces.Box(low=-1, high=1, shape=(obs,), dtype=np.float32), single_action_space=act)
    d8 = spaces.Discrete(8)
    assert not cleanrl.needs_general(models.Default(env(49, d8)), False)
    assert cleanrl.needs_general(models.Default(env(49, d8), hidden_size=256), False)
    assert cleanrl.needs_general(models.Default(env(300, d8)), False)
    assert cleanrl.needs_general(models.Default(env(49, spaces.Discrete(40))), False)
    assert not cleanrl.needs_general(models.LSTMWrapper(env(160, d8), models.Default(env(160, d8))), True)
    assert cleanrl.needs_general(models.LSTMWrapper(env(49, d8), models.Default(env(49, d8), hidden_size=256), input_size=256, hidden_size=256), True)
    frames = namespace(single_observation_space=spaces.Box(low=0, high=255, shape=(4, 84, 84), dtype=np.uint8), single_action_space=spaces.Discrete(4))
    assert not cleanrl.needs_general(models.Convolutional(frames), False)
    assert cleanrl.needs_general(models.LSTMWrapper(frames, models.Convolutional(frames), input_size=512, hidden_size=512), True)
    with pytest.raises(ValueError):
        models.Default(env(49, d8), hidden_size=100)


def test_experience_shape_checks_raise_before_touching_the_gpu():
    from pufferlib_amd.clean_pufferl import Experience
    with pytest.raises(ValueError):
        Experience(1000, 16, 300, 64, 10, 'cpu')       # batch not divisible by minibatch
    with pytest.raises(ValueError):
        Experience(1024, 16, 24, 64, 16, 'cpu')        # minibatch not divisible by bptt_horizon
    with pytest.raises(ValueError):
        Experience(1024, 16, 256, 64, 48, 'cpu')       # batch not divisible by the env count
    with pytest.raises(ValueError):
        Experience(16 * 24, 16, 128, 64, 16, 'cpu')    # rows per env not a whole number of bptt segments


def test_gae_halo_rows_travel_as_bit_patterns_and_come_from_as_many_later_shards_as_it_takes():
    """Host mirrors of csrc/gae.hip's gae_halo_publish / unpack kernels (pufferlib_amd.dist): a SUM of the ranks' zero-padded buffers
    returns every published float bit for bit (-0.0, denormals, inf), and a halo longer than a shard is filled from several."""
    from pufferlib_amd import dist as pdist
    assert pdist.gae_halo_rows(0.99, 0.95) == 544 and pdist.gae_halo_rows(0.999, 0.99) == 0 and pdist.gae_halo_rows(0.5, 0.0) == 16
    assert pdist.gae_halo_rows(0.995, 0.985) == 1640
    world, n, H = 4, 5, 12
    rng = np.random.RandomState(0)
    rows = [[rng.randn(n).astype(np.float32) for _ in range(3)] for _ in range(world)]
    rows[1][1][0], rows[2][2][4], rows[3][0][1] = np.float32(-0.0), np.float32(1e-42), np.float32(np.inf)
    total = sum(pdist.gae_halo_pack(*rows[q], q, world, H) for q in range(world))
    for q in range(world):
        halo = pdist.gae_halo_unpack(total, q, world, n, H)
        for k in range(3):
            want = np.concatenate([rows[p][k] for p in range(q + 1, world)] + [np.zeros(0, np.float32)])[:H]
            assert np.array_equal(halo[k].view(np.uint32), want.view(np.uint32)), (q, k)
    assert len(pdist.gae_halo_unpack(total, 0, world, n, H)[0]) == 12 and len(pdist.gae_halo_unpack(total, 3, world, n, H)[0]) == 0


@pytest.mark.parametrize('recurrent', [False, True])
def test_multidiscrete_policy_layout_head_packing_and_action_words(recurrent):
    """models.Default on a MultiDiscrete space: one decoder Linear per head with the reference's state_dict keys
    (models.py:29-35), all of them rows of ONE [A][H] block of the flat vector (what the kernels multiply), head sizes packed
    four bits each in pfa_mlp_dims.heads, and the kernel's action words unpacked to [rows, heads]."""
    from pufferlib_amd import _lib, cleanrl, models, namespace, spaces
    from pufferlib_amd.models import FlatParams
    nvec = [3, 4, 2, 5]
    env = namespace(single_observation_space=spaces.Box(low=-1, high=1, shape=(20,), dtype=np.float32),
                    single_action_space=spaces.MultiDiscrete(nvec))
    base = models.Default(env)
    pol = cleanrl.RecurrentPolicy(models.LSTMWrapper(env, base)) if recurrent else cleanrl.Policy(base)
    prefix = 'policy.policy.' if recurrent else 'policy.'
    keys = [k[len(prefix):] for k in pol.state_dict() if k.startswith(prefix)]
    assert keys == ['encoder.weight', 'encoder.bias'] + [f'decoder.{h}.{w}' for h in range(4) for w in ('weight', 'bias')] + \
        ['value_head.weight', 'value_head.bias']
    before = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    fp = FlatParams(pol.policy, 32, 'cpu')
    assert fp.multidiscrete and fp.nvec == nvec and fp.num_actions == 14
    assert fp.dims.heads == 3 | (4 << 4) | (2 << 8) | (5 << 12) and fp.dims.num_actions == 14
    L = _lib.lib()
    assert fp.count == (L.pfa_lstm_param_count if recurrent else L.pfa_mlp_param_count)(C.byref(fp.dims))
    for k, v in pol.state_dict().items():
        assert torch.equal(v, before[k]) and v.untyped_storage().data_ptr() == fp.flat.untyped_storage().data_ptr(), k
    # the heads are consecutive rows of one [14][128] block followed by one [14] bias vector
    o = 128 * 32 + 128
    block = fp.flat[o:o + 14 * 128].view(14, 128)
    assert torch.equal(block, torch.cat([before[f'{prefix}decoder.{h}.weight'] for h in range(4)]))
    assert torch.equal(fp.flat[o + 14 * 128:o + 14 * 128 + 14], torch.cat([before[f'{prefix}decoder.{h}.bias'] for h in range(4)]))
    words = torch.tensor([0, 2 | (3 << 4) | (1 << 8) | (4 << 12), 1 | (0 << 4) | (0 << 8) | (2 << 12)])
    assert fp.unpack_actions(words).tolist() == [[0, 0, 0, 0], [2, 3, 1, 4], [1, 0, 0, 2]]
    # a Discrete policy hands the words through unchanged
    single = FlatParams(_policy(False).policy, 64, 'cpu')
    assert not single.multidiscrete and single.dims.heads == 0 and single.unpack_actions(words) is words


def test_too_many_logits_or_heads_are_refused():
    from pufferlib_amd import cleanrl, models, namespace, spaces
    from pufferlib_amd.models import FlatParams
    box = spaces.Box(low=-1, high=1, shape=(4,), dtype=np.float32)
    for nvec in ([8, 8], [2] * 9):
        env = namespace(single_observation_space=box, single_action_space=spaces.MultiDiscrete(nvec))
        with pytest.raises(NotImplementedError):
            FlatParams(cleanrl.Policy(models.Default(env)).policy, 16, 'cpu')


def test_public_surface_has_the_names_reference_code_imports():
    """The drop-in boundary by name: what demo.py, policies and user code reach for in the reference modules this
    package mirrors."""
    import pufferlib_amd
    from pufferlib_amd import clean_pufferl, cleanrl, models, pytorch, vector
    for name in ('create', 'evaluate', 'train', 'close', 'Profile', 'make_losses', 'Experience', 'Utilization', 'save_checkpoint',
                 'try_load_checkpoint', 'count_params', 'seed_everything', 'print_dashboard'):
        assert hasattr(clean_pufferl, name), name                      # clean_pufferl.py:30-644
    for name in ('make', 'reset', 'step', 'make_seeds', 'RESET', 'SEND', 'RECV'):
        assert hasattr(vector, name), name                              # pufferlib/vector.py
    for name in ('Policy', 'RecurrentPolicy'):
        assert hasattr(cleanrl, name), name                             # pufferlib/frameworks/cleanrl.py:50-93
    for name in ('Default', 'LSTMWrapper'):
        assert hasattr(models, name), name                              # pufferlib/models.py
    for name in ('nativize_dtype', 'nativize_tensor', 'nativize_observation', 'flattened_tensor_size', 'layer_init',
                 'numpy_to_torch_dtype_dict'):
        assert hasattr(pytorch, name), name                             # pufferlib/pytorch.py
    assert hasattr(pufferlib_amd, 'namespace')
    pol = cleanrl.Policy(models.Default(vector.make_squared()))
    assert clean_pufferl.count_params(pol) == 128 * 49 + 128 + 8 * 128 + 8 + 128 + 1


def test_errors_are_also_the_reference_classes_once_pufferlib_exceptions_is_loaded():
    """pufferlib/exceptions.py:5-22: a caller that catches the reference's APIUsageError must catch ours, whichever package
    was imported first.  Uses the reference's own file when the tree is present (build container), a stand-in otherwise."""
    import importlib.util
    import os
    import sys
    import types
    from pufferlib_amd import exceptions as ours
    assert type(ours.APIUsageError('x')) is ours.APIUsageError                   # nothing loaded: plain class
    ref_file = '/root/reference/pufferlib/exceptions.py'
    saved = sys.modules.get('pufferlib.exceptions')
    try:
        if os.path.exists(ref_file):
            spec = importlib.util.spec_from_file_location('pufferlib.exceptions', ref_file)
            mod = importlib.util.module_from_spec(spec)
            sys.dont_write_bytecode, old = True, sys.dont_write_bytecode
            try:
                spec.loader.exec_module(mod)
            finally:
                sys.dont_write_bytecode = old
        else:
            mod = types.ModuleType('pufferlib.exceptions')

            class APIUsageError(RuntimeError):
                def __init__(self, message='API usage error.'):
                    self.message = message
                    super().__init__(self.message)

            class InvalidAgentError(ValueError):
                def __init__(self, agent_id, agents):
                    super().__init__(f'Invalid agent/team ({agent_id}) specified. Valid values:\n{agents}')
            mod.APIUsageError, mod.InvalidAgentError = APIUsageError, InvalidAgentError
        sys.modules['pufferlib.exceptions'] = mod
        with pytest.raises(mod.APIUsageError, match='Call reset before stepping') as ei:
            raise ours.APIUsageError('Call reset before stepping')
        assert isinstance(ei.value, ours.APIUsageError) and ei.value.message == 'Call reset before stepping'
        with pytest.raises(mod.InvalidAgentError):
            raise ours.InvalidAgentError(7, [1, 2])
    finally:
        if saved is None:
            sys.modules.pop('pufferlib.exceptions', None)
        else:
            sys.modules['pufferlib.exceptions'] = saved


def test_conv_params_flat_buffer_layout_and_aliasing():
    """models.ConvParams: one flat buffer in the reference's named_parameters order (models.py:126-140), every module parameter a
    view of it; the frame vecenv's creator token and spaces."""
    import torch
    from pufferlib_amd import models, vector
    spec = vector.make_frames(framestack=4, num_actions=4, episode_length=7)
    assert spec.single_observation_space.shape == (4, 84, 84) and spec.single_observation_space.dtype == np.uint8
    assert spec.single_action_space.n == 4 and spec.episode_length == 7
    net = models.Convolutional(spec, framestack=4)
    before = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cp = models.ConvParams(net, 'cpu')
    assert cp.count == 1686693 and cp.obs_dim == 4 * 84 * 84 and cp.num_actions == 4
    assert cp.names == ['network.0.weight', 'network.0.bias', 'network.2.weight', 'network.2.bias', 'network.4.weight', 'network.4.bias',
                        'network.7.weight', 'network.7.bias', 'actor.weight', 'actor.bias', 'value_fn.weight', 'value_fn.bias']
    o = 0
    for name, p in net.named_parameters():
        assert torch.equal(p.detach(), before[name])                                    # values kept
        assert p.data_ptr() == cp.flat.data_ptr() + 4 * o                              # and they now alias the flat buffer, in order
        o += p.numel()
    cp.flat.zero_()
    assert all(float(p.abs().sum()) == 0.0 for p in net.parameters())
    assert models.find_cnn(torch.nn.Sequential(net)) is net and models.find_cnn(torch.nn.Linear(2, 2)) is None
    with pytest.raises(NotImplementedError):
        models.Convolutional(spec, framestack=4, hidden_size=256)


class _FakePending:
    """Stands in for readback.Pending on a box without a GPU: counts resolutions, runs the continuation once."""

    def __init__(self, fn):
        self.fn, self.n = fn, 0

    def resolve(self):
        if self.fn is not None:
            fn, self.fn = self.fn, None
            self.n += 1
            fn()


def test_deferred_readback_containers_resolve_on_first_use():
    """readback.LazyDict / LazyLosses: nothing is waited for until the numbers are read; every way of reading them (item,
    attribute, iteration, unpacking, pickling, printing, truthiness) resolves exactly once and then behaves like the plain
    dict / namespace the reference returns (clean_pufferl.py:127-152, 369-378)."""
    import pickle
    from pufferlib_amd import readback as rb

    def lazy_dict(values):
        d = rb.LazyDict()
        p = _FakePending(lambda: d.fill(values))
        d._pending = p
        return d, p

    d, p = lazy_dict({'a': 1.0, 'b': 2.0})
    assert p.n == 0
    assert d['a'] == 1.0 and p.n == 1 and len(d) == 2 and dict(d) == {'a': 1.0, 'b': 2.0} and p.n == 1
    d, p = lazy_dict({'x': 3})
    assert {**d} == {'x': 3} and p.n == 1
    d, p = lazy_dict({'x': 3})
    assert pickle.loads(pickle.dumps(d)) == {'x': 3} and type(pickle.loads(pickle.dumps(d))) is dict
    d, p = lazy_dict({})
    assert not d and p.n == 1
    d, p = lazy_dict({'k': 1})
    assert [k for k in d] == ['k'] and 'k' in d and d.get('z', 7) == 7 and list(d.items()) == [('k', 1)] and p.n == 1
    d, p = lazy_dict({'k': 1})
    assert repr(d) == "{'k': 1}" and d == {'k': 1}
    d, p = lazy_dict({'k': 1})
    d2, p2 = lazy_dict({'k': 1})
    assert d == d2 and p.n == 1 and p2.n == 1    # both operands resolve (dict.__eq__ reads the other's storage directly)
    d, p = lazy_dict({'k': 1})
    d['j'] = 2                                   # a write resolves first, so the readback cannot overwrite it later
    assert dict(d) == {'k': 1, 'j': 2}

    def lazy_losses(**values):
        ns = rb.LazyLosses(**{k: 0 for k in values})
        p = _FakePending(lambda: ns.fill(**values))
        ns.attach(p)
        return ns, p

    ns, p = lazy_losses(policy_loss=1.5, value_loss=2.5)
    assert p.n == 0
    assert ns.policy_loss == 1.5 and p.n == 1 and ns['value_loss'] == 2.5 and dict(ns) == {'policy_loss': 1.5, 'value_loss': 2.5}
    ns, p = lazy_losses(policy_loss=9.0)
    assert list(ns.items()) == [('policy_loss', 9.0)] and p.n == 1
    ns, p = lazy_losses(policy_loss=9.0)
    assert 'policy_loss=9.0' in repr(ns)
    ns.policy_loss = 4
    assert ns.policy_loss == 4 and len(ns) == 1


def test_deferred_readback_error_surfaces_at_resolution():
    """The tape-underrun check travels with the readback: it raises where the statistics are first read."""
    from pufferlib_amd import readback as rb
    d = rb.LazyDict()

    def boom():
        raise RuntimeError('reset-target tape underrun')
    d._pending = _FakePending(boom)
    with pytest.raises(RuntimeError, match='underrun'):
        d.get('score')


def test_bench_self_spawn_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` (no torchrun) re-runs itself as N ranks; with fewer devices than ranks the ranks share devices over gloo."""
    import argparse
    import os
    import subprocess
    import sys as _sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _sys.path.insert(0, repo)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen.update(cmd=cmd, env=env)
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(_sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    monkeypatch.delenv('PFA_DIST_BACKEND', raising=False)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    assert bench.self_spawn(argparse.Namespace(gpus=4)) == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'PFA_DIST_BACKEND' not in seen['env']
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    assert bench.self_spawn(argparse.Namespace(gpus=4)) == 0
    assert seen['env']['PFA_DIST_BACKEND'] == 'gloo'
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 0)
    with pytest.raises(SystemExit):
        bench.self_spawn(argparse.Namespace(gpus=2))


def test_combined_exception_classes_survive_pickling(monkeypatch):
    """with pufferlib.exceptions loaded our errors are instances of BOTH packages' classes (a synthesised subclass);
    they must still cross a process boundary (multiprocessing / Ray / concurrent.futures pickle exceptions)."""
    import pickle
    import sys
    import types
    from pufferlib_amd import exceptions as ex
    fake = types.ModuleType('pufferlib.exceptions')

    class APIUsageError(RuntimeError):
        pass

    class InvalidAgentError(ValueError):
        pass
    fake.APIUsageError, fake.InvalidAgentError = APIUsageError, InvalidAgentError
    monkeypatch.setitem(sys.modules, 'pufferlib.exceptions', fake)
    e = ex.APIUsageError('Call reset before stepping')
    assert isinstance(e, APIUsageError) and isinstance(e, ex.APIUsageError) and type(e) is not ex.APIUsageError
    back = pickle.loads(pickle.dumps(e))
    assert isinstance(back, ex.APIUsageError) and back.message == 'Call reset before stepping'
    e2 = ex.InvalidAgentError(3, [1, 2])
    assert isinstance(e2, InvalidAgentError)
    back2 = pickle.loads(pickle.dumps(e2))
    assert isinstance(back2, ex.InvalidAgentError) and str(back2) == str(e2)


def test_general_params_layout_and_packed_operands_on_cpu():
    """general.GeneralParams / general.Net.pack need no GPU: the flat buffer holds every parameter in named_parameters() order with
    torch shapes (module parameters are views of it), and the packed operand forms are what the GEMM launches multiply — the
    encoder zero-padded to the row stride, the decoder rows of all heads + the value row stacked and padded to 16, [W_ih | W_hh]
    side by side with b_ih + b_hh."""
    from pufferlib_amd import general, models, namespace, spaces
    env = namespace(single_observation_space=spaces.Box(low=-1, high=1, shape=(37,), dtype=np.float32),
                    single_action_space=spaces.MultiDiscrete([3, 5, 2]))
    torch.manual_seed(0)
    base = models.Default(env, hidden_size=64)
    wrap = models.LSTMWrapper(env, base, input_size=64, hidden_size=96)
    # (the heads of `base` read 64 features, the LSTM emits 96: the reference would fail in decode_actions too — rebuild them)
    base.decoder = torch.nn.ModuleList([torch.nn.Linear(96, n) for n in (3, 5, 2)])
    base.value_head = torch.nn.Linear(96, 1)
    before = {k: v.detach().clone() for k, v in wrap.state_dict().items()}
    gp = general.GeneralParams(wrap, 'cpu')
    assert gp.kind == 'mlp' and gp.nvec == [3, 5, 2] and gp.multidiscrete and gp.heads == 3 | (5 << 4) | (2 << 8)
    assert gp.obs_dim == 37 and gp.obs_stride == 48 and gp.features == 64 and gp.head_in == 96
    assert gp.names == [n for n, _ in wrap.named_parameters()] and gp.count == sum(p.numel() for p in wrap.parameters())
    o = 0
    for name, p in wrap.named_parameters():                      # contiguous, in order, values unchanged, module params alias the buffer
        assert torch.equal(gp.flat[o:o + p.numel()].view(p.shape), before[name]) and p.data_ptr() == gp.flat[o:].data_ptr()
        o += p.numel()
    net = general._net_for_general(gp)
    net.pack()
    assert net.NO == 16 and net.FH == 96 and net.lstm == (64, 96)
    assert torch.equal(net.w1p[:, :37], before['policy.encoder.weight']) and float(net.w1p[:, 37:].abs().sum()) == 0.0
    want = torch.cat([before[f'policy.decoder.{h}.weight'] for h in range(3)] + [before['policy.value_head.weight']])
    assert torch.equal(net.w2v[:11], want) and float(net.w2v[11:].abs().sum()) == 0.0 and torch.equal(net.w2vT, net.w2v.t())
    assert torch.equal(net.b2v[:11], torch.cat([before[f'policy.decoder.{h}.bias'] for h in range(3)] + [before['policy.value_head.bias']]))
    assert torch.equal(net.wcat, torch.cat([before['recurrent.weight_ih_l0'], before['recurrent.weight_hh_l0']], dim=1))
    assert torch.equal(net.bcat, before['recurrent.bias_ih_l0'] + before['recurrent.bias_hh_l0'])
    # an optimizer step through the flat buffer + a version bump re-packs
    gp.flat.add_(1.0)
    net.version += 1
    net.pack()
    assert torch.equal(net.w1p[:, :37], before['policy.encoder.weight'] + 1.0) and float(net.w1p[:, 37:].abs().sum()) == 0.0
    assert gp.unpack_actions(torch.tensor([2 | (4 << 4) | (1 << 8)])).tolist() == [[2, 4, 1]]
    with pytest.raises(NotImplementedError):
        general.pack_heads([16, 2], True)


def test_six_term_bf16_split_is_as_close_to_f64_as_the_fp32_chain():
    """The arithmetic behind csrc/igemm.hip's opt-in product form, restated in numpy (tools/experiments/bf16_split_accuracy.py): an fp32
    value is the sum of its three bf16 pieces to 2^-23 relative, the six partial products above 2^-24 reproduce a K-long fp32 dot
    product about as well as an fp32 accumulation chain does, and the three-term form does not (which is why it is not offered)."""
    import importlib.util
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bf16_split_accuracy', os.path.join(repo, 'tools', 'experiments', 'bf16_split_accuracy.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rs = np.random.RandomState(3)
    x = (rs.standard_normal(4096) * np.exp(rs.uniform(-20, 20, 4096))).astype(np.float32)
    hi, mid, lo = m.split3(x)
    for piece in (hi, mid, lo):                        # every piece is a bf16 value: its low 16 bits are zero
        assert not (piece.view(np.uint32) & 0xFFFF).any()
    rebuilt = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.abs(rebuilt - x.astype(np.float64)).max() <= 2.0 ** -23 * np.abs(x).max() and np.all(np.abs(rebuilt - x) <= 2.0 ** -23 * np.abs(x))
    for K in (64, 576, 3136):
        a = np.maximum(rs.standard_normal((32, K)), 0).astype(np.float32)
        b = (rs.standard_normal((K, 32)) * np.sqrt(2.0 / K)).astype(np.float32)
        exact = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.abs(exact).max()
        ah, am, al = m.split3(a)
        bh, bm, bl = m.split3(b)
        x3 = m.matmul_f32_chain(ah, bh, 32) + m.matmul_f32_chain(ah, bm, 32) + m.matmul_f32_chain(am, bh, 32)
        x6 = x3 + m.matmul_f32_chain(ah, bl, 32) + m.matmul_f32_chain(al, bh, 32) + m.matmul_f32_chain(am, bm, 32)
        chain = m.matmul_f32_chain(a, b)
        e3, e6, ec = (np.abs(c.astype(np.float64) - exact).max() / scale for c in (x3, x6, chain))
        assert e6 <= 2.0 * ec + 1e-7 and e6 < 1e-6, (K, e6, ec)
        assert e3 > 4.0 * e6, (K, e3, e6)


class _FakeEvent:
    """An event that completes after `ready_after` queries (None: never by query); counts what the waiter did."""

    def __init__(self, ready_after):
        self.ready_after, self.queries, self.syncs = ready_after, 0, 0

    def query(self):
        self.queries += 1
        return self.ready_after is not None and self.queries > self.ready_after

    def synchronize(self):
        self.syncs += 1


def test_readback_wait_polls_first_and_blocks_after_the_window(monkeypatch):
    """readback.wait_event: the two per-iteration readbacks are polled (no parked thread to wake while the device idles); a wait
    that outlasts readback.SPIN_WAIT_US falls back to the runtime's blocking wait, and a window of 0 blocks at once."""
    from pufferlib_amd import readback as rb
    monkeypatch.setattr(rb, 'SPIN_WAIT_US', 100000.0)
    ev = _FakeEvent(ready_after=5)
    rb.wait_event(ev)
    assert ev.queries == 6 and ev.syncs == 0
    monkeypatch.setattr(rb, 'SPIN_WAIT_US', 200.0)
    ev = _FakeEvent(ready_after=None)
    rb.wait_event(ev)
    assert ev.queries >= 1 and ev.syncs == 1
    monkeypatch.setattr(rb, 'SPIN_WAIT_US', 0.0)
    ev = _FakeEvent(ready_after=0)
    rb.wait_event(ev)
    assert ev.queries == 0 and ev.syncs == 1


def test_early_gae_key_follows_in_place_edits_and_hyperparameters():
    """clean_pufferl._gae_key: evaluate() runs the update's GAE pass behind its statistics readback and train() reuses it only
    when this key still holds — an in-place write to rewards / dones / values through ANY view (reward shaping, value
    re-bootstrapping) or another gamma / lambda / partition makes train() run its own pass."""
    import torch
    from pufferlib_amd import clean_pufferl as cp
    from pufferlib_amd.namespace import Namespace
    B = 64
    rdv = torch.zeros(3, B + 1)
    ex = Namespace(_rdv=rdv, rewards=rdv[0, :B], dones=rdv[1, :B], values=rdv[2, :B], advantages=torch.zeros(B), returns=torch.zeros(B),
                   batch_size=B, num_envs=4, num_minibatches=2)
    data = Namespace(config=Namespace(gamma=0.99, gae_lambda=0.95, bptt_horizon=8, norm_adv=True), experience=ex)
    k0 = cp._gae_key(data)
    assert cp._gae_key(data) == k0
    ex.rewards[3] = 1.0                                   # a view of the shared storage: the storage's version counter moves
    k1 = cp._gae_key(data)
    assert k1 != k0
    ex.values.mul_(0.5)
    k2 = cp._gae_key(data)
    assert k2 != k1
    data.config.gamma = 0.98
    assert cp._gae_key(data) != k2
    data.config.gamma = 0.99
    assert cp._gae_key(data) == k2
    ex.num_minibatches = 4
    assert cp._gae_key(data) != k2
    ex.num_minibatches = 2
    assert cp._gae_key(data) == k2
    ex.values = ex.values.clone()                         # re-bound to another tensor (advisor, round 5): the address is part of the key
    assert cp._gae_key(data) != k2


def test_rank_affinity_plan_keeps_every_rank_next_to_its_gpu_and_off_its_neighbours_cores():
    """pufferlib_amd.dist.plan_affinity (the planning half of pin_rank): 8 ranks on a 2-socket host whose GPUs 0-3 / 4-7 hang off
    NUMA node 0 / 1 — every rank gets a quarter of ITS node's allowed CPUs, disjoint from its neighbours'; without NUMA information
    (VM, container) the allowed CPUs are split evenly; a mask too small to split is shared; cpulist parsing."""
    from pufferlib_amd import dist as pdist
    assert pdist._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = set(range(256))
    shares = [pdist.plan_affinity(r, 8, nodes, node_cpus, allowed) for r in range(8)]
    for r, (node, share) in enumerate(shares):
        assert node == nodes[r] and len(share) == 32 and set(share) <= set(node_cpus[node])
    assert len(set().union(*[set(s) for _, s in shares])) == 256               # disjoint and complete
    # a restricted mask (cgroup / taskset): only what is allowed, still on the right node
    node, share = pdist.plan_affinity(5, 8, nodes, node_cpus, set(range(60, 80)))
    assert node == 1 and set(share) <= set(range(64, 80)) and len(share) == 4
    # no NUMA information: an even split of the mask
    got = [pdist.plan_affinity(r, 2, [-1, -1], {}, set(range(8))) for r in range(2)]
    assert got == [(-1, [0, 1, 2, 3]), (-1, [4, 5, 6, 7])]
    # fewer allowed CPUs than ranks: everybody shares them (never an empty mask)
    assert pdist.plan_affinity(2, 4, [-1] * 4, {}, {3, 7}) == (-1, [3, 7])
    # the GPU's node has no allowed CPU at all: fall back to the mask
    assert pdist.plan_affinity(0, 2, [1, 1], node_cpus, {0, 1, 2, 3}) == (-1, [0, 1])
