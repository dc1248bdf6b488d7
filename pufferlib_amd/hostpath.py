"""Rollout against a HOST vecenv: any backend that speaks the reference's protocol —
``pufferlib.vector.Serial`` / ``Multiprocessing`` over arbitrary CPU envs, async EnvPool batches — feeds the device-resident
trainer.  The loop is the reference's (clean_pufferl.py:84-124):

    recv() -> [pinned staging, async H2D] -> policy forward + sample on device -> Experience.store on device
           -> actions D2H (the one sync per step, the host env needs them) -> send()

What differs from the reference: ``Experience.store`` + ``sort_training_data`` (clean_pufferl.py:436-464) are one kernel
(``pfa_store_rows``): every arriving row is written straight to its sorted position ``env_id*T + rows_so_far[env_id]`` of the
env-major device buffers, so out-of-order ``env_id`` batches need no sort and the update path is the same as for the
device-resident envs.  Rows of a batch are ordered by ``env_id`` before they are staged; the reference's backends hand out
contiguous ``env_id`` slices (vector.py:158-162, :395-398), for which the Philox noise row of an agent is its global index —
the same stream the fused device rollout uses, so both paths sample identical actions.
"""
import ctypes as C
from collections import defaultdict

import numpy as np
import torch

from . import _lib, utils
from .cleanrl import obs_stride_for  # noqa: F401  (re-exported: create() and the tests size host rows with it)


class HostBridge:
    """Pinned staging buffers of one host vecenv and their device mirrors."""

    def __init__(self, vecenv, obs_stride, device, frames=False):
        """`frames`: the observations stay bytes (uint8 rows of obs_stride = prod(shape) bytes, for the conv policy) instead of being
        cast to padded float rows."""
        space = vecenv.single_observation_space
        self.obs_dim = int(np.prod(space.shape))
        self.obs_stride = obs_stride
        obs_dtype = torch.uint8 if frames else torch.float32
        self.total_agents = int(vecenv.num_agents)
        n = int(getattr(vecenv, 'agents_per_batch', self.total_agents))
        self.max_rows = n
        pin = dict(pin_memory=True)
        self.obs_pin = torch.zeros(n, obs_stride, dtype=obs_dtype, **pin)
        self.rew_pin = torch.zeros(n, dtype=torch.float32, **pin)
        self.done_pin = torch.zeros(n, dtype=torch.uint8, **pin)
        self.mask_pin = torch.zeros(n, dtype=torch.uint8, **pin)
        self.ids_pin = torch.zeros(n, dtype=torch.int32, **pin)
        self.act_pin = torch.zeros(n, dtype=torch.int64, **pin)
        self.obs = torch.zeros(n, obs_stride, dtype=obs_dtype, device=device)
        self.rew = torch.zeros(n, dtype=torch.float32, device=device)
        self.done = torch.zeros(n, dtype=torch.uint8, device=device)
        self.mask = torch.zeros(n, dtype=torch.uint8, device=device)
        self.ids = torch.zeros(n, dtype=torch.int32, device=device)
        self.ids64 = torch.zeros(n, dtype=torch.int64, device=device)
        self.counters = torch.zeros(self.total_agents, dtype=torch.int32, device=device)
        self.stored_dropped = torch.zeros(2, dtype=torch.int32, device=device)
        self.stored_pin = torch.zeros(2, dtype=torch.int32, **pin)

    def upload(self, o, r, d, env_id, mask):
        """Stage one recv() (numpy arrays, any row order) and start its H2D copies.  Returns (rows, order) with
        ``order`` the permutation that sorts the rows by env_id (None when they already are)."""
        env_id = np.asarray(env_id)
        n = len(env_id)
        if n > self.max_rows:
            raise ValueError(f'recv() returned {n} rows, vecenv advertised {self.max_rows} agents per batch')
        order = None
        if n > 1 and np.any(env_id[1:] < env_id[:-1]):
            order = np.argsort(env_id, kind='stable')
        pick = (lambda x: x[order]) if order is not None else (lambda x: x)
        flat = np.asarray(o).reshape(n, -1)
        if flat.shape[1] != self.obs_dim:
            raise ValueError(f'observation rows of {flat.shape[1]} values, expected {self.obs_dim}')
        self.obs_pin.numpy()[:n, :self.obs_dim] = pick(flat)          # .float() of models.Default (models.py:50)
        self.rew_pin.numpy()[:n] = pick(np.asarray(r))
        self.done_pin.numpy()[:n] = pick(np.asarray(d)).astype(np.uint8)
        self.mask_pin.numpy()[:n] = pick(np.asarray(mask)).astype(np.uint8)
        self.ids_pin.numpy()[:n] = pick(env_id)
        for dev, host in ((self.obs, self.obs_pin), (self.rew, self.rew_pin), (self.done, self.done_pin),
                          (self.mask, self.mask_pin), (self.ids, self.ids_pin)):
            dev[:n].copy_(host[:n], non_blocking=True)
        return n, order

    def download_actions(self, actions_dev, n, order, nvec=None):
        """Device actions (env_id order) -> host numpy in the row order recv() used; ``nvec`` unpacks the per-head choices of
        a MultiDiscrete policy into [n, heads] (what vecenv.send expects, vector.py:139-141).  Synchronises."""
        self.act_pin[:n].copy_(actions_dev[:n], non_blocking=True)
        self.stored_pin.copy_(self.stored_dropped, non_blocking=True)   # rides on the same sync: rows stored so far
        torch.cuda.current_stream().synchronize()
        a = self.act_pin.numpy()[:n]
        if nvec is not None:
            a = (a[:, None] >> (4 * np.arange(len(nvec)))) & 15
        if order is None:
            return a.copy()
        out = np.empty_like(a)
        out[order] = a
        return out


def utils_cfg(config, key, default):
    return getattr(config, key, default) if not isinstance(config, dict) else config.get(key, default)


def _arrival_buffers(data, experience, bridge):
    """Arrival-order staging of one rollout for config.async_store = 'reference' (allocated once)."""
    a = getattr(data, '_arrival', None)
    if a is None:
        B, dev = experience.batch_size, experience.obs.device
        a = dict(obs=torch.zeros_like(experience.obs), rewards=torch.zeros(B, device=dev), dones=torch.zeros(B, device=dev),
                 actions=torch.zeros(B, dtype=torch.int32, device=dev), logprobs=torch.zeros(B, device=dev), values=torch.zeros(B, device=dev),
                 keys=torch.zeros(B, dtype=torch.int64, device=dev))
        data._arrival = a
    a['dropped'] = 0
    return a


def evaluate(data):
    """clean_pufferl.evaluate (clean_pufferl.py:76-154) for a host vecenv."""
    config, profile, experience, vecenv, policy = data.config, data.profile, data.experience, data.vecenv, data.policy
    L = _lib.lib()
    fp, bridge = data.flat_params, data.host_bridge
    A = fp.num_actions
    stream = _lib.stream_handle()
    infos = defaultdict(list)
    if getattr(data, 'lstm_engine', None) is not None:
        data.lstm_engine.invalidate_obs_cache()     # the experience rows are about to be rewritten
    bridge.counters.zero_()
    bridge.stored_dropped.zero_()
    recvs = 0
    stored = 0
    # config.async_store = 'reference': Experience.store / sort_training_data exactly as the reference runs them on a pool whose
    # agents report unevenly (clean_pufferl.py:436-464) — the first batch_size masked rows in ARRIVAL order, then one stable sort by
    # (env_id, step); agents then own runs of different lengths in the flat batch, which GAE / the minibatch partition / the update
    # kernels never look at (they walk flat rows).  Default 'balanced': every agent contributes exactly batch_size / num_agents
    # rows, written straight to their sorted place, surplus rows of fast agents dropped (one kernel, no sort).
    exact = utils_cfg(config, 'async_store', 'balanced') == 'reference'
    if exact:
        arrival = _arrival_buffers(data, experience, bridge)
        arr_ptr = 0
    eng = data.lstm_engine
    # `while not experience.full` (clean_pufferl.py:84).  "Full" = every agent has its batch_size / num_agents rows: a genuinely
    # async pool (fast workers return more often, vector.py:382-390) keeps being stepped until the slow agents have caught up;
    # surplus rows of an agent that is already complete are acted on but not stored (counted in stored_dropped[1]).
    max_recvs = 64 * max(1, experience.batch_size // max(1, bridge.max_rows)) + 64
    while stored < experience.batch_size:
        with profile.env:
            o, r, d, t, info, env_id, mask = vecenv.recv()
        with profile.eval_misc:
            n, order = bridge.upload(o, r, d, env_id, mask)
            nmask = int(np.sum(mask))
            data.global_step += nmask * data.world_size             # clean_pufferl.py:90, all ranks
        with profile.eval_forward:
            key = _lib.NoiseKey(policy.noise_seed, policy.noise_step)
            row0 = data.env_offset + int(bridge.ids_pin[0]) if n else data.env_offset
            noise = None
            if data.noise is not None:                               # explicit Exp(1) draws [recv][agent][A] (parity tests)
                bridge.ids64[:n].copy_(bridge.ids[:n])
                noise = data.noise[recvs].to(device=bridge.obs.device, dtype=torch.float32).index_select(0, bridge.ids64[:n]).contiguous()
            actions = torch.empty(n, dtype=torch.int64, device=bridge.obs.device)
            logprob = torch.empty(n, device=bridge.obs.device)
            value = torch.empty(n, device=bridge.obs.device)
            if data.gen_engine is not None:                          # a policy shape outside the fused kernels (general.py)
                ids = None
                if data.gen_engine.lstm_h is not None:               # state rows of these agents (clean_pufferl.py:100-105)
                    bridge.ids64[:n].copy_(bridge.ids[:n])
                    ids = bridge.ids64[:n]
                data.gen_engine.policy_step(bridge.obs, n, noise, key, row0, actions, logprob, None, value, ids=ids)
            elif data.cnn_engine is not None:                        # models.Convolutional on the staged uint8 frames
                data.cnn_engine.policy_step(bridge.obs, n, noise, key, row0, actions, logprob, None, value)
            elif eng is None:
                _lib.check(L.pfa_mlp_forward_sample(_lib.ptr(bridge.obs), n, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(noise),
                                                    C.byref(key), row0, _lib.ptr(actions), _lib.ptr(logprob), None, _lib.ptr(value),
                                                    stream), 'forward_sample')
            else:                                                    # state rows of these agents (clean_pufferl.py:100-105)
                from . import lstm as plstm
                bridge.ids64[:n].copy_(bridge.ids[:n])
                idx = bridge.ids64[:n]
                h, c = eng.lstm_h[0].index_select(0, idx), eng.lstm_c[0].index_select(0, idx)
                if recvs == 0:
                    plstm.pack_gates(fp, eng.wpack)
                _lib.check(L.pfa_lstm_policy_step(_lib.ptr(bridge.obs), n, _lib.ptr(fp.flat), C.byref(fp.dims), _lib.ptr(eng.wpack),
                                                  _lib.ptr(h), _lib.ptr(c), _lib.ptr(noise), C.byref(key), row0, _lib.ptr(actions),
                                                  _lib.ptr(logprob), None, _lib.ptr(value), stream), 'lstm_policy_step')
                eng.lstm_h[0].index_copy_(0, idx, h)
                eng.lstm_c[0].index_copy_(0, idx, c)
            policy.noise_step += 1
        if exact:
            with profile.eval_misc:
                m_host = bridge.mask_pin.numpy()[:n].astype(bool)
                take = np.nonzero(m_host)[0][:experience.batch_size - arr_ptr]     # indices = where(mask)[:batch_size - ptr] (clean_pufferl.py:439-440)
                k = len(take)
                if k:
                    idx = torch.as_tensor(take, dtype=torch.int64).to(bridge.obs.device, non_blocking=True)
                    sl = slice(arr_ptr, arr_ptr + k)
                    arrival['obs'][sl] = bridge.obs.index_select(0, idx)
                    arrival['rewards'][sl] = bridge.rew.index_select(0, idx)
                    arrival['dones'][sl] = bridge.done.index_select(0, idx).float()
                    arrival['actions'][sl] = actions.index_select(0, idx).to(torch.int32)
                    arrival['logprobs'][sl] = logprob.index_select(0, idx)
                    arrival['values'][sl] = value.index_select(0, idx)
                    arrival['keys'][sl] = (bridge.ids.index_select(0, idx).to(torch.int64) << 32) + recvs     # sort key (env_id, step)
                arrival['dropped'] += int(m_host.sum()) - k
                arr_ptr += k
                actions_np = bridge.download_actions(actions, n, order, fp.nvec if fp.multidiscrete else None)
                stored = arr_ptr
                recvs += 1
                if recvs > max_recvs:
                    raise RuntimeError(f'host rollout: {stored} of {experience.batch_size} rows after {recvs} recv() calls')
                for i in info:
                    for kk, v in utils.unroll_nested_dict(i):
                        infos[kk].append(v)
            with profile.env:
                vecenv.send(actions_np)
            continue
        with profile.eval_misc:
            # (frame rows are bytes: the copy moves them as obs_dim / 4 four-byte words)
            _lib.check(L.pfa_store_rows(C.byref(experience.c), n, bridge.total_agents,
                                        fp.obs_dim // 4 if (data.cnn_engine is not None or (data.gen_engine is not None and data.gen_engine.net.kind == 'cnn')) else fp.obs_stride, _lib.ptr(bridge.obs),
                                        _lib.ptr(bridge.rew), _lib.ptr(bridge.done), _lib.ptr(actions), _lib.ptr(logprob),
                                        _lib.ptr(value), _lib.ptr(bridge.ids), _lib.ptr(bridge.mask), _lib.ptr(bridge.counters),
                                        _lib.ptr(bridge.stored_dropped), stream), 'store_rows')
            actions_np = bridge.download_actions(actions, n, order, fp.nvec if fp.multidiscrete else None)
            stored = int(bridge.stored_pin[0])
            recvs += 1
            if recvs > max_recvs:
                raise RuntimeError(f'host rollout: {stored} of {experience.batch_size} rows after {recvs} recv() calls — some agents '
                                   'never report (env-major experience needs batch_size / num_agents rows from every agent)')
            for i in info:                                           # clean_pufferl.py:110-113
                for k, v in utils.unroll_nested_dict(i):
                    infos[k].append(v)
        with profile.env:
            vecenv.send(actions_np)

    if exact:
        with profile.eval_misc:
            # sort_training_data (clean_pufferl.py:452-464): one stable sort by (env_id, step), rows gathered into the trainer's buffers
            perm = torch.argsort(arrival['keys'], stable=True)
            for name in ('obs', 'rewards', 'dones', 'actions', 'logprobs', 'values'):
                getattr(experience, name).copy_(arrival[name].index_select(0, perm))
            data.arrival_values = arrival['values']                 # y_pred of the reference's explained-variance line is in storage order
            data.sort_perm = perm
    with profile.eval_misc:
        data.host_rows_dropped = arrival['dropped'] if exact else int(bridge.stored_pin[1])   # surplus rows (async pools)
        data.noise = None
        experience.ptr = experience.batch_size
        experience.step = experience.horizon
        data.stats = {}
        for k, v in infos.items():                                   # clean_pufferl.py:127-137
            if '_map' in k and data.wandb is not None:
                data.stats[f'Media/{k}'] = data.wandb.Image(v[0])
                continue
            try:
                data.stats[k] = np.mean(v)
            except Exception:
                continue
    return data.stats, infos
