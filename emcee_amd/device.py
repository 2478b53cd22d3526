"""DeviceEnsemble: the Python handle on one libemx context (one ensemble on one MI355X).

This is the host side of the C ABI in include/emx.h; it owns no algorithmic logic beyond
argument marshalling.  There is no CPU fallback: construction raises when the HIP library
or a GPU is missing.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EmxError, MoveDesc

__all__ = ["DeviceEnsemble", "EmxError"]

_STATUS_NAN_LOGP = 1
_STATUS_BAD_COORD = 2
_STATUS_EXCHANGE_OVERFLOW = 4
_STATUS_EXCHANGE_TIMEOUT = 8
_STATUS_PLAN_PRODUCER = 16


def _as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError("expected array of shape %s, got %s" % (shape, a.shape))
    return a


class DeviceEnsemble:
    def __init__(self, nwalkers, ndim, device=0):
        self.lib = _lib.load()
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        ctx = C.c_void_p()
        rc = self.lib.emx_create(int(device), self.nwalkers, self.ndim, C.byref(ctx))
        if rc != 0:
            raise EmxError((self.lib.emx_last_error(None) or b"emx_create failed").decode())
        self.ctx = ctx
        self._target_kind = _lib.TARGET_HOST
        self._moves = None
        # resident states (state.ResidentState): the object a run handed out that still IS the device state, and the snapshot
        # slots of older ones
        self._gen = 0
        self._world = 1
        self._resident = None            # weakref to the ResidentState mirroring generation _gen
        self._free_slots = list(range(8))
        import weakref
        self._snap_states = weakref.WeakSet()   # ResidentStates whose values sit in a snapshot slot

    def close(self):
        if getattr(self, "ctx", None):
            ref = getattr(self, "_resident", None)
            st = ref() if ref is not None else None
            if st is not None:                       # a State handed out earlier still lives on this context: bring it home first
                try:
                    st._materialise(live=True)
                except Exception:  # noqa: BLE001
                    pass
                self._resident = None
            for st in list(getattr(self, "_snap_states", ())):      # ... and the ones kept in snapshot slots
                try:
                    st._materialise()
                except Exception:  # noqa: BLE001
                    pass
            self.lib.emx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        exc, self._cb_exc = getattr(self, "_cb_exc", None), None
        if exc is not None:               # raised inside a device log-prob callback: the caller's own exception, not our code
            raise exc
        _lib.check(self.lib, self.ctx, rc)

    # ---- state ----
    def _touch(self):
        """Called before anything changes (coords, log_prob) on the device: a live, never-read ResidentState of the current
        generation keeps its values in a device-side snapshot (or on the host when the slots are used up)."""
        ref = self._resident
        st = ref() if ref is not None else None
        if st is not None:
            # detach FIRST: while the generation still matches, the state can take a device snapshot -- and if that fails (no
            # memory for one, a pending callback exception) it still can be read live over PCIe, because nothing has changed yet
            try:
                st._detach_before_change()
            except Exception:  # noqa: BLE001
                st._materialise(live=True)
        self._resident = None
        self._gen += 1

    def snapshot_save(self):
        """-> slot holding a device-side copy of the current (coords, log_prob), or None when all slots are taken"""
        if not self._free_slots:
            return None
        slot = self._free_slots.pop()
        self._ck(self.lib.emx_snapshot_save(self.ctx, slot))
        return slot

    def snapshot_read(self, slot):
        x, lp = np.empty((self.nwalkers, self.ndim)), np.empty(self.nwalkers)
        self._ck(self.lib.emx_snapshot_read(self.ctx, slot, x.ctypes.data, lp.ctypes.data))
        return x, lp

    def snapshot_restore(self, slot):
        self._touch()
        self._ck(self.lib.emx_snapshot_restore(self.ctx, slot))

    def snapshot_release(self, slot):
        """the slot may be reused (its buffer stays allocated for the next snapshot)"""
        if slot is not None and slot not in self._free_slots:
            self._free_slots.append(slot)

    def set_state(self, coords, log_prob=None):
        self._touch()
        coords = _as_f64(coords, (self.nwalkers, self.ndim))
        lp = None if log_prob is None else _as_f64(log_prob, (self.nwalkers,))
        self._ck(self.lib.emx_set_state(self.ctx, coords.ctypes.data, None if lp is None else lp.ctypes.data))

    def get_state(self, coords=True, log_prob=True):
        x = np.empty((self.nwalkers, self.ndim)) if coords else None
        lp = np.empty(self.nwalkers) if log_prob else None
        self._ck(self.lib.emx_get_state(self.ctx, None if x is None else x.ctypes.data,
                                        None if lp is None else lp.ctypes.data))
        return x, lp

    def accepted_mask(self):
        m = np.empty(self.nwalkers, dtype=np.uint8)
        self._ck(self.lib.emx_get_accepted(self.ctx, m))
        return m.astype(bool)

    def status(self):
        bits = C.c_uint32(0)
        self._ck(self.lib.emx_status(self.ctx, C.byref(bits)))
        return bits.value

    def raise_on_status(self):
        """Mirror the reference's ValueErrors (ensemble.py:476-479, 550-551)."""
        bits = self.status()
        if bits & _STATUS_PLAN_PRODUCER:
            raise EmxError("exact-mode plans (rng='mt19937'): the device producer stalled or its stream ran out under the tokenizer; "
                           "the steps taken from it are void and the generator stands where it stood before the run -- "
                           "EMX_TUNE=mt_device=0 selects the host pipeline")
        if bits & _STATUS_EXCHANGE_OVERFLOW:
            raise EmxError("pull exchange: record capacity exceeded; the sharded run is invalid")
        if bits & _STATUS_EXCHANGE_TIMEOUT:
            # attributed by the CURRENT configuration (one replica that qualifies for the persistent kernel), not by whether a
            # persistent launch ever ran in this context's lifetime -- a later sharded run's barrier is the direct exchange's
            if self._world == 1 and self.persist_info()["qualifies"]:
                raise EmxError("persistent kernel: a device-wide barrier in the middle of a launch was not met in time (a launch that "
                               "cannot become co-resident is redone on the per-half-step path instead); the run is invalid -- "
                               "EMX_TUNE=persist=0 selects the per-half-step launches, persist_timeout_ms raises the bound")
            raise EmxError("direct exchange: a peer did not reach the device-side barrier in time; the sharded run is invalid")
        if bits & _STATUS_BAD_COORD:
            raise ValueError("At least one parameter value was infinite or NaN")
        if bits & _STATUS_NAN_LOGP:
            raise ValueError("Probability function returned NaN")

    def sync(self):
        self._ck(self.lib.emx_sync(self.ctx))

    def set_stream(self, stream_ptr):
        self._ck(self.lib.emx_set_stream(self.ctx, stream_ptr))

    def set_tuning(self, key, value):
        self._ck(self.lib.emx_set_tuning(self.ctx, key.encode(), int(value)))

    # ---- target ----
    def set_target(self, kind, p0=None, p1=None, scale=0.0):
        a0 = None if p0 is None else _as_f64(p0)
        a1 = None if p1 is None else _as_f64(p1)
        self._ck(self.lib.emx_set_target(self.ctx, int(kind), None if a0 is None else a0.ctypes.data,
                                         None if a1 is None else a1.ctypes.data, float(scale)))
        self._target_kind = int(kind)

    def set_target_callback(self, fn, graph=False):
        """``fn(q) -> log_prob`` on device memory (include/emx.h, emx_set_target_callback): ``q`` is a float64 CUDA tensor
        ``(n, ndim)`` viewing the library's proposal block, the result a length-n float64 CUDA tensor (anything
        ``torch.as_tensor`` takes from the device).  Called once per half-step on the host thread that drives the run; its
        work is enqueued on the context's stream, nothing is synchronised and nothing crosses PCIe."""
        import torch
        from .parallel import _DevView
        streams = {}
        graphs = self._cb_graphs = {}   # graph=True: (block address, rows, result address) -> [eager calls so far, graph or False]

        def eager(q_ptr, n, ndim, lp_ptr, dev):
            q = torch.as_tensor(_DevView(q_ptr, n * ndim), device=dev).view(n, ndim)
            out = torch.as_tensor(_DevView(lp_ptr, n), device=dev)
            res = fn(q)
            res = torch.as_tensor(res, dtype=torch.float64, device=dev).reshape(-1)
            if res.numel() != n:
                raise ValueError("the device log_prob_fn returned %d values for %d walkers" % (res.numel(), n))
            out.copy_(res)

        def tramp(user, q_ptr, n, ndim, lp_ptr, stream):
            try:
                s = streams.get(stream)
                if s is None:
                    s = streams[stream] = torch.cuda.ExternalStream(stream or 0) if stream else torch.cuda.default_stream()
                dev = torch.device("cuda", torch.cuda.current_device())
                with torch.cuda.stream(s):
                    if not graph:
                        eager(q_ptr, n, ndim, lp_ptr, dev)
                        return 0
                    # The library hands every split the same buffers, so the kernels fn launches for a given (block, rows) are
                    # the same every time: after two eager calls they are captured once and replayed -- one graph launch per
                    # split instead of one Python dispatch per torch operation.
                    key = (q_ptr, n, lp_ptr)
                    ent = graphs.setdefault(key, [0, None])
                    if ent[1]:
                        ent[1].replay()
                        return 0
                    eager(q_ptr, n, ndim, lp_ptr, dev)
                    ent[0] += 1
                    if ent[1] is None and ent[0] == 2:
                        try:
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g):
                                eager(q_ptr, n, ndim, lp_ptr, dev)
                            ent[1] = g
                        except Exception:  # noqa: BLE001  (capture unsupported for what fn does: stay eager)
                            ent[1] = False
                return 0
            except BaseException as e:  # noqa: BLE001  (handed to the caller by _ck)
                self._cb_exc = e
                return -1

        self._cb_keep = _lib.DEVICE_LOG_PROB_FN(tramp)          # the library holds the pointer: keep the object alive
        self._ck(self.lib.emx_set_target_callback(self.ctx, self._cb_keep, None))
        self._target_kind = _lib.TARGET_CALLBACK

    def set_target_callback_c(self, fn_ptr, user_ptr=None):
        """A native ``emx_device_log_prob_fn`` (include/emx.h) -- e.g. a function of the user's own shared library that launches
        a HIP kernel -- as the target: ``fn_ptr`` a ctypes function pointer or address, ``user_ptr`` its opaque argument."""
        fn = fn_ptr if isinstance(fn_ptr, _lib.DEVICE_LOG_PROB_FN) else C.cast(fn_ptr, _lib.DEVICE_LOG_PROB_FN)
        self._cb_keep = fn
        self._ck(self.lib.emx_set_target_callback(self.ctx, fn, C.c_void_p(user_ptr) if not isinstance(user_ptr, C.c_void_p) else user_ptr))
        self._target_kind = _lib.TARGET_CALLBACK

    def walkers_independent(self):
        """The reference's initial-state check (ensemble.py:653-663) on the ensemble this context holds: nothing crosses PCIe."""
        verdict = C.c_int32(0)
        self._ck(self.lib.emx_walkers_independent_resident(self.ctx, C.byref(verdict), None))
        return bool(verdict.value)

    def eval_state_log_prob(self):
        self._touch()
        self._ck(self.lib.emx_eval_state_log_prob(self.ctx))

    def eval_log_prob(self, coords):
        coords = _as_f64(coords)
        if coords.ndim != 2 or coords.shape[1] != self.ndim:
            raise ValueError("coords must be (n, ndim)")
        out = np.empty(coords.shape[0])
        for lo in range(0, coords.shape[0], self.nwalkers):
            blk = np.ascontiguousarray(coords[lo:lo + self.nwalkers])
            o = np.empty(blk.shape[0])
            self._ck(self.lib.emx_eval_log_prob(self.ctx, blk, blk.shape[0], o))
            out[lo:lo + blk.shape[0]] = o
        return out

    # ---- moves / rng ----
    def set_moves(self, descs, cdf):
        arr = (MoveDesc * len(descs))(*descs)
        cdf = _as_f64(cdf)
        self._ck(self.lib.emx_set_moves(self.ctx, len(descs), arr, cdf))
        self._moves = list(descs)

    def set_move_scale(self, move_index, std):
        """Per-coordinate standard deviations of a Gaussian move (None: back to the isotropic sigma)."""
        if std is None:
            self._ck(self.lib.emx_set_move_scale(self.ctx, int(move_index), None, 0))
            return
        std = _as_f64(std)
        self._ck(self.lib.emx_set_move_scale(self.ctx, int(move_index), std.ctypes.data, len(std)))

    def get_move(self, move_index):
        d = MoveDesc()
        self._ck(self.lib.emx_get_move(self.ctx, int(move_index), C.byref(d)))
        return d

    def plan_set_noise(self, normals, factor=1.0):
        self._ck(self.lib.emx_plan_set_noise(self.ctx, _as_f64(normals), float(factor)))

    def set_rng_mode(self, mode):
        self._ck(self.lib.emx_set_rng_mode(self.ctx, int(mode)))

    def set_mt19937(self, state):
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self._ck(self.lib.emx_rng_set_mt19937(self.ctx, key, int(state[2]), int(state[3]), float(state[4])))

    def get_mt19937(self):
        key = np.empty(624, dtype=np.uint32)
        pos, hg, cached = C.c_int32(), C.c_int32(), C.c_double()
        self._ck(self.lib.emx_rng_get_mt19937(self.ctx, key, C.byref(pos), C.byref(hg), C.byref(cached)))
        return ("MT19937", key, pos.value, hg.value, cached.value)

    def set_philox(self, seed, step=0):
        self._ck(self.lib.emx_rng_set_philox(self.ctx, int(seed) & (2**64 - 1), int(step)))

    def get_philox(self):
        s, t = C.c_uint64(), C.c_uint64()
        self._ck(self.lib.emx_rng_get_philox(self.ctx, C.byref(s), C.byref(t)))
        return s.value, t.value

    # ---- chain / run ----
    def chain_config(self, capacity):
        self._ck(self.lib.emx_chain_config(self.ctx, int(capacity)))

    def chain_reset(self):
        self._ck(self.lib.emx_chain_reset(self.ctx))

    def run(self, nsteps, thin_by=1, store=True):
        self._touch()
        self._ck(self.lib.emx_run(self.ctx, int(nsteps), int(thin_by), int(bool(store))))

    def iteration(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.emx_iteration(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def graph_state(self):
        d, cap = C.c_int32(), C.c_int32()
        self._ck(self.lib.emx_graph_state(self.ctx, C.byref(d), C.byref(cap)))
        return bool(d.value), cap.value

    def chain_read(self, what, start, stop, stride=1):
        nsel = len(range(start, stop, stride))
        shape = (nsel, self.nwalkers, self.ndim) if what == 0 else (nsel, self.nwalkers)
        out = np.empty(shape)
        if nsel:
            self._ck(self.lib.emx_chain_read(self.ctx, what, start, stop, stride, out))
        return out

    def accepted_counts(self):
        out = np.empty(self.nwalkers)
        self._ck(self.lib.emx_accepted_counts(self.ctx, out))
        return out

    # ---- around the hot loop ----
    def autocorr(self, discard=0, thin=1, c=5.0):
        """-> (tau per parameter in units of the selected samples, windows, number of samples): emx_autocorr on the
        device-resident chain (reference autocorr.integrated_time over get_chain(discard, thin))."""
        DeviceEnsemble._load_hipfft(self.lib)
        tau = np.empty(self.ndim)
        win = np.empty(self.ndim, dtype=np.int32)
        nt = C.c_int64()
        self._ck(self.lib.emx_autocorr(self.ctx, int(discard), int(thin), float(c), tau, win, C.byref(nt)))
        return tau, win, nt.value

    @staticmethod
    def _load_hipfft(lib):
        import os
        path = os.environ.get("EMX_HIPFFT_LIB")
        if not path:
            try:   # share PyTorch's hipFFT / rocFFT when torch is in the process
                import torch
                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libhipfft.so")
                path = cand if os.path.exists(cand) else None
            except Exception:  # noqa: BLE001
                path = None
        rc = lib.emx_fft_load(path.encode() if path else None)
        if rc != 0:
            raise EmxError((lib.emx_last_error(None) or b"cannot load libhipfft").decode())

    # ---- split-phase ----
    def step_begin(self, store=False):
        self._touch()
        mv, S = C.c_int32(), C.c_int32()
        self._ck(self.lib.emx_step_begin(self.ctx, int(bool(store)), C.byref(mv), C.byref(S)))
        return mv.value, S.value

    def step_begin_with(self, move_index, store=False):
        self._touch()
        S = C.c_int32()
        self._ck(self.lib.emx_step_begin_with(self.ctx, int(bool(store)), int(move_index), C.byref(S)))
        return S.value

    def _split_size(self, split):
        """number of walkers `split` of the step begun updates (the library copies exactly that many values)"""
        return self.shard_slots(split)[2]

    def accept_proposals(self, split, q, factors, new_log_prob):
        ns = self._split_size(split)
        self._ck(self.lib.emx_accept_proposals(self.ctx, int(split), _as_f64(q, (ns, self.ndim)), _as_f64(factors, (ns,)),
                                               _as_f64(new_log_prob, (ns,))))

    def halfstep(self, split):
        self._ck(self.lib.emx_halfstep(self.ctx, int(split)))

    def propose(self, split, with_factors=False):
        ns = C.c_int64()
        q = np.empty((self.nwalkers, self.ndim))
        f = np.empty(self.nwalkers) if with_factors else None
        self._ck(self.lib.emx_propose(self.ctx, int(split), q.ctypes.data, None if f is None else f.ctypes.data,
                                      C.byref(ns)))
        if with_factors:
            return q[: ns.value], f[: ns.value]
        return q[: ns.value]

    def accept(self, split, new_log_prob):
        # a log_prob_fn returning the wrong number of values is a ValueError here, not a host over-read in the library
        self._ck(self.lib.emx_accept(self.ctx, int(split), _as_f64(new_log_prob, (self._split_size(split),))))

    def step_end(self):
        self._ck(self.lib.emx_step_end(self.ctx))

    def plan_set(self, move_index, plan):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        self._ck(self.lib.emx_plan_set(self.ctx, int(move_index), i32(plan["off"]), i32(plan["order"]), i32(plan["p0"]),
                                       i32(plan["p1"]), i32(plan["p2"]), _as_f64(plan["s0"]), _as_f64(plan["uacc"])))

    def plan_get(self, nsplits):
        N = self.nwalkers
        off = np.zeros(nsplits + 1, dtype=np.int32)
        order, p0, p1, p2 = (np.empty(N, dtype=np.int32) for _ in range(4))
        s0, uacc = np.empty(N), np.empty(N)
        self._ck(self.lib.emx_plan_get(self.ctx, off, order, p0, p1, p2, s0, uacc))
        return dict(off=off, order=order, p0=p0, p1=p1, p2=p2, s0=s0, uacc=uacc)

    # ---- sharding ----
    def set_shard(self, rank, world):
        self._ck(self.lib.emx_set_shard(self.ctx, int(rank), int(world)))
        self._world = int(world)

    def set_shard_buffers(self, sendbuf_ptr, gathered_ptr, rows):
        self._ck(self.lib.emx_set_shard_buffers(self.ctx, sendbuf_ptr, gathered_ptr, int(rows)))

    def device_ptr(self, which):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self.lib.emx_device_ptr(self.ctx, int(which), C.byref(p), C.byref(n)))
        return p.value, n.value

    def shard_slots(self, split):
        lo, hi, ns = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.emx_shard_slots(self.ctx, int(split), C.byref(lo), C.byref(hi), C.byref(ns)))
        return lo.value, hi.value, ns.value

    def scatter_gathered(self, split):
        self._ck(self.lib.emx_scatter_gathered(self.ctx, int(split)))

    # ---- pull exchange (walker-block ownership; include/emx.h) ----
    def set_exchange(self, kind):
        """'allgather' (every updated row to every rank), 'pull' (only the partner rows read), 'direct' (partner rows read in
        place over xGMI), 'logprob' (evaluations shared out) or 'replay' (decisions travel, accepted updates recomputed)."""
        k = {"allgather": _lib.EXCHANGE_ALLGATHER, "pull": _lib.EXCHANGE_PULL, "direct": _lib.EXCHANGE_DIRECT,
             "logprob": _lib.EXCHANGE_LOGPROB, "replay": _lib.EXCHANGE_REPLAY}.get(kind, kind)
        self._ck(self.lib.emx_set_exchange(self.ctx, int(k)))

    def exchange_layout(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.emx_exchange_layout(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_exchange_buffers(self, send_ptr, send_doubles, recv_ptr, recv_doubles):
        self._ck(self.lib.emx_set_exchange_buffers(self.ctx, send_ptr, int(send_doubles), recv_ptr, int(recv_doubles)))

    def own_walkers(self):
        lo, hi = C.c_int64(), C.c_int64()
        self._ck(self.lib.emx_own_walkers(self.ctx, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def pull_prepare(self, split):
        n = C.c_int64()
        self._ck(self.lib.emx_pull_prepare(self.ctx, int(split), C.byref(n)))
        return n.value

    def pull_apply(self, split):
        self._ck(self.lib.emx_pull_apply(self.ctx, int(split)))

    def logprob_begin(self, split):
        """log-prob exchange: proposals of the whole split + the log-probs of this rank's share -> doubles per rank of the
        in-place all-gather that follows (buffer: device_ptr(3))"""
        n = C.c_int64(0)
        self._ck(self.lib.emx_logprob_begin(self.ctx, int(split), C.byref(n)))
        return int(n.value)

    def logprob_finish(self, split):
        self._ck(self.lib.emx_logprob_finish(self.ctx, int(split)))

    def replay_begin(self, split):
        """own slots of `split` (fused), decisions into the send buffer -> doubles per rank for the all-gather"""
        n = C.c_int64()
        self._ck(self.lib.emx_replay_begin(self.ctx, int(split), C.byref(n)))
        return n.value

    def replay_exchange(self, split):
        """device-side exchange of the decisions (peers mapped): stores into every peer's receive buffer + the barrier kernel"""
        self._ck(self.lib.emx_replay_exchange(self.ctx, int(split)))

    def replay_finish(self, split):
        """after the all-gather of the decisions: the other ranks' accepted updates, recomputed on this replica"""
        self._ck(self.lib.emx_replay_finish(self.ctx, int(split)))

    # ---- direct exchange (partner rows read in place from the peers' HBM; include/emx.h) ----
    def direct_export(self):
        """128 bytes (IPC handles of the coordinate array and the barrier flags) for the peers' direct_import."""
        h = np.zeros(128, dtype=np.uint8)
        self._ck(self.lib.emx_direct_export(self.ctx, h))
        return h

    def direct_import(self, handles):
        """handles: (world, 128) uint8, every rank's direct_export in rank order (multi-process runs)."""
        h = np.ascontiguousarray(handles, dtype=np.uint8).reshape(-1)
        self._ck(self.lib.emx_direct_import(self.ctx, h))

    def direct_attach(self, coords_ptrs, flags_ptrs=None):
        """Same-process peers: raw device pointers (device_ptr(0) / device_ptr(8) of every rank's context)."""
        n = len(coords_ptrs)
        a = (C.c_void_p * n)(*[C.c_void_p(p) for p in coords_ptrs])
        f = None if flags_ptrs is None else (C.c_void_p * n)(*[C.c_void_p(p) for p in flags_ptrs])
        self._ck(self.lib.emx_direct_attach(self.ctx, a, f))

    def direct_halfstep(self, split, barrier=False):
        self._ck(self.lib.emx_direct_halfstep(self.ctx, int(split), int(bool(barrier))))

    def replica_pack(self):
        n = C.c_int64()
        self._ck(self.lib.emx_replica_pack(self.ctx, C.byref(n)))
        return n.value

    def replica_unpack(self):
        self._touch()
        self._ck(self.lib.emx_replica_unpack(self.ctx))

    # ---- library-driven RCCL ----
    @staticmethod
    def rccl_unique_id():
        """128-byte ncclUniqueId (rank 0); broadcast it with any host-side channel."""
        lib = _lib.load()
        DeviceEnsemble._load_rccl(lib)
        uid = np.zeros(128, dtype=np.uint8)
        rc = lib.emx_comm_get_unique_id(uid)
        if rc != 0:
            raise EmxError((lib.emx_last_error(None) or b"ncclGetUniqueId failed").decode())
        return uid

    @staticmethod
    def _load_rccl(lib):
        import os
        path = os.environ.get("EMX_RCCL_LIB")
        if not path:
            try:   # share PyTorch's RCCL when torch is in the process
                import torch
                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                path = cand if os.path.exists(cand) else None
            except Exception:  # noqa: BLE001
                path = None
        rc = lib.emx_comm_load(path.encode() if path else None)
        if rc != 0:
            raise EmxError((lib.emx_last_error(None) or b"cannot load librccl").decode())

    def comm_init(self, rank, world, unique_id):
        self._load_rccl(self.lib)
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        self._ck(self.lib.emx_comm_init(self.ctx, int(rank), int(world), uid))

    def comm_destroy(self):
        self._ck(self.lib.emx_comm_destroy(self.ctx))

    def pipeline_stats(self):
        """exact mode: us per produced step of the plan pipeline's stages (include/emx.h emx_pipeline_stats), or None"""
        out = np.zeros(6)
        n, k = C.c_int64(0), C.c_int32(0)
        self._ck(self.lib.emx_pipeline_stats(self.ctx, out, C.byref(n), C.byref(k)))
        if n.value <= 0:
            return None
        return {"steps_produced": n.value, "finisher_threads": k.value, "wall_us": out[0], "generator_us": out[1], "tokenizer_us": out[2],
                "finishers_us_summed": out[3], "tokenizer_waited_for_words_us": out[4], "tokenizer_waited_for_consumer_us": out[5]}

    def pipeline_handovers(self):
        """stretch steps the exact-mode host pipeline handed over raw / as generator states (include/emx.h emx_pipeline_handovers)"""
        raw, regen = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.emx_pipeline_handovers(self.ctx, C.byref(raw), C.byref(regen)))
        return {"raw_steps": int(raw.value), "regen_steps": int(regen.value)}

    def persist_info(self):
        """persistent half-steps (include/emx.h emx_persist_info): does the configuration qualify, launches, half-steps they ran"""
        out = (C.c_int64 * 4)()
        self._ck(self.lib.emx_persist_info(self.ctx, out))
        loc = C.c_int64(0)
        self._ck(self.lib.emx_persist_local_launches(self.ctx, C.byref(loc)))
        return {"qualifies": bool(out[0]), "launches": int(out[1]), "halfsteps": int(out[2]), "recovered": int(out[3]),
                "local_launches": int(loc.value)}

    def mtdev_info(self):
        """exact-mode plans made on the device (include/emx.h emx_mtdev_info)"""
        out = (C.c_int64 * 8)()
        self._ck(self.lib.emx_mtdev_info(self.ctx, out))
        return {"qualifies": bool(out[0]), "alive": bool(out[1]), "steps": int(out[2]), "rounds": int(out[3]), "segments": int(out[4]),
                "batches": int(out[5]), "windows": int(out[6]), "poly_us": int(out[7])}

    def mtdev_tok_stats(self):
        """what the device tokenizer has done so far (include/emx.h emx_mtdev_tok_stats)"""
        out = (C.c_int64 * 8)()
        self._ck(self.lib.emx_mtdev_tok_stats(self.ctx, out))
        return {"windows": int(out[0]), "rounds": int(out[1]), "tail_groups": int(out[2]), "tail_rounds": int(out[3]),
                "wait_us": out[4] / 100.0, "chunk_us": out[5] / 100.0, "tail_us": out[6] / 100.0, "kernel_us": out[7] / 100.0}

    def mtdev_debug(self, what, arg, n=0):
        """tests: stream words / Fisher-Yates targets / positions of the live device producer (include/emx.h emx_mtdev_debug)"""
        if what == 0:
            out = np.empty(int(n), dtype=np.uint32)
        elif what == 1:
            out = np.empty(self.nwalkers, dtype=np.uint32)
        else:
            out = np.empty(int(n), dtype=np.uint64)
        self._ck(self.lib.emx_mtdev_debug(self.ctx, int(what), int(arg), out.ctypes.data, out.size))
        return out

    def comm_count(self):
        """ranks of the library's RCCL communicator (ncclCommCount); 0 without one"""
        n = C.c_int32(0)
        self._ck(self.lib.emx_comm_count(self.ctx, C.byref(n)))
        return n.value

    # ---- measurement ----
    def timer_start(self):
        self._ck(self.lib.emx_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(self.lib.emx_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    def profile_enable(self, max_launches):
        self._ck(self.lib.emx_profile_enable(self.ctx, int(max_launches)))

    def profile_read(self, max_launches):
        out = np.empty(max_launches, dtype=np.float32)
        n = C.c_int32(max_launches)
        self._ck(self.lib.emx_profile_read(self.ctx, out, C.byref(n)))
        return out[: n.value]
