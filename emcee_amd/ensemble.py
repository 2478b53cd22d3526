"""EnsembleSampler: the drop-in driver of the GPU hot path (reference ``ensemble.py:32-713``).

Same constructor, ``sample()`` generator, ``run_mcmc()``, ``compute_log_prob()``, ``get_*``
accessors, errors and RNG-state semantics as reference emcee.  What runs where:

* ``log_prob_fn`` is a :class:`emcee_amd.targets.DeviceTarget` and every move is a built-in
  split-ensemble move  ->  **fused path**: one HIP launch per half-step does proposal, batched
  log-prob, Metropolis accept, commit and chain append; ``run_mcmc`` is a single C call.
* any other ``log_prob_fn`` with built-in moves  ->  **split-phase path**: proposal and
  accept/commit stay on the GPU, only the callable (optionally through ``pool.map``) runs on
  the host, exactly where reference emcee calls it (``moves/red_blue.py:93``).
* user-written moves  ->  their own ``propose(model, state)`` is called as in the reference.

``distributed=True`` (one process per GPU under ``torchrun``) shards the fused path over the ranks'
GPUs; ``torch.distributed`` only bootstraps RCCL and replicates rank 0's RNG state and inputs.

``rng="mt19937"`` (default) replays NumPy's legacy MT19937 stream: the same seed gives the same
chain as reference emcee.  ``rng="philox"`` generates all draws inside the kernels (counter
based), the throughput mode.  There is no CPU fallback: a missing GPU raises.
"""
import warnings
from itertools import count

import numpy as np

from . import _lib
from .backends import Backend
from .model import Model
from .moves import StretchMove
from .pbar import get_progress_bar
from .state import DeviceState, ResidentState, State
from .targets import DeviceTarget
from .utils import deprecation_warning

__all__ = ["EnsembleSampler", "walkers_independent"]

try:
    from collections.abc import Iterable
except ImportError:  # pragma: no cover
    from collections import Iterable

try:
    from numpy.exceptions import VisibleDeprecationWarning
except ImportError:  # pragma: no cover
    from numpy import VisibleDeprecationWarning


def _native_desc(move, ndim, can_fuse=True):
    """MoveDesc of a built-in move (ours, or a reference emcee instance of the same class name
    with an un-overridden get_proposal); None for anything else.  Whole-ensemble Metropolis moves
    (``_fused_only``) are only worth a device step when the log-prob is evaluated there too."""
    if getattr(move, "_fused_only", False) and not can_fuse:
        return None
    if hasattr(move, "_is_native"):
        return move._desc(ndim) if move._is_native() else None
    for klass in type(move).__mro__:
        if "get_proposal" in klass.__dict__:
            owner, mod = klass.__name__, klass.__module__
            break
    else:
        return None
    if not mod.startswith("emcee.moves"):
        return None
    try:
        ns, rs = int(move.nsplits), int(bool(move.randomize_split))
        if owner == "StretchMove":
            return _lib.MoveDesc(_lib.MOVE_STRETCH, ns, rs, 0, float(move.a), 0.0, 0.0, 0.0)
        if owner == "DEMove":
            g0 = move.gamma0 if move.gamma0 is not None else 2.38 / np.sqrt(2 * ndim)
            return _lib.MoveDesc(_lib.MOVE_DE, ns, rs, 0, 2.0, float(move.sigma), float(g0), 0.0)
        if owner == "DESnookerMove":
            return _lib.MoveDesc(_lib.MOVE_SNOOKER, ns, rs, 0, 2.0, 0.0, 0.0, float(move.gammas))
    except AttributeError:
        return None
    return None


class EnsembleSampler(object):
    """An ensemble MCMC sampler (see the module docstring; arguments as in reference
    ``ensemble.py:41-77``, plus ``rng`` and ``device``)."""

    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, moves=None, args=None, kwargs=None, backend=None,
                 vectorize=False, blobs_dtype=None, parameter_names=None,
                 # Deprecated...
                 a=None, postargs=None, threads=None, live_dangerously=None, runtime_sortingfn=None,
                 # emcee_amd extensions
                 rng="mt19937", device=None, distributed=False, exchange="allgather"):
        for value, text in ((a, "The 'a' argument is deprecated, use 'moves' instead"),
                            (threads, "The 'threads' argument is deprecated"),
                            (runtime_sortingfn, "The 'runtime_sortingfn' argument is deprecated"),
                            (live_dangerously, "The 'live_dangerously' argument is deprecated")):
            if value is not None:
                deprecation_warning(text)

        # move schedule (reference ensemble.py:115-129): a single move, a sequence of moves (equal weights), or a
        # sequence of (move, weight) pairs; the weights are normalised to probabilities
        self._moves, self._weights = _parse_move_schedule(moves)

        if rng not in ("mt19937", "philox"):
            raise ValueError("rng must be 'mt19937' or 'philox'")
        self.rng = rng
        # one process per GPU (torchrun): every rank builds the same sampler; the ensemble is sharded by walkers
        # and RCCL moves the updated rows between the half-step kernels
        self._dist = None
        if distributed:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("distributed=True needs an initialised torch.distributed process group "
                                   "(it is only used to bootstrap RCCL and to replicate the inputs)")
            if exchange not in ("allgather", "pull", "direct", "logprob", "replay", "replay_push"):
                raise ValueError("exchange must be 'allgather', 'pull', 'direct', 'logprob', 'replay' or 'replay_push'")
            self._dist = dist
            self._exchange = exchange
            self._comm_ready = False
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0")) if distributed else 0
        self.device = int(device)
        self.pool = pool
        self.vectorize = vectorize
        self.blobs_dtype = blobs_dtype
        self.ndim = ndim
        self.nwalkers = nwalkers
        self.backend = Backend() if backend is None else backend
        self._ens = None
        self._philox_step = 0

        if not self.backend.initialized:
            self._previous_state = None
            self.reset()
            state = np.random.get_state()
        else:
            if self.backend.shape != (self.nwalkers, self.ndim):
                raise ValueError(("the shape of the backend ({0}) is incompatible with the shape of the sampler ({1})"
                                  ).format(self.backend.shape, (self.nwalkers, self.ndim)))
            state = self.backend.random_state
            if state is None:
                state = np.random.get_state()
            it = self.backend.iteration
            if it > 0:
                self._previous_state = self.get_last_sample()
            else:
                self._previous_state = None

        # private generator, seeded from the global NumPy state (reference ensemble.py:164-167)
        self._random = np.random.mtrand.RandomState()
        self._random.set_state(state)
        self._rng_on_device = None      # the DeviceEnsemble whose MT19937 state is newer than self._random's
        if self._dist is not None:            # replicated decisions need ONE stream: rank 0's
            self._random.set_state(self._replicate(self._random.get_state()))

        self._device_target = log_prob_fn if isinstance(log_prob_fn, DeviceTarget) else None
        self.log_prob_fn = _FunctionWrapper(log_prob_fn, args, kwargs)

        self.params_are_named = parameter_names is not None
        if self.params_are_named:
            assert not self.vectorize, "named parameters with vectorization unsupported for now"
            self.parameter_names = _normalize_parameter_names(parameter_names, ndim)

    # ------------------------------------------------------------------ RNG / bookkeeping
    @property
    def random_state(self):
        """``get_state()`` of the sampler's private ``numpy.random.RandomState``."""
        self._flush_rng()
        return self._random.get_state()

    @random_state.setter  # NOQA
    def random_state(self, state):
        """Try to set the generator state; fails silently like the reference (ensemble.py:228-238): a state that
        NumPy refuses (None included) leaves the stream where it is -- which, after a device run, is wherever
        libemx left it, so that copy is brought home first."""
        self._flush_rng()
        try:
            self._random.set_state(state)
        except Exception:  # noqa: BLE001
            pass

    def _flush_rng(self):
        """Bring self._random up to date with the MT19937 state a device run left in libemx (copied lazily: a
        624-word get_state/set_state round trip per yielded step would cost more than the step)."""
        ens = self._rng_on_device
        if ens is not None:
            self._rng_on_device = None
            self._random.set_state(ens.get_mt19937())

    @property
    def iteration(self):
        return self.backend.iteration

    def reset(self):
        """Reset the bookkeeping parameters"""
        self.backend.reset(self.nwalkers, self.ndim)

    def __getstate__(self):
        self._flush_rng()
        d = dict(self.__dict__)
        d["pool"] = None
        d["_rng_on_device"] = None
        d["_ens"] = None            # device contexts are not picklable; re-created on demand
        d["_dist"] = None           # nor are process groups: an unpickled sampler is a single-GPU one
        return d

    # ------------------------------------------------------------------ multi-GPU plumbing
    def _replicate(self, obj):
        """rank 0's copy of a (picklable) host object on every rank"""
        box = [obj]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def _refuse_partial_chain(self, store):
        if self._dist is not None and self._exchange in ("pull", "direct") and store:
            raise RuntimeError("exchange='%s' keeps only each rank's block of walkers current between steps, so a stored "
                               "chain would be partial: run with store=False, or use exchange='allgather' / 'logprob'"
                               % self._exchange)

    def _join_communicator(self, ens):
        """RCCL communicator for this ensemble (once, after the moves are installed: the exchange buffers are sized
        for them); from here on emx_run exchanges the updated rows itself."""
        if self._dist is None or self._comm_ready:
            return
        from .device import DeviceEnsemble
        rank, world = self._dist.get_rank(), self._dist.get_world_size()
        if self._exchange == "replay_push":
            # the replay exchange with the decisions stored straight into the peers' buffers: no collective library involved,
            # the ranks only map each other's receive buffers and barrier flags (IPC handles over the process group)
            from .parallel import import_direct_peers
            ens.set_exchange("replay")
            ens.set_shard(rank, world)
            import_direct_peers(ens, self._dist)
            self._dist.barrier()
            self._comm_ready = True
            return
        uid = self._replicate(DeviceEnsemble.rccl_unique_id() if rank == 0 else None)
        ens.set_exchange(self._exchange)
        ens.comm_init(rank, world, uid)
        if self._exchange == "direct":           # map the peers' coordinate arrays and barrier flags (IPC handles over the group)
            from .parallel import import_direct_peers
            import_direct_peers(ens, self._dist)
        self._comm_ready = True

    # ------------------------------------------------------------------ device plumbing
    def _device_ensemble(self):
        if self._ens is None:
            from .device import DeviceEnsemble
            self._ens = DeviceEnsemble(self.nwalkers, self.ndim, device=self.device)
        return self._ens

    def _philox_seed(self):
        self._flush_rng()
        key = self._random.get_state()[1]
        return (int(key[0]) << 32 | int(key[1])) ^ (int(key[2]) << 16)

    def _configure_device(self, descs, fused):
        ens = self._device_ensemble()
        if fused:
            self._device_target.bind(ens)
        else:
            ens.set_target(_lib.TARGET_HOST)
        cdf = np.cumsum(self._weights)
        cdf /= cdf[-1]
        ens.set_moves(descs, cdf)
        for i, m in enumerate(self._moves):
            vec = getattr(m, "_scale_vector", None)
            if vec is not None and descs[i].kind == _lib.MOVE_GAUSS:
                ens.set_move_scale(i, vec())
        if self._dist is not None:
            if fused:
                self._join_communicator(ens)
            elif self._exchange != "logprob":
                raise RuntimeError("distributed=True with a Python log_prob_fn needs exchange='logprob' (every rank proposes and "
                                   "commits, the calls of log_prob_fn are shared out); the other exchanges shard the fused step")
        if self.rng == "mt19937":
            self._flush_rng()
            ens.set_rng_mode(_lib.RNG_MT19937)
            ens.set_mt19937(self._random.get_state())
        else:
            ens.set_rng_mode(_lib.RNG_PHILOX)
            ens.set_philox(self._philox_seed(), self._philox_step)
        return ens

    def _sync_rng_from_device(self, ens):
        for i, m in enumerate(self._moves):          # the sequential Gaussian mode's cursor lives in the move
            step = getattr(m, "get_proposal", None)
            if hasattr(m, "_scale_vector") and getattr(step, "mode", None) == "sequential" and m._is_native():
                step.index = int(ens.get_move(i).gammas)
        if self.rng == "mt19937":
            self._rng_on_device = ens
        else:
            self._philox_step = ens.get_philox()[1]

    def _raise_on_device_status(self, ens, store):
        """The reference's ValueErrors for a NaN log-prob / non-finite proposal (ensemble.py:476-479, 550-551).

        Divergence from the reference, by design: it stops AT the offending step; a native call covers a block of
        steps (the ``thin_by`` proposals between two yields, or a whole ``run_mcmc``), the offending proposals are
        rejected on the device, the block completes, and the error is raised when the block returns -- so
        ``backend.iteration`` may stand past the step that misbehaved.  The backend's ``random_state`` is pinned
        to the generator's real position before the exception leaves."""
        try:
            ens.raise_on_status()
        except Exception:
            if store:
                self.backend.random_state = self.random_state
            raise

    # ------------------------------------------------------------------ sampling
    def sample(self, initial_state, log_prob0=None, rstate0=None, blobs0=None, iterations=1, tune=False,
               skip_initial_state_check=False, thin_by=1, thin=None, store=True, progress=False,
               progress_kwargs=None):
        """Advance the chain as a generator; yields the :class:`State` every ``thin_by`` steps.

        Arguments and error behaviour as reference ``ensemble.py:258-424``."""
        if iterations is None and store:
            raise ValueError("'store' must be False when 'iterations' is None")
        state = State(initial_state, copy=True)
        if self._dist is not None:
            state = self._replicate(state)
        state_shape = np.shape(state.coords)
        if state_shape != (self.nwalkers, self.ndim):
            raise ValueError(f"incompatible input dimensions {state_shape}")
        _refuse_extended_precision(state.coords)
        if (not skip_initial_state_check) and (not walkers_independent(state.coords)):
            raise ValueError("Initial state has a large condition number. Make sure that your walkers are "
                             "linearly independent for the best performance")

        if rstate0 is not None:
            deprecation_warning("The 'rstate0' argument is deprecated, use a 'State' instead")
            state.random_state = rstate0
        self.random_state = state.random_state

        if log_prob0 is not None:
            deprecation_warning("The 'log_prob0' argument is deprecated, use a 'State' instead")
            state.log_prob = log_prob0
        if blobs0 is not None:
            deprecation_warning("The 'blobs0' argument is deprecated, use a 'State' instead")
            state.blobs = blobs0
        if state.log_prob is None:
            state.log_prob, state.blobs = self.compute_log_prob(state.coords)
        if np.shape(state.log_prob) != (self.nwalkers,):
            raise ValueError("incompatible input dimensions")
        if np.any(np.isnan(state.log_prob)):
            raise ValueError("The initial log_prob was NaN")

        # which execution path
        can_fuse = self._device_target is not None and state.blobs is None
        descs = [_native_desc(m, self.ndim, can_fuse) for m in self._moves]
        native = all(d is not None for d in descs)
        fused = native and can_fuse
        own_backend = isinstance(self.backend, Backend)
        for m in self._moves:
            live = getattr(m, "live_dangerously", False) or not hasattr(m, "nsplits")   # MH moves have no such guard
            if native and self.nwalkers < 2 * self.ndim and not live:       # reference red_blue.py:64-70
                raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice "
                                   "the number of dimensions.")

        yield_step, checkpoint_step, nsaves = _thinning_plan(iterations, thin_by, thin)

        if self._dist is not None and not fused and not (native and self._exchange == "logprob"):
            raise RuntimeError("distributed=True needs built-in moves, and a DeviceTarget log_prob_fn or exchange='logprob' "
                               "(a Python log_prob_fn: its calls are shared out over the ranks)")
        self._refuse_partial_chain(store)
        ens = None
        if native:
            ens = self._configure_device(descs, fused)
            ens.set_state(state.coords, np.asarray(state.log_prob, dtype=np.float64))
            if state.blobs is None:
                # from here on the state lives in HBM; the yielded object copies it back lazily
                state = DeviceState(ens, random_state=state.random_state)
            if own_backend and store:
                if self.backend._dev is not ens:
                    if self.backend.iteration > 0:
                        self.backend._detach()          # host samples from an earlier (custom-move) run
                    else:
                        self.backend._attach(ens)
        elif own_backend:
            self.backend._detach()
        dev_store = native and store and own_backend and self.backend._dev is ens
        if store:
            self.backend.grow(nsaves, state.blobs)

        map_fn = self.pool.map if self.pool is not None else map
        self._flush_rng()
        model = Model(self.log_prob_fn, self.compute_log_prob, map_fn, self._random)
        if progress_kwargs is None:
            progress_kwargs = {}

        total = None if iterations is None else iterations * yield_step
        with get_progress_bar(progress, total, **progress_kwargs) as pbar:
            i = 0
            # fused path: the yield_step proposals between two yields are ONE native call (emx_run(1, yield_step, store)
            # keeps the block's last step, which is the step the reference stores: ensemble.py:416)
            block_call = fused and (not store or dev_store) and checkpoint_step == yield_step
            lazy_rs = lambda: self.random_state  # noqa: E731
            for _ in count() if iterations is None else range(iterations):
                if block_call:
                    ens.run(1, yield_step, store)
                    self._sync_rng_from_device(ens)
                    self._raise_on_device_status(ens, store)
                    state._invalidate()
                    state.random_state = lazy_rs             # resolved when somebody reads it
                    if store:
                        self.backend._device_step_saved(None, lazy_rs)
                    pbar.update(yield_step)
                    i += yield_step
                    try:
                        yield state
                    except GeneratorExit:
                        if store:
                            self.backend.random_state = self.random_state    # pin what the provider stood for
                        raise
                    continue
                for _ in range(yield_step):
                    save = store and (i + 1) % checkpoint_step == 0
                    if native:
                        accepted = self._device_step(ens, state, fused, save and dev_store,
                                                     need_mask=save and not dev_store)
                        move = None
                        if isinstance(state, DeviceState):
                            state._invalidate()
                    else:
                        move = self._random.choice(self._moves, p=self._weights)     # ensemble.py:406
                        state, accepted = move.propose(model, state)
                    # device steps leave the MT19937 state in libemx; a DeviceState resolves it when it is read
                    state.random_state = lazy_rs if isinstance(state, DeviceState) else self.random_state
                    if tune and move is not None:
                        move.tune(state, accepted)
                    if save:
                        if dev_store:
                            self.backend._device_step_saved(state.blobs, lazy_rs if isinstance(state, DeviceState)
                                                            else state.random_state)
                        else:
                            if native and not isinstance(state, DeviceState):
                                state.coords, state.log_prob = ens.get_state()
                            self.backend.save_step(state, accepted)
                    pbar.update(1)
                    i += 1
                if native and not isinstance(state, DeviceState):
                    state.coords, state.log_prob = ens.get_state()
                try:
                    yield state
                except GeneratorExit:
                    if store and dev_store:
                        self.backend.random_state = self.random_state
                    raise
            if store and (block_call or dev_store):
                self.backend.random_state = self.random_state            # pin what the provider stood for

    def _device_step(self, ens, state, fused, store, need_mask=True):
        """One full step of the built-in moves on the device; returns the accepted mask (None when
        nobody on the host needs it: the device chain keeps its own accept counters)."""
        if fused:
            ens.run(1, 1, store)
            self._sync_rng_from_device(ens)
            self._raise_on_device_status(ens, store)
            return ens.accepted_mask() if need_mask else None
        # split-phase: the callable runs on the host between propose and accept (red_blue.py:90-104)
        k, nsplits = ens.step_begin(store)
        pending = []
        for split in range(nsplits):
            q = ens.propose(split)
            new_lp, new_blobs = self.compute_log_prob(q)
            ens.accept(split, np.asarray(new_lp, dtype=np.float64))
            if new_blobs is not None:
                pending.append((split, new_blobs))
        plan = ens.plan_get(nsplits) if pending else None
        ens.step_end()
        ens.raise_on_status()
        self._sync_rng_from_device(ens)
        # the mask crosses PCIe only when somebody on the host consumes it (host backend, blobs)
        accepted = ens.accepted_mask() if (need_mask or pending) else None
        for split, new_blobs in pending:
            if state.blobs is None:
                raise ValueError("If you start sampling with a given log_prob, you also need to provide the "
                                 "current list of blobs at that position.")
            members = plan["order"][plan["off"][split]:plan["off"][split + 1]]
            acc = accepted[members]
            state.blobs[members[acc]] = np.asarray(new_blobs)[acc]
        return accepted

    def run_mcmc(self, initial_state, nsteps, **kwargs):
        """Iterate :func:`sample` for ``nsteps`` iterations and return the result.

        ``initial_state=None`` resumes from the last state (reference ensemble.py:426-456).  With a
        device target, built-in moves and no progress bar the whole run is one native call."""
        if initial_state is None:
            if self._previous_state is None:
                raise ValueError("Cannot have `initial_state=None` if run_mcmc has never been called.")
            initial_state = self._previous_state

        fast = self._fast_run(initial_state, nsteps, **kwargs)
        if fast is not None:
            self._previous_state = fast
            return fast

        results = None
        for results in self.sample(initial_state, iterations=nsteps, **kwargs):
            pass
        if isinstance(results, DeviceState):
            # hand back a plain snapshot: later runs must not change what the caller holds
            results = State(results.coords, log_prob=results.log_prob, blobs=results.blobs,
                            random_state=results.random_state)
        self._previous_state = results
        return results

    def _fast_run(self, initial_state, nsteps, **kw):
        """``run_mcmc`` as ONE emx_run call when nothing on the host has to see the intermediate
        steps.  Returns None when the general generator path is required."""
        allowed = {"skip_initial_state_check", "thin_by", "store", "tune", "progress"}
        if set(kw) - allowed or kw.get("progress", False) or self._device_target is None:
            return None
        if not isinstance(self.backend, Backend) or nsteps is None or nsteps < 1:
            return None
        descs = [_native_desc(m, self.ndim, True) for m in self._moves]
        if any(d is None for d in descs):
            return None
        thin_by = int(kw.get("thin_by", 1))
        store = kw.get("store", True)
        if thin_by <= 0:
            raise ValueError("Invalid thinning argument")
        # the State a device run returned, handed back unread (or None -> _previous_state): it already is -- or, from its
        # snapshot, can again become -- the device state without crossing PCIe (reference semantics: ensemble.py:312,441-447)
        resident = None
        if isinstance(initial_state, ResidentState) and self._dist is None and self._ens is not None and \
                initial_state._ens is self._ens and initial_state._c is None and initial_state._lp is None:
            resident = initial_state
        if resident is not None:
            state = resident
            # (the reference's conditioning check of a continuation, ensemble.py:316-323: below, on the ensemble where it lives)
        else:
            state = State(initial_state, copy=True)
            if self._dist is not None:
                state = self._replicate(state)
            if state.blobs is not None:
                return None
            if np.shape(state.coords) != (self.nwalkers, self.ndim):
                raise ValueError(f"incompatible input dimensions {np.shape(state.coords)}")
            _refuse_extended_precision(state.coords)
            if (not kw.get("skip_initial_state_check", False)) and (not walkers_independent(state.coords)):
                raise ValueError("Initial state has a large condition number. Make sure that your walkers are "
                                 "linearly independent for the best performance")
        for m in self._moves:
            if self.nwalkers < 2 * self.ndim and hasattr(m, "nsplits") and not getattr(m, "live_dangerously", False):
                raise RuntimeError("It is unadvisable to use a red-blue move with fewer walkers than twice "
                                   "the number of dimensions.")
        self.random_state = state.random_state
        self._refuse_partial_chain(store)
        ens = self._configure_device(descs, True)
        check = not kw.get("skip_initial_state_check", False)
        ill = ("Initial state has a large condition number. Make sure that your walkers are "
               "linearly independent for the best performance")
        on_device = resident is not None and (resident._is_device_state(ens) or resident._restore_on_device(ens))
        # The reference re-checks a continuation too (ensemble.py:316-323): every path below checks first, then installs the state.
        if on_device:
            # on the device (emx_walkers_independent_resident: Householder QR where the ensemble is, ndim^2 numbers come back):
            # nothing crosses PCIe on a path whose point is that
            if check and not ens.walkers_independent():
                raise ValueError(ill)
            lp0 = None                # log-probs of a state a run produced: finite by construction (NaN proposals are rejected)
        else:
            if resident is not None and check and not walkers_independent(state.coords):
                raise ValueError(ill)
            if state.log_prob is None:
                ens.set_state(state.coords)
                ens.eval_state_log_prob()
                ens.raise_on_status()
                lp0 = ens.get_state(coords=False)[1]
            else:
                lp0 = np.asarray(state.log_prob, dtype=np.float64)
                if np.shape(lp0) != (self.nwalkers,):
                    raise ValueError("incompatible input dimensions")
                ens.set_state(state.coords, lp0)
        if lp0 is not None and np.any(np.isnan(lp0)):
            raise ValueError("The initial log_prob was NaN")
        if store:
            if self.backend._dev is not ens:
                if self.backend.iteration > 0:
                    return None
                self.backend._attach(ens)
            self.backend.grow(nsteps, None)
        ens.run(nsteps, thin_by, store and self.backend._dev is ens)
        self._sync_rng_from_device(ens)
        self._raise_on_device_status(ens, store)
        if self._dist is None:
            out = ResidentState(ens, random_state=self.random_state)      # the arrays cross PCIe when (if) they are read
        else:
            coords, lp = ens.get_state()
            out = State(coords, log_prob=lp, random_state=self.random_state)
        if store:
            self.backend.random_state = out.random_state
        return out

    # ------------------------------------------------------------------ batched log-prob
    def compute_log_prob(self, coords):
        """Calculate the vector of log-probability for the walkers -> ``(log_prob, blobs)``.

        Device targets are evaluated by the batched kernel; other callables as in reference
        ``ensemble.py:458-553`` (``vectorize``, ``pool.map``, blobs, NaN / inf guards)."""
        p = coords
        if not np.isfinite(p).all():               # one pass; the reference's two messages on the rare failure
            if np.any(np.isinf(p)):
                raise ValueError("At least one parameter value was infinite")
            raise ValueError("At least one parameter value was NaN")

        if self._device_target is not None:
            ens = self._device_ensemble()
            self._device_target.bind(ens) if ens._target_kind != self._device_target.kind else None
            log_prob = ens.eval_log_prob(np.atleast_2d(p))
            ens.status()
            if np.any(np.isnan(log_prob)):
                raise ValueError("Probability function returned NaN")
            return log_prob, None

        if self._dist is not None and self._exchange == "logprob":
            log_prob, blob = self._shared_log_prob(np.atleast_2d(p))
        else:
            log_prob, blob = self._call_log_prob_fn(p)
        if np.any(np.isnan(log_prob)):
            raise ValueError("Probability function returned NaN")
        return log_prob, blob

    def _call_log_prob_fn(self, p):
        if self.params_are_named:
            p = ndarray_to_list_of_dicts(p, self.parameter_names)
        if self.vectorize:
            results = self.log_prob_fn(p)
        else:
            mapper = map if self.pool is None else self.pool.map
            results = list(mapper(self.log_prob_fn, p))
        return _split_log_prob_and_blobs(results, self.blobs_dtype)

    def _shared_log_prob(self, p):
        """exchange='logprob' with a Python callable: the reference's ``pool.map`` (ensemble.py:486-496) over the ranks of
        the process group -- every rank holds the same proposals, calls ``log_prob_fn`` on its share of the rows and all
        ranks gather the results (rank order = row order)."""
        dist = self._dist
        rank, world = dist.get_rank(), dist.get_world_size()
        n = len(p)
        per = -(-n // world)
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        # an exception in one rank's share must reach EVERY rank (pool.map re-raises in the parent, ensemble.py:496): a rank that
        # raised before the collective would leave the others waiting in it for ever
        import hashlib
        digest = hashlib.sha1(np.ascontiguousarray(p, dtype=np.float64).tobytes()).hexdigest()[:16]
        local_exc = None
        try:
            mine = self._call_log_prob_fn(p[lo:hi]) if hi > lo else (np.empty(0), None)
            payload = (digest, None, np.asarray(mine[0], dtype=np.float64), mine[1])
        except Exception as e:  # noqa: BLE001  (re-raised on all ranks below; KeyboardInterrupt / SystemExit take their usual way)
            # only text travels: an exception object may not pickle (locals, device tensors, custom __init__ signatures), and a
            # rank that fails to pickle before the collective leaves the others waiting in it -- the hang this exists to prevent
            import traceback
            local_exc = e
            payload = (digest, (type(e).__name__, repr(e), traceback.format_exc()), None, None)
        parts = [None] * world
        dist.all_gather_object(parts, payload)
        for r, (_, err, _, _) in enumerate(parts):
            if err is not None:
                if r == rank and local_exc is not None:
                    raise local_exc
                raise RuntimeError("log_prob_fn failed on rank %d: %s: %s\n%s" % (r, err[0], err[1], err[2]))
        if len({d for d, _, _, _ in parts}) != 1:
            raise RuntimeError("exchange='logprob': the ranks hold different coordinates (every rank must pass the same initial "
                               "state and seed)")
        parts = [(a, b) for _, _, a, b in parts]
        log_prob = np.concatenate([np.atleast_1d(a) for a, _ in parts])
        blobs = [b for a, b in parts if len(np.atleast_1d(a))]
        if any(b is None for b in blobs):
            blob = None
        else:
            blob = np.concatenate([np.asarray(b) for b in blobs]) if blobs else None
        return log_prob, blob

    # ------------------------------------------------------------------ results
    @property
    def acceptance_fraction(self):
        """The fraction of proposed steps that were accepted"""
        return self.backend.accepted / float(self.backend.iteration)

    def get_chain(self, **kwargs):
        return self.get_value("chain", **kwargs)

    def get_blobs(self, **kwargs):
        return self.get_value("blobs", **kwargs)

    def get_log_prob(self, **kwargs):
        return self.get_value("log_prob", **kwargs)

    def get_last_sample(self, **kwargs):
        return self.backend.get_last_sample()

    def get_value(self, name, **kwargs):
        return self.backend.get_value(name, **kwargs)

    def get_autocorr_time(self, **kwargs):
        return self.backend.get_autocorr_time(**kwargs)

    get_chain.__doc__ = Backend.get_chain.__doc__
    get_blobs.__doc__ = Backend.get_blobs.__doc__
    get_log_prob.__doc__ = Backend.get_log_prob.__doc__
    get_last_sample.__doc__ = Backend.get_last_sample.__doc__
    get_autocorr_time.__doc__ = Backend.get_autocorr_time.__doc__

    # deprecated attribute spellings of emcee 2.x (reference ensemble.py:560-595)
    def __getattr__(self, name):
        legacy = {
            "chain": ("get_chain()", lambda s: np.swapaxes(s.get_chain(), 0, 1)),
            "flatchain": ("get_chain(flat=True)", lambda s: s.get_chain(flat=True)),
            "lnprobability": ("get_log_prob()", lambda s: np.swapaxes(s.get_log_prob(), 0, 1)),
            "flatlnprobability": ("get_log_prob(flat=True)", lambda s: s.get_log_prob(flat=True)),
            "blobs": ("get_blobs()", lambda s: s.get_blobs()),
            "flatblobs": ("get_blobs(flat=True)", lambda s: s.get_blobs(flat=True)),
            "acor": ("get_autocorr_time", lambda s: s.get_autocorr_time()),
        }
        if name in legacy:
            replacement, getter = legacy[name]
            deprecation_warning("{0} is deprecated, use {1} instead".format(name, replacement))
            return getter(self)
        raise AttributeError("{0!r} object has no attribute {1!r}".format(type(self).__name__, name))


class _FunctionWrapper(object):
    """The user's callable with its extra positional / keyword arguments bound.  A plain class (not a closure or a
    functools.partial holding a lambda) so that ``pool.map`` can pickle it.  When the callable raises, the walker
    position and the bound arguments are reported before the exception continues (reference ``ensemble.py:632-652``
    prints the same facts; the wording is ours)."""

    def __init__(self, f, args, kwargs):
        self.f = f
        self.args = [] if args is None else args
        self.kwargs = {} if kwargs is None else kwargs

    def __call__(self, x):
        try:
            return self.f(x, *self.args, **self.kwargs)
        except BaseException:  # pragma: no cover
            import sys
            import traceback
            report = ["emcee: Exception while calling your likelihood function:",
                      "  params: %s" % (x,), "  args: %s" % (self.args,), "  kwargs: %s" % (self.kwargs,), "  exception:"]
            print("\n".join(report))
            traceback.print_exception(*sys.exc_info(), file=sys.stdout)
            raise


def _parse_move_schedule(moves):
    """-> (list of moves, weight vector summing to one) from the ``moves`` constructor argument."""
    if moves is None:
        seq, w = [StretchMove()], [1.0]
    elif not isinstance(moves, Iterable):
        seq, w = [moves], [1.0]
    else:
        items = list(moves)
        pairs = [it for it in items if isinstance(it, (tuple, list)) and len(it) == 2]
        if len(pairs) == len(items) and items:
            seq, w = [m for m, _ in items], [wt for _, wt in items]
        else:
            seq, w = items, [1.0] * len(items)
    w = np.asarray(w, dtype=float).reshape(-1)
    return seq, w / w.sum()


def _refuse_extended_precision(coords):
    """The ensemble lives in HBM as float64.  The reference carries ``np.longdouble`` coordinates through its host
    arrays (tests/integration/test_longdouble.py); here they would be truncated silently, so they are refused."""
    dt = np.asarray(coords).dtype
    if dt.kind == "f" and dt.itemsize > 8:
        raise TypeError("emcee_amd keeps the ensemble in float64 on the GPU; %s coordinates would lose precision "
                        "(cast them explicitly if that is acceptable)" % dt)


def _thinning_plan(iterations, thin_by, thin):
    """-> (steps per yield, steps per stored sample, number of samples to allocate).

    ``thin_by=k`` makes k proposals per yielded / stored sample; the deprecated ``thin=k`` yields
    every step but stores only every k-th (reference ``ensemble.py:360-386``)."""
    if thin is None:
        k = int(thin_by)
        if k <= 0:
            raise ValueError("Invalid thinning argument")
        return k, k, iterations
    deprecation_warning("The 'thin' argument is deprecated. Use 'thin_by' instead.")
    k = int(thin)
    if k <= 0:
        raise ValueError("Invalid thinning argument")
    return 1, k, (None if iterations is None else iterations // k)


def _normalize_parameter_names(parameter_names, ndim):
    """Validate ``parameter_names`` (list of names, or dict name -> index / list of indices) and
    return the dict form.  Same rules as the reference (``ensemble.py:173-214``): no duplicate
    names, a list must name every dimension, and the indices must cover 0..ndim-1 exactly."""
    assert isinstance(parameter_names, (list, dict))
    names = list(parameter_names)
    dupes = {n for n in names if names.count(n) > 1}
    assert not dupes, f"duplicate parameters: {dupes}"
    if isinstance(parameter_names, list):
        assert len(names) == ndim, "name all parameters or set `parameter_names` to `None`"
        parameter_names = dict(zip(names, range(ndim)))
    assert len(parameter_names) <= ndim, "too many names"
    covered = set()
    for idx in parameter_names.values():
        covered.update(idx if isinstance(idx, list) else [idx])
    assert covered == set(range(ndim)), f"not all values appear -- set should be 0 to {ndim-1}"
    return parameter_names


def _blob_dtype(first_blob):
    """dtype for the blob array when the user gave none: the first blob's own dtype, falling back
    to ``object`` for ragged or string blobs (reference ``ensemble.py:514-539``)."""
    try:
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("error", VisibleDeprecationWarning)
            try:
                dt = np.atleast_1d(first_blob).dtype
            except Warning:
                deprecation_warning("You have provided blobs that are not all the same shape or size. This means "
                                    "they must be placed in an object array. Numpy has deprecated this automatic "
                                    "detection, so please specify blobs_dtype=np.dtype('object')")
                return np.dtype("object")
    except ValueError:
        return np.dtype("object")
    return np.dtype("object") if dt.kind in "US" else dt


def _split_log_prob_and_blobs(results, blobs_dtype):
    """``log_prob_fn`` may return a scalar or ``(log_prob, blob, ...)`` per walker; separate the two
    (reference ``ensemble.py:498-547``).  Returns ``(log_prob array, blob array or None)``."""
    if isinstance(results, np.ndarray) and results.ndim == 1 and results.dtype.kind == "f":
        return results.astype(np.float64, copy=False), None      # a vectorised callable's plain log-prob vector
    try:
        blobs = [r[1:] for r in results if len(r) > 1]
        has_blobs = len(blobs) > 0
    except (IndexError, TypeError):          # scalars have no len()
        has_blobs = False
    if not has_blobs:
        return np.array([_scalar(r) for r in results]), None
    log_prob = np.array([_scalar(r[0]) for r in results])
    dt = blobs_dtype if blobs_dtype is not None else _blob_dtype(blobs[0])
    blob = np.array(blobs, dtype=dt)
    # (nwalkers, 1, ...) -> (nwalkers, ...): a single blob per walker is not wrapped
    unit_axes = tuple(ax for ax in range(1, blob.ndim) if blob.shape[ax] == 1)
    if unit_axes:
        blob = np.squeeze(blob, unit_axes)
    return log_prob, blob


_DEVICE_CHECK_MIN_SIZE = 1 << 21      # elements; below this the host SVD is quicker than the PCIe round trip


def walkers_independent(coords):
    """Initial-state conditioning check (reference ``ensemble.py:653-663``): the scaled, centred
    walker matrix must have condition number <= 1e8.  One-off check outside the step loop.

    Large ensembles (>= 2^21 coordinates) on a machine with a GPU are checked there
    (``emx_walkers_independent`` in libemx): the same centring / scaling passes, then a Householder QR
    whose (ndim, ndim) triangular factor has the singular values of the tall matrix -- backward stable
    like the SVD the reference takes, so the verdict is the same (a Gram matrix
    would square the condition number and could not resolve the 1e8 threshold in float64)."""
    if np.size(coords) >= _DEVICE_CHECK_MIN_SIZE and np.asarray(coords).dtype == np.float64:
        verdict = _walkers_independent_device(coords)
        if verdict is not None:
            return verdict
    if not np.all(np.isfinite(coords)):
        return False
    C = coords - np.mean(coords, axis=0)[None, :]
    colmax = np.amax(np.abs(C), axis=0)
    if np.any(colmax == 0):
        return False
    C /= colmax
    C /= np.sqrt(np.sum(C ** 2, axis=0))
    return np.linalg.cond(C.astype(float)) <= 1e8


def _walkers_independent_device(coords):
    """The check on the GPU (``emx_walkers_independent``: Householder QR there, extreme singular values of the small
    triangular factor on the host); None when no GPU is available (the caller then uses the host)."""
    import ctypes as C
    try:
        lib = _lib.load()
        if _lib.device_count() < 1:
            return None
        x = np.ascontiguousarray(coords, dtype=np.float64)
        verdict = C.c_int32(0)
        rc = lib.emx_walkers_independent(0, x, x.shape[0], x.shape[1], C.byref(verdict), None)
    except Exception:  # noqa: BLE001
        return None
    if rc != 0:
        return None
    return bool(verdict.value)


def ndarray_to_list_of_dicts(x, key_map):
    """(N, ndim) array -> list of {name: value(s)} dicts (reference ``ensemble.py:685-700``)."""
    return [{key: xi[val] for key, val in key_map.items()} for xi in x]


def _scalar(fx):
    """One walker's log-prob as a Python float: NumPy / Python scalars directly, size-one arrays through ``item()``;
    anything longer is the user returning a vector where a number is expected (reference ``ensemble.py:703-713``)."""
    if np.isscalar(fx):
        return float(fx)
    arr = np.asarray(fx)
    if arr.size != 1:
        raise ValueError("log_prob_fn should return scalar")
    try:
        return float(arr.reshape(()).item())
    except (TypeError, ValueError) as err:
        raise ValueError("log_prob_fn should return scalar") from err
