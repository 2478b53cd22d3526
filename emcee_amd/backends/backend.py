"""Backend: chain storage (reference ``backends/backend.py:11-237``), device-resident by default.

When a sampler drives the GPU path it attaches its :class:`DeviceEnsemble`: the chain
``(nsteps, nwalkers, ndim)``, the log-prob chain and the per-walker accept counters then live in
HBM (the half-step kernel appends to them directly) and ``get_value`` copies only the requested
``discard/thin`` slice to the host.  Without a device attached (user-written moves running
through ``Move.propose``) it is a plain in-memory store with the reference's ``save_step``."""
import numpy as np

from .. import autocorr
from .._lib import EmxError
from ..state import State

__all__ = ["Backend"]


class Backend(object):
    """A backend that keeps the chain in memory (HBM when attached to a device ensemble)."""

    def __init__(self, dtype=None):
        self.initialized = False
        self.dtype = np.float64 if dtype is None else dtype
        self._dev = None

    # ---- lifecycle ----
    def reset(self, nwalkers, ndim):
        """Forget every stored sample and size the backend for ``(nwalkers, ndim)``."""
        self.nwalkers = int(nwalkers)
        self.ndim = int(ndim)
        self._iteration = 0
        self._accepted = np.zeros(self.nwalkers, dtype=self.dtype)
        self._chain = np.empty((0, self.nwalkers, self.ndim), dtype=self.dtype)
        self._log_prob = np.empty((0, self.nwalkers), dtype=self.dtype)
        self.blobs = None
        self.random_state = None
        self.initialized = True
        if self._dev is not None:
            self._dev.chain_reset()

    def _attach(self, ens):
        """Move storage to the device ensemble ``ens`` (idempotent)."""
        if self._dev is ens:
            return
        if self._dev is not None:
            self._detach()
        ens.chain_reset()
        if self._iteration > 0:
            raise RuntimeError("cannot attach a device to a backend that already holds host samples")
        self._dev = ens

    def _detach(self):
        """Pull everything to host arrays and drop the device."""
        if self._dev is None:
            return
        it = self.iteration
        self._chain = self._dev.chain_read(0, 0, it)
        self._log_prob = self._dev.chain_read(1, 0, it)
        self._accepted = self._dev.accepted_counts().astype(self.dtype)
        self._iteration = it
        self._dev = None

    # ---- reference attributes ----
    @property
    def iteration(self):
        if self._dev is not None:
            return self._dev.iteration()[0]
        return self._iteration

    @iteration.setter
    def iteration(self, v):
        self._iteration = v

    @property
    def accepted(self):
        if self._dev is not None:
            return self._dev.accepted_counts().astype(self.dtype)
        return self._accepted

    @accepted.setter
    def accepted(self, v):
        self._accepted = v

    @property
    def chain(self):
        if self._dev is not None:
            return self._dev.chain_read(0, 0, self.iteration)
        return self._chain

    @chain.setter
    def chain(self, v):
        self._chain = v

    @property
    def log_prob(self):
        if self._dev is not None:
            return self._dev.chain_read(1, 0, self.iteration)
        return self._log_prob

    @log_prob.setter
    def log_prob(self, v):
        self._log_prob = v

    def has_blobs(self):
        """Whether blob storage has been set up (the log-prob function returns metadata)."""
        return self.blobs is not None

    def get_value(self, name, flat=False, thin=1, discard=0):
        it = self.iteration
        if it <= 0:
            raise AttributeError("you must run the sampler with 'store == True' before accessing the results")
        if name == "blobs" and not self.has_blobs():
            return None
        start = discard + thin - 1                     # reference backend.py:53
        if self._dev is not None and name in ("chain", "log_prob"):
            v = self._dev.chain_read(0 if name == "chain" else 1, min(start, it), it, thin)
        else:
            v = getattr(self, name)[start:it:thin]
        if flat:
            s = list(v.shape[1:])
            s[0] = int(np.prod(v.shape[:2]))
            return v.reshape(s)
        return v

    def get_chain(self, **kwargs):
        """Stored samples, ``(nsteps, nwalkers, ndim)``.

        Keyword arguments (all optional): ``flat`` merges the step and walker axes, ``thin`` keeps
        every thin-th stored step, ``discard`` drops that many initial steps (burn-in)."""
        return self.get_value("chain", **kwargs)

    def get_blobs(self, **kwargs):
        """Stored blobs, one per walker and step (``None`` when the model has none); same
        ``flat`` / ``thin`` / ``discard`` keywords as :meth:`get_chain`."""
        return self.get_value("blobs", **kwargs)

    def get_log_prob(self, **kwargs):
        """Stored log-probabilities, ``(nsteps, nwalkers)``; same keywords as :meth:`get_chain`."""
        return self.get_value("log_prob", **kwargs)

    def get_last_sample(self):
        """The most recently stored step as a :class:`State` (with the RNG state saved alongside)."""
        if (not self.initialized) or self.iteration <= 0:
            raise AttributeError("you must run the sampler with 'store == True' before accessing the results")
        it = self.iteration
        blobs = self.get_blobs(discard=it - 1)
        if blobs is not None:
            blobs = blobs[0]
        return State(self.get_chain(discard=it - 1)[0], log_prob=self.get_log_prob(discard=it - 1)[0],
                     blobs=blobs, random_state=self.random_state)

    def get_autocorr_time(self, discard=0, thin=1, **kwargs):
        """Integrated autocorrelation time per parameter, in steps (reference backend.py:130-150).

        A device-resident chain is analysed where it lives (``emx_autocorr``: batched FFTs next to the chain, Sokal window,
        only ``ndim`` numbers come back); the reference's ``tol`` / ``quiet`` handling (autocorr.py:110-121) is applied here.
        Host-side chains (custom moves, blobs) go through :func:`emcee_amd.autocorr.integrated_time`."""
        only = {k: v for k, v in kwargs.items() if k in ("c", "tol", "quiet")}
        if self._dev is not None and kwargs.get("has_walkers", True) and len(only) == len(kwargs):
            c, tol, quiet = only.get("c", 5), only.get("tol", 50), only.get("quiet", False)
            try:
                tau_est, _, n_t = self._dev.autocorr(discard=discard, thin=thin, c=c)
            except EmxError as e:
                # libhipfft not loadable, no room for the work buffers next to a long chain, an empty selection ...: the host
                # estimator below computes the same numbers from a copy of the chain (and raises the reference's own errors)
                autocorr.logger.debug("device autocorrelation unavailable (%s): host estimator", e)
                tau_est = None
            if tau_est is not None:
                flag = tol * tau_est > n_t
                if np.any(flag):
                    msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). "
                           "Use this estimate with caution and run a longer chain!\n").format(tol, np.sum(flag))
                    msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
                    if not quiet:
                        raise autocorr.AutocorrError(tau_est, msg)
                    autocorr.logger.warning(msg)
                return thin * tau_est
        x = self.get_chain(discard=discard, thin=thin)
        return thin * autocorr.integrated_time(x, **kwargs)

    @property
    def shape(self):
        """``(nwalkers, ndim)`` of the ensemble this backend was reset for."""
        return self.nwalkers, self.ndim

    # ---- growth / saving ----
    def _check_blobs(self, blobs):
        has_blobs = self.has_blobs()
        if has_blobs and blobs is None:
            raise ValueError("inconsistent use of blobs")
        if self.iteration > 0 and blobs is not None and not has_blobs:
            raise ValueError("inconsistent use of blobs")

    def grow(self, ngrow, blobs):
        """Make room for ``ngrow`` more steps (reference backend.py:164-185); ``blobs`` (the current
        blob array or None) fixes the blob dtype on first use."""
        self._check_blobs(blobs)
        it = self.iteration
        if self._dev is not None:
            self._dev.chain_config(it + ngrow)
            have = 0 if self.blobs is None else len(self.blobs)
        else:
            i = ngrow - (len(self._chain) - it)
            a = np.empty((i, self.nwalkers, self.ndim), dtype=self.dtype)
            self._chain = np.concatenate((self._chain, a), axis=0)
            a = np.empty((i, self.nwalkers), dtype=self.dtype)
            self._log_prob = np.concatenate((self._log_prob, a), axis=0)
            have = len(self._chain) - i
        if blobs is not None:
            i = it + ngrow - (0 if self.blobs is None else len(self.blobs))
            dt = np.dtype((blobs.dtype, blobs.shape[1:]))
            a = np.empty((max(i, 0), self.nwalkers), dtype=dt)
            self.blobs = a if self.blobs is None else np.concatenate((self.blobs, a), axis=0)
        del have

    def _check(self, state, accepted):
        """Shape / blob consistency of a step about to be saved (reference backend.py:187-212)."""
        self._check_blobs(state.blobs)
        nwalkers, ndim = self.shape
        expect = (
            (state.coords.shape == (nwalkers, ndim), "invalid coordinate dimensions; expected {0}".format((nwalkers, ndim))),
            (state.log_prob.shape == (nwalkers,), "invalid log probability size; expected {0}".format(nwalkers)),
            (state.blobs is None or self.has_blobs(), "unexpected blobs"),
            (state.blobs is not None or not self.has_blobs(), "expected blobs, but none were given"),
            (state.blobs is None or len(state.blobs) == nwalkers, "invalid blobs size; expected {0}".format(nwalkers)),
            (accepted.shape == (nwalkers,), "invalid acceptance size; expected {0}".format(nwalkers)),
        )
        for ok, message in expect:
            if not ok:
                raise ValueError(message)

    def save_step(self, state, accepted):
        """Append ``state`` and add ``accepted`` to the per-walker counters -- the host-side path
        (user-written moves, foreign samplers); reference backend.py:214-231."""
        if self._dev is not None:
            self._detach()
        self._check(state, accepted)
        self._chain[self._iteration, :, :] = state.coords
        self._log_prob[self._iteration, :] = state.log_prob
        if state.blobs is not None:
            self.blobs[self._iteration, :] = state.blobs
        self._accepted = self._accepted + accepted
        self.random_state = state.random_state
        self._iteration += 1

    @property
    def random_state(self):
        """RNG state after the last stored step.  During a device run the sampler hands over a provider (the MT19937
        state then lives in libemx and is only copied out when somebody asks)."""
        v = self._rstate
        return v() if callable(v) else v

    @random_state.setter
    def random_state(self, value):
        self._rstate = value

    def _device_step_saved(self, state_blobs, random_state):
        """Book-keeping after the kernel appended a step to the device chain."""
        if state_blobs is not None:
            self.blobs[self.iteration - 1, :] = state_blobs
        self.random_state = random_state

    def __getstate__(self):
        """Pickling materialises the device chain on the host (contexts are process-local)."""
        d = dict(self.__dict__)
        d["_rstate"] = self.random_state          # resolve a lazy provider
        if self._dev is not None:
            it = self.iteration
            d["_chain"] = self._dev.chain_read(0, 0, it)
            d["_log_prob"] = self._dev.chain_read(1, 0, it)
            d["_accepted"] = self._dev.accepted_counts().astype(self.dtype)
            d["_iteration"] = it
            d["_dev"] = None
        return d

    def __enter__(self):
        return self

    def __exit__(self, exception_type, exception_value, traceback):
        pass
