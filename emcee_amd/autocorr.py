"""Integrated autocorrelation time, host side (API of the reference's ``autocorr`` module).

Estimator (reference ``autocorr.py:49-123``, Sokal's recipe): normalised autocorrelation
function of every walker's series by FFT, averaged over walkers for each parameter, cumulative
sum ``tau(M) = 2 sum_{t<=M} rho(t) - 1`` and the smallest window ``M >= c tau(M)``.  Here the FFTs
of all walkers of one parameter are taken in a single batched real transform instead of one
Python call per walker.  Chains that live in HBM are analysed there by ``emx_autocorr`` (csrc/emx_aux.hip, behind the C
ABI), which ``Backend.get_autocorr_time`` prefers; this module is its fallback and the estimator for host-side chains."""
import logging

import numpy as np

__all__ = ["function_1d", "integrated_time", "AutocorrError"]

logger = logging.getLogger(__name__)


class AutocorrError(Exception):
    """The chain is too short for a reliable estimate; ``.tau`` holds the current one."""

    def __init__(self, tau, *args, **kwargs):
        super(AutocorrError, self).__init__(*args, **kwargs)
        self.tau = tau


def next_pow_two(n):
    """Smallest power of two that is >= n."""
    p = 1
    while p < n:
        p *= 2
    return p


def _batched_acf(series):
    """Normalised ACF along axis 0 of a (n_t, m) array, all m columns in one real FFT pair."""
    n_t = series.shape[0]
    size = 2 * next_pow_two(n_t)
    centred = series - series.mean(axis=0)
    spectrum = np.fft.rfft(centred, n=size, axis=0)
    power = spectrum.real ** 2 + spectrum.imag ** 2
    acf = np.fft.irfft(power, n=size, axis=0)[:n_t]
    return acf / acf[0]


def function_1d(x):
    """Normalised autocorrelation function of one 1-d time series."""
    x = np.atleast_1d(x)
    if x.ndim != 1:
        raise ValueError("invalid dimensions for 1D autocorrelation function")
    return _batched_acf(np.asarray(x, dtype=float).reshape(-1, 1)).ravel()


def auto_window(taus, c):
    """Sokal's automatic window: the first lag M with M >= c * tau(M).

    Degenerate inputs follow the reference (``autocorr.py:42-46``) exactly: if no lag is below
    ``c * tau`` at all the last lag is used, and if every lag is below it the window is 0."""
    below = np.arange(len(taus)) < c * taus
    if not below.any():
        return len(taus) - 1
    reached = np.flatnonzero(~below)
    return int(reached[0]) if len(reached) else 0


def _as_steps_walkers_params(x, has_walkers):
    x = np.atleast_1d(x)
    if x.ndim == 1:
        return x[:, None, None]
    if x.ndim == 2:
        return x[:, :, None] if has_walkers else x[:, None, :]
    if x.ndim == 3:
        return x
    raise ValueError("invalid dimensions")


def tau_from_mean_acf(rho, c):
    """(window, tau) from a walker-averaged ACF of one parameter."""
    taus = 2.0 * np.cumsum(rho) - 1.0
    w = auto_window(taus, c)
    return w, taus[w]


def integrated_time(x, c=5, tol=50, quiet=False, has_walkers=True):
    """Integrated autocorrelation time per parameter of ``x``.

    ``x`` is ``(n_step,)``, ``(n_step, n_walker)`` (or ``(n_step, n_param)`` with
    ``has_walkers=False``) or ``(n_step, n_walker, n_param)``.  ``c`` is the window step, ``tol`` the
    number of autocorrelation times the chain must span; shorter chains raise
    :class:`AutocorrError` (or only warn when ``quiet``)."""
    chain = _as_steps_walkers_params(x, has_walkers)
    n_t, _, n_d = chain.shape
    tau_est = np.empty(n_d)
    for d in range(n_d):
        rho = _batched_acf(np.asarray(chain[:, :, d], dtype=float)).mean(axis=1)
        _, tau_est[d] = tau_from_mean_acf(rho, c)
    too_short = tol * tau_est > n_t
    if too_short.any():
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). "
               "Use this estimate with caution and run a longer chain!\n").format(tol, int(too_short.sum()))
        msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
        logger.warning(msg)
    return tau_est
