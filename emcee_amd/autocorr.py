"""Integrated autocorrelation time (reference ``autocorr.py:11-136``), host side.

Same estimator (FFT autocorrelation per walker, averaged over walkers per dimension, Sokal's
automatic window with step ``c``), but batched: one real FFT over all walkers of a dimension
instead of the reference's Python loop over walkers.  Not part of the step loop;
a rocFFT version is the section-8f "next" item."""
import logging

import numpy as np

__all__ = ["function_1d", "integrated_time", "AutocorrError"]

logger = logging.getLogger(__name__)


def next_pow_two(n):
    """Smallest power of two >= n."""
    i = 1
    while i < n:
        i = i << 1
    return i


def _acf_columns(x):
    """Normalised ACF of every column of x (n_t, m)."""
    n_t = x.shape[0]
    n = next_pow_two(n_t)
    f = np.fft.rfft(x - np.mean(x, axis=0), n=2 * n, axis=0)
    acf = np.fft.irfft(f * np.conjugate(f), n=2 * n, axis=0)[:n_t]
    acf /= acf[0]
    return acf


def function_1d(x):
    """Normalised autocorrelation function of a 1-D series."""
    x = np.atleast_1d(x)
    if len(x.shape) != 1:
        raise ValueError("invalid dimensions for 1D autocorrelation function")
    return _acf_columns(x[:, None].astype(float))[:, 0]


def auto_window(taus, c):
    m = np.arange(len(taus)) < c * taus
    if np.any(m):
        return np.argmin(m)
    return len(taus) - 1


def integrated_time(x, c=5, tol=50, quiet=False, has_walkers=True):
    """Estimate the integrated autocorrelation time of a (n_step, n_walker, n_param) series.

    Same arguments, return value and :class:`AutocorrError` behaviour as the reference
    (``autocorr.py:49-123``)."""
    x = np.atleast_1d(x)
    if len(x.shape) == 1:
        x = x[:, np.newaxis, np.newaxis]
    if len(x.shape) == 2:
        x = x[:, np.newaxis, :] if not has_walkers else x[:, :, np.newaxis]
    if len(x.shape) != 3:
        raise ValueError("invalid dimensions")
    n_t, n_w, n_d = x.shape
    tau_est = np.empty(n_d)
    windows = np.empty(n_d, dtype=int)
    for d in range(n_d):
        f = np.mean(_acf_columns(np.asarray(x[:, :, d], dtype=float)), axis=1)
        taus = 2.0 * np.cumsum(f) - 1.0
        windows[d] = auto_window(taus, c)
        tau_est[d] = taus[windows[d]]
    flag = tol * tau_est > n_t
    if np.any(flag):
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). "
               "Use this estimate with caution and run a longer chain!\n").format(tol, np.sum(flag))
        msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
        if not quiet:
            raise AutocorrError(tau_est, msg)
        logger.warning(msg)
    return tau_est


class AutocorrError(Exception):
    """Raised if the chain is too short to estimate an autocorrelation time; the current
    estimate is available as ``.tau``."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super(AutocorrError, self).__init__(*args, **kwargs)
