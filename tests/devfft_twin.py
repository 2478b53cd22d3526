"""TEST-ONLY twin of emx_autocorr (csrc/emx_aux.hip): the same estimator through torch.fft, kept as a cross-check of the
library's hipFFT path (tests/test_gpu_sampler_api.py).  Not part of the product package.

Device-side integrated autocorrelation time (SURVEY.md 8f "next" item 2).

The chain already lives in HBM (``(nsteps, nwalkers, ndim)``, written by the half-step kernel);
the reference estimator (``autocorr.py:20-123``: FFT autocorrelation of every walker's series,
averaged over walkers per dimension, Sokal window) is run there with batched rocFFT through
``torch.fft`` -- a plain library op -- on zero-copy views of the library's buffers, chunked over
walkers so that the padded spectra fit.  Only the (nsteps, ndim) mean ACF crosses PCIe; the
window search is the reference's host code.
"""
import numpy as np


class _DevView(object):
    """Minimal __cuda_array_interface__ carrier for a library-owned device buffer."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def chain_tensor(ens, nstored):
    import torch
    ptr, nbytes = ens.device_ptr(4)
    if not ptr or nbytes < nstored * ens.nwalkers * ens.ndim * 8:
        raise RuntimeError("device chain not available")
    ens.sync()
    dev = torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DevView(ptr, (nstored, ens.nwalkers, ens.ndim)), device=dev)


def mean_acf(ens, nstored, discard=0, thin=1, max_bytes=8 << 30):
    """Walker-averaged normalised ACF per dimension, (n_t, ndim) float64 on the host."""
    import torch
    x = chain_tensor(ens, nstored)[discard + thin - 1::thin]        # Backend.get_value slice
    n_t, n_w, n_d = x.shape
    n = 1
    while n < n_t:
        n <<= 1
    per_walker = 2 * n * n_d * 8 * 4                                 # padded input + spectrum + output, roughly
    chunk = max(1, min(n_w, int(max_bytes // per_walker)))
    acc = torch.zeros((n_t, n_d), dtype=torch.float64, device=x.device)
    for lo in range(0, n_w, chunk):
        blk = x[:, lo:lo + chunk, :]
        blk = blk - blk.mean(dim=0, keepdim=True)
        f = torch.fft.rfft(blk, n=2 * n, dim=0)
        acf = torch.fft.irfft(f * f.conj(), n=2 * n, dim=0)[:n_t]
        acf = acf / acf[0:1]
        acc += acf.sum(dim=1)
        del f, acf, blk
    return (acc / n_w).cpu().numpy()


def integrated_time_device(ens, nstored, discard=0, thin=1, c=5, tol=50, quiet=False):
    """Same return value / errors as ``autocorr.integrated_time`` for the device-resident chain."""
    from emcee_amd import autocorr
    f = mean_acf(ens, nstored, discard=discard, thin=thin)
    n_t, n_d = f.shape
    tau_est = np.empty(n_d)
    for d in range(n_d):
        _, tau_est[d] = autocorr.tau_from_mean_acf(f[:, d], c)
    flag = tol * tau_est > n_t
    if np.any(flag):
        msg = ("The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). "
               "Use this estimate with caution and run a longer chain!\n").format(tol, np.sum(flag))
        msg += "N/{0} = {1:.0f};\ntau: {2}".format(tol, n_t / tol, tau_est)
        if not quiet:
            raise autocorr.AutocorrError(tau_est, msg)
        autocorr.logger.warning(msg)
    return tau_est
