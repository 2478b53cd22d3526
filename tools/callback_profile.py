"""Kernel-level view of the device-callback path at C2's shape (run under rocprofv3 --kernel-trace --stats):
  python tools/callback_profile.py user|torch|fused [steps]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import emcee_amd  # noqa: E402
from emcee_amd import _lib, targets  # noqa: E402
from bench import dense_gaussian  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "user"
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N, D = 65536, 64
mu, cov, icov = dense_gaussian(D)
p0 = mu + np.random.RandomState(1).randn(N, D) @ np.linalg.cholesky(cov).T
if which == "user":
    so = "/tmp/libuser_logprob.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "tests/c/user_logprob.hip", "-o", so],
                   check=True, capture_output=True)
    _lib.load()
    user = C.CDLL(so)
    user.user_setup.restype = C.c_void_p
    user.user_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    h = user.user_setup(np.ascontiguousarray(mu).ctypes.data, np.ascontiguousarray(icov).ctypes.data, D)
    target = targets.DeviceKernel(user.user_log_prob, h)
elif which == "torch":
    import torch
    mu_t, icov_t = torch.as_tensor(mu, device="cuda"), torch.as_tensor(icov, device="cuda")

    def lp(q):
        d = q - mu_t
        return -0.5 * ((d @ icov_t) * d).sum(1)
    target = targets.DeviceCallable(lp)
else:
    target = targets.DenseGaussian(mu, icov)
s = emcee_amd.EnsembleSampler(N, D, target, rng="philox")
s._random.seed(3)
st = s.run_mcmc(p0, 10, store=False, skip_initial_state_check=True)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    st = s.run_mcmc(st, 100, store=False, skip_initial_state_check=True)
t0 = time.perf_counter()
st = s.run_mcmc(st, nst, store=False, skip_initial_state_check=True)
s._ens.sync()
print("%s: %.1f us/step" % (which, (time.perf_counter() - t0) / nst * 1e6))
