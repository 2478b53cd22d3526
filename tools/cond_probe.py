"""Probe: tall-skinny R factor on the GPU (torch.linalg.qr, rocSOLVER) vs NumPy's SVD-based cond."""
import time
import numpy as np
import torch

for N, D in [(65536, 64), (262144, 32), (16384, 1024), (4096, 16)]:
    rs = np.random.RandomState(0)
    x = rs.randn(N, D)
    x[:, 1] = x[:, 0] * (1 - 1e-7) + 1e-7 * x[:, 1]
    t0 = time.perf_counter()
    c_host = np.linalg.cond(x)
    t_host = time.perf_counter() - t0
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xd = torch.as_tensor(x, device="cuda")
        r = torch.linalg.qr(xd, mode="r").R
        rh = r.cpu().numpy()
        c_dev = np.linalg.cond(rh)
        torch.cuda.synchronize()
        t_dev = time.perf_counter() - t0
    print(N, D, "host cond %.6e in %.2fs | device-R cond %.6e in %.4fs | rel diff %.2e" % (c_host, t_host, c_dev, t_dev, abs(c_dev - c_host) / c_host), flush=True)
