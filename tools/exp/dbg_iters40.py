"""debug: which combination of (store, thin_by, steps per call, half-steps per launch) breaks the persistent path"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from test_gpu_persist import dense_spec, native_ens

spec = dense_spec(8192, 64, seed=4)
for maxhs in (40, 32, 36):
    for store, thin, nst, calls in ((True, 1, 63, 1), (True, 1, 45, 2), (True, 3, 7, 1), (True, 3, 19, 2), (False, 3, 19, 2), (True, 2, 21, 2), (True, 1, 19, 2)):
        out = []
        for persist in (1, 0):
            ens = native_ens(spec, persist)
            ens.set_tuning("persist_max_halfsteps", maxhs)
            if store:
                ens.chain_config(nst * calls)
            for _ in range(calls):
                ens.run(nst, thin, store)
            x, lp = ens.get_state()
            rec = [x, lp]
            if store:
                rec.append(ens.chain_read(0, 0, nst * calls))
            info = ens.persist_info()
            ens.close()
            out.append((rec, info))
        (a, ia), (b, ib) = out
        eq = [bool(np.array_equal(p, q)) for p, q in zip(a, b)]
        first_bad = None
        if store and not eq[2]:
            bad = np.where(np.any(a[2] != b[2], axis=(1, 2)))[0]
            first_bad = int(bad[0])
        print("max half-steps %d store %d thin %d steps/call %d calls %d: launches %d  x/lp/chain equal %s  first differing stored row %s" % (
            maxhs, store, thin, nst, calls, ia["launches"], eq, first_bad), flush=True)
