import sys, time
import numpy as np
sys.path.insert(0, ".")
from emcee_amd import _lib
from tools.quick_bench import run
for target in ("dense", "iso", "dense"):
    for steps in (50, 200):
        r = run(65536, 64, target, rng=_lib.RNG_MT19937, steps=steps)
        print(target, steps, "gpu-timer %.1f us/step  wall %.1f us/step" % (r["ms_per_step"] * 1e3, r["wall_ms_per_step"] * 1e3), flush=True)
