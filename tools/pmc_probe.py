"""One configuration, a known number of steps, nothing else: the process rocprofv3 --pmc wraps (tools/pmc_r05.sh).  Every counter
value of the process's half-step-type kernels divided by (walkers x steps) is the traffic per walker-update.
usage: pmc_probe.py <config> [steps]     configs: c2 c2_store c3 c4 c5 w128 w128_de valu2048 local2048 exact_c3"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from emcee_amd import _lib  # noqa: E402
from emcee_amd.device import DeviceEnsemble  # noqa: E402
from tools.benchkit.model import Workload  # noqa: E402

cfg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
store, rng, tuning = False, "philox", {}
if cfg in ("c2", "c2_store"):
    wl, store = Workload("c2", 65536), cfg == "c2_store"
elif cfg == "c3":
    wl = Workload("c3", 262144)
elif cfg == "c4":
    wl = Workload("c4", 65536)
elif cfg == "c5":
    wl = Workload("c5", 16384)
elif cfg in ("w128", "w128_de"):
    wl = Workload("w128", 65536)
    if cfg == "w128_de":
        wl.moves, wl.weights = [("de", _lib.MoveDesc(1, 2, 1, 0, 2.0, 1e-5, 2.38 / np.sqrt(2 * 128), 1.7))], [1.0]
elif cfg == "valu2048":
    wl = Workload("c3", 2048)                     # k_persist_valu (one-XCD form), Rosenbrock ndim 32
elif cfg == "local2048":
    wl = Workload("c2", 2048)                     # k_persist<..., LOCAL>, dense ndim 64
elif cfg == "exact_c3":
    wl, rng = Workload("c3", 262144), "mt19937"   # the device producer of exact-mode plans (k_mt_*, k_fin_*)
else:
    raise SystemExit("unknown config " + cfg)
ens = DeviceEnsemble(wl.N, wl.D, device=0)
wl.install(ens, rng)
for k, v in tuning.items():
    ens.set_tuning(k, v)
if store:
    ens.chain_config(steps)
ens.run(steps, 1, store)
ens.sync()
st = ens.status()
print(json.dumps({"config": cfg, "N": wl.N, "D": wl.D, "steps": steps, "store": store, "rng": rng, "status": st,
                  "algorithmic_bytes_per_walker_update": wl.bytes_per_update(store), "accept_frac": float(np.mean(ens.accepted_counts())) / steps
                  if store else None, "persist": ens.persist_info()}))
ens.close()
