"""Driver-clock step time of bench configurations, K-step blocks exactly as bench.py times them (median of >= 50 ms of blocks).
  usage: python tools/ab_cfg.py [K] [cfg ...]       cfg in c2 c3 c4 c5 hbm_dense hbm_wide w512 w128; env EMX_LIB picks the build,
  EMX_AB_TUNE='{"key": value, ...}' sets tuning keys"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (rocprofv3 runs it from /tmp)
import bench  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfgs = sys.argv[2:] or ["c2", "c3"]
sizes = {"c2": 65536, "c3": 262144, "c4": 65536, "c5": 16384, "hbm_dense": 1048576, "hbm_wide": 262144, "w512": 65536, "w128": 65536}
tune = json.loads(os.environ.get("EMX_AB_TUNE", "{}"))
tag = os.path.basename(os.environ.get("EMX_LIB", "cur")) + (" " + json.dumps(tune) if tune else "")
tag = tag + (" spin0" if os.environ.get("EMX_SPIN_SYNC") == "0" else "")
for key in cfgs:
    store = key.endswith("+store")              # e.g. c2+store: the chain appended every step
    key = key.replace("+store", "")
    wl = bench.Workload(key, sizes[key])
    r = bench.measure_single(wl, K, 5, want_kernel=True, store=store, tuning=tune)
    B = wl.bytes_per_update(store)
    key += "+store" if store else ""
    us = r["wall_s"] * 1e6 / K
    print("%-22s %-10s K=%d  %.3f us/step (best %.3f)  events %.3f us/step  kernel %.2f us  frac_wall %.4f  blocks %d" % (
        tag, key, K, us, r["wall_min_s"] * 1e6 / K, r["gpu_ms"] * 1e3 / K, r["per_launch_us"] or 0.0,
        wl.N * B / (us * 1e-6) / 1e9 / 8000.0, r["blocks"]), flush=True)
