import sys, json
sys.path.insert(0, '.')
import bench
bench._claim_stdout()
wl = bench.Workload("c2", 65536)
for k in (100, 400):
    e = bench.exact_mode_entry(wl, k, 40, 0)
    print(json.dumps({q: e[q] for q in ("steps", "ms_per_step", "best_block_ms_per_step", "wu_per_s", "kernel_us")}), file=sys.stderr)
