import sys, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench._claim_stdout()
wl = bench.Workload("c2", 65536)
for k in (100, 400):
    e = bench.exact_mode_entry(wl, k, 40, 0)
    print(json.dumps({q: e[q] for q in ("steps", "ms_per_step", "best_block_ms_per_step", "wu_per_s", "kernel_us", "pipeline_stage_us_per_step")}), file=sys.stderr)
