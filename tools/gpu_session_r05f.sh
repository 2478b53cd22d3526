#!/bin/bash
# round 5, session f: the wide dense path's fused proposal (tuning wide_fuse) in the bench's own measurement (no per-launch events)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05f
O=$PWD/gpurun_out/r05f
export TMPDIR=/tmp
for r in 1 2; do
  for v in 0 1; do
    EMX_TUNE=wide_fuse=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --config w512 --no-cpu-baseline > $O/b_${v}_${r}.json 2> $O/b_${v}_${r}.err
    python - "$O/b_${v}_${r}.json" $v <<'PY' | tee -a $O/ab.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("configs", {}).get("wide_65536x512_dense") or {}
print("wide_fuse=%s: %.1f us/step, roofline %s" % (sys.argv[2], (c.get("ms_per_step") or 0) * 1e3, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (c.get("roofline") or {}).items() if k in ("frac", "frac_wall_clock", "avg_launch_us")}))
PY
  done
done
