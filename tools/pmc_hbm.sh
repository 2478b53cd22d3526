#!/bin/bash
# HBM traffic of the beyond-Infinity-Cache configurations: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md),
# medians per launch of the half-step kernel -> gpurun_out/<dir>/pmc_hbm.json (copied into profiles/pmc_traffic.json by hand)
set -u
export TMPDIR=/tmp
O=${1:-gpurun_out/r03b}
mkdir -p $O/pmc
for cfg in hbm_dense hbm_wide c2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc/${cfg}_$ctr
    timeout 300 rocprofv3 --pmc $ctr -d $O/pmc/${cfg}_$ctr -o p -f csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --config $cfg > $O/pmc/${cfg}_$ctr.log 2>&1
  done
done
python - "$O" <<'PY'
import csv, collections, glob, json, statistics, sys
O = sys.argv[1]
out = {}
for d in sorted(glob.glob(O + "/pmc/*_*_SIZE")):
    cfg, ctr = d.split("/")[-1].rsplit("_", 2)[0], "_".join(d.split("/")[-1].rsplit("_", 2)[1:])
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_halfstep" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                agg[(r["Kernel_Name"], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for (k, g), v in agg.items():
        out.setdefault(cfg, {}).setdefault("%s grid=%s" % (k, g), {})[ctr] = {"median_kb": statistics.median(v), "n": len(v)}
json.dump(out, open(O + "/pmc_hbm.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c and c["FETCH_SIZE"]["n"] > 8:
            print(cfg, k, "bytes/launch = %.4g (2 x FETCH + WRITE)" % ((2 * c["FETCH_SIZE"]["median_kb"] + c["WRITE_SIZE"]["median_kb"]) * 1024), c)
PY
find $O/pmc -name "*.csv" -size +1M -delete
