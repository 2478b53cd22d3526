#!/bin/bash
# round 5, session g: device finish: kernel + copy statistics of the exact-mode run at C2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
rm -rf $O/prof_exact_g
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_exact_g -o x -f csv -- python $GRAFT_REPO_ROOT/tools/exact_mode_probe.py > $O/prof_exact_g.log 2>&1; cd $GRAFT_REPO_ROOT
ls $O/prof_exact_g
for f in $O/prof_exact_g/*stats*.csv; do echo "== $f"; head -n 8 "$f" | cut -c1-200; done
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05/prof_exact_g"
for f in glob.glob(O + "/*memory_copy_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    import collections
    by = collections.defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        by[(r.get("Direction"), )].append(d)
    for k, v in by.items():
        v.sort()
        print(k, "n", len(v), "median us %.1f" % v[len(v)//2], "p90 %.1f" % v[int(len(v)*0.9)], "max %.1f" % v[-1], "sum ms %.1f" % (sum(v)/1e3))
PY
grep ms_per_step $O/prof_exact_g.log | cut -c1-120
