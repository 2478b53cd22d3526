#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04n
O=$PWD/gpurun_out/r04n
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o mtdev -- python $R/tools/mtdev_probe.py 262144 32 100 1 > $O/prof_c3.log 2>&1; echo "prof rc=$?"
cd $R
grep "mt_device" $O/prof_c3.log | cut -c1-700
python - <<PY
import sqlite3, glob
for f in glob.glob("$O/prof_c3/*.db"):
    c = sqlite3.connect(f)
    for r in c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc limit 16"):
        print("%-60s n=%6d total=%10.1f us avg=%9.2f" % (r[0][:60], *r[1:]))
    rows=list(c.execute("select name,start,end,stream_id,queue_id from kernels order by start"))
    toks=[r for r in rows if 'k_mt_tok' in r[0]]
    a=toks[15][1]-100000; b=toks[16][2]+100000
    for r in rows:
        if r[1]>=a and r[1]<=b and 'halfstep' not in r[0]:
            n=r[0].split('(')[0].split('::')[-1][:14]
            print("%-14s start %9.1f us  dur %8.1f us  end %9.1f q %s"%(n,(r[1]-a)/1e3,(r[2]-r[1])/1e3,(r[2]-a)/1e3,r[4]))
PY
