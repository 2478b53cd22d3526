#!/bin/bash
# round 4, session d: 1024-thread tokenizer with shrinking windows and a one-wave tail: tests, kernel trace, window-rule sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04d
R=$PWD
timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -x -p no:cacheprovider > $O/mtdev_tests.log 2>&1; echo "mtdev tests rc=$?" | tee -a $O/summary.txt
tail -n 15 $O/mtdev_tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_mtdev -o mtdev -- python $R/tools/mtdev_probe.py 65536 64 200 1 > $O/prof_mtdev.log 2>&1; echo "prof rc=$?" | tee -a $O/summary.txt
cd $R
grep "mt_device" $O/prof_mtdev.log | cut -c1-420
python - <<PY
import sqlite3, glob
for f in glob.glob("$O/prof_mtdev/*.db"):
    c = sqlite3.connect(f)
    for r in c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc limit 20"):
        print("%-60s n=%6d total=%10.1f us avg=%9.2f min=%9.2f max=%9.2f" % (r[0][:60], *r[1:]))
PY
for tune in "mt_tok_wshift=12,mt_tok_tail=2048" "mt_tok_wshift=12,mt_tok_tail=0" "mt_tok_wshift=12,mt_tok_tail=512" "mt_tok_wshift=12,mt_tok_tail=8192" "mt_tok_wshift=11,mt_tok_tail=2048" "mt_tok_wshift=13,mt_tok_tail=2048" "mt_tok_wshift=14,mt_tok_tail=4096" "mt_tok_wshift=12,mt_tok_tail=2048,persist=0"; do
  echo "== $tune" | tee -a $O/sweep.log
  EMX_TUNE="$tune" timeout 300 python tools/mtdev_probe.py 65536 64 400 1 2>&1 | grep "mt_device" | cut -c1-330 | tee -a $O/sweep.log
done
