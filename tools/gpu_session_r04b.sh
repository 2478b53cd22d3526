#!/bin/bash
# round 4, session b: the device producer's tests after the debug-window fix + where its time goes (kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -p no:cacheprovider > $O/mtdev_tests.log 2>&1; echo "mtdev tests rc=$?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_persist.py -q -p no:cacheprovider > $O/persist_tests.log 2>&1; echo "persist tests rc=$?" | tee -a $O/summary.txt
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_mtdev -o mtdev -- python $R/tools/mtdev_probe.py 65536 64 200 1 > $O/prof_mtdev.log 2>&1; echo "prof rc=$?" | tee -a $O/summary.txt
cd $R
tail -n 5 $O/mtdev_tests.log $O/persist_tests.log
tail -n 3 $O/prof_mtdev.log
find $O/prof_mtdev -name "*kernel_stats*" | head -1 | xargs -r head -n 20
find $O/prof_mtdev -name "*kernel_trace*" -size +1M -delete
