#!/bin/bash
# round 5, session q2: vectorised rejection randint of the host pipeline (ensembles whose complement is no power of two)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_mtdev.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider ) > $O/tests_q.log 2>&1; echo "tests q rc=$?" | tee -a $O/summary_q.txt
tail -n 5 $O/tests_q.log
for n in 100000 196608; do
  timeout 200 python tools/mtdev_probe.py $n 32 100 0,1 2>&1 | grep mt_device | cut -c1-110 | tee -a $O/mtdev_sizes_r05b.txt
done
