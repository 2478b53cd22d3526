#!/bin/bash
# round 5, session c: slab kernel (no scratch, skewed start): parity + A/B; the device-side exchange tests between processes; the host
# pipeline's stage times on this host
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_wide_dense.py -q -x -p no:cacheprovider ) > $O/slab_tests.log 2>&1; echo "slab tests rc=$?" | tee -a $O/summary_c.txt
tail -n 6 $O/slab_tests.log
timeout 300 python tools/exp/slab_ab.py slab_skew 3 > $O/slab_skew_ab.txt 2>&1; echo "slab ab rc=$?" | tee -a $O/summary_c.txt
cat $O/slab_skew_ab.txt
( time timeout 900 python -m pytest tests/test_gpu_sharded.py -q -x -p no:cacheprovider -k "device_side" ) > $O/device_side_tests.log 2>&1; echo "device_side tests rc=$?" | tee -a $O/summary_c.txt
tail -n 6 $O/device_side_tests.log
( nproc; taskset -p $$; EMX_PIPE_STATS=1 timeout 120 python tools/mt_pipe_bench.py 65536 400 0 ) > $O/mt_pipe_host.txt 2>&1
tail -n 12 $O/mt_pipe_host.txt
