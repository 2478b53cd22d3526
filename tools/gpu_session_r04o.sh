#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04o
O=$PWD/gpurun_out/r04o
timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -x -p no:cacheprovider > $O/mtdev_tests.log 2>&1; echo "mtdev tests rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/mtdev_tests.log
for cfg in "65536 64 400" "262144 32 200"; do
 for tune in "mt_tok_wshift=11,mt_tok_tail=2048" "mt_tok_wshift=10,mt_tok_tail=2048" "mt_tok_wshift=12,mt_tok_tail=2048"; do
  echo "== $cfg $tune" | tee -a $O/sweep.log
  EMX_TUNE="$tune" timeout 300 python tools/mtdev_probe.py $cfg 1 2>&1 | grep "mt_device" | cut -c1-560 | tee -a $O/sweep.log
 done
done
