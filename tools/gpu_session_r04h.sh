#!/bin/bash
# device exact-plan producer against the host pipeline over ensemble sizes (both on the per-half-step kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04h
O=$PWD/gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -x -p no:cacheprovider > $O/mtdev_tests.log 2>&1; echo "mtdev tests rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/mtdev_tests.log
for cfg in "32768 64 400" "65536 64 400" "131072 64 200" "262144 32 200" "524288 32 100" "1048576 64 60"; do
  echo "== $cfg" | tee -a $O/sizes.log
  EMX_TUNE="mt_tok_wshift=11,mt_tok_tail=2048" timeout 600 python tools/mtdev_probe.py $cfg 1,0 2>&1 | grep "mt_device" | cut -c1-200 | tee -a $O/sizes.log
done
