#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04g
O=$PWD/gpurun_out/r04g
for tune in "mt_tok_wshift=11,mt_tok_tail=4096,persist=0"; do
  echo "== $tune" | tee -a $O/sweep.log
  EMX_TUNE="$tune" timeout 300 python tools/mtdev_probe.py 65536 64 400 1 2>&1 | grep "mt_device\|profile" | cut -c1-600 | tee -a $O/sweep.log
done
