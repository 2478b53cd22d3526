#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04p
O=$PWD/gpurun_out/r04p
for cfg in "262144 32 200"; do
  EMX_TUNE="mt_tok_wshift=11,mt_tok_tail=2048" timeout 300 python tools/mtdev_probe.py $cfg 1 2>&1 | grep "mt_device\|profile" | cut -c1-560 | tee -a $O/sweep.log
done
