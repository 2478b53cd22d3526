#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04e
O=$PWD/gpurun_out/r04e
for tune in "mt_tok_wshift=12,mt_tok_tail=2048" ; do
  echo "== $tune" | tee -a $O/sweep3.log
  EMX_MTDEV_TRACE=1 EMX_TUNE="$tune" timeout 300 python tools/mtdev_probe.py 65536 64 400 1 2>&1 | grep "mt_device\|mtdev tok" | cut -c1-600 | tee -a $O/sweep3.log
done
