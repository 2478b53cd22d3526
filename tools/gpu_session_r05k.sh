#!/bin/bash
# round 5, session k: exact-mode mixtures on the persistent kernels; the resident conditioning check; the hardware-assumption test; rates
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_walkers_independent.py tests/test_gpu_hw_assumptions.py -q -x -p no:cacheprovider -k "exact_mode or resident or hw or xcd or assume" ) > $O/tests_k.log 2>&1; echo "tests k rc=$?" | tee -a $O/summary_k.txt
tail -n 12 $O/tests_k.log
timeout 300 python tools/exp/exact_mix_probe.py > $O/exact_mix_probe.txt 2>&1; echo "mix probe rc=$?" | tee -a $O/summary_k.txt
cat $O/exact_mix_probe.txt | tail -n 8
