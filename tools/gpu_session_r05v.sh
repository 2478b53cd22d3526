#!/bin/bash
# round 5, session v: exact mode after the tokenizer's DE / snooker scans: the exact-mode GPU tests, the mixture probe, C4 exact
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05v
O=$PWD/gpurun_out/r05v
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_parity.py tests/test_gpu_small_run.py tests/test_gpu_sampler_api.py tests/test_gpu_mtdev.py -q -p no:cacheprovider -x ) > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/tests.log
timeout 600 python tools/exp/exact_mix_probe.py > $O/exact_mix_probe.txt 2>&1; echo "probe rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/exact_mix_probe.txt
