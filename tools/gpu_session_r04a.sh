#!/bin/bash
# round 4, session a: the device-side exact-plan producer's first contact with hardware + the persistent kernel's handshake
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
O=gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_mtdev.py -q -p no:cacheprovider > $O/mtdev_tests.log 2>&1; echo "mtdev tests rc=$?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider > $O/persist_tests.log 2>&1; echo "persist tests rc=$?" | tee -a $O/summary.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "lean or split_phase" -p no:cacheprovider > $O/lean_tests.log 2>&1; echo "lean tests rc=$?" | tee -a $O/summary.txt
timeout 300 python tools/mtdev_probe.py 65536 64 400 > $O/probe_c2.log 2>&1; echo "probe rc=$?" | tee -a $O/summary.txt
tail -5 $O/mtdev_tests.log $O/persist_tests.log $O/lean_tests.log; cat $O/probe_c2.log | tail -8
