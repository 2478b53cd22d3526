#!/bin/bash
# round 5, first session: k_persist_p2p (no barrier between the half-steps) -- parity, stress, A/B against the barrier form, phase clock
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider -k "without_a_barrier or stress or gives_the_bits or headline or stores_the_chain or interleave or co_resident" ) > $O/p2p_tests.log 2>&1; echo "p2p tests rc=$?" | tee -a $O/summary_a.txt
tail -n 15 $O/p2p_tests.log
timeout 600 python tools/exp/p2p_ab.py 1600 4 > $O/p2p_ab.txt 2>&1; echo "ab rc=$?" | tee -a $O/summary_a.txt
cat $O/p2p_ab.txt
timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 > $O/persist_phase_c2_p2p.txt 2>&1; echo "phase rc=$?" | tee -a $O/summary_a.txt
cat $O/persist_phase_c2_p2p.txt
timeout 300 python tools/persist_phase_clock.py 65536 64 0 0 > $O/persist_phase_c2_barrier.txt 2>&1
cat $O/persist_phase_c2_barrier.txt
