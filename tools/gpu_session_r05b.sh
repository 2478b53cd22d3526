#!/bin/bash
# round 5: k_persist_p2p variants (scalar polls; speculative partner rows or not) -- A/B + phase clocks + parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for v in "" p10 p00; do
  L=$PWD/emcee_amd/libemx${v:+_$v}.so
  EMX_LIB=$L timeout 300 python tools/exp/p2p_ab.py 800 3 1 2>&1 | grep -v amdgpu.ids | tee -a $O/p2p_ab_variants.txt
done
timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_11.txt
sed -i 's/libemx_stamps.so/libemx_p10s.so/' tools/persist_phase_clock.py
timeout 300 python tools/persist_phase_clock.py 65536 64 0 1 2>&1 | grep -v amdgpu.ids | tee $O/persist_phase_c2_p2p_10.txt
( time timeout 600 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider -k "without_a_barrier or persistent_kernel_coherence_stress or headline" ) > $O/p2p_tests_b.log 2>&1; echo "p2p tests (default lib) rc=$?" | tee -a $O/summary_b.txt
tail -n 4 $O/p2p_tests_b.log
( time EMX_LIB=$PWD/emcee_amd/libemx_p10.so timeout 600 python -m pytest tests/test_gpu_persist.py -q -x -p no:cacheprovider -k "without_a_barrier or persistent_kernel_coherence_stress or headline" ) > $O/p2p_tests_b10.log 2>&1; echo "p2p tests (p10) rc=$?" | tee -a $O/summary_b.txt
tail -n 4 $O/p2p_tests_b10.log
