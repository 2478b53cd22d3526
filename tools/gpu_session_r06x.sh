#!/bin/bash
# round 6, session x: exact mode at C2 with the generator on a core of its own (default when the threads outnumber the cores) and without,
# five fresh processes each
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06x
O=$PWD/gpurun_out/r06x
lscpu | grep -i "model name\|^CPU(s)\|Thread\|Core\|Socket\|L3" | tee $O/host.txt
for rep in 1 2 3 4 5; do
  for g in own shared; do
    if [ $g = shared ]; then export EMX_PIPE_NO_GEN_CORE=1; else unset EMX_PIPE_NO_GEN_CORE; fi
    echo "generator core: $g" | tee -a $O/exact_c2_gen_core.txt
    timeout 300 python tools/exact_mode_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee -a $O/exact_c2_gen_core.txt
  done
done
