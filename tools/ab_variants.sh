#!/bin/bash
# Build experiment variants of libemx (compile-time switches in emx_kernels.hpp) next to the shipped one.
#   usage: tools/ab_variants.sh name "-DEMX_OPT_X=0 ..." [name flags ...]
cd "$(dirname "$0")/../emcee_amd/csrc" || exit 1
HOSTCXX=/opt/rocm/lib/llvm/bin/clang++
$HOSTCXX -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -pthread -c emx_mtpipe.cpp -o /tmp/emx_mtpipe_ab.o || exit 1
$HOSTCXX -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -pthread -c emx_mtjump.cpp -o /tmp/emx_mtjump_ab.o || exit 1
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-constant-logical-operand -fPIC -fvisibility=hidden $flags -mllvm -amdgpu-sched-strategy=max-ilp -c emx_hot.hip -o /tmp/emx_hot_ab_$name.o || exit 1
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-constant-logical-operand -fPIC -fvisibility=hidden -shared $flags /tmp/emx_mtpipe_ab.o /tmp/emx_mtjump_ab.o /tmp/emx_hot_ab_$name.o emx.hip emx_small.hip emx_aux.hip emx_wide.hip emx_mtdev.hip emx_slab.hip emx_pvalu.hip emx_pmix.hip emx_pslab.hip emx_podd.hip -o ../libemx_$name.so -ldl -pthread &
done
wait
ls -la ../libemx_*.so
