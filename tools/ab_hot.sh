#!/bin/bash
# Fast variants of libemx that differ in the headline translation unit only (emx_hot.hip: k_halfstep<8,2,4,STRETCH,4,*>, k_persist,
# k_persist_p2p): the other units are compiled once per BASE (a name + flags: "plain" "" or "stamps" "-DEMX_OPT_STAMPS=1") into
# /tmp/emx_base_<base>/, every variant compiles emx_hot.hip with its flags and links.  ~1 min per batch instead of ~3.
#   usage: tools/ab_hot.sh <base> "<base flags>" name "flags" [name "flags" ...]
cd "$(dirname "$0")/../emcee_amd/csrc" || exit 1
BASE=$1; BFLAGS=$2; shift 2
B=/tmp/emx_base_$BASE
HOSTCXX=/opt/rocm/lib/llvm/bin/clang++
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-constant-logical-operand -fPIC -fvisibility=hidden"
stamp=$(cat emx.hip emx_small.hip emx_aux.hip emx_wide.hip emx_mtdev.hip emx_slab.hip emx_pvalu.hip emx_pmix.hip emx_mtpipe.cpp emx_mtjump.cpp *.hpp ../../include/emx.h | grep -v EMX_P2P | md5sum | cut -c1-12)
if [ ! -f $B/.stamp ] || [ "$(cat $B/.stamp)" != "$stamp$BFLAGS" ]; then
  rm -rf $B; mkdir -p $B
  $HOSTCXX -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -pthread -c emx_mtpipe.cpp -o $B/emx_mtpipe.o &
  $HOSTCXX -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -pthread -c emx_mtjump.cpp -o $B/emx_mtjump.o &
  for u in emx emx_small emx_aux emx_wide emx_mtdev emx_slab emx_pvalu emx_pmix; do
    hipcc $COMMON $BFLAGS -c $u.hip -o $B/$u.o &
  done
  wait
  echo "$stamp$BFLAGS" > $B/.stamp
fi
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc $COMMON $BFLAGS $flags -mllvm -amdgpu-sched-strategy=max-ilp -c emx_hot.hip -o $B/hot_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC $B/emx_mtpipe.o $B/emx_mtjump.o $B/emx.o $B/emx_small.o $B/emx_aux.o $B/emx_wide.o $B/emx_mtdev.o $B/emx_slab.o $B/emx_pvalu.o $B/emx_pmix.o $B/hot_$name.o -o ../libemx_$name.so -ldl -pthread ) &
done
wait
ls -la ../libemx_*.so
