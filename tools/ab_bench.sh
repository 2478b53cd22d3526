#!/bin/bash
# A/B on ONE box: alternate library builds (emcee_amd/libemx_<name>.so; "cur" = the shipped libemx.so), R rounds each
#   usage: tools/ab_bench.sh [rounds] name [name ...]          (env AB_ARGS: extra bench.py arguments)
R=${1:-3}; shift
for i in $(seq $R); do
  for which in "$@"; do
    if [ $which = cur ]; then unset EMX_LIB; else export EMX_LIB=$PWD/emcee_amd/libemx_$which.so; fi
    v=$(timeout 100 python bench.py --no-cpu-baseline --no-extras $AB_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4e  %.3f us/step  kernel-avg %.2f us  per-launch %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_launch_us'], d['roofline']['per_launch_event_us'] or 0))")
    echo "$which $v"
  done
done
