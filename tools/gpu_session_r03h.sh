#!/bin/bash
set -u
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
export TMPDIR=/tmp
for w in user torch; do
  rm -rf $O/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o p -f csv -- python tools/callback_profile.py $w 400 > $O/prof_$w.log 2>&1
  grep "us/step" $O/prof_$w.log
  find $O/prof_$w -name "*kernel_stats.csv" -exec head -12 {} \;
  find $O/prof_$w -name "*kernel_trace.csv" -exec sh -c 'tail -40 "$1" > "$1.tail"; rm "$1"' _ {} \;
done
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "device_side or two_processes" > $O/pytest_push.log 2>&1; tail -4 $O/pytest_push.log
