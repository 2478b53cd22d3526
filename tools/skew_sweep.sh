#!/bin/bash
# EMX_OPT_SKEW experiments: extra delay of the staging waves (s_sleep(8) units) against the bench headline
for r in 1 2; do for s in 0 2 4 8 16; do
  v=$(EMX_SKEW_SLEEP=$s timeout 100 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4e  %.3f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "skew_sleep=$s $v"
done; done
