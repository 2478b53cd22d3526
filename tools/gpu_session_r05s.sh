#!/bin/bash
# round 5, session s: finisher count with device finish (threads vs physical cores of the L3 domain), alternating processes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
for cfg in "EMX_TUNE=mt_pipeline=6" "EMX_TUNE=mt_pipeline=4" "EMX_TUNE=mt_pipeline=5"; do
  echo "== $cfg" | tee -a $O/exact_c2_s.txt
  env $cfg timeout 120 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['pipeline_stage_us_per_step']
print('%.2f us/step  generator %.1f tokenizer %.1f finishers %.0f  waited: words %.1f consumer %.1f' % (d['ms_per_step'] * 1e3, p['generator_us'], p['tokenizer_us'], p['finishers_us_summed'], p['tokenizer_waited_for_words_us'], p['tokenizer_waited_for_consumer_us']))" | tee -a $O/exact_c2_s.txt
done
done
