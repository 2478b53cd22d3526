#!/bin/bash
# round 5, session s2: the tokenizer's copy into the pinned staging buffer with non-temporal stores, alternating processes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
O=$PWD/gpurun_out/r05
export TMPDIR=/tmp
for rep in 1 2 3 4; do
for cfg in "EMX_PIPE_NT_COPY=0" "EMX_PIPE_NT_COPY=1"; do
  echo "== $cfg" | tee -a $O/exact_c2_s2.txt
  env $cfg timeout 120 python tools/exact_mode_probe.py 2>&1 | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
p = d['pipeline_stage_us_per_step']
print('%.2f us/step  generator %.1f tokenizer %.1f finishers %.0f  waited: words %.1f consumer %.1f' % (d['ms_per_step'] * 1e3, p['generator_us'], p['tokenizer_us'], p['finishers_us_summed'], p['tokenizer_waited_for_words_us'], p['tokenizer_waited_for_consumer_us']))" | tee -a $O/exact_c2_s2.txt
done
done
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -p no:cacheprovider -k "exact or mt or golden or c2 or full" ) > $O/tests_s2.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests_s2.log
