#!/bin/bash
# round 6, session a: the compact bench line measured for real (driver's command and the no-flag default), the GPU suite on the
# code as it stands, the tests this round added so far first (fail fast), rocprofv3 kernel statistics of the headline.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06a
O=$PWD/gpurun_out/r06a
R=$PWD
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_persist.py -q -x -m gpu -p no:cacheprovider -k "restarted_when_its_consumer_changes" ) > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "mid_size" ) >> $O/new_tests.log 2>&1; echo "digest tests rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/new_tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n1.json 2>/dev/null
wc -c $O/bench_n1.json | tee -a $O/summary.txt
grep -v "bench-detail" $O/bench_n1.err | tail -n 5
python - <<'PY' | tee -a $O/summary.txt
import json, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06a"
try:
    d = json.loads(open(O + "/bench_n1.json").read().strip().splitlines()[-1])
    print("value %.4g %s  ms/step %.5f  frac %.4f (value x 1553 / 8e12 = %.4f)  event-clock frac %.4f  blocks %s timed_ms %s" % (
        d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["value"] * 1553 / 8e12, d["roofline"]["frac_event_clock"], d.get("timed_blocks"), d.get("timed_ms")))
    for k, c in d["configs"].items():
        print("  %-28s %.2f us/step frac %.3f" % (k, c["ms_per_step"] * 1e3, c["frac"]))
    print("  exact:", json.dumps(d["exact_mode"]))
    print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"])
except Exception as e:
    print("parse failed", repr(e))
PY
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/summary.txt
wc -c $O/bench_default.json | tee -a $O/summary.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o c2 -f csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/trace_c2.log 2>&1; echo "trace c2 rc=$?" | tee -a $O/summary.txt
cd $R
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/gpu_tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/gpu_tests.log
find $O -name "*kernel_trace.csv" -size +2M -exec sh -c 'head -300 "$1" > "$1.head"; rm "$1"' _ {} \;
find $O -name "*.db" -delete
du -sh $O
