#!/bin/bash
set -u
mkdir -p gpurun_out/r03s
O=gpurun_out/r03s
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "replay or direct_ipc or sharded" > $O/pytest_part.log 2>&1; tail -3 $O/pytest_part.log
for i in 1 2; do
timeout 600 python bench.py --gpus 2 --all-on-device 0 --config c2 --exchange replay_push --steps 20 --warmup 5 2> $O/bench_n2.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); e=d['multi_gpu']['c2_weak_65536_per_gpu']['exchange']['replay_push']; print('2 ranks on one GPU, C2 weak replay_push: %.1f us/step, agree %s status %s' % (e['ms_per_step']*1e3, e['replicas_agree'], e['device_status']))"
done
timeout 300 python bench.py --force-dist --config c2 --exchange replay --steps 20 --warmup 5 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print({k: round(v['ms_per_step']*1e3,2) for k,v in d['multi_gpu']['c2_weak_65536_per_gpu']['exchange'].items()})"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
