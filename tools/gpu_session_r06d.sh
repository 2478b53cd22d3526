#!/bin/bash
# round 6, session d: the persistent slab kernel -- bit-equality (tests/test_gpu_persist_slab.py), then us/step against the
# per-half-step launches over ndim x walkers, then the bench's w128 configuration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06d
O=$PWD/gpurun_out/r06d
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_persist_slab.py -q -m gpu -p no:cacheprovider ) > $O/pslab_tests.log 2>&1; echo "persistent slab tests rc=$?" | tee -a $O/summary.txt
tail -n 25 $O/pslab_tests.log
timeout 1500 python tools/pslab_bench.py 200 2>/dev/null | tee $O/pslab_bench.txt
timeout 300 python tools/ab_cfg.py 20 w128 2>/dev/null | tee -a $O/pslab_bench.txt
timeout 300 python tools/ab_cfg.py 400 w128 2>/dev/null | tee -a $O/pslab_bench.txt
du -sh $O
