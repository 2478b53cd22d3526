#!/bin/bash
# round 6, session c: the flat form of the device-wide barrier (persist_hier = 2) -- bit-equality of the persistent suite under it,
# A/B of the three barriers on C2 / C4 / C2 with the chain stored, the phase clock of each; the live-reference test as corrected;
# bench.py --gpus 2 --all-on-device 0 (the N > 1 emitter on a real box: control flow only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06c
O=$PWD/gpurun_out/r06c
export TMPDIR=/tmp
( time EMX_TUNE=persist_hier=2 timeout 900 python -m pytest tests/test_gpu_persist.py -q -x -m gpu -p no:cacheprovider ) > $O/persist_tests_flat.log 2>&1; echo "persist tests under persist_hier=2 rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/persist_tests_flat.log
for rep in 1 2; do
for h in 0 1 2; do
  for K in 20 400; do
    EMX_TUNE=persist_hier=$h timeout 300 python tools/ab_cfg.py $K c2 c4 c2+store 2>/dev/null | sed "s/^/rep=$rep hier=$h /" | tee -a $O/ab_hier.txt
  done
done
done
bash tools/ab_variants.sh stamps "-DEMX_OPT_STAMPS=1" > $O/stamps_build.log 2>&1; echo "stamps build rc=$?" | tee -a $O/summary.txt
for h in 0 1 2; do timeout 300 python tools/persist_phase_clock.py 65536 64 0 $h 2>&1 | grep -v amdgpu.ids | tee -a $O/persist_phase_hier.txt; done
( time timeout 600 python -m pytest tests/test_gpu_live_reference.py -q -m gpu -p no:cacheprovider ) > $O/live_ref.log 2>&1; echo "live reference rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/live_ref.log
( time timeout 1200 python bench.py --gpus 2 --steps 20 --warmup 5 --all-on-device 0 ) > $O/bench_n2_one_device.json 2> $O/bench_n2_one_device.err; echo "bench --gpus 2 --all-on-device rc=$?" | tee -a $O/summary.txt
wc -c $O/bench_n2_one_device.json | tee -a $O/summary.txt
cp gpurun_out/bench_detail.json $O/bench_detail_n2.json 2>/dev/null
grep -v "bench-detail" $O/bench_n2_one_device.err | tail -n 15
rm -f emcee_amd/libemx_stamps.so
du -sh $O
