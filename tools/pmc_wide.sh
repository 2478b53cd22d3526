#!/bin/bash
# SQ counters of k_wide_lp at 65536 x 512 (two passes of <= 8 SQ counters)
export TMPDIR=/tmp
CFG=${1:-65536x512}
mkdir -p gpurun_out/pmcw
rm -rf gpurun_out/pmcw/*
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d gpurun_out/pmcw/sq1 -o w -f csv -- python tools/wide_bench.py --steps 4 --configs $CFG > gpurun_out/pmcw/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA -d gpurun_out/pmcw/sq2 -o w -f csv -- python tools/wide_bench.py --steps 4 --configs $CFG > gpurun_out/pmcw/sq2.log 2>&1
python - <<PY
import csv, collections, statistics, glob
for f in sorted(glob.glob("gpurun_out/pmcw/*/w_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_wide_lp" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f.split("/")[2], k, "median=%.4g n=%d"%(statistics.median(v), len(v)))
PY
tail -2 gpurun_out/pmcw/sq1.log
