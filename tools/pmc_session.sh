#!/bin/bash
# SQ counters for the bench kernel (one pass; <= 8 SQ counters)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rm -rf gpurun_out/pmc/*
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS -d gpurun_out/pmc/sq1 -o c2 -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/pmc/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES -d gpurun_out/pmc/sq2 -o c2 -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/pmc/sq2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d gpurun_out/pmc/grbm -o c2 -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/pmc/grbm.log 2>&1
python - <<PY
import csv, collections, statistics, glob
for f in sorted(glob.glob("gpurun_out/pmc/*/c2_counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_halfstep" in r["Kernel_Name"] and ", 0, " in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f.split("/")[2], k, "median=%.4g n=%d"%(statistics.median(v), len(v)))
PY
tail -3 gpurun_out/pmc/sq1.log
