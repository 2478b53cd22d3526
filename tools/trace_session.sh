#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/trace; rm -rf gpurun_out/trace/*
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/trace/t -o c2 -f csv -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/trace/log.txt 2>&1
cat gpurun_out/trace/t/c2_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/trace/t/c2_kernel_trace.csv")))
rows=[r for r in rows if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
mid=rows[len(rows)//2: len(rows)//2+9]
t0=int(mid[0]["Start_Timestamp"])
for r in mid:
    print("%-40s start=%7.2f us dur=%6.2f us grid=%s wg=%s"%(r["Kernel_Name"][:40],(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r["Grid_Size"],r["Workgroup_Size"]))
PY
