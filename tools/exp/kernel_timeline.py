"""a window of the GPU timeline of a rocprofv3 --kernel-trace csv (steady state: from the middle of the trace)
   usage: python tools/exp/kernel_timeline.py <dir> [rows] [min_dur_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
t0 = int(rows[0]["Start_Timestamp"])
n = 0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < mind:
        continue
    name = r["Kernel_Name"].replace("emx::", "").replace("void ", "")[:34]
    print("%-34s q%-3s start %10.1f end %10.1f dur %8.1f" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, d))
    n += 1
    if n >= nrows:
        break
