"""GPU timeline of a rocprofv3 --kernel-trace csv: per kernel name calls / mean duration, and the idle time between consecutive kernels
   usage: python tools/exp/kernel_gaps.py <dir> [skip_fraction]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
rows = rows[int(len(rows) * float(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2):]        # steady state: the second half
busy = collections.defaultdict(lambda: [0, 0.0]); gap_after = collections.defaultdict(lambda: [0, 0.0])
end = rows[0][0]
for (s, e, name), nxt in zip(rows, rows[1:] + [None]):
    k = name.split("(")[0][:60]
    busy[k][0] += 1; busy[k][1] += (e - s) / 1e3
    if nxt:
        g = max(0.0, (nxt[0] - max(e, end)) / 1e3)
        gap_after[k][0] += 1; gap_after[k][1] += g
    end = max(end, e)
span = (rows[-1][1] - rows[0][0]) / 1e3
print("span %.1f us, %d kernels" % (span, len(rows)))
for k, (n, t) in sorted(busy.items(), key=lambda x: -x[1][1]):
    g = gap_after[k]
    print("%-62s calls %6d  mean %8.2f us  share %5.1f %%   idle after it: mean %6.2f us (%4.1f %% of span)" % (k, n, t / n, 100 * t / span, g[1] / max(1, g[0]), 100 * g[1] / span))
