"""tools/solve_trace.py <kernel_trace.csv> <launches per solve> -- per position in the solve's launch sequence: kernel, grid, median / min duration
(rocprofv3 --kernel-trace of tools/time_solves_graph.py; durations there run from the end of the launch before, i.e. they include the boundary)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_sfw" in r["Kernel_Name"] or "k_sbw" in r["Kernel_Name"]]
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = rows[-100 * n:]
agg = collections.defaultdict(list)
for i, r in enumerate(seq):
    agg[(i % n, r["Kernel_Name"].split("(")[0].replace("sdm::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0.0
for k in sorted(agg):
    v = sorted(agg[k])
    tot += v[len(v) // 2]
    print("%2d %-12s workgroups %6d  median %7.2f us  min %7.2f us" % (k[0], k[1], k[2], v[len(v) // 2] / 1e3, v[0] / 1e3))
print("sum of medians %.2f us" % (tot / 1e3))
