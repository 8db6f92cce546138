"""tools/unit_timeline.py <kernel_trace.csv> [first kernel of a unit] -- the kernels of the last complete bench units of a rocprofv3 --kernel-trace run as a
timeline: start and duration of every launch relative to the unit's first kernel, and the gaps between them (median over the units)."""
import csv
import sys
import statistics

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("sdm::", "").replace("void ", "") for r in rows]
first = "k_lq_q_prep" if any(n.startswith("k_lq_q_prep") for n in names) else names[0]
# units = runs starting with the unit's first kernel; keep those of the most common length
delim = sys.argv[2] if len(sys.argv) > 2 else "k_psd_stage1_mfma"
starts = [i for i, n in enumerate(names) if n.startswith(delim)]
units = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
lens = [b - a for a, b in units]
common = statistics.mode(lens)
units = [(a, b) for a, b in units if b - a == common][-40:]
print("units of", common, "launches;", len(units), "used")
for k in range(common):
    st = [int(rows[a + k]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in units]
    du = [int(rows[a + k]["End_Timestamp"]) - int(rows[a + k]["Start_Timestamp"]) for a, b in units]
    gap = [int(rows[a + k]["Start_Timestamp"]) - max(int(rows[a + j]["End_Timestamp"]) for j in range(k)) if k else 0 for a, b in units]
    print("%-28s start %8.2f us  dur %8.2f us  gap after the latest end before it %7.2f us" % (names[units[0][0] + k][:28], statistics.median(st) / 1e3, statistics.median(du) / 1e3, statistics.median(gap) / 1e3))
tot = [int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in units]
print("unit period %.2f us" % (statistics.median(tot) / 1e3))
