"""Condense a tools/profile_round.sh capture into the two files kept under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel summary
  <tag>_pmc_traffic.json   HBM bytes per launch per kernel = (2*FETCH_SIZE + WRITE_SIZE) * 1024
                           (FETCH doubled per the gfx950 note of MI355X_MICROARCH.md, HBM section; the counters
                           come from separate passes)."""
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    hits = glob.glob(os.path.join(out, "**", pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<\d+>)?)", name)
    return m.group(1) if m else name


stats = find("*kernel_stats.csv")
if stats:
    os.makedirs(os.path.join(out, "keep"), exist_ok=True)
    dst = os.path.join(out, "keep", f"{tag}_kernel_stats.csv")
    with open(stats) as f, open(dst, "w") as g:
        g.write(f.read())
    print("kernel stats ->", dst)


def per_kernel(counter_dir, counter):
    f = find(os.path.join(counter_dir, "**", "*counter_collection.csv")) or find("*" + counter_dir + "*counter_collection.csv")
    files = glob.glob(os.path.join(out, counter_dir, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for fn in files:
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row.get("Kernel_Name", ""))
                v = float(row.get("Counter_Value", 0.0))
                d = row.get("Dispatch_Id")
                a = acc.setdefault(k, {})
                a[d] = a.get(d, 0.0) + v          # sum over XCDs / instances of one dispatch
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in acc.items() if v}


fetch = per_kernel("fetch", "FETCH_SIZE")
write = per_kernel("write", "WRITE_SIZE")
traffic = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, bench.py --steps 10): "
                    "(2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section); "
                    "_raw holds (FETCH_SIZE KB, WRITE_SIZE KB, launches seen)."}
raw = {}
for k in sorted(set(fetch) | set(write)):
    fk, wk = fetch.get(k, (0.0, 0)), write.get(k, (0.0, 0))
    traffic[k] = int(round((2.0 * fk[0] + wk[0]) * 1024))
    raw[k] = [round(fk[0], 2), round(wk[0], 2), max(fk[1], wk[1])]
traffic["_raw"] = raw
dst = os.path.join(out, "keep", f"{tag}_pmc_traffic.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(traffic, open(dst, "w"), indent=1)
print("pmc traffic ->", dst, {k: v for k, v in traffic.items() if not k.startswith("_")})


# ---- matrix-core utilisation per kernel (the "mfma" pass of tools/profile_round.sh)
def per_kernel_multi(counter_dir):
    files = glob.glob(os.path.join(out, counter_dir, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for fn in files:
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name")
                a = acc.setdefault(k, {}).setdefault(c, {})
                dsp = row.get("Dispatch_Id")
                a[dsp] = a.get(dsp, 0.0) + float(row.get("Counter_Value", 0.0))      # summed over XCDs / shader engines of one dispatch
    return {k: {c: (sum(v.values()) / len(v), len(v)) for c, v in cs.items() if v} for k, cs in acc.items()}


mf = per_kernel_multi("mfma")
if mf:
    NCU = 256
    res = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass), averages per launch, "
                    "summed over the 8 XCDs.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs): the gfx94x MfmaUtil formula "
                    "(ROCm 7.2 has no gfx950 derived counters, MI355X_MICROARCH.md), GRBM_GUI_ACTIVE divided by the 8 XCD instances it is summed over.  "
                    "flops_from_mops = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 (one MOP = 512 flops in the gfx94x definition): compare with the algorithmic flops."}
    for k, cs in sorted(mf.items()):
        busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", (0.0, 0))[0]
        gui = cs.get("GRBM_GUI_ACTIVE", (0.0, 0))[0]
        mops = cs.get("SQ_INSTS_VALU_MFMA_MOPS_F64", (0.0, 0))[0]
        sqb = cs.get("SQ_BUSY_CYCLES", (0.0, 0))[0]
        n = max(v[1] for v in cs.values())
        if busy <= 0 and mops <= 0:
            continue
        res[k] = {"launches_seen": n, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_VALU_MFMA_MOPS_F64": mops, "SQ_BUSY_CYCLES": sqb, "GRBM_GUI_ACTIVE": gui,
                  "mfma_util": (busy / (gui / 8.0 * NCU * 4.0)) if gui > 0 else None, "flops_from_mops": mops * 512.0}
    dst = os.path.join(out, "keep", f"{tag}_pmc_mfma.json")
    json.dump(res, open(dst, "w"), indent=1)
    print("pmc mfma ->", dst, {k: (v.get("mfma_util") if isinstance(v, dict) else None) for k, v in res.items() if not k.startswith("_")})
