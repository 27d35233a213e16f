#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile_bench.sh into a short text summary
(per-kernel average duration; HBM bytes per launch from FETCH_SIZE / WRITE_SIZE with the gfx950
correction from MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x and is
in KiB... see the unit note printed below; SQ counters per launch)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]


def rows(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    r = []
    for f in fs:
        r += list(csv.DictReader(open(f)))
    return r


print("== kernel trace (ns) ==")
d = collections.defaultdict(list)
meta = {}
for r in rows("trace", "*kernel_trace.csv"):
    n = r["Kernel_Name"].split("(")[0][-48:]
    d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta[n] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"], r["Grid_Size_Y"],
               r["Workgroup_Size_X"])
tot = sum(sum(v) for v in d.values())
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"{n:50s} calls {len(v):4d} avg {sum(v) / len(v) / 1e3:10.1f} us  {100 * sum(v) / tot:5.1f}%  vgpr/agpr/sgpr/lds/grid/wg {meta[n]}")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_grbm"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(sub, "*counter_collection.csv"):
        n = r["Kernel_Name"].split("(")[0][-48:]
        if "trace_" not in n:
            continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if acc:
        print(f"== {sub} (mean per launch) ==")
    for n, c in acc.items():
        print(f"{n:50s} " + "  ".join(f"{k}={sum(v) / len(v):.4g}" for k, v in c.items()))
print("note: FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts a wide "
      "coalesced read stream at half its bytes (MI355X_MICROARCH.md, HBM section) - dword-per-lane streams "
      "as used here are uncalibrated there, so the raw and the doubled value are both quoted in DESIGN.md.")
