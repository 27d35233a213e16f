#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the rocprofv3 PMC passes of tools/profile_bench.sh.

  usage: tools/make_pmc_traffic.py OUT.json cfg3:strict=gpurun_out/prof_X cfg3:fast=gpurun_out/prof_Y cfg3a:strict=...

Per workload / arithmetic mode: mean FETCH_SIZE and WRITE_SIZE per launch (KiB, separate passes) of the forward
kernel and of the walk-back backward kernel, and hbm_bytes = (2 x FETCH + WRITE) x 1024 -- FETCH_SIZE doubled as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 (confirmed on this access pattern: the forward kernel's
read stream is exactly P x 8 B = 134.2 MB and reads 65,600 KiB raw).  bench.py quotes hbm_bytes as roofline.traffic.
"""
import collections
import csv
import glob
import json
import os
import sys


def mean_counter(prof, sub, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(prof, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(d, *needles):
    hits = [k for k in d if any(n in k for n in needles)]
    assert len(hits) == 1, (needles, list(d))
    return hits[0], d[hits[0]]


out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh <tag> fetch|write; "
                 "bench.py --steps 3 --warmup 1 --repeats 1, main workload only), assembled by tools/make_pmc_traffic.py; "
                 "FETCH_SIZE / WRITE_SIZE in KiB, hbm_bytes = (2 x FETCH + WRITE) x 1024 per MI355X_MICROARCH.md 'HBM'"}
for spec in sys.argv[2:]:
    key, prof = spec.split("=")
    wl, mode = key.split(":")
    fe, wr = mean_counter(prof, "pmc_fetch", "FETCH_SIZE"), mean_counter(prof, "pmc_write", "WRITE_SIZE")
    ent = {}
    for tag, needles in (("fwd", ("trace_fwd_kernel", "trace_fwd_plain_kernel")), ("bwd", ("trace_bwd_inv_kernel", "trace_bwd_inv_unrolled_kernel"))):
        kn, f = pick(fe, *needles)
        _, w = pick(wr, *needles)
        ent[tag] = {"kernel": kn.replace("void ", ""), "fetch_KiB_raw": f, "write_KiB": w, "hbm_bytes": int(round((2 * f + w) * 1024))}
    out.setdefault(wl, {})[mode] = ent
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
