#!/usr/bin/env python3
"""Table of registers / scratch / LDS / occupancy from a hipcc -Rpass-analysis=kernel-resource-usage log."""
import re, subprocess, sys
rows, cur = [], None
for ln in open(sys.argv[1]):
    m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        name = re.sub(r"\(tl_problem.*", "", name).replace("void ", "")
        cur = {"name": re.sub(r"tl_\w+_impl::", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
print(f"{'kernel':58s} {'SGPR':>5s} {'VGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if pat and not re.search(pat, r['name']):
        continue
    print(f"{r['name'][:58]:58s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
