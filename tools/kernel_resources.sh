#!/bin/bash
# Register / LDS / occupancy table of every kernel of one translation unit, as the compiler sees it
# (hipcc -Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.sh [strict|fast]
MODE=${1:-strict}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/res
CONTRACT=$([ "$MODE" = fast ] && echo fast || echo off)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -mllvm -memssa-check-limit=4000 \
  -mllvm -amdgpu-sched-strategy=iterative-ilp -ffp-contract=$CONTRACT -Rpass-analysis=kernel-resource-usage \
  -c $R/torchoptics_amd/csrc/tl_$MODE.hip -o $R/build/res/tl_$MODE.o 2> $R/build/res/${MODE}_res.txt
python3 - "$R/build/res/${MODE}_res.txt" <<'PY'
import re, subprocess, sys
rows, cur = [], None
for ln in open(sys.argv[1]):
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", ln) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        cur = {"name": re.sub(r"^tl_\w+_impl::", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ(waves/SIMD)':>16s}")
for r in rows:
    print(f"{r['name'][:58]:58s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>16s}")
PY
