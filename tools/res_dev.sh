#!/bin/bash
# Resource table of a DEVELOPMENT build of one mode TU (row counts 7/11/20 only), with extra -D flags:
#   tools/res_dev.sh strict -DTL_INVU_ASPH_WAVES=4
MODE=${1:-strict}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/res
CONTRACT=$([ "$MODE" = fast ] && echo fast || echo off)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -mllvm -memssa-check-limit=4000 \
  -mllvm -amdgpu-sched-strategy=iterative-ilp -ffp-contract=$CONTRACT -Rpass-analysis=kernel-resource-usage -DTL_INVU_DEV "$@" \
  -c $R/torchoptics_amd/csrc/tl_$MODE.hip -o $R/build/res/tl_${MODE}_dev.o 2> $R/build/res/${MODE}_dev_res.txt
python3 $R/tools/res_table.py "$R/build/res/${MODE}_dev_res.txt" "${RES_PAT:-fwd|inv}"
