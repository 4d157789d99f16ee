#!/bin/bash
# usage (on the GPU box): tools/small_step_ab.sh <outdir> : the small-complex optimiser step with the one-launch kernel (scn_small_step)
# and with the layer-by-layer kernels (SCN_SMALL_STEP=0), wall clock per step and the rocprofv3 kernel table of each.
set -euo pipefail
OUT=$1; mkdir -p "$OUT"
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
python3 tools/small_step.py 2000 > "$OUT/small_on.txt" 2>&1
SCN_SMALL_STEP=0 python3 tools/small_step.py 2000 > "$OUT/small_off.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/prof_on" -o small_on -- python3 tools/small_step.py 300 > "$OUT/prof_on.txt" 2>&1
cat "$OUT/small_on.txt" "$OUT/small_off.txt"
