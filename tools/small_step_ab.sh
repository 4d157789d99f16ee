#!/bin/bash
# usage (on the GPU box): tools/small_step_ab.sh <outdir> : the small-complex optimiser step (graph-replayed, tools/small_step.py) with the
# one-launch kernel (scn_small_step; SCN_SMALL_STEP=force lifts the size rule) and with the layer-by-layer kernels (SCN_SMALL_STEP=0) at
# |E| = 1001 / 100 trajectories (TE:86-90) and |E| = 319 / 160 trajectories (the drifter complex's size): wall clock per step, the
# rocprofv3 kernel table of the one-launch step, and -- when tools/build_stamps.sh has been run -- the kernel's phase stamps.
set -euo pipefail
OUT=$1; mkdir -p "$OUT"
cd "$(dirname "${BASH_SOURCE[0]}")/.."
export TMPDIR=/tmp
{
for cfg in "400 100" "130 160"; do
  set -- $cfg
  export SCN_POINTS=$1 SCN_TRAJ=$2
  echo "== $1 points, $2 trajectories"
  echo -n "one launch     : "; SCN_SMALL_STEP=force python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
  echo -n "layer by layer : "; SCN_SMALL_STEP=0 python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
  SCN_SMALL_STEP=force rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$1" -o small -- python3 tools/small_step.py 300 > "$OUT/prof_$1.txt" 2>&1
  f=$(find "$OUT/prof_$1" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "rocprofv3 --kernel-trace --stats of 300 steps:"; cut -d, -f1-4 "$f" | head -6; }
  if [ -f tools/ubench/libscone_hip_stamps.so ]; then
    SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so python3 tools/small_stamps.py $1 $2 2>&1 | grep -v amdgpu.ids | tail -11
  fi
done
} > "$OUT/small_step_ab.txt" 2>&1
cat "$OUT/small_step_ab.txt"
