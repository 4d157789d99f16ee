#!/bin/bash
# usage (GPU box): tools/pmc_all.sh <outdir>  -- the rocprofv3 --pmc traffic passes behind bench.py's `roofline.traffic` (FETCH_SIZE, WRITE_SIZE,
# TCC hit / miss in separate runs: tools/pmc_run.sh), one section per launch shape, merged by tools/pmc_merge.py into <outdir>/pmc_traffic.json
# and stamped there with the hash of the kernel sources the passes ran on (copy it to profiles/rNN_pmc_traffic.json).
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
OUT=${1:?usage: tools/pmc_all.sh <outdir>}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() {
  local name=$1 limit=$2; shift 2
  local t0=$(date +%s)
  local rc=0
  timeout -k 10 $limit "$@" > $OUT/$name.log 2>&1 || rc=$?
  echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/status.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping" | tee -a $OUT/status.txt; exit 1; fi
}
step pmc_main_bench 400 bash tools/pmc_run.sh $OUT/pmc_main_bench fetch,write,tcc tools/skip_step.py dense 2 512
step pmc_main 400 bash tools/pmc_run.sh $OUT/pmc_main fetch,write,tcc tools/prof_kernels.py --which spmm,fwd,bwd,fwd1,bwdf --reps 2
step pmc_cfg1 300 bash tools/pmc_run.sh $OUT/pmc_cfg1 fetch,write,tcc tools/prof_kernels.py --edges 50000 --hidden 16 --slabs 256 --which fwd,bwd,fwd1,bwdf --reps 2
step pmc_ebli 400 bash tools/pmc_run.sh $OUT/pmc_ebli fetch,write,tcc tools/prof_ebli.py --reps 2
step pmc_bunch 400 bash tools/pmc_run.sh $OUT/pmc_bunch fetch,write,tcc tools/prof_bunch.py --reps 2 --slabs 32
python3 tools/pmc_merge.py $OUT/pmc_traffic.json \
  "main_bench=$OUT/pmc_main_bench/pmc.json:|E|=996634, hidden 32, the benchmark's own trajectories: 2 optimiser steps of 512 trajectories = 8 micro-batches of 128 (tools/skip_step.py dense 2 512)" \
  "main=$OUT/pmc_main/pmc.json:|E|=996634, hidden 32, 32 slabs = 128 trajectories of dense random data (tools/prof_kernels.py --which spmm,fwd,bwd,fwd1,bwdf)" \
  "configs[1]=$OUT/pmc_cfg1/pmc.json:|E|=49616, hidden 16, 256 slabs = 1024 trajectories of dense random data (tools/prof_kernels.py --edges 50000 --hidden 16 --slabs 256)" \
  "ebli=$OUT/pmc_ebli/pmc.json:|E|=996634, hidden 32, 32 slabs of dense random data (tools/prof_ebli.py)" \
  "bunch=$OUT/pmc_bunch/pmc.json:|E|=996634, hidden 32, 32 slabs = 128 trajectories of dense random data, the model's third-layer launches (tools/prof_bunch.py --slabs 32)" \
  > $OUT/merge.log 2>&1
echo "merge rc=$?" | tee -a $OUT/status.txt
