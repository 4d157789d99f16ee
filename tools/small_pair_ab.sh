# usage (on the GPU box): tools/small_pair_ab.sh [outdir] -- the one-launch step with two workgroups per trajectory (default where it applies),
# with one (SCN_SMALL_PAIRING=0) and the layer-by-layer kernels, graph-replayed (tools/small_step.py), after the small-step tests.
set -euo pipefail
export TMPDIR=/tmp
O=${1:-gpurun_out/pair}; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_small_step.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
for cfg in ${CFGS:-400:100 400:128 300:100 250:100 200:100}; do
  set -- ${cfg/:/ }
  export SCN_POINTS=$1 SCN_TRAJ=$2
  echo "== $1 points, $2 trajectories"
  echo -n "paired   : "; SCN_SMALL_STEP=force timeout -k 10 120 python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
  echo -n "single   : "; SCN_SMALL_PAIRING=0 SCN_SMALL_STEP=force timeout -k 10 120 python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
  echo -n "layers   : "; SCN_SMALL_STEP=0 timeout -k 10 120 python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
done > $O/ab.txt 2>&1
cat $O/ab.txt
