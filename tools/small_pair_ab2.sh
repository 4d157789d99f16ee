# usage (on the GPU box): tools/small_pair_ab2.sh -- what the hand-over of the paired one-launch step costs: the step with the collect skipped
# (-DSM_AB_NO_COLLECT, wrong results), with plain instead of agent-scope stores (-DSM_AB_PLAIN_STORE), and with both, at |E| = 1001 / 100 trajectories.
set -euo pipefail
export TMPDIR=/tmp
O=gpurun_out/pair3; mkdir -p $O
bash tools/ab_build.sh nocollect "-DSM_AB_NO_COLLECT" > $O/build.txt 2>&1
bash tools/ab_build.sh plainstore "-DSM_AB_PLAIN_STORE" >> $O/build.txt 2>&1
bash tools/ab_build.sh both "-DSM_AB_PLAIN_STORE -DSM_AB_NO_COLLECT" >> $O/build.txt 2>&1
export SCN_POINTS=400 SCN_TRAJ=100
{
for v in cur nocollect plainstore both; do
  [ $v = cur ] && unset SCN_LIB_PATH || export SCN_LIB_PATH=tools/ab/lib_$v.so
  echo -n "$v: "; SCN_SMALL_STEP=force timeout -k 10 120 python3 tools/small_step.py 2000 2>&1 | grep -v amdgpu.ids
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
