#!/bin/bash
# usage: tools/r3_steps.sh <outdir> <step> [<step> ...]   -- GPU-box driver: runs the named measurement steps one after another,
# each under its own timeout, logs under <outdir>; stops at the first step that had to be killed.
OUT=$1; shift
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
step() {
  local name=$1 limit=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $limit "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc $(( $(date +%s) - t0 ))s" | tee -a $OUT/status.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping" | tee -a $OUT/status.txt; exit 1; fi
}
for s in "$@"; do
  case $s in
    tests_dense) step tests_dense 700 python3 -m pytest tests/test_gpu_fullsize_dense.py tests/test_gpu_rccl.py -x -q -m gpu ;;
    tests_rest)  step tests_rest 1100 python3 -m pytest tests -x -q -m gpu --ignore=tests/test_gpu_fullsize_dense.py ;;
    ab_noslp)    step ab_noslp_build 300 bash tools/ab_build.sh noslp "-fno-slp-vectorize"
                 step ab_base 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_noslp.so step ab_noslp 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 3 ;;
    ab_scalar)   step ab_scalar_build 300 bash tools/ab_build.sh scalar "-DSCN_AB_GATHER_SCALAR -fno-slp-vectorize"
                 step ab_base 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_scalar.so step ab_scalar 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 3 ;;
    tests_spmm)  step tests_spmm 600 python3 -m pytest tests/test_gpu_fullsize_dense.py tests/test_gpu_parity.py -x -q -m gpu -k "spmm or rectangular or full_size_properties" ;;
    terms_floor) step tf_build1 300 bash tools/ab_build.sh tfloor1 "-DSCN_TERMS_FLOOR=1"
                 step tf_build2 300 bash tools/ab_build.sh tfloor2 "-DSCN_TERMS_FLOOR=2"
                 step tf_base 300 python3 tools/prof_bunch.py --which fwd_nf --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_tfloor1.so step tf_floor1 300 python3 tools/prof_bunch.py --which fwd_nf --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_tfloor2.so step tf_floor2 300 python3 tools/prof_bunch.py --which fwd_nf --reps 3 ;;
    fwd_floor)   step ff_build1 300 bash tools/ab_build.sh ffloor1 "-DSCN_FWD_FLOOR=1"
                 step ff_build2 300 bash tools/ab_build.sh ffloor2 "-DSCN_FWD_FLOOR=2"
                 step ff_base 300 python3 tools/prof_kernels.py --which fwd --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_ffloor1.so step ff_floor1 300 python3 tools/prof_kernels.py --which fwd --reps 3
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_ffloor2.so step ff_floor2 300 python3 tools/prof_kernels.py --which fwd --reps 3 ;;
    layout_check) step tests_quick 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu
                 step ab_new 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 4
                 step bq_new 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                 step pmc_new 300 bash tools/pmc_run.sh $OUT/pmc_new sq2 tools/prof_kernels.py --which fwd,bwd --reps 2 ;;
    ab_act)      step ab_act_tanh 200 python3 tools/prof_kernels.py --which fwd,fwd1 --reps 4
                 step ab_act_relu 200 python3 tools/prof_kernels.py --which fwd,fwd1 --reps 4 --act relu ;;
    terms_check) step tests_bunch 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_dense.py -x -q -m gpu -k "bunch or Bunch or terms or poisoned"
                 step prof_bunch 300 python3 tools/prof_bunch.py --reps 3
                 step pmc_bunch_sq2 300 bash tools/pmc_run.sh $OUT/pmc_bunch_sq2 sq2 tools/prof_bunch.py --reps 2 ;;
    ab_layout)   step abl_new_bunch 300 python3 tools/prof_bunch.py --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_legacy.so step abl_old_bunch 300 python3 tools/prof_bunch.py --reps 4
                 step abl_new_bunch2 300 python3 tools/prof_bunch.py --reps 4
                 step abl_new_c32 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_legacy.so step abl_old_c32 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 4
                 step abl_new_bq 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_legacy.so step abl_old_bq 400 python3 bench.py --extras 0 --steps 5 --warmup 1 ;;
    ab_layout2)  step abl2_new_bunch 300 python3 tools/prof_bunch.py --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_legacy.so step abl2_old_bunch 300 python3 tools/prof_bunch.py --reps 4
                 step abl2_new_bunch2 300 python3 tools/prof_bunch.py --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_legacy.so step abl2_old_bunch2 300 python3 tools/prof_bunch.py --reps 4
                 step tests_bunch 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_dense.py -x -q -m gpu -k "bunch or Bunch or terms or poisoned" ;;
    ab_base)     for r in 1 2; do
                   step abb_new_c32_$r 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 4
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step abb_old_c32_$r 200 python3 tools/prof_kernels.py --which fwd,bwd --reps 4
                   step abb_new_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step abb_old_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                 done ;;
    ab_base_bunch) for r in 1 2; do
                   step abbb_new_$r 300 python3 tools/prof_bunch.py --reps 4
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step abbb_old_$r 300 python3 tools/prof_bunch.py --reps 4
                 done ;;
    ab_three)    for r in 1 2; do
                   step ab3_new_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step ab3_old_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_peel.so step ab3_peel_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                 done
                 step ab3_new_c32 200 python3 tools/prof_kernels.py --which fwd,bwd,fwd1 --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step ab3_old_c32 200 python3 tools/prof_kernels.py --which fwd,bwd,fwd1 --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_peel.so step ab3_peel_c32 200 python3 tools/prof_kernels.py --which fwd,bwd,fwd1 --reps 4 ;;
    ab_four)     for r in 1 2; do
                   step ab4_new_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step ab4_old_bq_$r 400 python3 bench.py --extras 0 --steps 5 --warmup 1
                 done
                 step ab4_new_c32 200 python3 tools/prof_kernels.py --which spmm,fwd,bwd,fwd1 --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step ab4_old_c32 200 python3 tools/prof_kernels.py --which spmm,fwd,bwd,fwd1 --reps 4
                 step ab4_new_bunch 300 python3 tools/prof_bunch.py --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_base.so step ab4_old_bunch 300 python3 tools/prof_bunch.py --reps 4 ;;
    tests_bunch) step tests_bunch 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_dense.py -x -q -m gpu -k "bunch or Bunch or terms or poisoned" ;;
    floors_prebuilt) SCN_CEILING_PREBUILT=1 step spmm_ceiling 500 bash tools/spmm_ceiling.sh $OUT/spmm
                 step ff_base 200 python3 tools/prof_kernels.py --which fwd --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_ffloor1.so step ff_floor1 200 python3 tools/prof_kernels.py --which fwd --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_ffloor2.so step ff_floor2 200 python3 tools/prof_kernels.py --which fwd --reps 4
                 step tf_base 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_tfloor1.so step tf_floor1 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4
                 SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_tfloor2.so step tf_floor2 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4 ;;
    smoke)       step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ;;
    ab_pf)       for r in 1 2; do
                   step abpf_new_$r 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_pf2.so step abpf_pf2_$r 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4
                   SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_pfoff.so step abpf_off_$r 300 python3 tools/prof_bunch.py --which fwd_nf --reps 4
                 done ;;
    tests_quick) step tests_quick 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu ;;
    tests_all)   step tests_all 1100 python3 -m pytest tests -x -q -m gpu ;;
    pmc_bunch)   step pmc_bunch 500 bash tools/pmc_run.sh $OUT/pmc_bunch fetch,write,tcc,sq1,sq2,sq4 tools/prof_bunch.py --reps 2 ;;
    pmc_c32)     step pmc_c32 400 bash tools/pmc_run.sh $OUT/pmc_c32 sq1,sq2,sq4 tools/prof_kernels.py --which fwd,bwd --reps 2 ;;
    pmc_c32_mem) step pmc_c32_mem 400 bash tools/pmc_run.sh $OUT/pmc_c32_mem fetch,write,tcc tools/prof_kernels.py --which spmm,fwd,bwd,fwd1 --reps 2 ;;
    pmc_main)    step pmc_main 400 bash tools/pmc_run.sh $OUT/pmc_main fetch,write,tcc tools/prof_kernels.py --which spmm,fwd,bwd,fwd1,bwdf --reps 2 ;;
    pmc_cfg1)    step pmc_cfg1 300 bash tools/pmc_run.sh $OUT/pmc_cfg1 fetch,write,tcc tools/prof_kernels.py --edges 50000 --hidden 16 --slabs 256 --which fwd,bwd,fwd1,bwdf --reps 2 ;;
    pmc_ebli)    step pmc_ebli 400 bash tools/pmc_run.sh $OUT/pmc_ebli fetch,write,tcc tools/prof_ebli.py --reps 2 ;;
    pmc_bunch_mem) step pmc_bunch_mem 400 bash tools/pmc_run.sh $OUT/pmc_bunch fetch,write,tcc,sq1,sq2 tools/prof_bunch.py --reps 2 ;;
    pmc_main_bench) step pmc_main_bench 400 bash tools/pmc_run.sh $OUT/pmc_main_bench fetch,write,tcc tools/skip_step.py dense 2 512 ;;
    bench_prof)  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
                 step bench_prof 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -- python3 bench.py --extras 0 ;;
    gloo2)       step gloo2 600 python3 bench.py --gpus 2 --backend gloo --extras 0 --steps 2 --warmup 1 --global-batch 512 ;;
    tests_durations) step tests_all 1150 python3 -m pytest tests -x -q -m gpu --durations=25 ;;
    pmc_skip)    step pmc_skip 900 bash tools/pmc_skip.sh $OUT/pmc_skip ;;
    cfg0)        step cfg0 300 python3 tools/cfg1_step_time.py dense breakdown ;;
    small_step)  step small_graph 200 python3 tools/small_step.py
                 step small_eager 200 python3 tools/small_step.py eager
                 cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
                 step small_graph_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small_graph_prof -- python3 tools/small_step.py
                 step small_eager_prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small_eager_prof -- python3 tools/small_step.py eager ;;
    spmm_ceiling) step spmm_ceiling 600 bash tools/spmm_ceiling.sh $OUT/spmm ;;
    bench)       step bench 900 python3 bench.py ;;
    bench_quick) step bench_quick 400 python3 bench.py --extras 0 --steps 5 --warmup 1 ;;
    prof_kernels) step prof_kernels 300 python3 tools/prof_kernels.py --which spmm,fwd,bwd,fwd1 --reps 3 ;;
    bunch_scale) step bunch_scale 300 python3 tools/model_scale.py bunch 1000000 64 32 ;;
    *) echo "unknown step $s" ;;
  esac
done
cat $OUT/status.txt
