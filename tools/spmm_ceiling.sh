#!/bin/bash
# usage (on the GPU box): tools/spmm_ceiling.sh <outdir>
# Makes the dual SpMM's bound reproducible: the product ring kernel against two diagnostic builds of the SAME
# kernel on the same box and data -- SCN_SPMM_FLOOR=1: LDS-DMA + stores only (no gather: what the memory system alone takes for
# this block structure), SCN_SPMM_FLOOR=2: gather + stores only (no LDS-DMA) -- plus the fabric-side counters of the product
# kernel (FETCH / WRITE, TCC hit rate, TCC_EA0_RDREQ vs TCC_EA0_RDREQ_DRAM).  Writes <outdir>/spmm_ceiling.json.
set -e
OUT=$1
cd $GRAFT_REPO_ROOT
mkdir -p $OUT
if [ -z "$SCN_CEILING_PREBUILT" ]; then     # (set it when tools/ab/lib_floor{1,2}.so were cross-compiled from this tree before the snapshot)
  bash tools/ab_build.sh floor1 "-DSCN_SPMM_FLOOR=1" > $OUT/build1.log 2>&1
  bash tools/ab_build.sh floor2 "-DSCN_SPMM_FLOOR=2" > $OUT/build2.log 2>&1
fi
python3 tools/prof_kernels.py --which spmm --reps 5 > $OUT/time_product.log 2>&1
SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_floor1.so python3 tools/prof_kernels.py --which spmm --reps 5 > $OUT/time_floor1.log 2>&1
SCN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/lib_floor2.so python3 tools/prof_kernels.py --which spmm --reps 5 > $OUT/time_floor2.log 2>&1
bash tools/pmc_run.sh $OUT/pmc fetch,write,tcc,ea tools/prof_kernels.py --which spmm --reps 2 > $OUT/pmc.log 2>&1
python3 - <<PY
import json, re
def ms(f):
    for l in open(f):
        m = re.match(r"spmm_dual k128 (\d+) ([0-9.]+) ms", l)
        if m:
            return float(m.group(2))
E, K, S = 996634, 128, 32
alg = 12.0 * E * K * S
pmc = json.load(open("$OUT/pmc/pmc.json"))
k = [v for n, v in pmc.items() if "spmm_ring" in n][0]
t = {"product": ms("$OUT/time_product.log"), "dma_and_stores_only": ms("$OUT/time_floor1.log"), "gather_and_stores_only": ms("$OUT/time_floor2.log")}
res = {"workload": "[L_low X, L_up X], X = [32, 996634, 128] dense random fp32 (tools/prof_kernels.py --which spmm)",
       "algorithmic_bytes": alg, "ms": t,
       "algorithmic_GBps": {n: alg / (v * 1e-3) / 1e9 for n, v in t.items() if v},
       "frac_of_8TBps": {n: alg / (v * 1e-3) / 8e12 for n, v in t.items() if v},
       "pmc_product_kernel": k,
       "fabric_GBps_product": k.get("hbm_bytes_per_launch", 0) / (t["product"] * 1e-3) / 1e9 if t["product"] else None,
       "fabric_GBps_dma_and_stores_only": k.get("hbm_bytes_per_launch", 0) / (t["dma_and_stores_only"] * 1e-3) / 1e9 if t["dma_and_stores_only"] else None}
json.dump(res, open("$OUT/spmm_ceiling.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
