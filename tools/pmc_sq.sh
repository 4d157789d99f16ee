#!/bin/bash
# usage: tools/pmc_sq.sh <outdir> <which> ; the two SQ passes of tools/pmc.sh only (instruction mix, LDS, MFMA busy)
set -e
OUT=$1; WHICH=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 tools/prof_kernels.py --which $WHICH --reps 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "scn::" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt","w") as o:
    for k,d in agg.items():
        o.write(k+"\n")
        for c,v in sorted(d.items()):
            o.write("   %-28s n=%d mean=%.4g\n"%(c,len(v),sum(v)/len(v)))
print(open("$OUT/summary.txt").read())
PY
