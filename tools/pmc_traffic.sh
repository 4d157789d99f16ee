#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> <which> ; the three memory-side passes of tools/pmc.sh only (FETCH_SIZE, WRITE_SIZE, TCC hit/miss)
set -e
OUT=$1; WHICH=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 tools/prof_kernels.py --which $WHICH --reps 2 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "scn::" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={}
for k,d in agg.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        res[k]={"FETCH_SIZE_KB":m["FETCH_SIZE"],"WRITE_SIZE_KB":m["WRITE_SIZE"],
                "hbm_bytes_per_launch":(2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024,
                "tcc_hit_rate":m.get("TCC_HIT_sum",0)/max(1.0,m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0))}
json.dump(res,open("$OUT/traffic.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
