#!/bin/bash
# usage: tools/pmc_run.sh <outdir> <passes> <script.py> [args...]
# rocprofv3 --pmc passes (one counter group per run, the program itself after `--`) of a python driver under tools/, aggregated
# per kernel into <outdir>/pmc.json: mean counter values per launch, plus hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB
# (gfx950: FETCH_SIZE reports half of wide coalesced reads, MI355X_MICROARCH.md section HBM) and the TCC hit rate.
# <passes>: comma list out of  fetch,write,tcc,ea,sq1,sq2,sq3,sq4
set -e
OUT=$1; PASSES=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
declare -A CTRS
CTRS[fetch]="FETCH_SIZE"
CTRS[write]="WRITE_SIZE"
CTRS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
CTRS[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"
CTRS[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
CTRS[sq2]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
CTRS[sq3]="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
CTRS[sq4]="SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU"
for p in ${PASSES//,/ }; do
  rocprofv3 --pmc ${CTRS[$p]} --output-format csv -d $OUT/$p -- python3 "$@" > $OUT/$p.log 2>&1 || echo "pass $p failed (see $OUT/$p.log)"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        if "scn::" not in k:
            continue
        k = k.split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    m["launches_seen"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    if "TCC_HIT_sum" in m:
        m["tcc_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in m and m["TCC_EA0_RDREQ_sum"]:
        m["ea_rdreq_dram_share"] = m.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / m["TCC_EA0_RDREQ_sum"]
    res[k] = m
json.dump(res, open("$OUT/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
