#!/usr/bin/env python3
"""Static instruction budget of one kernel from hipcc's device assembly: per basic block the count of VALU
(by class), MFMA, LDS, vector-memory and scalar instructions, so that the per-slab instruction stream of the fused kernels can be
itemised next to the PMC totals (profiles/r03_isa_budget_*.txt).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Iscone_gcn_amd/csrc --cuda-device-only -S scone_gcn_amd/csrc/scn_blocked.hip -o /tmp/scn_blocked.s
    python tools/isa_budget.py /tmp/scn_blocked.s fwd_c32_w16_kernelILi1ELb0
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32")):
        return "valu_fma"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_max_f32", "v_min_f32")):
        return "valu_f32_other"
    if op.startswith(("v_perm", "v_and", "v_or", "v_xor", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_not", "v_alignb")):
        return "valu_bit"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap")):
        return "valu_mov"
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
        return "valu_cmp_sel"
    if op.startswith(("v_add", "v_sub", "v_mul", "v_mad", "v_mbcnt")):
        return "valu_int"
    if op.startswith("v_"):
        return "valu_misc"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load" + ("_lds" if "lds" in op else "")
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vmem_store"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_barrier", "s_cbranch", "s_branch", "s_setprio", "s_nop", "s_sleep", "s_endpgm")):
        return "s_ctl"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3scn") and key in l and l.rstrip().endswith(("; @" + l.split(":")[0])) or
                 (l.startswith("_ZN3scn") and key in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    cats = ["valu_fma", "valu_f32_other", "valu_trans", "valu_bit", "valu_mov", "valu_cmp_sel", "valu_int", "valu_packed", "valu_misc",
            "mfma", "lds_read", "lds_other", "vmem_load_lds", "vmem_load", "vmem_store", "salu", "smem", "s_waitcnt", "s_ctl"]
    print("%-12s %5s | " % ("block", "VALU") + " ".join("%6s" % c.replace("valu_", "")[:6] for c in cats) + " | branches")
    tot = collections.Counter()
    for name, ins in blocks.items():
        c = collections.Counter(classify(op) for op, _ in ins)
        valu = sum(v for k, v in c.items() if k.startswith("valu_"))
        br = [s.split()[-1] for op, s in ins if op.startswith(("s_cbranch", "s_branch"))]
        print("%-12s %5d | " % (name, valu) + " ".join("%6d" % c.get(k, 0) for k in cats) + " | " + ",".join(br))
        tot.update(c)
    print("total static: VALU %d" % sum(v for k, v in tot.items() if k.startswith("valu_")), dict(tot))


if __name__ == "__main__":
    main()
