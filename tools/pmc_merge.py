#!/usr/bin/env python3
"""Merge tools/pmc_run.sh outputs into profiles/rNN_pmc_traffic.json, keyed the way bench.py looks traffic up (and stamped
with the hash of the kernel sources the passes ran on: bench.py cites the bytes only while the checkout's hash is the same):
    python tools/pmc_merge.py out.json  main=<dir>/pmc.json:"launch text"  "configs[1]"=...  ebli=...  bunch=...
Every kernel row gets `timer_keys`: the ops.KernelTimer families it runs under."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import _lib


def timer_keys(name):
    m = re.match(r"scn::(\w+)(<(.*)>)?", name)
    if not m:
        return []
    k, targs = m.group(1), [t.strip() for t in (m.group(3) or "").split(",") if t.strip()]
    if k == "fwd_c32_w16_kernel":
        return ["conv_fwd_power c32"] if targs[1:2] == ["true"] else ["conv_fwd c32->32"]
    if k == "fwd_c16_w16_kernel":
        return ["conv_fwd c16->16"]
    if k == "fwd_c1_kernel":
        return ["conv_fwd c1->%s" % targs[0]]
    if k == "bwd_c32_bf16_kernel":
        ext0, pair, first = (targs + ["false"] * 4)[1:4]
        if ext0 == "true":
            return ["conv_bwd_power c32"]
        base = "conv_bwd c16->16" if pair == "true" else "conv_bwd c32->32"
        return [base + (" + dW_first" if first == "true" else "")]
    if k == "spmm_ring_kernel":
        kk = 64 * int(targs[1])
        return ["spmm_dual k%d" % kk if targs[0] == "true" else "spmm k%d" % kk]
    if k == "terms_fwd_c32_kernel":
        return ["terms_fwd c32"]
    if k == "terms_bwd_c32_kernel":
        return ["terms_bwd c32" + (" + dW_first" if targs[1:2] == ["true"] else "")]
    return []


out, res = sys.argv[1], {}
for spec in sys.argv[2:]:
    sec, rest = spec.split("=", 1)
    path, _, launch = rest.partition(":")
    d = json.load(open(path))
    res[sec] = {"launch": launch, "source": path,
                "kernels": {k: dict(v, timer_keys=timer_keys(k)) for k, v in d.items() if "hbm_bytes_per_launch" in v}}
res["note"] = ("rocprofv3 --pmc passes (tools/pmc_run.sh: FETCH_SIZE, WRITE_SIZE, TCC hit/miss in separate runs), mean per launch; "
               "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KB: gfx950 reports half of wide coalesced reads "
               "(MI355X_MICROARCH.md, section HBM)")
res["kernel_sources_sha"] = _lib.sources_sha()
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, {s: list(v["kernels"]) for s, v in res.items() if isinstance(v, dict)})
