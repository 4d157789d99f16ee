#!/usr/bin/env python3
"""Run each hot kernel a few times on the |E|~1M complex (for rocprofv3 --kernel-trace / --pmc passes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te   # noqa: E402
from scone_gcn_amd.complex import SimplicialComplex                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--hidden", type=int, default=32)
ap.add_argument("--slabs", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--which", default="spmm,fwd,bwd,fwd1,bwd1")
ap.add_argument("--act", default="tanh", help="activation of the timed layers (tanh | relu | leaky_relu | none)")
a = ap.parse_args()
cx = g.random_SC_graph(g.calibrate_n_points(a.edges))
sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "scone")
plan = ops.get_scone_plan(shifts[0], shifts[1], readout, a.act, ops.default_device())
print("plan blocks, sources/row:", plan.conv.plan_info(), flush=True)
E, C, S = cx.n_edges, a.hidden, a.slabs
torch.manual_seed(0)
dev = "cuda"
W = [torch.randn(C, C, device=dev) * 0.1 for _ in range(3)]
W1 = [torch.randn(1, C, device=dev) * 0.1 for _ in range(3)]
x = torch.randn(S, E, 4, C, device=dev)
x1 = torch.randn(S, E, 4, 1, device=dev)
aux = torch.tanh(torch.randn(S, E, 4, C, device=dev)) if ("bwd" in a.which.split(",") or "bwdf" in a.which.split(",")) else x   # a layer output, distinct from dz
which = a.which.split(",")
yrec = torch.randn(S, E, 4, 4, device=dev) if "bwdf" in which else None
def one_pass():
    if "spmm" in which:
        plan.conv.spmm_dual(x.view(S, E, 4 * C))
    if "fwd" in which:
        plan.conv.forward([x], W, C, a.act)
    if "fwd1" in which:
        plan.conv.forward([x1], W1, C, a.act)
    if "bwd" in which:
        plan.conv.backward([x], W, aux, a.act, True, [torch.zeros_like(w) for w in W])
    if "bwd1" in which:
        plan.conv.backward([x], W1, x1, a.act, False, [torch.zeros_like(w) for w in W1])
    if "bwdf" in which:             # the layer after the first one: fused with the first layer's weight gradient
        assert plan.conv.backward_fused_first(x, W, aux, a.act, yrec, [torch.zeros_like(w) for w in W], [torch.zeros_like(w) for w in W1])
    if "dwf" in which:
        assert plan.conv.dw_first(x1, None, x, [torch.zeros_like(w) for w in W1])


if "fwd" in which:              # checksums: A/B builds of a kernel must agree on these
    print("checksum fwd %.9e" % float(plan.conv.forward([x], W, C, a.act).double().abs().sum()), flush=True)
if "bwd" in which:
    dWs = [torch.zeros_like(w) for w in W]
    dxx = plan.conv.backward([x], W, aux, a.act, True, dWs)
    print("checksum bwd %.9e %.9e" % (float(dxx.double().abs().sum()), float(sum(d.double().abs().sum() for d in dWs))), flush=True)
one_pass()                      # untimed warm-up (first launches set function attributes, fault in pages)
torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    for _ in range(a.reps):
        one_pass()
for k, (n, ms) in kt.summary().items():
    print(k, n, "%.3f ms" % ms, flush=True)
