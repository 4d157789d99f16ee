#!/usr/bin/env python3
"""Run the fused Bunch layer kernels (scn_terms_forward / _backward / _backward_fused_first) a few times on dense random slabs of
the |E|~1M complex (for rocprofv3 --kernel-trace / --pmc passes, tools/pmc_run.sh): 16 slabs = 64 trajectories, hidden 32 -- the
launch shape of BASELINE configs[4]."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te   # noqa: E402
from scone_gcn_amd.complex import SimplicialComplex                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--slabs", type=int, default=16)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--which", default="fwd_nf,bwd_nf",
                help="fwd / bwd / bwdf: all three levels in and out; fwd_nf / bwd_nf: the launches of the model's third layer (no face output, no face gradient)")
a = ap.parse_args()
cx = g.random_SC_graph(g.calibrate_n_points(a.edges))
sc = SimplicialComplex(cx)
shifts, nbr, _ = te.setup_from_complex(sc, "bunch")
plan = ops.get_bunch_plan(shifts, nbr, ops.default_device())
fwd, bwd = plan._terms_ops()
print("terms plan: fwd blocks %d src/row %.2f, bwd blocks %d src/row %.2f, rows %d" % (*fwd.plan_info(), *bwd.plan_info(), sum(plan.sizes)),
      flush=True)
S, sizes = a.slabs, plan.sizes
SRC, DST = ops.BUNCH_SRC, ops.BUNCH_DST
torch.manual_seed(0)
xs = [torch.randn((S, n, 4, 32), device="cuda") for n in sizes]
auxs = [torch.relu(torch.randn((S, n, 4, 32), device="cuda")) for n in sizes]
ys = [torch.randn((S, n, 4, 1), device="cuda") for n in sizes]
Wf = [[None] * 3 for _ in range(3)]
Wb = [[None] * 3 for _ in range(3)]
dWb = [[None] * 3 for _ in range(3)]
for k in range(7):
    w = torch.randn(32, 32, device="cuda") * 0.1
    Wf[DST[k]][SRC[k]] = w
    Wb[SRC[k]][DST[k]] = w
    dWb[SRC[k]][DST[k]] = torch.zeros(32, 32, device="cuda")
dWf = [torch.zeros(1, 32, device="cuda") for _ in range(3)]
which = a.which.split(",")


def one_pass():
    if "fwd" in which:
        fwd.forward(xs, Wf, "relu", [True] * 3)
    if "bwd" in which:
        ops._terms_backward(bwd, xs, Wb, auxs, "relu", [True] * 3, dWb)
    if "fwd_nf" in which:               # the layer before the last one: its face output is never asked for (own plan)
        want = [True, True, False]
        Wn = [[Wf[l][j] if l != 2 else None for j in range(3)] for l in range(3)]
        plan._terms_fwd_for(want).forward(xs, Wn, "relu", want)
    if "bwd_nf" in which:               # ... and the faces carry no gradient back into it
        dzs = [xs[0], xs[1], None]
        Wn = [[Wb[a][b] if b != 2 else None for b in range(3)] for a in range(3)]
        dWn = [[dWb[a][b] if b != 2 else None for b in range(3)] for a in range(3)]
        ops._terms_backward(plan._terms_bwd_for([True, True, False]), dzs, Wn, auxs, "relu", [True] * 3, dWn)
    if "bwdf" in which:
        ops._terms_backward_first(bwd, xs, Wb, auxs, "relu", ys, dWb, dWf)


one_pass()
torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    for _ in range(a.reps):
        one_pass()
for k, r in kt.table().items():
    print(k, r["launches"], "%.3f ms" % r["avg_ms"], "alg %.2f GB" % (r["alg_bytes"] / 1e9), "%.0f GB/s" % r["GB/s"], flush=True)
