#!/usr/bin/env python3
"""Plan statistics and timing of the fused Bunch layer operator (scn_terms_*) on the |E|~1M complex."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te
from scone_gcn_amd.complex import SimplicialComplex
edges = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cx = g.random_SC_graph(g.calibrate_n_points(edges)); sc = SimplicialComplex(cx)
shifts, nbr, _ = te.setup_from_complex(sc, "bunch")
plan = ops.get_bunch_plan(shifts, nbr, ops.default_device())
fwd = plan._terms_ops()[0]
print("terms fwd plan: blocks %d, sources/row %.2f, rows %d, nnz %d" % (*fwd.plan_info(), sum(plan.sizes), fwd.nnz), flush=True)
for k in range(7):
    print("  shift", k, "blocks, src/row", plan.term_fwd[k].plan_info())
xs = [torch.randn((S, n, 4, 32), device="cuda") for n in plan.sizes]
Ws = [[torch.randn(32, 32, device="cuda") * 0.1 if plan._slot(l, j) is not None else None for j in range(3)] for l in range(3)]
for want in ([True, True, True], [True, False, False], [False, True, False], [False, False, True]):
    fwd.forward(xs, Ws, "relu", want); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): fwd.forward(xs, Ws, "relu", want)
    torch.cuda.synchronize()
    print("want", want, "%.2f ms" % ((time.perf_counter() - t) / 3 * 1e3), flush=True)
