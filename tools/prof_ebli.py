#!/usr/bin/env python3
"""Run the Ebli (SNN) power kernels -- scn_spmm_dual (single operator), scn_conv_forward_power, scn_conv_backward_power -- a few
times on dense random slabs of the |E|~1M complex (for rocprofv3 --kernel-trace / --pmc passes, tools/pmc_run.sh)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te   # noqa: E402
from scone_gcn_amd.complex import SimplicialComplex                                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--slabs", type=int, default=32)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
cx = g.random_SC_graph(g.calibrate_n_points(a.edges))
sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "ebli")
plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "leaky_relu", ops.default_device())
assert isinstance(plan, ops.PowerPlan), type(plan)
E, C, S = cx.n_edges, 32, a.slabs
torch.manual_seed(0)
W = [torch.randn(C, C, device="cuda") * 0.1 for _ in range(3)]
x = torch.randn(S, E, 4, C, device="cuda")
aux = torch.randn(S, E, 4, C, device="cuda")


def one_pass():
    g1 = plan._shift(plan.op, x)
    plan.op.forward_power(x, g1, W, "leaky_relu")
    plan.op_T.backward_power(x, g1, W, aux, "leaky_relu", True, [torch.zeros_like(w) for w in W])


one_pass()
torch.cuda.synchronize()
with ops.KernelTimer() as kt:
    for _ in range(a.reps):
        one_pass()
for k, r in kt.table().items():
    print(k, r["launches"], "%.3f ms" % r["avg_ms"], "alg %.2f GB" % (r["alg_bytes"] / 1e9), "%.0f GB/s" % r["GB/s"], flush=True)
