#!/usr/bin/env python3
"""Ebli (SNN) gradient step on the |E|~1M synthetic complex: L1^2 does not fit the block plan there, so the composed plan
(ops.PowerPlan: S (S H) on the blocked SpMM + dense term kernels) carries it.  Prints step time and kernel split."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
edges = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
t0 = time.perf_counter()
cx = g.random_SC_graph(g.calibrate_n_points(edges)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "ebli")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
net.setup(te.ebli_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="ebli")
plan = net._plan(inputs)
print("plan:", type(plan).__name__, "setup %.1f s" % (time.perf_counter() - t0), "nnz L1 %d, L1^2 %d" % (shifts[0].csr.nnz, shifts[1].csr.nnz), flush=True)
staged = net.stage(inputs, y, np.arange(B))
net.grad_step_staged(inputs, staged, B); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    net.grad_step_staged(inputs, staged, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / 3
print("ebli hidden 32, %d trajectories: %.1f ms/step = %.0f trajectories/s" % (B, dt * 1e3, B / dt), flush=True)
with ops.KernelTimer() as kt:
    net.grad_step_staged(inputs, staged, B)
for k, (n, ms) in kt.summary().items():
    print("  %-28s x%-3d %.3f ms" % (k, n, ms))
