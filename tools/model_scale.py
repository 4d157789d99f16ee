#!/usr/bin/env python3
"""Gradient step of the Ebli (SNN) or Bunch (SCCONV) model on the |E|~1M synthetic complex (BASELINE configs[4] is the Bunch
one).  Ebli: L1^2 does not fit the block plan there, so the composed plan (ops.PowerPlan: S (S H) on the blocked SpMM + dense
term kernels) carries it; Bunch: seven per-shift SpMM operators + dense term kernels.  Prints step time and kernel split.
    python tools/model_scale.py [ebli|bunch] [edges] [batch] [hidden]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
model = sys.argv[1] if len(sys.argv) > 1 else "ebli"
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H = int(sys.argv[4]) if len(sys.argv) > 4 else 32
t0 = time.perf_counter()
cx = g.random_SC_graph(g.calibrate_n_points(edges)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, model)
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
net.setup(te.MODEL_FUNCS[model], [(7 if model == "bunch" else 3, H)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type=model)
plan = net._plan(inputs)
print("plan:", type(plan).__name__, "setup %.1f s" % (time.perf_counter() - t0), "shift nnz", [s_.csr.nnz for s_ in shifts], flush=True)
staged = net.stage(inputs, y, np.arange(B))
net.grad_step_staged(inputs, staged, B); torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(3):
    net.grad_step_staged(inputs, staged, B)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / 3
print("%s hidden %d, %d trajectories: %.1f ms/step = %.0f trajectories/s" % (model, H, B, dt * 1e3, B / dt), flush=True)
with ops.KernelTimer() as kt:
    net.grad_step_staged(inputs, staged, B)
for k, (n, ms) in kt.summary().items():
    print("  %-28s x%-3d %.3f ms" % (k, n, ms))
