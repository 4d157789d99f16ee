#!/usr/bin/env python3
"""The launch-amortised small step under a profiler: BASELINE configs[0]-sized problem (|E| = 1001, hidden 16, 100 trajectories
resident on the device), N optimiser steps through Scone_GCN.grad_step_staged -- graph replay (default) or plain launches
(`eager`).  Prints wall-clock per step; under `rocprofv3 --kernel-trace --stats` the kernel table shows what the GPU spends."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
eager = "eager" in sys.argv
if os.environ.get("SCN_SMALL_PAIRING") == "0":                  # A/B: one workgroup per trajectory even where two would be taken
    from scone_gcn_amd import _lib
    _lib.load().scn_small_step_pairing(1)
steps = int(next((a for a in sys.argv[1:] if a.isdigit()), 300))
pts = int(os.environ.get("SCN_POINTS", "400"))                  # 400 points: |E| = 1001 (TE:86-90); 130: |E| ~ 320 (the drifter complex's size)
cx = g.random_SC_graph(pts); sc = SimplicialComplex(cx)
N = int(os.environ.get("SCN_TRAJ", "100"))
M = min(N, 160)                                                 # walks drawn; larger batches repeat them
paths = g.generate_random_walks(cx, m=M, seed=1)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=1)
rep = np.arange(N) % M
flows, choice, last = flows.select(rep), np.asarray(choice)[rep], np.asarray(last)[rep]
y = np.zeros((N, sc.max_degree, 1)); y[np.arange(N), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False)
net.use_graph = not eager
net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
staged = net.stage(inputs, y, np.arange(N))
for _ in range(3):
    net.grad_step_staged(inputs, staged, N)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    net.grad_step_staged(inputs, staged, N)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("%s: %d steps, |E|=%d: host loop %.3f ms/step, to completion %.3f ms/step" % ("eager" if eager else "graph", steps, cx.n_edges,
      t_host / steps * 1e3, t_all / steps * 1e3), flush=True)
