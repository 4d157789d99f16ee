#!/usr/bin/env python3
"""Run K gradient steps of the benchmark workload in one execution mode (dense | zeros | field) and nothing else, so that
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this script measure that mode's HBM traffic per trajectory.
    python3 tools/skip_step.py <mode> [steps] [batch]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
mode = sys.argv[1] if len(sys.argv) > 1 else "field"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False, skip_mode=mode)
net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
staged = net.stage(inputs, y, np.arange(B))
for _ in range(steps):
    net.grad_step_staged(inputs, staged, B)
torch.cuda.synchronize()
print("mode %s: %d steps x %d trajectories" % (mode, steps, B), flush=True)
