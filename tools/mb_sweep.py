#!/usr/bin/env python3
"""Headline step (|E|~1M, hidden 32) at different micro-batch sizes: trajectories/s of grad_step_staged on 1024 resident trajectories.
    python tools/mb_sweep.py 128 256 512"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
B = 1024
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
orig = ops.micro_batch_size
for mb in sizes:
    ops.micro_batch_size = lambda *a, **k: mb
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
    staged = net.stage(inputs, y, np.arange(B))
    net.grad_step_staged(inputs, staged, B); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        net.grad_step_staged(inputs, staged, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("micro-batch %d: %d launches of %d slabs, %.1f ms/step, %.0f trajectories/s, peak memory %.1f GB"
          % (mb, len(staged), staged[0][0].shape[0], dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del net, staged
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
