#!/usr/bin/env python3
"""Headline step (|E|~1M, hidden 32) at different micro-batch sizes: trajectories/s of grad_step_staged on resident trajectories.
    python tools/mb_sweep.py 128 256 512            (scone, 1024 trajectories)
    SCN_MODEL=bunch python tools/mb_sweep.py 64 128 (configs[4]: 7_32_7_32_7_32, 256 trajectories)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
model = os.environ.get("SCN_MODEL", "scone")
B = 1024 if model == "scone" else 256
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, model)
inputs = [readout, last, flows]
orig = ops.micro_batch_size
for mb in sizes:
    ops.micro_batch_size = lambda *a, **k: mb
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    layers = [(3, 32)] * 3 if model != "bunch" else [(7, 32)] * 3
    net.setup(te.MODEL_FUNCS[model], layers, shifts, inputs, y, None, np.ones(B, int), model_type=model)
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
