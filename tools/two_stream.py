#!/usr/bin/env python3
"""Experiment: do the tails of the step's kernels overlap when independent micro-batches run on TWO streams?  Two trainers share
the plan (operators) and split the micro-batches of one batch of 1024 trajectories; timed sequentially on one stream and
interleaved on two.  (Every kernel holds a CU's whole LDS, so a second kernel only gets CUs the first one has left: tail filling.)
    python tools/two_stream.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
B = 1024
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
nets, stageds = [], []
for h in range(2):
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
    nets.append(net)
    stageds.append(net.stage(inputs, y, np.arange(h * (B // 2), (h + 1) * (B // 2))))
plan = nets[0]._plan(inputs)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def seq():
    for h in range(2):
        nets[h]._flat_g.zero_()
        nets[h]._accumulate_staged(plan, stageds[h], B)

def par():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    its = [iter(stageds[0]), iter(stageds[1])]
    for h in range(2):
        with torch.cuda.stream(streams[h]):
            nets[h]._flat_g.zero_()
    for mb0, mb1 in zip(stageds[0], stageds[1]):          # interleave the launches of the two halves
        with torch.cuda.stream(streams[0]):
            nets[0]._accumulate_staged(plan, [mb0], B)
        with torch.cuda.stream(streams[1]):
            nets[1]._accumulate_staged(plan, [mb1], B)
    for s in streams:
        cur.wait_stream(s)

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-12s %.1f ms per %d trajectories = %.0f trajectories/s (peak memory %.0f GB)" % (name, dt * 1e3, B, B / dt, torch.cuda.max_memory_allocated() / 1e9), flush=True)
