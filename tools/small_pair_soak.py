#!/usr/bin/env python3
"""Soak of the paired one-launch step (two workgroups per trajectory, hand-over flags, scn_small_step_adam): optimiser steps on the
400-point complex (|E| = 1001) with batches of changing size -- each size its own staged buffers and captured graph, visited in a
shuffled order so that launches of different grids alternate on the same flag words -- and a check after every step that the loss is
finite (a hand-over that timed out would make it NaN) plus, at the end, that a replay of the first step's batch still gives the
first net's bits on a fresh net.  usage: python tools/small_pair_soak.py [steps=20000] [eager]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
steps = int(next((v for v in sys.argv[1:] if v.isdigit()), 20000))
eager = "eager" in sys.argv                                     # plain launches: the workspaces of different batch sizes share memory blocks
cx = g.random_SC_graph(400); sc = SimplicialComplex(cx)
M = 128
paths = g.generate_random_walks(cx, m=M, seed=1)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=1)
y = np.zeros((M, sc.max_degree, 1)); y[np.arange(M), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]


def make():
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, M, 5e-5, verbose=False)
    net.GRAPH_CACHE = 16
    net.use_graph = not eager
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(M, int), model_type="scone")
    return net


net = make()
sizes = [1, 3, 4, 17, 32, 33, 64, 100, 127, 128]
staged = {n: net.stage(inputs, y, np.arange(n)) for n in sizes}
rs = np.random.RandomState(0)
order = rs.choice(sizes, size=steps)
t0 = time.perf_counter()
losses = []
for i, n in enumerate(order):
    losses.append(net.grad_step_staged(inputs, staged[int(n)], int(n)))
    if i % 256 == 255:
        vals = torch.stack(losses).cpu().numpy(); losses = []
        assert np.isfinite(vals).all(), "step %d: loss %r" % (i, vals)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
w = net._flat_w.cpu().numpy()
assert np.isfinite(w).all()
print("%d optimiser steps over batch sizes %s in shuffled order: every loss finite, %.3f ms per step, %d graphs" % (steps, sizes, dt / steps * 1e3, len(net._graphs)))
# determinism across the soak: the same sequence on a fresh net gives the same weights bit for bit
net2 = make()
staged2 = {n: net2.stage(inputs, y, np.arange(n)) for n in sizes}
for n in order:
    net2.grad_step_staged(inputs, staged2[int(n)], int(n))
torch.cuda.synchronize()
same = np.array_equal(net2._flat_w.cpu().numpy(), w)
print("the same %d steps on a fresh net: weights %s" % (steps, "identical bit for bit" if same else "DIFFER"))
assert same
