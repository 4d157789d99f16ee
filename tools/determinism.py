#!/usr/bin/env python3
"""Claim check: the gradient step is bitwise reproducible (per-workgroup partials, fixed-order reductions, no float atomics on
the path except the input scatter).  Runs the step of every model R times on the |E|~1M complex and compares loss and the flat
gradient buffer bit for bit.    python tools/determinism.py [repeats]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
ok = True
for model, hidden, B in (("scone", 32, 256), ("scone", 16, 256), ("ebli", 32, 128), ("bunch", 32, 64), ("scone", 64, 64)):
    paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
    y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
    shifts, operand, _ = te.setup_from_complex(sc, model)
    inputs = [operand, last, flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    net.setup(te.MODEL_FUNCS[model], [(7 if model == "bunch" else 3, hidden)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type=model)
    for w in net.weights:
        w.mul_(10.0)
    staged = net.stage(inputs, y, np.arange(B))
    ref = None
    same = True
    for r in range(R):
        loss = net.grad_step_staged(inputs, staged, B, apply=False).detach().clone()
        gsnap = net._flat_g.detach().clone()
        torch.cuda.synchronize()
        if ref is None:
            ref = (loss, gsnap)
        else:
            same = same and bool(torch.equal(loss, ref[0])) and bool(torch.equal(gsnap, ref[1]))
    print("%-6s hidden %-3d batch %-4d: %d repeats bitwise identical: %s   (loss %.9f, |g|_1 %.6e)"
          % (model, hidden, B, R, same, float(ref[0]), float(ref[1].abs().sum())), flush=True)
    ok = ok and same
    del net, staged
    torch.cuda.empty_cache()
sys.exit(0 if ok else 1)
