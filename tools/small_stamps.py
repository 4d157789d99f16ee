#!/usr/bin/env python3
"""Where the one-launch small step spends its time: wall-clock stamps (100 MHz) of workgroup 0 at the phase boundaries of
small_step_kernel, from the diagnostic build (tools/build_stamps.sh -> tools/ubench/libscone_hip_stamps.so).
usage: SCN_LIB_PATH=tools/ubench/libscone_hip_stamps.so python3 tools/small_stamps.py [n_points=400] [n_traj=100]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import _lib, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
pts = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cx = g.random_SC_graph(pts); sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=N, seed=1)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=1)
y = np.zeros((N, sc.max_degree, 1)); y[np.arange(N), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False)
net.use_graph = False
net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
staged = net.stage(inputs, y, np.arange(N))
lib = _lib.load()
if os.environ.get("SCN_SMALL_PAIRING") == "0":
    lib.scn_small_step_pairing(1)
lib.scn_debug_small_stamps.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 16)()
names = {0: "requests issued; x, W1 -> LDS", 1: "layer 1: y = (x, S_lo x, S_up x), H1", 3: "layer 2", 4: "layer 3 (or: readout, if 2 layers)",
         5: "readout + cross-entropy (wave 0) | zero dz", 9: "scatter dH, dz = dH act'(H)", 10: "backward of the last layer",
         11: "backward of the layer before (+ dW1 if it is layer 2)", 12: "backward", 13: "backward"}
for rep in range(3):
    net.grad_step_staged(inputs, staged, N, apply=False)
    torch.cuda.synchronize()
    assert lib.scn_debug_small_stamps(out) == 0
    st = [int(v) for v in out]
    idx = [k for k in range(14) if st[k]]
    print("   s_memtime counter: %.0f ticks per us of s_memrealtime" % ((st[15] - st[14]) / ((st[idx[-1]] - st[idx[0]]) / 100.0)))
    print("|E| = %d, %d trajectories, run %d: total %.2f us" % (cx.n_edges, N, rep, (st[idx[-1]] - st[idx[0]]) / 100.0))
    cyc = (ctypes.c_ulonglong * 16)()
    if lib.scn_debug_small_cycles(cyc) == 0:
        c = [int(v) for v in cyc]
        print("   tile 1 of wave 0, layer 2 (shader cycles, every phase drained): gather %d, 12 MFMA %d, activation %d, stores %d%s"
              % (c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], "; whole next tile %d" % (c[5] - c[4]) if c[5] > c[4] else ""))
    for a, b in zip(idx[:-1], idx[1:]):
        print("   %6.2f us  %s" % ((st[b] - st[a]) / 100.0, names.get(a, "")))
