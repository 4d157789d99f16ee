import sys, numpy as np
sys.path.insert(0, ".")
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
B = 256
paths = g.generate_random_walks(cx, m=B, seed=5, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=9)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
res = {}
for mode in ("dense", "zeros", "field"):
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-2, B, 5e-5, verbose=False, skip_mode=mode)
    net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
    for w in net.weights: w.mul_(10.0)
    losses = []
    for step in range(6):
        idx = np.arange(B) if step % 2 == 0 else np.arange(B)[::-1].copy()      # different staging each step
        staged = net.stage(inputs, y, idx)
        losses.append(float(net.grad_step_staged(inputs, staged, B)))
    res[mode] = (losses, [w.clone() for w in net.weights])
    print(mode, ["%.6f" % l for l in losses], flush=True)
for mode in ("zeros", "field"):
    d = max(float((a - b).abs().max()) for a, b in zip(res[mode][1], res["dense"][1]))
    print(mode, "max |w - w_dense| after 6 Adam steps: %.3e" % d)
    assert d < 5e-5
print("soak ok")
