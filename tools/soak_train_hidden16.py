import sys, numpy as np
sys.path.insert(0, ".")
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(g.calibrate_n_points(50_000)); sc = SimplicialComplex(cx)
N = 600
paths = g.generate_random_walks(cx, m=N, seed=5, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=9)
y = np.zeros((N, sc.max_degree, 1)); y[np.arange(N), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
train = np.array([1] * 480 + [0] * 120); test = 1 - train
n_nbrs = sc.n_nbrs(last)
res = {}
for mode in ("dense", "field"):
    stm.reseed(1030)
    net = stm.Scone_GCN(3, 5e-3, 100, 5e-5, verbose=False, skip_mode=mode)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, train, model_type="scone")
    out = net.train(inputs, y, train, test, n_nbrs)
    res[mode] = (out, [w.clone() for w in net.weights])
    print(mode, "train loss/acc, test loss/acc:", ["%.6f" % float(v) for v in out], flush=True)
d = max(float((a - b).abs().max()) for a, b in zip(res["field"][1], res["dense"][1]))
print("max |w_field - w_dense| after 3 epochs: %.3e" % d)
assert d < 1e-4 and all(abs(float(a) - float(b)) < 1e-4 for a, b in zip(res["field"][0], res["dense"][0]))
print("train() soak ok")
