import sys, numpy as np, torch, cProfile, pstats, time
sys.path.insert(0, ".")
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
B = 512
paths = g.generate_random_walks(cx, m=B, seed=1030, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
y = np.zeros((B, sc.max_degree, 1)); y[np.arange(B), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False, skip_mode="field")
net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
t0 = time.perf_counter(); staged = net.stage(inputs, y, np.arange(B)); print("stage %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); staged = net.stage(inputs, y, np.arange(B)); print("stage again %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): net.stage(inputs, y, np.arange(B))
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
for _ in range(3): net.grad_step_staged(inputs, staged, B)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): net.grad_step_staged(inputs, staged, B)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
