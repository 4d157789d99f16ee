"""BASELINE configs[0] (the reference's own case: 400 points, |E|=1001, 1000 trajectories, hidden 16, batch 100): wall-clock
per optimiser step and per epoch of Scone_GCN.train(), with the share the GPU is busy (launch-bound regime)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(400); sc = SimplicialComplex(cx)
N = 1000
paths = g.generate_random_walks(cx, m=N, seed=1)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=1)
y = np.zeros((N, sc.max_degree, 1)); y[np.arange(N), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
train = np.array([1] * 800 + [0] * 200); test = 1 - train
for mode in [a for a in sys.argv[1:] if a != "breakdown"] or ["dense"]:
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, 100, 5e-5, verbose=False, skip_mode=mode)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, train, model_type="scone")
    rng = np.random.RandomState(0)
    def step():
        m = np.array([1] * 100 + [0] * (N - 100)); rng.shuffle(m)
        net.grad_step(inputs, y, np.logical_and(m, train))
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    # host-side breakdown of the graph-replayed step (wall-clock of the Python calls, no device synchronisation inside)
    if net._graphs and "breakdown" in sys.argv:
        import collections
        acc = collections.defaultdict(float)
        def wrap(obj, name, label):
            f = getattr(obj, name)
            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); acc[label] += time.perf_counter() - t; return r
            setattr(obj, name, g)
        st = next(iter(net._static.values()))
        wrap(st, "load", "stage.load (numpy + one H2D copy)")
        wrap(net, "_graph_accumulate", "graph replay")
        wrap(net, "_adam", "adam launch")
        wrap(net, "_plan", "plan lookup")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print("%s: 200 steps: host loop %.3f ms/step, with final sync %.3f ms/step" % (mode, t_host / 200 * 1e3, t_all / 200 * 1e3))
        for k, v in acc.items():
            print("    %-40s %.1f us/step" % (k, v / 200 * 1e6))
        t0 = time.perf_counter()
        for _ in range(100):
            step(); torch.cuda.synchronize()
        print("    step + synchronize (serial latency): %.3f ms" % ((time.perf_counter() - t0) / 100 * 1e3), flush=True)
    with ops.KernelTimer() as kt:
        for _ in range(50): step()
    gpu = sum(n * ms for n, ms in kt.summary().values()) / 50
    print("%s: %.3f ms/step wall, %.3f ms/step in scn kernels, |E|=%d" % (mode, dt * 1e3, gpu, cx.n_edges), flush=True)
    for rep in range(2):
        t0 = time.perf_counter(); net.train(inputs, y, train, test, sc.n_nbrs(last)); torch.cuda.synchronize()
        print("%s: one epoch (8 steps + train/test loss and accuracy): %.1f ms" % (mode, (time.perf_counter() - t0) * 1e3), flush=True)
    t0 = time.perf_counter(); net.loss(net.weights, inputs, y, train); t1 = time.perf_counter(); net.accuracy(shifts, inputs, y, train, sc.n_nbrs(last)); t2 = time.perf_counter()
    print("%s: loss(train) %.1f ms, accuracy(train) %.1f ms" % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
