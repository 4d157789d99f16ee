"""Seeded random cases of the whole path against the oracle: random small complexes (with and without holes, so isolated nodes
and blocks without sources occur), all three models, hidden widths on every code path (zero-padded promotion, slab pairs, the
32-channel-block decomposition, the generic kernels), batch sizes that are not multiples of the slab, masks, activations as the
models fix them (TE:137-203; loss STM:42-56).  Tolerance: the north_star's 1e-5 (relative to max(1, |reference|))."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import scone_oracle as so

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _maxdiff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


CASES = [(seed, model) for seed in range(14) for model in ("scone", "ebli", "bunch")]


@pytest.mark.parametrize("seed,model", CASES)
def test_random_case_matches_oracle(seed, model):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    rs = np.random.RandomState(1000 * seed + {"scone": 1, "ebli": 2, "bunch": 3}[model])
    n_pts = int(rs.randint(80, 260))                    # (the generator's point set is the reference's seed-1 stream: the size varies the complex)
    cx = g.random_SC_graph(n_pts, holes=bool(rs.randint(2)))
    sc = SimplicialComplex(cx)
    n = int(rs.choice([1, 3, 5, 8, 13]))
    paths = g.generate_random_walks(cx, m=n, seed=int(rs.randint(1 << 20)))
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=int(rs.randint(1 << 20)))
    n = len(paths)
    X = flows.todense()[:n].astype(np.float64)
    last = np.asarray(last[:n])
    D = sc.max_degree
    y = so.onehot_targets(np.asarray(choice[:n]), D)
    mask = (rs.rand(n) < 0.8).astype(int)
    mask[0] = 1
    if model == "bunch":
        widths = [int(rs.choice([4, 8, 16, 32, 40, 64]))] * int(rs.choice([2, 3]))
        if rs.randint(2):
            widths[-1] = int(rs.choice([8, 16, 32, 48]))
        layers = [(7, c) for c in widths]
        w = [0.3 * rs.randn(*s) for s in so.weight_shapes(1, layers, 1, "bunch")]
    else:
        widths = [int(rs.choice([4, 8, 16, 24, 32, 48, 64]))] * int(rs.choice([2, 3]))
        if rs.randint(2):
            widths[int(rs.randint(len(widths)))] = int(rs.choice([8, 16, 32, 64]))
        layers = [(3, c) for c in widths]
        w = [(0.2 if model == "scone" else 0.04) * rs.randn(*s) for s in so.weight_shapes(1, layers, 1)]
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    nb, D2 = so.neighborhoods(cx.edges, cx.n_nodes)
    assert D2 == D
    if model == "bunch":
        S = [m.tocsr() for m in compute_shift_matrices(*g.incidence_matrices(cx))]
        ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, mask, 0.0)
        ref_out = so.bunch_forward(w, S, sc.nbrhoods, last, X)
        shifts, operand, _ = te.setup_from_complex(sc, "bunch")
    else:
        # every other case under random edge-orientation flips (-flip_edges, TE:214-219: F L F, B1 F, X F)
        flips = rs.choice([1.0, -1.0], size=cx.n_edges, p=[0.8, 0.2]) if seed % 2 else None
        F = np.diag(flips) if flips is not None else None
        sh = so.scone_shifts(B1, B2, F) if model == "scone" else so.ebli_shifts(B1, B2, F)
        act = "tanh" if model == "scone" else "leaky_relu"
        Bc = so.make_Bconds(B1, nb, F)
        Xf = X * flips[None, :, None] if flips is not None else X
        ref_loss, ref_g = so.scone_loss_and_grad(w, sh[0], sh[1], Bc, last, Xf, y, mask, 0.0, act)
        ref_out = so.scone_forward(w, sh[0], sh[1], Bc, last, Xf, act)
        shifts, operand, _ = te.setup_from_complex(sc, model, flips=flips)
        X = te.apply_flips(X, flips)
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.MODEL_FUNCS[model](wt, *shifts, operand, last, X)
    m = torch.as_tensor(mask, device="cuda").bool()
    yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
    loss = -(out[m] * yt[m]).sum() / m.sum()
    loss.backward()
    what = "%s %s n_pts %d E %d batch %d" % (model, widths, n_pts, cx.n_edges, n)
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL, what
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss)), what
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "%s: weight %d" % (what, k)


@pytest.mark.parametrize("seed", range(8))
def test_random_zero_skipping_step_equals_dense(seed):
    """Exact zero-skipping (work lists) == the dense step on random complexes, widths, batch sizes and modes: loss and every
    weight gradient (to the summation order of the weight-gradient partials), twice in a row (the pooled buffers must come back
    all-zero), hidden 16 (slab pairs) and 32."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    rs = np.random.RandomState(77 + seed)
    cx = g.random_SC_graph(int(rs.randint(2500, 9000)), holes=bool(rs.randint(2)))
    sc = SimplicialComplex(cx)
    N = int(rs.choice([3, 6, 10, 17, 30]))
    paths = g.generate_random_walks(cx, m=N, seed=int(rs.randint(1 << 20)), waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=int(rs.randint(1 << 20)))
    N = len(paths)
    y = so.onehot_targets(np.asarray(choice[:N]), sc.max_degree)
    hidden = int(rs.choice([16, 32]))
    mode = str(rs.choice(["zeros", "field"]))
    model = str(rs.choice(["scone", "ebli"])) if int(np.diff(sc.ebli_shifts()[1].csr.indptr).max()) < 128 else "scone"
    shifts, readout, _ = te.setup_from_complex(sc, model)
    inputs = [readout, last, flows]
    res = {}
    for m in ("dense", mode):
        stm.reseed(1030)
        net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False, skip_mode=m)
        net.setup(te.MODEL_FUNCS[model], [(3, hidden)] * int(2 + seed % 2), shifts, inputs, y, None, np.ones(N, int), model_type=model)
        for w in net.weights:
            w.mul_(20.0 if model == "scone" else 3.0)
        staged = net.stage(inputs, y, np.arange(N))
        loss = float(net.grad_step_staged(inputs, staged, N, apply=False))
        g1 = [t.detach().cpu().numpy().astype(np.float64) for t in net._grads]
        assert float(net.grad_step_staged(inputs, staged, N, apply=False)) == loss
        res[m] = (loss, g1, staged[0][3])
    what = "%s hidden %d mode %s E %d batch %d" % (model, hidden, mode, cx.n_edges, N)
    if res[mode][2] is None:
        pytest.skip("work lists not served for " + what)
    assert abs(res[mode][0] - res["dense"][0]) <= 1e-6 * max(1.0, abs(res["dense"][0])), what
    gmax = max(float(np.abs(a).max()) for a in res["dense"][1])
    for a, b in zip(res[mode][1], res["dense"][1]):
        assert float(np.abs(a - b).max()) <= 2e-5 * max(gmax, 1e-30), what


@pytest.mark.parametrize("seed", range(8))
def test_random_trainer_steps_match_oracle_adam(seed):
    """Scone_GCN.grad_step (STM:306-326: masked batch, gradient of the loss with its ridge term, Adam with the (i + 1) bias
    correction) for three consecutive steps on random complexes / models / widths / batch masks, through whatever path the
    trainer picks (HIP-graph replay on fixed staging buffers for these small complexes, eager otherwise), against the oracle's
    gradient + oracle Adam: weights after every step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    rs = np.random.RandomState(4242 + seed)
    model = ["scone", "ebli", "bunch"][seed % 3]
    cx = g.random_SC_graph(int(rs.randint(90, 400)), holes=bool(rs.randint(2)))
    sc = SimplicialComplex(cx)
    N = int(rs.choice([9, 14, 23, 40]))
    paths = g.generate_random_walks(cx, m=N, seed=int(rs.randint(1 << 20)))
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=int(rs.randint(1 << 20)))
    N = len(paths)
    last = np.asarray(last[:N])
    D = sc.max_degree
    y = so.onehot_targets(np.asarray(choice[:N]), D)
    hidden = int(rs.choice([8, 16, 32]))
    k = 7 if model == "bunch" else 3
    layers = [(k, hidden)] * int(rs.choice([2, 3]))
    bs = max(2, N // 2)
    train_mask = (rs.rand(N) < 0.8).astype(int)
    train_mask[:2] = 1
    wd = 5e-5
    shifts, operand, _ = te.setup_from_complex(sc, model)
    inputs = [operand, last, flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, bs, wd, verbose=False)
    net.setup(te.MODEL_FUNCS[model], layers, shifts, inputs, y, None, train_mask, model_type=model)
    for w in net.weights:
        w.mul_(15.0 if model != "ebli" else 3.0)                   # gradients well above rounding
    w0 = [w.detach().cpu().numpy().astype(np.float64) for w in net.weights]
    adam = so.Adam(w0, 1e-3)
    X = flows.todense()[:N].astype(np.float64)
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    nb, _ = so.neighborhoods(cx.edges, cx.n_nodes)
    if model == "bunch":
        S = [m.tocsr() for m in compute_shift_matrices(*g.incidence_matrices(cx))]
        grad = lambda w, bm: so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, bm, wd)[1]
    else:
        sh = so.scone_shifts(B1, B2) if model == "scone" else so.ebli_shifts(B1, B2)
        Bc = so.make_Bconds(B1, nb)
        act = "tanh" if model == "scone" else "leaky_relu"
        grad = lambda w, bm: so.scone_loss_and_grad(w, sh[0], sh[1], Bc, last, X, y, bm, wd, act)[1]
    for i in range(3):
        bm = np.array([1] * bs + [0] * (N - bs))
        rs.shuffle(bm)
        bm = np.logical_and(bm, train_mask).astype(int)
        if bm.sum() == 0:
            bm[0] = 1
        net._step = i
        net.grad_step(inputs, y, bm)
        adam.update(i, grad(adam.x, bm))
        for a, b in zip(net.weights, adam.x):
            assert _maxdiff(a.cpu().numpy(), b) <= 5e-6, "%s %s step %d" % (model, layers, i)


@pytest.mark.parametrize("model", ["scone", "ebli", "bunch"])
def test_complex_without_faces(model):
    """A grid graph: edges, no triangle at all (B2 has no columns, L_up = 0, the Bunch face level is empty) -- the degenerate
    input of TE:240-257 / BMM:71-135.  Forward, loss and every weight gradient against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    n = 7
    idx = np.arange(n * n).reshape(n, n)
    edges = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1), np.stack([idx[:-1].ravel(), idx[1:].ravel()], 1)])
    edges = np.unique(np.sort(edges, axis=1), axis=0)
    xy = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2) / (n - 1.0)
    cx = g.Complex(n_nodes=n * n, edges=edges, faces=np.zeros((0, 3), np.int64), coords=xy)
    sc = SimplicialComplex(cx)
    rs = np.random.RandomState(5)
    N = 6
    X = np.zeros((N, len(edges), 1))
    for i in range(N):
        sel = rs.choice(len(edges), 5, replace=False)
        X[i, sel, 0] = rs.choice([-1.0, 1.0], 5)
    last = rs.randint(0, n * n, N)
    D = sc.max_degree
    nb, D2 = so.neighborhoods(edges, n * n)
    assert D == D2 == 4
    choice = np.array([rs.randint(0, max(1, int((nb[v] >= 0).sum()))) for v in last])
    y = so.onehot_targets(choice, D)
    mask = np.ones(N, int)
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    assert B2.shape == (len(edges), 0)
    if model == "bunch":
        pytest.importorskip("scipy")
        from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
        w = [0.4 * rs.randn(*s) for s in so.weight_shapes(1, [(7, 16)] * 2, 1, "bunch")]
        S = [m.tocsr() for m in compute_shift_matrices(*g.incidence_matrices(cx))]
        ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, mask, 0.0)
        shifts, operand, _ = te.setup_from_complex(sc, "bunch")
    else:
        w = [(0.3 if model == "scone" else 0.05) * rs.randn(*s) for s in so.weight_shapes(1, [(3, 16)] * 2, 1)]
        sh = so.scone_shifts(B1, B2) if model == "scone" else so.ebli_shifts(B1, B2)
        act = "tanh" if model == "scone" else "leaky_relu"
        ref_loss, ref_g = so.scone_loss_and_grad(w, sh[0], sh[1], so.make_Bconds(B1, nb), last, X, y, mask, 0.0, act)
        shifts, operand, _ = te.setup_from_complex(sc, model)
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.MODEL_FUNCS[model](wt, *shifts, operand, last, X)
    loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / N
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


@pytest.mark.parametrize("model", ["scone", "ebli", "bunch"])
@pytest.mark.parametrize("shape", ["triangle", "two_triangles", "path"])
def test_tiny_complexes(model, shape):
    """Complexes smaller than one wave's rows: a single triangle (V = E = 3, F = 1), two triangles sharing an edge (4 / 5 / 2) and
    a three-edge path without faces -- one trajectory and three, every model."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    if shape == "triangle":
        edges, faces, xy = [[0, 1], [0, 2], [1, 2]], [[0, 1, 2]], [[0, 0], [1, 0], [0, 1]]
    elif shape == "two_triangles":
        edges, faces, xy = [[0, 1], [0, 2], [1, 2], [1, 3], [2, 3]], [[0, 1, 2], [1, 2, 3]], [[0, 0], [1, 0], [0, 1], [1, 1]]
    else:
        edges, faces, xy = [[0, 1], [1, 2], [2, 3]], np.zeros((0, 3), np.int64), [[0, 0], [0.3, 0], [0.6, 0], [1, 0]]
    cx = g.Complex(n_nodes=len(xy), edges=np.asarray(edges, np.int64), faces=np.asarray(faces, np.int64).reshape(-1, 3),
                   coords=np.asarray(xy, float))
    sc = SimplicialComplex(cx)
    E, V = len(edges), len(xy)
    rs = np.random.RandomState(len(shape))
    for N in (1, 3):
        X = rs.choice([-1.0, 0.0, 1.0], size=(N, E, 1))
        last = rs.randint(0, V, N)
        nb, D = so.neighborhoods(cx.edges, V)
        choice = np.array([rs.randint(0, max(1, int((nb[v] >= 0).sum()))) for v in last])
        y = so.onehot_targets(choice, D)
        B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
        if model == "bunch":
            w = [0.5 * rs.randn(*s) for s in so.weight_shapes(1, [(7, 8)] * 2, 1, "bunch")]
            S = [m.tocsr() for m in compute_shift_matrices(*g.incidence_matrices(cx))]
            ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, np.ones(N, int), 0.0)
            shifts, operand, _ = te.setup_from_complex(sc, "bunch")
        else:
            w = [(0.5 if model == "scone" else 0.1) * rs.randn(*s) for s in so.weight_shapes(1, [(3, 16)] * 2, 1)]
            sh = so.scone_shifts(B1, B2) if model == "scone" else so.ebli_shifts(B1, B2)
            act = "tanh" if model == "scone" else "leaky_relu"
            ref_loss, ref_g = so.scone_loss_and_grad(w, sh[0], sh[1], so.make_Bconds(B1, nb), last, X, y, np.ones(N, int), 0.0, act)
            shifts, operand, _ = te.setup_from_complex(sc, model)
        wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
        out = te.MODEL_FUNCS[model](wt, *shifts, operand, last, X)
        loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / N
        loss.backward()
        assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss)), (shape, N)
        for k in range(len(w)):
            assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "%s N=%d weight %d" % (shape, N, k)


@pytest.mark.parametrize("model,scale,gtol", [("scone", 4.0, 2e-5), ("scone", 40.0, 2e-4), ("ebli", 2.0, 2e-5), ("bunch", 6.0, 2e-5)])
def test_saturating_and_large_activations(cfg1, model, scale, gtol):
    """Weights far from the 0.01-scale initialisation: tanh deep in saturation (pre-activations of several hundred: the fast tanh
    must not overflow its exponential), leaky_relu / relu with activations of 1e3...1e6 (the exact bf16 split has to carry the
    dynamic range).  Log-probabilities to 2e-5 relative to their magnitude, gradients relative to the largest entry.  At scale 40
    nearly every unit sits at |y| = 1 - O(2^-24): the backward takes tanh' = 1 - y^2 from the fp32 OUTPUT (as jax's tanh gradient
    does in the reference), which is quantised in steps of 2^-23 there, while the fp64 oracle is not -- 4e-5 of the largest gradient
    measured, so that case is held to 2e-4 (finite values and the forward are held to the same bar as everywhere)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.Complex(n_nodes=cfg1["n_nodes"], edges=cfg1["edges"].astype(np.int64), faces=cfg1["faces"].astype(np.int64),
                   coords=cfg1["coords"])
    sc = SimplicialComplex(cx)
    sel = np.arange(500, 512)
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    rs = np.random.RandomState(3)
    if model == "bunch":
        w = [scale * 0.1 * rs.randn(*s) for s in so.weight_shapes(1, [(7, 32)] * 3, 1, "bunch")]
        S = so.bunch_shifts(cfg1["B1"], cfg1["B2"])
        nb, _ = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
        ref_out = so.bunch_forward(w, S, nb, last, X)
        ref_loss, ref_g = so.bunch_loss_and_grad(w, S, nb, last, X, y, np.ones(len(sel), int), 0.0)
        shifts, operand, _ = te.setup_from_complex(sc, "bunch")
    else:
        w = [scale * 0.1 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 32)] * 3, 1)]
        sh = so.scone_shifts(cfg1["B1"], cfg1["B2"]) if model == "scone" else so.ebli_shifts(cfg1["B1"], cfg1["B2"])
        nb, _ = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
        Bc = so.make_Bconds(cfg1["B1"], nb)
        act = "tanh" if model == "scone" else "leaky_relu"
        ref_out = so.scone_forward(w, sh[0], sh[1], Bc, last, X, act)
        ref_loss, ref_g = so.scone_loss_and_grad(w, sh[0], sh[1], Bc, last, X, y, np.ones(len(sel), int), 0.0, act)
        shifts, operand, _ = te.setup_from_complex(sc, model)
    assert np.isfinite(ref_out).all()
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.MODEL_FUNCS[model](wt, *shifts, operand, last, X)
    loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / len(sel)
    loss.backward()
    o = out.detach().cpu().numpy().astype(np.float64)
    assert np.isfinite(o).all()
    assert np.abs(o - ref_out).max() <= 2e-5 * max(1.0, np.abs(ref_out).max()), (np.abs(o - ref_out).max(), np.abs(ref_out).max())
    assert abs(float(loss.detach()) - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss))
    gmax = max(float(np.abs(r).max()) for r in ref_g)
    for k in range(len(w)):
        gk = wt[k].grad.cpu().numpy().astype(np.float64)
        assert np.isfinite(gk).all()
        assert np.abs(gk - ref_g[k]).max() <= gtol * max(1.0, gmax), "weight %d: %g of %g" % (k, np.abs(gk - ref_g[k]).max(), gmax)


@pytest.mark.parametrize("model,hidden", [("scone", 32), ("scone", 16), ("bunch", 32), ("ebli", 8)])
def test_ragged_micro_batches(monkeypatch, model, hidden):
    """A batch cut into several micro-batches with a ragged last one (27 trajectories as 8 + 8 + 8 + 3, the last slab half padding):
    the accumulated loss and weight gradients of Scone_GCN.grad_step_staged equal the one-micro-batch result (summation order) and
    the oracle's (STM:42-56: one mean over the whole batch)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    cx = g.random_SC_graph(2000)
    sc = SimplicialComplex(cx)
    N = 27
    paths = g.generate_random_walks(cx, m=N, seed=9, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=4)
    N = len(paths)
    last = np.asarray(last[:N])
    y = so.onehot_targets(np.asarray(choice[:N]), sc.max_degree)
    k = 7 if model == "bunch" else 3
    shifts, operand, _ = te.setup_from_complex(sc, model)
    inputs = [operand, last, flows]
    res = {}
    real = ops.micro_batch_size
    for name, mb in (("one", None), ("ragged", 8)):
        if mb:
            monkeypatch.setattr(ops, "micro_batch_size", lambda *a, **kw: mb)
        stm.reseed(1030)
        net = stm.Scone_GCN(1, 1e-3, N, 0.0, verbose=False)
        net.setup(te.MODEL_FUNCS[model], [(k, hidden)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type=model)
        for w in net.weights:
            w.mul_(15.0 if model != "ebli" else 3.0)
        staged = net.stage(inputs, y, np.arange(N))
        assert len(staged) == (1 if mb is None else 4)
        loss = float(net.grad_step_staged(inputs, staged, N, apply=False))
        res[name] = (loss, [t.detach().cpu().numpy().astype(np.float64) for t in net._grads],
                     [w.detach().cpu().numpy().astype(np.float64) for w in net.weights])
        monkeypatch.setattr(ops, "micro_batch_size", real)
    gmax = max(float(np.abs(a).max()) for a in res["one"][1])
    assert abs(res["one"][0] - res["ragged"][0]) <= 1e-6 * max(1.0, abs(res["one"][0]))
    for a, b in zip(res["one"][1], res["ragged"][1]):
        assert np.abs(a - b).max() <= 2e-6 * gmax
    w = res["one"][2]
    X = flows.todense()[:N].astype(np.float64)
    B1, B2 = g.incidence_matrices(cx)
    if model == "bunch":
        S = [m.tocsr() for m in compute_shift_matrices(B1, B2)]
        ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, np.ones(N, int), 0.0)
    else:
        import scipy.sparse as sp
        L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
        if model == "ebli":
            L1 = (L_lo + L_up).tocsr()
            L_lo, L_up = L1, (L1 @ L1).tocsr()
        B1x = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()
        Bc = lambda v: B1x[sc.nbrhoods[v]].toarray()
        ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last, X, y, np.ones(N, int), 0.0,
                                                 "tanh" if model == "scone" else "leaky_relu")
    assert abs(res["ragged"][0] - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for a, b in zip(res["ragged"][1], ref_g):
        assert _maxdiff(a, b) <= TOL


@pytest.mark.parametrize("model,hidden", [("scone", 32), ("scone", 16), ("ebli", 32), ("bunch", 32), ("scone", 64)])
def test_gradient_step_is_bitwise_reproducible(model, hidden):
    """Per-workgroup partials and fixed-order reductions: the same step twice gives the same loss and the same flat gradient
    buffer BIT FOR BIT, for every model and for the 32-channel-block path (tools/determinism.py checks it at |E| = 1M)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(8000)
    sc = SimplicialComplex(cx)
    B = 40
    paths = g.generate_random_walks(cx, m=B, seed=11, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=3)
    B = len(paths)
    y = so.onehot_targets(np.asarray(choice[:B]), sc.max_degree)
    shifts, operand, _ = te.setup_from_complex(sc, model)
    inputs = [operand, np.asarray(last[:B]), flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    net.use_graph = False
    net.setup(te.MODEL_FUNCS[model], [(7 if model == "bunch" else 3, hidden)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type=model)
    for w in net.weights:
        w.mul_(10.0)
    staged = net.stage(inputs, y, np.arange(B))
    snaps = []
    for _ in range(3):
        loss = net.grad_step_staged(inputs, staged, B, apply=False).detach().clone()
        snaps.append((loss, net._flat_g.detach().clone()))
    assert float(snaps[0][1].abs().sum()) > 0
    for loss, gsnap in snaps[1:]:
        assert torch.equal(loss, snaps[0][0]) and torch.equal(gsnap, snaps[0][1])
