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
        sh = so.scone_shifts(B1, B2) if model == "scone" else so.ebli_shifts(B1, B2)
        act = "tanh" if model == "scone" else "leaky_relu"
        Bc = so.make_Bconds(B1, nb)
        ref_loss, ref_g = so.scone_loss_and_grad(w, sh[0], sh[1], Bc, last, X, y, mask, 0.0, act)
        ref_out = so.scone_forward(w, sh[0], sh[1], Bc, last, X, act)
        shifts, operand, _ = te.setup_from_complex(sc, model)
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
