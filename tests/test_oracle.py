"""CPU tests of the oracle itself: pinned against the golden fixtures generated from the reference's NumPy side
(tests/golden/make_golden.py) and against independent derivations (finite differences, torch autograd)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import scone_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_boundary_of_boundary_is_zero_and_laplacians_symmetric(cfg1):
    B1, B2 = cfg1["B1"], cfg1["B2"]
    assert np.abs(B1 @ B2).max() == 0.0
    L_lo, L_up = so.hodge_laplacians(B1, B2)
    assert np.array_equal(L_lo, L_lo.T) and np.array_equal(L_up, L_up.T)
    assert set(np.unique(L_lo)) <= {-1.0, 0.0, 1.0, 2.0}
    assert np.all(np.diag(L_lo) == 2)
    assert np.all((L_up != 0) <= (L_lo != 0))                # pattern(L_upper) is inside pattern(L_lower)


def test_incidence_sign_convention(cfg1):
    B1, edges = cfg1["B1"], cfg1["edges"]
    for e in (0, 17, 500, 1000):
        a, b = edges[e]
        assert a < b and B1[a, e] == -1 and B1[b, e] == 1     # tail -> head, smaller -> larger (SDG:149)
    B2, faces = cfg1["B2"], cfg1["faces"]
    E = {tuple(e): i for i, e in enumerate(map(tuple, edges.tolist()))}
    for f in (0, 100, 648):
        a, b, c = faces[f]
        assert B2[E[(a, b)], f] == 1 and B2[E[(b, c)], f] == 1 and B2[E[(a, c)], f] == -1   # SDG:155-160


def test_bunch_shifts_match_reference_output(cfg1):
    g = np.load(os.path.join(GOLDEN, "cfg1_bunch.npz"))
    S = so.bunch_shifts(cfg1["B1"], cfg1["B2"])
    for name, M in zip(["S_00", "S_10", "S_01", "S_11", "S_21", "S_12", "S_22"], S):
        ref = so.dense_from_coo(g[name + "_row"], g[name + "_col"], g[name + "_val"], g[name + "_shape"])
        assert np.abs(M - ref).max() <= 1e-12, name


def test_tiny4_known_answer_graph():
    """4-node graph of projection_model.test_dataset (PM:128-151): B1/B2 and the seven Bunch shifts."""
    t = np.load(os.path.join(GOLDEN, "tiny4_complex.npz"))
    S = so.bunch_shifts(t["B1"], t["B2"])
    for name, M in zip(["S_00", "S_10", "S_01", "S_11", "S_21", "S_12", "S_22"], S):
        assert np.abs(M - t[name]).max() <= 1e-12, name
    assert np.abs(t["B1"] @ t["B2"]).max() == 0.0
    assert t["faces"].shape == (2, 3)                        # triangles (0,1,2) and (0,2,3)


def test_weights_follow_seed_1030_stream():
    w = so.generate_weights(1, [(3, 16)] * 3, 1)
    assert [x.shape for x in w] == [(1, 16)] * 3 + [(16, 16)] * 6 + [(16, 1)]
    # first draws of RandomState(1030), recorded in SURVEY.md section 5
    assert np.allclose(w[0][0, :4], [-0.02233836, -0.00954615, 0.00552181, -0.01315112], atol=5e-9)
    wb = so.generate_weights(1, [(7, 32)] * 3, 1, "bunch")
    assert len(wb) == 28 and wb[-1].shape == (32, 1) and wb[7].shape == (32, 32)


def _small(cfg1, n=6):
    sel = np.arange(n)
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    return sel, nb, D, cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]


def test_neighbourhood_table(cfg1):
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    assert D == cfg1["D"] == 13
    deg = (nb >= 0).sum(1)
    assert (deg == 0).sum() == 49                            # the isolated hole nodes
    v = int(np.argmax(deg))
    assert list(nb[v, :deg[v]]) == sorted(nb[v, :deg[v]]) and np.all(nb[v, deg[v]:] == -1)
    # targets are one-hot over the sorted neighbours of the last node and point at the recorded target node
    for i in range(50):
        assert nb[cfg1["last_nodes"][i], cfg1["target_choice"][i]] == cfg1["target_nodes"][i]


@pytest.mark.parametrize("model", ["scone", "ebli"])
def test_scone_gradient_finite_differences(cfg1, model):
    sel, nb, D, X, y, last = _small(cfg1)
    B1, B2 = cfg1["B1"], cfg1["B2"]
    shifts = so.scone_shifts(B1, B2) if model == "scone" else so.ebli_shifts(B1, B2)
    act = "tanh" if model == "scone" else "leaky_relu"
    rs = np.random.RandomState(0)
    scale = 0.3 if model == "scone" else 0.03
    w = [scale * rs.randn(*s) for s in so.weight_shapes(1, [(3, 5)] * 3, 1)]
    Bc = so.make_Bconds(B1, nb)
    mask = np.array([1, 1, 0, 1, 1, 1])
    loss, g = so.scone_loss_and_grad(w, shifts[0], shifts[1], Bc, last, X, y, mask, 5e-5, act)

    def f(ww):
        out = so.scone_forward(ww, shifts[0], shifts[1], Bc, last, X, act)
        return so.loss_from_preds(out, y, mask, ww, 5e-5)
    assert abs(f(w) - loss) < 1e-12
    for k in range(len(w)):
        i = np.unravel_index(rs.randint(w[k].size), w[k].shape)
        h = 1e-6
        wp = [a.copy() for a in w]; wp[k][i] += h
        wm = [a.copy() for a in w]; wm[k][i] -= h
        fd = (f(wp) - f(wm)) / (2 * h)
        assert abs(fd - g[k][i]) <= 1e-6 * max(1.0, abs(fd)), (k, fd, g[k][i])


def test_bunch_gradient_finite_differences(cfg1):
    sel, nb, D, X, y, last = _small(cfg1)
    S = so.bunch_shifts(cfg1["B1"], cfg1["B2"])
    rs = np.random.RandomState(1)
    w = [0.5 * rs.randn(*s) for s in so.weight_shapes(1, [(7, 4)] * 3, 1, "bunch")]
    mask = np.array([1, 0, 1, 1, 1, 1])
    loss, g = so.bunch_loss_and_grad(w, S, nb, last, X, y, mask, 5e-5)

    def f(ww):
        return so.loss_from_preds(so.bunch_forward(ww, S, nb, last, X), y, mask, ww, 5e-5)
    assert abs(f(w) - loss) < 1e-12
    for k in (0, 1, 3, 8, 10, 15, 22, 27):
        i = np.unravel_index(rs.randint(w[k].size), w[k].shape)
        h = 1e-6
        wp = [a.copy() for a in w]; wp[k][i] += h
        wm = [a.copy() for a in w]; wm[k][i] -= h
        fd = (f(wp) - f(wm)) / (2 * h)
        assert abs(fd - g[k][i]) <= 2e-7, (k, fd, g[k][i])


def test_hand_gradients_match_torch_autograd_dense_restatement(cfg1):
    """oracle/torch_dense.py (op-for-op dense restatement, autograd) vs the hand-derived NumPy backward."""
    torch = pytest.importorskip("torch")
    from oracle import torch_dense as td
    sel, nb, D, X, y, last = _small(cfg1, 5)
    B1, B2 = cfg1["B1"], cfg1["B2"]
    L_lo, L_up = so.scone_shifts(B1, B2)
    rs = np.random.RandomState(3)
    w = [0.3 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 6)] * 2, 1)]
    mask = np.ones(5, int)
    loss, g = so.scone_loss_and_grad(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X, y, mask, 5e-5)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    wt = [t(a).requires_grad_(True) for a in w]
    B1x = t(np.concatenate([B1, np.zeros((1, B1.shape[1]))]))
    out = td.scone_func(wt, t(L_lo), t(L_up), B1x, torch.as_tensor(nb), torch.as_tensor(last), t(X))
    tl = td.loss_fn(out, t(y), torch.as_tensor(mask), wt, 5e-5)
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-10
    for a, b in zip(wt, g):
        assert np.abs(a.grad.numpy() - b).max() < 1e-10
    # bunch
    S = so.bunch_shifts(B1, B2)
    wb = [0.5 * rs.randn(*s) for s in so.weight_shapes(1, [(7, 3)] * 2, 1, "bunch")]
    loss, g = so.bunch_loss_and_grad(wb, S, nb, last, X, y, mask, 0.0)
    wt = [t(a).requires_grad_(True) for a in wb]
    out = td.bunch_func(wt, [t(s) for s in S], torch.as_tensor(nb), torch.as_tensor(last), t(X))
    tl = td.loss_fn(out, t(y), torch.as_tensor(mask), wt, 0.0)
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-10
    for a, b in zip(wt, g):
        assert np.abs(a.grad.numpy() - b).max() < 1e-10


def test_dense_and_csr_shifts_agree(cfg1):
    sel, nb, D, X, y, last = _small(cfg1, 4)
    B1, B2 = cfg1["B1"], cfg1["B2"]
    L_lo, L_up = so.scone_shifts(B1, B2)
    rs = np.random.RandomState(5)
    w = [0.3 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 4)] * 3, 1)]
    Bc = so.make_Bconds(B1, nb)
    mask = np.ones(4, int)
    l1, g1 = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last, X, y, mask, 5e-5)
    l2, g2 = so.scone_loss_and_grad(w, sp.csr_matrix(L_lo), sp.csr_matrix(L_up), Bc, last, X, y, mask, 5e-5)
    assert abs(l1 - l2) < 1e-12
    for a, b in zip(g1, g2):
        assert np.abs(a - b).max() < 1e-12


def test_orientation_flip_invariance(cfg1):
    """-flip_edges (TE:43): log-probabilities are orientation invariant for odd activations, not for leaky_relu."""
    sel, nb, D, X, y, last = _small(cfg1, 5)
    B1, B2 = cfg1["B1"], cfg1["B2"]
    F = so.flip_matrix(cfg1["E"])
    assert 0.15 < (np.diag(F) == -1).mean() < 0.25
    rs = np.random.RandomState(7)
    w = [0.3 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 4)] * 3, 1)]
    XF = X * np.diag(F)[None, :, None]
    for act, invariant in (("tanh", True), ("leaky_relu", False)):
        a = so.scone_forward(w, *so.scone_shifts(B1, B2), so.make_Bconds(B1, nb), last, X, act)
        b = so.scone_forward(w, *so.scone_shifts(B1, B2, F), so.make_Bconds(B1, nb, F), last, XF, act)
        assert (np.abs(a - b).max() < 1e-12) == invariant


def test_zero_support_propagation_and_padding_rows_in_logsumexp(cfg1):
    sel, nb, D, X, y, last = _small(cfg1, 3)
    B1, B2 = cfg1["B1"], cfg1["B2"]
    L_lo, L_up = so.scone_shifts(B1, B2)
    rs = np.random.RandomState(9)
    w = [0.3 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 4)] * 3, 1)]
    H, saved = so.conv_forward(w, L_lo, L_up, X, keep=True)
    supp = [(np.abs(s[0]).sum(-1) > 0).mean() for s in saved] + [(np.abs(H).sum(-1) > 0).mean()]
    assert supp[0] < supp[1] < supp[2] < supp[3] < 0.5       # support grows by one hop per layer
    out = so.scone_forward(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X)
    deg = (nb[last] >= 0).sum(1)
    for i in range(3):
        pad = out[i, deg[i]:, 0]
        assert np.allclose(pad, pad[0]) and np.isclose(np.exp(out[i, :, 0]).sum(), 1.0)   # zero logits stay inside the lse


def test_adam_matches_closed_form_first_step():
    w = [np.array([[1.0, -2.0]])]
    adam = so.Adam(w, 1e-3)
    g = [np.array([[0.5, -0.25]])]
    x = adam.update(0, g)
    # first step of Adam moves every weight by lr * sign(g) (up to eps)
    assert np.allclose(x[0], w[0] - 1e-3 * np.sign(g[0]), atol=1e-9)


def test_draw_batch_mask_follows_reference_stream():
    rs = np.random.RandomState(1030)
    tm = np.array([1] * 8 + [0] * 2)
    bm = so.draw_batch_mask(rs, 10, 4, tm)
    assert bm.dtype == bool and bm.sum() <= 4 and not bm[8:].any()


def test_sparse_fp32_torch_restatement_matches_the_fp64_oracle(cfg1):
    """oracle/torch_sparse.py (the CPU baseline B2 of bench.py: fp32, torch.sparse_csr shifts, autograd) against the fp64
    NumPy oracle with its hand-derived backward."""
    import warnings
    import scipy.sparse as sp
    torch = pytest.importorskip("torch")
    from oracle import torch_sparse as ts
    warnings.filterwarnings("ignore", message="Sparse CSR tensor support is in beta")
    B1, B2 = cfg1["B1"], cfg1["B2"]
    sel = np.arange(10)
    L_lo, L_up = so.scone_shifts(B1, B2)
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    rs = np.random.RandomState(3)
    w = [0.2 * rs.randn(*s) for s in so.weight_shapes(1, [(3, 8)] * 3, 1)]
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X, y, np.ones(10, int), 5e-5)
    Sl, Su = ts.csr_tensor(sp.csr_matrix(L_lo)), ts.csr_tensor(sp.csr_matrix(L_up))
    loss, grads = ts.loss_and_grad([torch.tensor(a, dtype=torch.float32) for a in w], Sl, Su, Sl, Su,
                                   ts.make_inc_rows(sp.csr_matrix(B1), nb), last, torch.tensor(X, dtype=torch.float32),
                                   torch.tensor(y, dtype=torch.float32), 5e-5)
    assert abs(loss - ref_loss) < 1e-5
    for a, b in zip(grads, ref_g):
        assert np.abs(a.numpy() - b).max() < 2e-5 * max(1.0, np.abs(b).max())


def test_two_target_accuracy_restatement(cfg1):
    """STM:73-108 on fixed log-probabilities: the cached random targets differ from the PREDICTED choice afterwards (the
    reference's quirk), ties count one half, and the second call reuses the draw."""
    rs = np.random.RandomState(5)
    N, D = 12, cfg1["D"]
    n_nbrs = rs.randint(2, D, size=N)
    preds = np.log(rs.dirichlet(np.ones(D), size=N))[:, :, None]
    y = so.onehot_targets(rs.randint(0, 2, size=N), D)
    mask = np.array([1, 1, 0, 1, 0, 1, 1, 1, 0, 1, 1, 0])
    acc, rt = so.two_target_accuracy_from_preds(preds, y, mask, n_nbrs, np.random.RandomState(9))
    p = preds.copy()
    for i in range(N):
        p[i, n_nbrs[i]:] = -100
    pc = np.argmax(p[mask == 1], axis=1).reshape(-1)
    for i in range(N):
        assert rt[i] != pc[min(i, len(pc) - 1)] and 0 <= rt[i] < n_nbrs[i]
    t = p[np.arange(N), np.argmax(y, axis=1).reshape(N), 0][mask == 1]
    r = p[np.arange(N), rt, 0][mask == 1]
    assert acc == (np.sum(t > r) + 0.5 * np.sum(t == r)) / mask.sum()
    acc2, rt2 = so.two_target_accuracy_from_preds(preds, y, mask, n_nbrs, np.random.RandomState(1), random_targets=rt.copy())
    assert acc2 == acc and np.array_equal(rt, rt2)
