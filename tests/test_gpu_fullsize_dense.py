"""GPU parity AT THE BENCHMARK SIZE (|E| = 996 634) ON DENSE RANDOM SLABS: every kernel bench.py times on dense data -- the ring
dual SpMM, the fused C=32 / paired C=16 forward and backward, the fused-first backward, the Ebli power kernels and the fused
Bunch layer -- against an fp64 scipy-CSR evaluation of the same formula (TE:145-147, TE:183-195) on the same slabs.

The trajectories of tests/test_gpu_fullsize.py touch a few hundred edges of a million; here EVERY row of EVERY plan block
(the ELL-capped ones, the width-sorted row groups, every XCD's share of the assignment tables) produces a non-zero output that
is compared.  Error bars are relative to each output's own sum of |terms| (an fp32 evaluation cannot do better than a few ulp of
that); weight gradients (sums over 8M points) relative to the largest entry of the gradient matrix.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

REL_TERMS = 4e-6       # |err| <= REL_TERMS * sum |terms| of that output (fp32: ~30 terms, three exact-split products each)
REL_DW = 1e-4          # |dW err| <= REL_DW * max |dW| (one missing plan block of 16 505 moves dW by ~1e-2 of it)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _shift(m, xs):
    """(R_dst x R_src) csr @ one slab [R_src, ns, C] -> [R_dst, ns, C], fp64."""
    ns, C = xs.shape[1], xs.shape[2]
    return np.asarray(m @ xs.reshape(xs.shape[0], ns * C)).reshape(m.shape[0], ns, C)


def _assert_close(got, ref, scale, what):
    err = np.abs(got.astype(np.float64) - ref)
    bad = err > REL_TERMS * np.maximum(scale, 1e-3)
    assert not bad.any(), "%s: %d outputs off, worst %.3e of its terms' sum" % (
        what, int(bad.sum()), float((err / np.maximum(scale, 1e-3)).max()))
    assert float(np.abs(ref).max()) > 0.1, what + ": reference is trivially small"


@pytest.fixture(scope="module")
def scone_big(big_complex):
    from scone_gcn_amd import ops, trajectory_experiments as te
    cx, sc = big_complex
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
    lo, up = shifts[0].device_csr().astype(np.float64), shifts[1].device_csr().astype(np.float64)
    nb = plan.conv.plan_info()[0]
    assert nb > 15000                                   # the LDS-blocked plan serves it (16 505 blocks on this complex)
    return cx.n_edges, plan, lo, up


@pytest.mark.parametrize("K,S", [(128, 3), (64, 3), (64, 17)])
def test_dual_spmm_every_row_against_scipy_at_one_million_edges(scone_big, K, S):
    """scn_spmm_dual (the four-stage ring kernel at K = 128 / 64; S = 3: the ring wraps inside one block visit, S = 17: a
    second visit of every block with a single slab) -- [L_low X, L_up X] (TE:146-147), all rows, all columns."""
    E, plan, lo, up = scone_big
    rs = np.random.RandomState(100 + K + S)
    alo, aup = abs(lo), abs(up)
    check = range(S) if S <= 3 else (0, 15, 16)         # fp64 products cost ~2 s per slab: first, last of visit 1, visit 2
    x = rs.randn(S, E, K).astype(np.float32)
    ya, yb = plan.conv.spmm_dual(_t(x))
    for s in check:
        xs = x[s].astype(np.float64)
        _assert_close(ya[s].cpu().numpy(), lo @ xs, alo @ np.abs(xs), "L_low X, K=%d slab %d" % (K, s))
        _assert_close(yb[s].cpu().numpy(), up @ xs, aup @ np.abs(xs), "L_up X, K=%d slab %d" % (K, s))
    ya1, _ = plan.conv.spmm_dual(_t(x[:1]), dual=False)
    assert torch.equal(ya1[0], ya[0])                   # the single-operator form runs the same sums


def _layer_reference(lo, up, x, W, aux=None):
    """g = [x, lo x, up x] per slab; forward z = sum_k g_k W_k; with aux: dx = (sum_k g_k W_k^T) (1 - aux^2), dW_k = aux^T g_k."""
    Wd = [w.astype(np.float64) for w in W]
    out, scale, dW = [], [], [np.zeros(W[0].shape[::-1] if aux is not None else W[0].shape) for _ in W]
    alo, aup = abs(lo), abs(up)
    for s in range(x.shape[0]):
        xs = x[s].astype(np.float64)
        g = [xs, _shift(lo, xs), _shift(up, xs)]
        ga = [np.abs(xs), _shift(alo, np.abs(xs)), _shift(aup, np.abs(xs))]
        if aux is None:
            out.append(sum(gk @ wk for gk, wk in zip(g, Wd)))
            scale.append(sum(gk @ np.abs(wk) for gk, wk in zip(ga, Wd)))
        else:
            a = aux[s].astype(np.float64)
            out.append(sum(gk @ wk.T for gk, wk in zip(g, Wd)) * (1.0 - a ** 2))
            scale.append(sum(gk @ np.abs(wk).T for gk, wk in zip(ga, Wd)))
            for k in range(3):
                dW[k] += np.einsum("rnc,rnd->cd", a, g[k])
    return np.stack(out), np.stack(scale), dW


@pytest.mark.parametrize("C,S", [(32, 2), (16, 3)])
def test_fused_layer_forward_and_backward_on_dense_slabs_at_one_million_edges(scone_big, C, S):
    """scn_conv_forward / scn_conv_backward on the benchmark operator with dense random slabs: fwd_c32_w16 (C = 32), the paired
    fwd_c16_w16 on an odd slab count (C = 16), and bwd_c32_bf16 / its PAIR form: dx and the three weight gradients."""
    E, plan, lo, up = scone_big
    rs = np.random.RandomState(7 + C)
    x = rs.randn(S, E, 4, C).astype(np.float32)
    W = [(0.05 * rs.randn(C, C)).astype(np.float32) for _ in range(3)]
    Wt = [_t(w) for w in W]
    # forward: linear part against the terms' sum, then tanh on top (1-Lipschitz)
    z, zscale, _ = _layer_reference(lo, up, x, W)
    out = plan.conv.forward([_t(x)], Wt, C, "none").cpu().numpy()
    _assert_close(out, z, zscale, "forward C=%d" % C)
    out = plan.conv.forward([_t(x)], Wt, C, "tanh").cpu().numpy()
    _assert_close(out, np.tanh(z), zscale, "forward tanh C=%d" % C)
    del out, z, zscale
    # backward of the same layer: dz := x (any dense tensor), aux = the layer's input as tanh values
    aux = np.tanh(rs.randn(S, E, 4, C)).astype(np.float32)
    dWs = [torch.full((C, C), 0.25, device="cuda") for _ in range(3)]          # accumulated INTO
    dx = plan.conv_T.backward([_t(x)], Wt, _t(aux), "tanh", True, dWs).cpu().numpy()
    refdx, sdx, refdW = _layer_reference(lo, up, x, W, aux)
    _assert_close(dx, refdx, sdx, "backward dx C=%d" % C)
    for k in range(3):
        err = np.abs(dWs[k].cpu().numpy().astype(np.float64) - 0.25 - refdW[k]).max()
        assert err <= REL_DW * np.abs(refdW[k]).max(), "dW%d C=%d: %.3e of max" % (k, C, err / np.abs(refdW[k]).max())


@pytest.mark.parametrize("C,S", [(32, 2), (16, 2)])
def test_fused_first_backward_on_dense_slabs_at_one_million_edges(scone_big, C, S):
    """scn_conv_backward_fused_first (the layer after the first one: its input gradient is contracted with the first layer's
    shifted input y in registers, dW_first[g][c] = sum_p y[p][g] dx[p][c], and never written) against fp64 AND against
    scn_conv_backward + scn_conv_dw_first on the same dense slabs."""
    E, plan, lo, up = scone_big
    rs = np.random.RandomState(70 + C)
    dz = rs.randn(S, E, 4, C).astype(np.float32)
    aux = np.tanh(rs.randn(S, E, 4, C)).astype(np.float32)
    y = rs.randn(S, E, 4, 4).astype(np.float32)
    y[..., 3] = 0.0
    W = [(0.05 * rs.randn(C, C)).astype(np.float32) for _ in range(3)]
    Wt, dzt, auxt, yt = [_t(w) for w in W], _t(dz), _t(aux), _t(y)
    dWs = [torch.zeros((C, C), device="cuda") for _ in range(3)]
    dWf = [torch.zeros((1, C), device="cuda") for _ in range(3)]
    assert plan.conv_T.backward_fused_first(dzt, Wt, auxt, "tanh", yt, dWs, dWf), "fused-first backward not served"
    refdx, _, refdW = _layer_reference(lo, up, dz, W, aux)
    ref_first = np.einsum("srng,srnc->gc", y[..., :3].astype(np.float64), refdx)
    for k in range(3):
        e = np.abs(dWs[k].cpu().numpy() - refdW[k]).max() / np.abs(refdW[k]).max()
        assert e <= REL_DW, "dW%d: %.3e" % (k, e)
        e = np.abs(dWf[k].cpu().numpy()[0] - ref_first[k]).max() / np.abs(ref_first).max()
        assert e <= REL_DW, "dW_first%d: %.3e" % (k, e)
    # the separate entry points on the same data
    dWs2 = [torch.zeros((C, C), device="cuda") for _ in range(3)]
    dWf2 = [torch.zeros((1, C), device="cuda") for _ in range(3)]
    dx = plan.conv_T.backward([dzt], Wt, auxt, "tanh", True, dWs2)
    assert plan.conv.dw_first(None, yt, dx, dWf2)
    for a, b in zip(dWs + dWf, dWs2 + dWf2):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize("C,S", [(32, 2), (16, 3)])
def test_ebli_power_kernels_on_dense_slabs_at_one_million_edges(big_complex, C, S):
    """scn_conv_forward_power / scn_conv_backward_power (Ebli on large complexes, TE:161-167 with L1^2 never formed):
    out = act(x0 W0 + x W1 + (L1 x) W2) and its gradient given g1 = L1^T dz, dense random slabs, C = 32 and C = 16 (the
    slab-pair form; three slabs: one pair and a lone last slab)."""
    from scone_gcn_amd import ops, trajectory_experiments as te
    cx, sc = big_complex
    shifts, readout, _ = te.setup_from_complex(sc, "ebli")
    plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "leaky_relu", ops.default_device())
    assert isinstance(plan, ops.PowerPlan)
    L1 = shifts[0].device_csr().astype(np.float64)
    aL1 = abs(L1)
    E = cx.n_edges
    rs = np.random.RandomState(5)
    x0 = rs.randn(S, E, 4, C).astype(np.float32)
    x = rs.randn(S, E, 4, C).astype(np.float32)
    W = [(0.05 * rs.randn(C, C)).astype(np.float32) for _ in range(3)]
    Wd = [w.astype(np.float64) for w in W]
    Wt = [_t(w) for w in W]
    out = plan.op.forward_power(_t(x0), _t(x), Wt, "leaky_relu")
    assert out is not None, "power forward not served"
    out = out.cpu().numpy()
    for s in range(S):
        a, b = x0[s].astype(np.float64), x[s].astype(np.float64)
        z = a @ Wd[0] + b @ Wd[1] + _shift(L1, b) @ Wd[2]
        sc_ = np.abs(a) @ np.abs(Wd[0]) + np.abs(b) @ np.abs(Wd[1]) + _shift(aL1, np.abs(b)) @ np.abs(Wd[2])
        _assert_close(out[s], np.where(z >= 0, z, 0.01 * z), sc_, "power forward slab %d" % s)
    # backward: dx = (dz W0^T + g1 W1^T + (L1 g1) W2^T) act'(aux), dW = aux^T [dz, g1, L1 g1]   (L1 symmetric)
    dz, g1 = x0, x
    aux = rs.randn(S, E, 4, C).astype(np.float32)
    dWs = [torch.zeros((C, C), device="cuda") for _ in range(3)]
    served, dx = plan.op_T.backward_power(_t(dz), _t(g1), Wt, _t(aux), "leaky_relu", True, dWs)
    assert served
    dx = dx.cpu().numpy()
    refdW = [np.zeros((C, C)) for _ in range(3)]
    for s in range(S):
        a, b, h = dz[s].astype(np.float64), g1[s].astype(np.float64), aux[s].astype(np.float64)
        g = [a, b, _shift(L1, b)]
        ga = [np.abs(a), np.abs(b), _shift(aL1, np.abs(b))]
        ref = sum(gk @ wk.T for gk, wk in zip(g, Wd)) * np.where(h >= 0, 1.0, 0.01)
        _assert_close(dx[s], ref, sum(gk @ np.abs(wk).T for gk, wk in zip(ga, Wd)), "power backward slab %d" % s)
        for k in range(3):
            refdW[k] += np.einsum("rnc,rnd->cd", h, g[k])
    for k in range(3):
        e = np.abs(dWs[k].cpu().numpy() - refdW[k]).max() / np.abs(refdW[k]).max()
        assert e <= REL_DW, "power dW%d: %.3e" % (k, e)


def test_fused_bunch_layer_on_dense_slabs_at_one_million_edges(big_complex):
    """scn_terms_forward / scn_terms_backward on the 1M-edge Bunch operator (R = V + E + F = 2.03 M rows, 53 664 patches across
    the three levels) with dense random slabs: out_l = relu(sum_j (S_{j->l} x_j) W[l][j]) (TE:183-195) for all three levels and,
    on the transposed operator, dx_l and all seven weight gradients."""
    from scone_gcn_amd import ops, trajectory_experiments as te
    cx, sc = big_complex
    shifts, nbr, _ = te.setup_from_complex(sc, "bunch")
    plan = ops.get_bunch_plan(shifts, nbr, ops.default_device())
    fwd, bwd = plan._terms_ops()
    assert fwd.plan_info()[0] > 40000
    S, sizes = 2, plan.sizes
    SRC, DST = ops.BUNCH_SRC, ops.BUNCH_DST
    rs = np.random.RandomState(9)
    dev = [s.device_csr().astype(np.float64) for s in shifts]
    xs = [rs.randn(S, n, 4, 32).astype(np.float32) for n in sizes]
    Wk = [(0.1 * rs.randn(32, 32)).astype(np.float32) for _ in range(7)]
    Wd = [w.astype(np.float64) for w in Wk]
    Ws = [[None] * 3 for _ in range(3)]
    for k in range(7):
        Ws[DST[k]][SRC[k]] = _t(Wk[k])
    outs = fwd.forward([_t(x) for x in xs], Ws, "relu", [True] * 3)
    for l in range(3):
        got = outs[l].cpu().numpy()
        for s in range(S):
            ks = [k for k in range(7) if DST[k] == l]
            ref = sum(_shift(dev[k], xs[SRC[k]][s].astype(np.float64)) @ Wd[k] for k in ks)
            scale = sum(_shift(abs(dev[k]), np.abs(xs[SRC[k]][s]).astype(np.float64)) @ np.abs(Wd[k]) for k in ks)
            _assert_close(got[s], np.maximum(ref, 0), scale, "bunch forward level %d slab %d" % (l, s))
    del outs
    # backward on the transposed operator: rows = the layer's input rows (level a), terms = the levels b they feed
    dzs = xs
    auxs = [np.maximum(rs.randn(S, n, 4, 32), 0).astype(np.float32) for n in sizes]
    Wb = [[None] * 3 for _ in range(3)]
    dWb = [[None] * 3 for _ in range(3)]
    for k in range(7):
        Wb[SRC[k]][DST[k]], dWb[SRC[k]][DST[k]] = _t(Wk[k]), torch.zeros((32, 32), device="cuda")
    dxs = ops._terms_backward(bwd, [_t(d) for d in dzs], Wb, [_t(a) for a in auxs], "relu", [True] * 3, dWb)
    devT = [m.T.tocsr() for m in dev]
    for a in range(3):
        got = dxs[a].cpu().numpy()
        ks = [k for k in range(7) if SRC[k] == a]
        refw = {k: np.zeros((32, 32)) for k in ks}
        for s in range(S):
            gk = {k: _shift(devT[k], dzs[DST[k]][s].astype(np.float64)) for k in ks}
            ref = sum(gk[k] @ Wd[k].T for k in ks) * (auxs[a][s] > 0)
            scale = sum(_shift(abs(devT[k]), np.abs(dzs[DST[k]][s]).astype(np.float64)) @ np.abs(Wd[k]).T for k in ks)
            _assert_close(got[s], ref, scale, "bunch backward level %d slab %d" % (a, s))
            for k in ks:
                refw[k] += np.einsum("rnc,rnd->cd", auxs[a][s].astype(np.float64), gk[k])
        for k in ks:
            e = np.abs(dWb[a][DST[k]].cpu().numpy() - refw[k]).max() / np.abs(refw[k]).max()
            assert e <= REL_DW, "bunch dW slot %d: %.3e" % (k, e)
