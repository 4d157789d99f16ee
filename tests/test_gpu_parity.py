"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Tolerance: BASELINE.json's north_star asks for <= 1e-5 max deviation (fp32) on forward and backward outputs.
Inputs are the committed golden fixtures of config 1 (tests/golden) plus seeded synthetic complexes.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import scone_oracle as so

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def sc1(cfg1):
    _need_gpu()
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.synthetic_data_gen import Complex
    cx = Complex(n_nodes=cfg1["n_nodes"], edges=cfg1["edges"].astype(np.int64), faces=cfg1["faces"].astype(np.int64),
                 coords=cfg1["coords"])
    return SimplicialComplex(cx)


def _rand_weights(shapes, scale, seed):
    rs = np.random.RandomState(seed)
    return [scale * rs.randn(*s) for s in shapes]


def _maxdiff(a, b):
    """max |a - b| in units of max(1, max|b|): the 1e-5 bar is absolute for O(1) log-probabilities / gradients and
    relative to the reference's magnitude when that is larger (fp32 carries ~7 digits)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


def _oracle_scone(cfg1, weights, sel, model, flips=None):
    B1, B2 = cfg1["B1"], cfg1["B2"]
    F = np.diag(flips) if flips is not None else None
    shifts = so.scone_shifts(B1, B2, F) if model == "scone" else so.ebli_shifts(B1, B2, F)
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    Bc = so.make_Bconds(B1, nb, F)
    X = cfg1["flows"][sel]
    if flips is not None:
        X = X * flips[None, :, None]
    act = "tanh" if model == "scone" else "leaky_relu"
    return shifts, Bc, X, act


@pytest.mark.parametrize("model,hidden", [("scone", 16), ("scone", 32), ("ebli", 16), ("scone", 8)])
def test_forward_and_gradients_match_oracle(cfg1, sc1, model, hidden):
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(0, 27)            # not a multiple of the slab width on purpose
    shapes = so.weight_shapes(1, [(3, hidden)] * 3, 1)
    w = _rand_weights(shapes, 0.25 if model == "scone" else 0.03, 7)   # ebli: L1^2 entries reach 31, keep outputs O(1)
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, model)
    y, last = cfg1["targets"][sel], cfg1["last_nodes"][sel]
    mask = np.ones(len(sel), int)
    mask[[3, 11]] = 0
    ref_out = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, last, X, act)
    ref_loss, ref_g = so.scone_loss_and_grad(w, shifts_o[0], shifts_o[1], Bc, last, X, y, mask, 0.0, act)

    shifts, readout, _ = te.setup_from_complex(sc1, model)
    fn = te.MODEL_FUNCS[model]
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = fn(wt, *shifts, readout, last, X)
    assert out.shape == (len(sel), cfg1["D"], 1)
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL
    m = torch.as_tensor(mask, device="cuda").bool()
    yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
    loss = -(out[m] * yt[m]).sum() / m.sum()
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


@pytest.mark.parametrize("model,layers", [("scone", [(3, 64)] * 3), ("scone", [(3, 48), (3, 64)]), ("scone", [(3, 64), (3, 32)]),
                                          ("ebli", [(3, 96), (3, 40)]),         # widths above 32: 32-channel blocks (ops.SconePlan._wide_stack)
                                          ("scone", [(3, 32), (3, 16)]), ("scone", [(3, 16), (3, 32)]),
                                          ("ebli", [(3, 32), (3, 16)]), ("ebli", [(3, 16), (3, 32)]),
                                          ("scone", [(3, 8), (3, 8)]), ("scone", [(3, 16), (3, 24), (3, 8)])])
def test_reference_documented_layer_shapes_match_oracle(cfg1, sc1, model, layers):
    """The stacks the reference's own documentation names -- `-hidden_layers [(3, 32), (3, 16)]` (TE:51), 3_8_3_8 (TE:82) -- and
    other mixed widths: log-probabilities, loss and every weight gradient against the oracle.  They run on the MFMA kernels with
    the hidden widths zero-padded to one promoted width (ops.promote_weights), never on the one-row-per-workgroup generic path."""
    from scone_gcn_amd import ops, trajectory_experiments as te
    sel = np.arange(40, 67)
    shapes = so.weight_shapes(1, layers, 1)
    w = _rand_weights(shapes, 0.25 if model == "scone" else 0.03, 11)
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, model)
    y, last = cfg1["targets"][sel], cfg1["last_nodes"][sel]
    mask = np.ones(len(sel), int)
    mask[[0, 20]] = 0
    ref_out = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, last, X, act)
    ref_loss, ref_g = so.scone_loss_and_grad(w, shifts_o[0], shifts_o[1], Bc, last, X, y, mask, 0.0, act)
    shifts, readout, _ = te.setup_from_complex(sc1, model)
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    with ops.KernelTimer() as kt:
        out = te.MODEL_FUNCS[model](wt, *shifts, readout, last, X)
        m = torch.as_tensor(mask, device="cuda").bool()
        yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
        loss = -(out[m] * yt[m]).sum() / m.sum()
        loss.backward()
    P = min(32, ops.promoted_width([c for _, c in layers], wide=True))         # widths above 32 run in 32-channel blocks
    keys = list(kt.summary())
    assert any(k == "conv_fwd c%d->%d" % (P, P) for k in keys), keys           # the uniform-width MFMA kernels carried it
    assert not any("c48" in k or "c64" in k or "c96" in k or "c40" in k for k in keys), keys   # never the generic one-row-per-workgroup path
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert wt[k].grad.shape == wt[k].shape
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


def test_mixed_width_bunch_matches_oracle(cfg1, sc1):
    """Bunch with 7_32_7_16 hidden layers: promoted to 32 everywhere, so the fused three-level kernels carry the middle layer."""
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(200, 219)
    w = _rand_weights(so.weight_shapes(1, [(7, 32), (7, 16)], 1, "bunch"), 0.4, 5)
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    shifts, nbrhoods, _ = te.setup_from_complex(sc1, "bunch")
    ref_loss, ref_g = so.bunch_loss_and_grad(w, [s_.csr for s_ in shifts], nbrhoods, last, X, y, np.ones(len(sel), int), 0.0)
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
    loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / len(sel)
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


def test_single_sample_call(cfg1, sc1):
    from scone_gcn_amd import trajectory_experiments as te
    shapes = so.weight_shapes(1, [(3, 16)] * 3, 1)
    w = _rand_weights(shapes, 0.25, 3)
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, np.array([5]), "scone")
    ref = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, cfg1["last_nodes"][[5]], X)[0]
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    out = te.scone_func(w, *shifts, readout, int(cfg1["last_nodes"][5]), cfg1["flows"][5])
    assert out.shape == (cfg1["D"], 1)
    assert _maxdiff(out.cpu().numpy(), ref) <= TOL


def test_sparse_and_dense_flows_agree(cfg1, sc1):
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.synthetic_data_gen import SparseFlows
    sel = np.arange(40, 72)
    shapes = so.weight_shapes(1, [(3, 16)] * 3, 1)
    w = _rand_weights(shapes, 0.25, 5)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    dense = te.scone_func(w, *shifts, readout, cfg1["last_nodes"][sel], cfg1["flows"][sel])
    sparse = te.scone_func(w, *shifts, readout, cfg1["last_nodes"][sel], SparseFlows.fromdense(cfg1["flows"][sel]))
    assert torch.equal(dense, sparse)


def test_flip_invariance_tanh(cfg1, sc1):
    """-flip_edges experiment (TE:43, 214-219, 242-244, 288-296): with an odd activation the log-probabilities do
    not depend on edge orientation; with leaky_relu they do."""
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(100, 116)
    shapes = so.weight_shapes(1, [(3, 16)] * 3, 1)
    w = _rand_weights(shapes, 0.25, 9)
    last = cfg1["last_nodes"][sel]
    outs = {}
    for model in ("scone", "ebli"):
        if model == "ebli":
            w = _rand_weights(shapes, 0.03, 9)
        for flip in (False, True):
            shifts, readout, flips = te.setup_from_complex(sc1, model, flip_edges=flip)
            X = te.apply_flips(cfg1["flows"][sel], flips)
            outs[(model, flip)] = te.MODEL_FUNCS[model](w, *shifts, readout, last, X).cpu().numpy()
    assert _maxdiff(outs[("scone", False)], outs[("scone", True)]) <= TOL
    assert _maxdiff(outs[("ebli", False)], outs[("ebli", True)]) > 1e-4
    # and the flipped scone path matches the oracle run on flipped operators
    w = _rand_weights(shapes, 0.25, 9)
    flips = sc1.flip_vector(1)
    assert np.array_equal(np.diag(so.flip_matrix(cfg1["E"])), flips)
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, "scone", flips)
    ref = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, last, X)
    assert _maxdiff(outs[("scone", True)], ref) <= TOL


@pytest.mark.parametrize("hidden", [8, 16, 32, 40])   # 40: ns*C > 128 -> generic multi-group kernels
def test_bunch_matches_oracle(cfg1, sc1, hidden):
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(200, 214)
    shapes = so.weight_shapes(1, [(7, hidden)] * 3, 1, "bunch")
    w = _rand_weights(shapes, 0.4, 11)
    S = so.bunch_shifts(cfg1["B1"], cfg1["B2"])
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    mask = np.ones(len(sel), int)
    mask[2] = 0
    ref_out = so.bunch_forward(w, S, nb, last, X)
    ref_loss, ref_g = so.bunch_loss_and_grad(w, S, nb, last, X, y, mask, 0.0)

    shifts, nbrhoods, _ = te.setup_from_complex(sc1, "bunch")
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL
    m = torch.as_tensor(mask, device="cuda").bool()
    yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
    loss = -(out[m] * yt[m]).sum() / m.sum()
    from scone_gcn_amd import ops
    with ops.KernelTimer() as kt:
        loss.backward()
    if hidden == 32 and ops.FUSE_BUNCH and ops.FOLD_BUNCH:
        # the third layer ran the fused three-level backward, the first two the rank-one fold (one stream over dZ2 per level)
        assert {"terms_bwd c32", "dense_bwd x6", "dense_bwd x4"} <= set(kt.summary())
        assert "terms_bwd c32 + dW_first" not in kt.summary()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


def test_bunch_fused_first_gradient_equals_the_separate_kernels(cfg1, sc1):
    """scn_terms_backward_fused_first (the second layer's backward contracts its input gradient with the first layer's shifted
    input and never writes it) against scn_terms_backward + scn_conv_dw_first on the same batch: all 28 weight gradients."""
    from scone_gcn_amd import ops
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(300, 322)
    w = _rand_weights(so.weight_shapes(1, [(7, 32)] * 3, 1, "bunch"), 0.4, 3)
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    shifts, nbrhoods, _ = te.setup_from_complex(sc1, "bunch")
    yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
    grads = {}
    keep, keep_fold = ops.FUSE_FIRST, ops.FOLD_BUNCH
    try:
        ops.FOLD_BUNCH = False                                  # (the default path folds the first two layers and needs neither)
        for fused in (True, False):
            ops.FUSE_FIRST = fused
            wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
            out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
            with ops.KernelTimer() as kt:
                (-(out * yt).sum() / len(sel)).backward()
            assert ("terms_bwd c32 + dW_first" in kt.summary()) == fused
            grads[fused] = [t.grad.cpu().numpy().astype(np.float64) for t in wt]
    finally:
        ops.FUSE_FIRST, ops.FOLD_BUNCH = keep, keep_fold
    for k, (a, b) in enumerate(zip(grads[True], grads[False])):
        assert _maxdiff(a, b) <= 2e-6 * max(1.0, np.abs(b).max()), "weight %d" % k
    assert max(np.abs(g).max() for g in grads[True][:7]) > 1e-4           # the first layer's gradients are not trivially zero


@pytest.mark.parametrize("hidden", [32, 16])
def test_bunch_rank_one_fold_equals_the_unfolded_layers(cfg1, sc1, hidden):
    """BunchPlan._fold_forward / _fold_backward (the first two layers as shifts of one-channel tensors + rank-one expansions:
    relu(g w) = g+ relu(w) + g- min(w, 0), the first layer's output never formed) against the same model with the two layers
    run as ordinary layers: log-probabilities and all 28 weight gradients, and both against the oracle."""
    from scone_gcn_amd import ops
    from scone_gcn_amd import trajectory_experiments as te
    sel = np.arange(500, 523)
    w = _rand_weights(so.weight_shapes(1, [(7, hidden)] * 3, 1, "bunch"), 0.4, 13)
    X, y, last = cfg1["flows"][sel], cfg1["targets"][sel], cfg1["last_nodes"][sel]
    shifts, nbrhoods, _ = te.setup_from_complex(sc1, "bunch")
    ref_loss, ref_g = so.bunch_loss_and_grad(w, [s_.csr for s_ in shifts], nbrhoods, last, X, y, np.ones(len(sel), int), 0.0)
    yt = torch.as_tensor(y, dtype=torch.float32, device="cuda")
    res = {}
    keep = ops.FOLD_BUNCH
    try:
        for fold in (True, False):
            ops.FOLD_BUNCH = fold
            wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
            with ops.KernelTimer() as kt:
                out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
                loss = -(out * yt).sum() / len(sel)
                loss.backward()
            assert any(k.startswith("dense_fwd x6") or k.startswith("dense_fwd x4") for k in kt.summary()) == fold
            res[fold] = (out.detach().cpu().numpy(), float(loss.detach()), [t.grad.cpu().numpy().astype(np.float64) for t in wt])
    finally:
        ops.FOLD_BUNCH = keep
    assert _maxdiff(res[True][0], res[False][0]) <= 2e-6
    for k, (a, b) in enumerate(zip(res[True][2], res[False][2])):
        assert _maxdiff(a, b) <= 2e-6 * max(1.0, np.abs(b).max()), "weight %d" % k
    assert abs(res[True][1] - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(res[True][2][k], ref_g[k]) <= TOL, "weight %d" % k
    assert max(np.abs(g_).max() for g_ in res[True][2][:14]) > 1e-4          # the two folded layers' gradients are not trivially zero


def test_spmm_dual_matches_scipy(cfg1, sc1):
    from scone_gcn_amd import ops
    shifts = sc1.scone_shifts()
    plan = ops.get_scone_plan(shifts[0], shifts[1], sc1.bconds(), "tanh", ops.default_device())
    rs = np.random.RandomState(0)
    for k in (1, 4, 64, 128, 100):
        x = rs.randn(3, cfg1["E"], k).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        ya, yb = plan.conv.spmm_dual(xd)
        lo, up = shifts[0].device_csr(), shifts[1].device_csr()
        for s in range(3):
            assert _maxdiff(ya[s].cpu().numpy(), lo @ x[s].astype(np.float64)) <= 2e-5
            assert _maxdiff(yb[s].cpu().numpy(), up @ x[s].astype(np.float64)) <= 2e-5


@pytest.mark.parametrize("hidden,slabs", [(16, 1), (16, 3), (16, 4), (32, 3)])
def test_layer_forward_on_random_slabs_matches_scipy(cfg1, sc1, hidden, slabs):
    """One layer act(X W0 + L_low X W1 + L_up X W2) (TE:145-147) through scn_conv_forward on dense random slabs -- the C=16
    kernel works on PAIRS of slabs, so odd and single slab counts are the edge cases."""
    from scone_gcn_amd import ops
    shifts = sc1.scone_shifts()
    plan = ops.get_scone_plan(shifts[0], shifts[1], sc1.bconds(), "tanh", ops.default_device())
    rs = np.random.RandomState(3)
    E, C = cfg1["E"], hidden
    x = rs.randn(slabs, E, 4, C).astype(np.float32)
    W = [(0.3 * rs.randn(C, C)).astype(np.float32) for _ in range(3)]
    out = plan.conv.forward([torch.from_numpy(x).cuda()], [torch.from_numpy(w).cuda() for w in W], C, "tanh").cpu().numpy()
    lo, up = shifts[0].device_csr(), shifts[1].device_csr()
    for s in range(slabs):
        xs = x[s].astype(np.float64).reshape(E, 4 * C)
        ref = np.tanh(xs.reshape(E, 4, C) @ W[0].astype(np.float64) + (lo @ xs).reshape(E, 4, C) @ W[1].astype(np.float64)
                      + (up @ xs).reshape(E, 4, C) @ W[2].astype(np.float64))
        assert _maxdiff(out[s], ref) <= TOL


def test_trainer_step_matches_oracle_adam(cfg1, sc1):
    """Scone_GCN.grad_step == oracle gradient + oracle Adam, for three consecutive steps on reference batches."""
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    stm.reseed(1030)
    N = 200
    sel = np.arange(N)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    inputs = [readout, cfg1["last_nodes"][sel], cfg1["flows"][sel]]
    y, train_mask = cfg1["targets"][sel], cfg1["train_mask"][sel]
    net = stm.Scone_GCN(1, 1e-3, 50, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, train_mask, model_type="scone")
    w0 = so.generate_weights(1, [(3, 16)] * 3, 1)
    for a, b in zip(net.weights, w0):
        assert np.array_equal(a.cpu().numpy(), b.astype(np.float32))   # same seed-1030 stream as STM:15, 237
    shifts_o, Bc, _, act = _oracle_scone(cfg1, w0, sel, "scone")
    adam = so.Adam([w.astype(np.float32).astype(np.float64) for w in w0], 1e-3)
    rs_o = np.random.RandomState(1030)
    [rs_o.randn(*s) for s in so.weight_shapes(1, [(3, 16)] * 3, 1)]   # the oracle's copy of the seed-1030 stream
    for i in range(3):
        bm_o = so.draw_batch_mask(rs_o, N, 50, train_mask)
        bm = np.array([1] * 50 + [0] * (N - 50))
        stm._RNG.shuffle(bm)
        bm = np.logical_and(bm, train_mask)
        assert np.array_equal(bm, bm_o)
        ridge = 5e-5 * so.ridge(adam.x)
        loss = float(net.grad_step(inputs, y, bm))
        ref_loss, g = so.scone_loss_and_grad(adam.x, shifts_o[0], shifts_o[1], Bc, inputs[1], inputs[2], y, bm, 5e-5)
        assert abs(loss + ridge - ref_loss) <= TOL
        adam.update(i, g)
        for a, b in zip(net.weights, adam.x):
            assert _maxdiff(a.cpu().numpy(), b) <= 2e-6, "step %d" % i


def test_adam_with_the_step_index_in_device_memory_gives_the_bits_of_the_host_index_form():
    """scn_adam_step_dev (reads i from device memory, leaves i + 1: the launch a captured graph can end with) against scn_adam_step
    with the same index from the host: the same weights and moments bit for bit over consecutive steps, from a late index as well
    (bias corrections 1 - b^(i+1) near their limits), and against the update formula in fp64 (STM:300-326)."""
    import ctypes
    from scone_gcn_amd import _lib, ops
    lib = _lib.load()
    dev = ops.default_device()
    rs = np.random.RandomState(5)
    n = 6272
    w0 = torch.tensor(rs.randn(n).astype(np.float32) * 0.1, device=dev)
    for start in (0, 4096):
        bufs = {}
        for form in ("host", "dev"):
            w, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
            step = torch.full((1,), start, device=dev, dtype=torch.int32)
            rg = np.random.RandomState(9)
            for i in range(start, start + 4):
                g = torch.tensor(rg.randn(n).astype(np.float32) * 1e-2, device=dev)
                if form == "host":
                    st = lib.scn_adam_step(n, ops._dev(w), ops._dev(g), ops._dev(m), ops._dev(v), 1e-3, 0.9, 0.999, 1e-8, i, 5e-5, 1.0,
                                           ops._stream())
                else:
                    st = lib.scn_adam_step_dev(n, ops._dev(w), ops._dev(g), ops._dev(m), ops._dev(v), 1e-3, 0.9, 0.999, 1e-8,
                                               ctypes.c_void_p(step.data_ptr()), 5e-5, 1.0, ops._stream())
                assert st == 0
            bufs[form] = (w.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), int(step.item()))
        assert bufs["dev"][3] == start + 4 and bufs["host"][3] == start
        for a, b in zip(bufs["host"][:3], bufs["dev"][:3]):
            assert np.array_equal(a, b)
        # fp64 restatement of the four updates
        w, m, v = w0.cpu().numpy().astype(np.float64), np.zeros(n), np.zeros(n)
        rg = np.random.RandomState(9)
        for i in range(start, start + 4):
            g = (rg.randn(n).astype(np.float32) * 1e-2).astype(np.float64) + 2 * 5e-5 * w
            m = 0.1 * g + 0.9 * m
            v = 0.001 * g * g + 0.999 * v
            w = w - 1e-3 * (m / (1 - 0.9 ** (i + 1))) / (np.sqrt(v / (1 - 0.999 ** (i + 1))) + 1e-8)
        assert np.abs(bufs["dev"][0] - w).max() <= 2e-6


def test_loss_accuracy_and_reverse_inputs(cfg1, sc1):
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    stm.reseed(1030)
    sel = np.arange(120)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    inputs = [readout, cfg1["last_nodes"][sel], cfg1["flows"][sel]]
    y, mask = cfg1["targets"][sel], cfg1["test_mask"][sel]
    net = stm.Scone_GCN(1, 1e-3, 50, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, mask, model_type="scone")
    w = [a.cpu().numpy().astype(np.float64) for a in net.weights]
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, "scone")
    out = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, inputs[1], X)
    n_nbrs = sc1.n_nbrs(inputs[1])
    assert abs(net.loss(net.weights, inputs, y, mask) - so.loss_from_preds(out, y, mask, w, 5e-5)) <= TOL
    assert net.accuracy(shifts, inputs, y, mask, n_nbrs) == so.accuracy_from_preds(out, y, mask, n_nbrs)
    rev = [readout, cfg1["rev_last_nodes"][sel], cfg1["rev_flows"][sel]]
    out_r = so.scone_forward(w, shifts_o[0], shifts_o[1], Bc, rev[1], rev[2])
    assert abs(net.loss(net.weights, rev, cfg1["rev_targets"][sel], mask)
               - so.loss_from_preds(out_r, cfg1["rev_targets"][sel], mask, w, 5e-5)) <= TOL


@pytest.mark.parametrize("hidden,n_traj", [(32, 12), (16, 21)])
def test_larger_synthetic_complex_against_csr_oracle(hidden, n_traj):
    """E ~ 13k synthetic complex (own generator), oracle run with scipy CSR shifts.  Enough blocks for the launch grid to split
    the slabs over several workgroup rows; at hidden 16 the slab-pair kernels see odd and partial slab ranges."""
    _need_gpu()
    import scipy.sparse as sp
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(5000)
    sc = SimplicialComplex(cx)
    paths = g.generate_random_walks(cx, m=n_traj, seed=5)
    flows, choice, last, tnodes, _ = g.path_dataset(cx, paths, seed=2)
    D = sc.max_degree
    y = so.onehot_targets(choice, D)
    shapes = so.weight_shapes(1, [(3, hidden)] * 3, 1)
    w = _rand_weights(shapes, 0.15, 21)
    B1, B2 = g.incidence_matrices(cx)
    L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
    B1x = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()
    Bc = lambda n: B1x[sc.nbrhoods[n]].toarray()
    X = flows.todense().astype(np.float64)
    mask = np.ones(len(paths), int)
    ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last, X, y, mask, 0.0)
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.scone_func(wt, *shifts, readout, last, flows)
    loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / len(paths)
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL


def test_dense_terms_kernels_match_numpy():
    """scn_dense_terms_forward/backward (the Bunch per-level contraction) against fp64 NumPy."""
    _need_gpu()
    from scone_gcn_amd import ops
    rs = np.random.RandomState(3)
    for cs, c_out, c_aux, act in (([1, 1], 8, 1, "relu"), ([1, 1, 1], 32, 1, "leaky_relu"), ([1], 16, 1, "relu"), ([16, 16, 16], 16, 16, "relu"), ([32, 32], 32, 32, "tanh"),
                                  ([8, 8, 8], 1, 8, "none"), ([5, 3], 7, 5, "leaky_relu"), ([32, 32, 32], 1, 32, "relu"),
                                  ([16, 16], 1, 16, "relu"), ([32, 32, 32], 32, 32, "relu"),
                                  ([16, 16, 16], 16, 16, "relu"), ([16, 16], 16, 16, "tanh")):   # 16-wide: paired points, 32-wide kernels
        S, R, ns = 3, 157, 4
        Gs = [rs.randn(S, R, ns, c) for c in cs]
        Ws = [rs.randn(c, c_out) * 0.3 for c in cs]
        z = sum(g @ w for g, w in zip(Gs, Ws))
        ref = {"relu": np.maximum(z, 0), "tanh": np.tanh(z), "none": z, "leaky_relu": np.where(z >= 0, z, 0.01 * z)}[act]
        dev = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
        out = ops.dense_terms_forward([dev(g) for g in Gs], [dev(w) for w in Ws], c_out, act)
        assert _maxdiff(out.cpu().numpy(), ref) <= TOL
        # backward: G'_k [.., c_k] are the transposed-shift terms of the upstream gradient, W_k [c_aux, c_k]
        aux = rs.randn(S, R, ns, c_aux)
        if act == "relu":
            aux = np.maximum(aux, 0)
        if act == "tanh":
            aux = np.tanh(aux)
        Wb = [rs.randn(c_aux, c) * 0.3 for c in cs]
        dact = {"relu": (aux > 0) * 1.0, "tanh": 1 - aux ** 2, "none": np.ones_like(aux),
                "leaky_relu": np.where(aux >= 0, 1.0, 0.01)}[act]
        dx_ref = sum(g @ w.T for g, w in zip(Gs, Wb)) * dact
        dW_ref = [np.einsum("srna,srnc->ac", aux, g) for g in Gs]
        dWs = [torch.full((c_aux, c), 0.5, device="cuda") for c in cs]
        dx = ops.dense_terms_backward([dev(g) for g in Gs], [dev(w) for w in Wb], dev(aux), act, True, dWs)
        assert _maxdiff(dx.cpu().numpy(), dx_ref) <= TOL
        for a, b in zip(dWs, dW_ref):
            assert _maxdiff(a.cpu().numpy(), b + 0.5) <= 2e-5 * max(1.0, np.abs(b).max())


def test_rectangular_single_operator_spmm(cfg1, sc1):
    """The LDS-blocked SpMM on the Bunch shifts (rectangular, one value array, no identity) and their transposes."""
    from scone_gcn_amd import ops
    rs = np.random.RandomState(1)
    for sh in sc1.bunch_shifts():
        for m in (sh.device_csr(), sh.device_csr().T.tocsr()):
            op = ops.ConvOp(m.shape[0], [{"mats": [m], "identity": False, "n_cols": m.shape[1]}])
            for k in (4, 64, 128):
                x = rs.randn(2, m.shape[1], k).astype(np.float32)
                y, _ = op.spmm_dual(torch.from_numpy(x).cuda(), dual=False)
                for s in range(2):
                    assert _maxdiff(y[s].cpu().numpy(), m @ x[s].astype(np.float64)) <= 2e-5


def test_spmm_blocks_without_sources():
    """Rows without entries (isolated nodes cluster inside the holes of a complex, SDG:98-114) form plan blocks with NO source
    rows: nothing may be staged for them and their outputs are zeros.  K = 64 / 128 run the ring kernel, 100 the two-buffer one."""
    _need_gpu()
    import scipy.sparse as sp
    from scone_gcn_amd import ops
    rs = np.random.RandomState(4)
    n_rows, n_cols = 700, 300
    m = sp.random(n_rows, n_cols, density=0.02, random_state=rs, format="lil", dtype=np.float64)
    m[:192] = 0                                            # the first three 64-row blocks are empty
    m[400:530] = 0
    m = m.tocsr(); m.eliminate_zeros()
    op = ops.ConvOp(n_rows, [{"mats": [m], "identity": False, "n_cols": n_cols}])
    for k in (64, 128, 100):
        x = rs.randn(5, n_cols, k).astype(np.float32)
        y, _ = op.spmm_dual(torch.from_numpy(x).cuda(), dual=False)
        y = y.cpu().numpy()
        for s_ in range(5):
            assert _maxdiff(y[s_], m @ x[s_].astype(np.float64)) <= 2e-5
        assert not y[:, :192].any() and not y[:, 400:530].any()


def test_bunch_on_a_complex_with_a_hub_node_falls_back_to_the_per_shift_path():
    """A node of degree 60: its row of the concatenated Bunch operator has 121 > 112 distinct sources (itself, 60 neighbours, 60
    incident edges; the readout kernels take neighbourhoods up to 64 wide, so 60 it is), which the fused-layer plan
    cannot hold (scn_terms_create -> SCN_ERR_UNSUPPORTED).  BunchPlan must then run every layer, forward AND backward, on the
    per-shift SpMM + dense-term path (it used to raise at hidden 32): loss and all 28 weight gradients against the oracle."""
    _need_gpu()
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    base = g.random_SC_graph(700)
    hub = base.n_nodes
    near = np.argsort(np.linalg.norm(base.coords - [0.5, 0.5], axis=1))[:60]
    inset = np.zeros(hub + 1, bool)
    inset[near] = True
    spokes = np.stack([near, np.full(len(near), hub)], axis=1)
    rim = base.edges[inset[base.edges[:, 0]] & inset[base.edges[:, 1]]]
    cones = np.concatenate([rim, np.full((len(rim), 1), hub)], axis=1)            # (a, b, hub): sorted, hub is the largest id
    cx = g.Complex(n_nodes=hub + 1, edges=np.unique(np.concatenate([base.edges, spokes]), axis=0),
                   faces=np.unique(np.concatenate([base.faces, cones]), axis=0),
                   coords=np.concatenate([base.coords, [[0.5, 0.5]]]))
    sc = SimplicialComplex(cx)
    assert int(np.bincount(cx.edges.ravel()).max()) == 60
    paths = g.generate_random_walks(base, m=14, seed=5)                           # walks on the base complex (its edges all exist)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=2)
    n = len(paths)
    X = flows.todense()[:n].astype(np.float64)
    last = np.asarray(last[:n])
    y = so.onehot_targets(choice[:n], sc.max_degree)
    w = _rand_weights(so.weight_shapes(1, [(7, 32)] * 3, 1, "bunch"), 0.2, 5)
    B1, B2 = g.incidence_matrices(cx)
    S = [m.tocsr() for m in compute_shift_matrices(B1, B2)]
    ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, np.ones(n, int), 0.0)
    shifts, nbrhoods, _ = te.setup_from_complex(sc, "bunch")
    plan = ops.get_bunch_plan(shifts, nbrhoods, ops.default_device())
    assert plan._terms_ops() is None                                              # the fused-layer plan refused the hub row
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    with ops.KernelTimer() as kt:
        out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
        loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / n
        loss.backward()
    assert not any(k.startswith("terms_") for k in kt.summary())
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k


def test_fused_bunch_layer_with_an_absent_level_and_poisoned_lds():
    """scn_terms_forward / _backward with level tensors that are not given (x[0] = None, dz[0] = None): the staging skips their
    source rows, so the padded ELL entries of the PRESENT terms must not read those LDS pieces -- a preceding launch leaves
    NaN bit patterns there (a dual SpMM over an all-NaN tensor fills both staging buffers of every CU)."""
    _need_gpu()
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(2500)
    sc = SimplicialComplex(cx)
    shifts, nbr, _ = te.setup_from_complex(sc, "bunch")
    plan = ops.get_bunch_plan(shifts, nbr, ops.default_device())
    fwd, bwd = plan._terms_ops()
    S, sizes = 2, plan.sizes
    SRC, DST = ops.BUNCH_SRC, ops.BUNCH_DST
    rs = np.random.RandomState(3)
    dev = [s_.device_csr().astype(np.float64) for s_ in shifts]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def poison():
        nan = torch.full((4, sizes[1], 128), float("nan"), device="cuda")
        plan.term_fwd[3].spmm_dual(nan, dual=False)
        torch.cuda.synchronize()

    def shift(m, x):
        y = m @ x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1).astype(np.float64)
        return y.reshape(m.shape[0], x.shape[0], 4, 32).transpose(1, 0, 2, 3)
    xs = [None, rs.randn(S, sizes[1], 4, 32).astype(np.float32), rs.randn(S, sizes[2], 4, 32).astype(np.float32)]
    Wk = [(0.2 * rs.randn(32, 32)).astype(np.float32) for _ in range(7)]
    Ws = [[None] * 3 for _ in range(3)]
    for k in range(7):
        if xs[SRC[k]] is not None:
            Ws[DST[k]][SRC[k]] = t(Wk[k])
    poison()
    outs = fwd.forward([None if x is None else t(x) for x in xs], Ws, "relu", [True] * 3)
    for l in range(3):
        ref = sum(shift(dev[k], xs[SRC[k]]) @ Wk[k].astype(np.float64) for k in range(7) if DST[k] == l and xs[SRC[k]] is not None)
        got = outs[l].cpu().numpy()
        assert np.isfinite(got).all()
        assert _maxdiff(got, np.maximum(ref, 0)) <= 2e-5
    dzs = [None, rs.randn(S, sizes[1], 4, 32).astype(np.float32), rs.randn(S, sizes[2], 4, 32).astype(np.float32)]
    auxs = [np.maximum(rs.randn(S, n_, 4, 32), 0).astype(np.float32) for n_ in sizes]
    Wb = [[None] * 3 for _ in range(3)]
    dWb = [[None] * 3 for _ in range(3)]
    for k in range(7):
        if dzs[DST[k]] is not None:
            Wb[SRC[k]][DST[k]], dWb[SRC[k]][DST[k]] = t(Wk[k]), torch.zeros((32, 32), device="cuda")
    poison()
    dxs = ops._terms_backward(bwd, [None if d is None else t(d) for d in dzs], Wb, [t(a) for a in auxs], "relu", [True] * 3, dWb)
    for a in range(3):
        gk = {k: shift(dev[k].T.tocsr(), dzs[DST[k]]) for k in range(7) if SRC[k] == a and dzs[DST[k]] is not None}
        ref = sum(gk[k] @ Wk[k].astype(np.float64).T for k in gk) * (auxs[a] > 0)
        got = dxs[a].cpu().numpy()
        assert np.isfinite(got).all()
        assert _maxdiff(got, ref) <= 2e-5
        for k in gk:
            refw = np.einsum("srnc,srnd->cd", auxs[a].astype(np.float64), gk[k])
            assert _maxdiff(dWb[a][DST[k]].cpu().numpy(), refw) <= 2e-5 * max(1.0, np.abs(refw).max())


def test_bunch_two_layers_larger_complex_against_csr_oracle():
    """Bunch, two layers x hidden 16, on an E ~ 13k complex: blocked per-shift SpMM + dense kernels, node/face orders."""
    _need_gpu()
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    cx = g.random_SC_graph(5000)
    sc = SimplicialComplex(cx)
    paths = g.generate_random_walks(cx, m=12, seed=5)
    flows, choice, last, tnodes, _ = g.path_dataset(cx, paths, seed=2)
    n = len(paths)
    X = flows.todense()[:n].astype(np.float64)
    last = np.asarray(last[:n])
    nb, D = sc.nbrhoods, sc.max_degree
    y = so.onehot_targets(choice[:n], D)
    ynode = y                                # bunch targets use the same neighbour-slot one-hot layout
    shapes = so.weight_shapes(1, [(7, 16), (7, 16)], 1, "bunch")
    w = _rand_weights(shapes, 0.3, 5)
    B1, B2 = g.incidence_matrices(cx)
    S = [m.tocsr() for m in compute_shift_matrices(B1, B2)]
    mask = np.ones(n, int)
    ref_out = so.bunch_forward(w, S, nb, last, X)
    ref_loss, ref_g = so.bunch_loss_and_grad(w, S, nb, last, X, ynode, mask, 0.0)
    shifts, nbrhoods, _ = te.setup_from_complex(sc, "bunch")
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.bunch_func(wt, *shifts, nbrhoods, last, X)
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL
    loss = -(out * torch.as_tensor(ynode, dtype=torch.float32, device="cuda")).sum() / n
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL * max(1.0, np.abs(ref_g[k]).max()), "weight %d" % k


@pytest.mark.parametrize("hidden", [16, 32])
def test_first_layer_fast_path_matches_generic_entry_points(cfg1, sc1, hidden):
    """scn_conv_forward_first / scn_conv_dw_first (shift on the 1-channel side) against scn_conv_forward / scn_conv_backward
    and against scipy for the shifted input."""
    from scone_gcn_amd import ops
    shifts = sc1.scone_shifts()
    plan = ops.get_scone_plan(shifts[0], shifts[1], sc1.bconds(), "tanh", ops.default_device())
    rs = np.random.RandomState(4)
    S, E, C = 3, cfg1["E"], hidden
    x = torch.from_numpy(rs.randn(S, E, 4, 1).astype(np.float32)).cuda()
    dz = torch.from_numpy(rs.randn(S, E, 4, C).astype(np.float32)).cuda()
    W = [torch.from_numpy((0.3 * rs.randn(1, C)).astype(np.float32)).cuda() for _ in range(3)]
    out_ref = plan.conv.forward([x], W, C, "tanh")
    out, y = plan.conv.forward_first(x, W, C, "tanh")
    assert torch.equal(out, out_ref)
    lo, up = shifts[0].device_csr(), shifts[1].device_csr()
    xs = x.cpu().numpy().astype(np.float64)[..., 0]                    # (S, E, 4)
    for s in range(S):
        ref = np.stack([xs[s], lo @ xs[s], up @ xs[s]], axis=-1)       # (E, 4, 3)
        assert _maxdiff(y[s].cpu().numpy()[..., :3], ref) <= 2e-5
        assert float(y[s][..., 3].abs().max()) == 0.0                  # 16-byte records (x, S_lo x, S_up x, 0)
    g_ref = [torch.zeros_like(w) for w in W]
    plan.conv_T.backward([dz], W, x, "tanh", False, g_ref)
    for yy in (None, y):
        g = [torch.full_like(w, 0.25) for w in W]                      # accumulated into
        assert plan.conv.dw_first(x, yy, dz, g)
        for a, b in zip(g, g_ref):
            bb = b.cpu().numpy().astype(np.float64)
            assert _maxdiff(a.cpu().numpy(), bb + 0.25) <= 2e-5 * max(1.0, np.abs(bb).max())


@pytest.mark.parametrize("model,mode,hidden", [("scone", "zeros", 32), ("scone", "field", 32), ("scone", "zeros", 16),
                                               ("scone", "field", 16)])   # ebli: L1^2 rows exceed the block plan
def test_zero_skipping_modes_match_the_dense_step(model, mode, hidden):
    """Work-list (zero-skipping) execution of a gradient step == the dense execution, on a complex large enough to have
    inactive blocks: same loss, same weight gradients (to rounding of the summation order), buffers all-zero afterwards."""
    _need_gpu()
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(20000)
    sc = SimplicialComplex(cx)
    paths = g.generate_random_walks(cx, m=24, seed=3, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=2)
    N, D = len(paths), sc.max_degree
    y = so.onehot_targets(choice, D)
    shifts, readout, _ = te.setup_from_complex(sc, model)
    inputs = [readout, last, flows]
    res = {}
    for m in ("dense", mode):
        stm.reseed(1030)
        net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False, skip_mode=m)
        net.setup(te.MODEL_FUNCS[model], [(3, hidden)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type=model)
        for w in net.weights:                                  # larger weights than 0.01 randn: gradients well above noise
            w.mul_(20.0 if model == "scone" else 3.0)
        staged = net.stage(inputs, y, np.arange(N))
        assert (staged[0][3] is None) == (m == "dense")
        loss = float(net.grad_step_staged(inputs, staged, N, apply=False))
        loss2 = float(net.grad_step_staged(inputs, staged, N, apply=False))     # pooled buffers must be clean again
        assert loss2 == loss
        res[m] = (loss, [gr.clone() for gr in net._grads], staged, net)
    act = res[mode][2][0][3]
    assert act is not None and max(act["active_fraction"]["fwd"]) < 0.9
    assert abs(res[mode][0] - res["dense"][0]) <= 1e-6 * max(1.0, abs(res["dense"][0]))
    for a, b in zip(res[mode][1], res["dense"][1]):
        bb = b.cpu().numpy().astype(np.float64)
        assert _maxdiff(a.cpu().numpy(), bb) <= 2e-5 * max(1.0, np.abs(bb).max())
    # prediction path (forward only, buffers released without a backward): same loss and accuracy as the dense net
    mask = np.ones(N, int)
    n_nbrs = sc.n_nbrs(last)
    l_skip = res[mode][3].loss(res[mode][3].weights, inputs, y, mask)
    l_dense = res["dense"][3].loss(res["dense"][3].weights, inputs, y, mask)
    assert abs(l_skip - l_dense) <= 1e-6 * max(1.0, abs(l_dense))
    assert res[mode][3].accuracy(shifts, inputs, y, mask, n_nbrs) == res["dense"][3].accuracy(shifts, inputs, y, mask, n_nbrs)
    plan = res[mode][3]._plan(inputs)
    for pool in plan._zero_pool.values():
        for t in pool:
            assert float(t.abs().max()) == 0.0


@pytest.mark.parametrize("hidden", [8, 16, 32])
def test_ebli_composed_plan_matches_fused_operator(cfg1, sc1, hidden):
    """PowerPlan (S (S H) on the blocked SpMM + dense term kernels; what Ebli uses when L1^2 outgrows the block plan) against
    the plan that applies the stored L1^2: same log-probabilities, same gradients."""
    from scone_gcn_amd import ops
    L1, L1sq = sc1.ebli_shifts()
    assert ops._is_square_of(L1sq, L1) and not ops._is_square_of(L1, L1sq)
    dev = ops.default_device()
    fused = ops.SconePlan(L1, L1sq, sc1.bconds(), "leaky_relu", dev)
    comp = ops.PowerPlan(L1, L1sq, sc1.bconds(), "leaky_relu", dev)
    sel = np.arange(40, 52)
    x, n = ops.flows_to_slabs(cfg1["flows"][sel], sc1.layout, dev)
    last = ops._last_nodes_dev(cfg1["last_nodes"][sel], x.shape[0] * ops.NS, dev)
    shapes = so.weight_shapes(1, [(3, hidden)] * 2, 1)
    w = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in _rand_weights(shapes, 0.05, 9)]
    rs = np.random.RandomState(2)
    res = []
    for plan in (fused, comp):
        logp, saved = plan.forward(x, last, w)
        d_logp = torch.tensor(rs.randn(*logp.shape).astype(np.float32), device="cuda") if not res else res[0][2]
        grads = [torch.zeros_like(a) for a in w]
        plan.backward(saved, logp, d_logp, last, w, grads)
        res.append((logp, grads, d_logp))
    assert _maxdiff(res[1][0].cpu().numpy(), res[0][0].cpu().numpy().astype(np.float64)) <= TOL
    for a, b in zip(res[1][1], res[0][1]):
        bb = b.cpu().numpy().astype(np.float64)
        assert _maxdiff(a.cpu().numpy(), bb) <= 2e-5 * max(1.0, np.abs(bb).max())


def test_full_size_properties_at_one_million_edges(big_complex):
    """Size-independent properties on the benchmark complex itself (|E| = 996 634, hidden 32) -- the oracle comparison at this
    size is tests/test_gpu_fullsize.py; here:
    (1) the dual SpMM is linear, (2) with tanh the log-probabilities do not depend on edge orientation (-flip_edges,
    TE:214-219, 242-244, 288-296), (3) the zero-skipping modes reproduce the dense gradient step, (4) a row of zeros in,
    a row of zeros out: padding trajectories get the uniform-over-D log-probabilities of all-zero logits."""
    _need_gpu()
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    cx, sc = big_complex
    E = cx.n_edges
    N = 6                                                          # 2 slabs, the second one half padding
    paths = g.generate_random_walks(cx, m=N, seed=11, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=3)
    D = sc.max_degree
    y = so.onehot_targets(choice, D)
    shapes = so.weight_shapes(1, [(3, 32)] * 3, 1)
    w = [torch.tensor(a, dtype=torch.float32, device="cuda") for a in _rand_weights(shapes, 0.12, 5)]
    outs = {}
    for flip in (False, True):
        shifts, readout, flips = te.setup_from_complex(sc, "scone", flip_edges=flip)
        outs[flip] = te.scone_func(w, *shifts, readout, last, te.apply_flips(flows, flips)).cpu().numpy()
        if not flip:
            plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
            # (1) linearity of [L_lo X, L_up X]
            rs = np.random.RandomState(0)
            xa = torch.from_numpy(rs.randn(1, E, 8).astype(np.float32)).cuda()
            xb = torch.from_numpy(rs.randn(1, E, 8).astype(np.float32)).cuda()
            ya, yb = plan.conv.spmm_dual(xa), plan.conv.spmm_dual(xb)
            yc = plan.conv.spmm_dual(2.0 * xa - 3.0 * xb)
            for i in range(2):
                ref = 2.0 * ya[i].double() - 3.0 * yb[i].double()
                assert float((yc[i].double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
            # (4) padding rows: all-zero flow -> all-zero logits -> log(1/D) in every slot
            x, n = ops.flows_to_slabs(flows, sc.layout, ops.default_device())
            ld = ops._last_nodes_dev(last, x.shape[0] * ops.NS, ops.default_device())
            logp, _ = plan.forward(x, ld, w)
            assert float((logp[n:] + np.log(D)).abs().max()) <= 1e-6
            # (3) zero-skipping == dense on a gradient step
            inputs = [readout, last, flows]
            grads = {}
            for mode in ("dense", "zeros", "field"):
                stm.reseed(1030)
                net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False, skip_mode=mode)
                net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
                for a, b in zip(net.weights, w):
                    a.copy_(b)
                staged = net.stage(inputs, y, np.arange(N))
                loss = float(net.grad_step_staged(inputs, staged, N, apply=False))
                grads[mode] = (loss, [t.clone() for t in net._grads])
            for mode in ("zeros", "field"):
                assert abs(grads[mode][0] - grads["dense"][0]) <= 1e-6 * max(1.0, abs(grads["dense"][0]))
                for a, b in zip(grads[mode][1], grads["dense"][1]):
                    bb = b.cpu().numpy().astype(np.float64)
                    assert _maxdiff(a.cpu().numpy(), bb) <= 2e-5 * max(1.0, np.abs(bb).max())
    assert _maxdiff(outs[True], outs[False].astype(np.float64)) <= TOL      # (2)


def test_errors_are_loud(cfg1, sc1):
    from scone_gcn_amd import trajectory_experiments as te
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    w = _rand_weights(so.weight_shapes(1, [(3, 16)] * 3, 1), 0.1, 1)
    with pytest.raises(AssertionError):
        te.scone_func(w[:-2], *shifts, readout, cfg1["last_nodes"][:4], cfg1["flows"][:4])   # wrong number of weights
    with pytest.raises(ValueError):
        te.scone_func(w, *shifts, readout, cfg1["last_nodes"][:3], cfg1["flows"][:4])


def test_ocean_drifter_config_full_batch():
    """BASELINE.json config 3: buoy complex (133 nodes / 320 edges / 186 faces), 3-layer SCoNe hidden 16, all 160
    training trajectories as one batch: loss, every gradient and test-set accuracy against the oracle."""
    _need_gpu()
    import os
    from scone_gcn_amd import buoy_data as bd
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    gld = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "buoy.npz"))
    trajs = [gld["traj_nodes"][gld["traj_ptr"][i]:gld["traj_ptr"][i + 1]].astype(int).tolist()
             for i in range(len(gld["traj_ptr"]) - 1)]
    cx, paths, flows, choice, last, tnodes, train_mask, test_mask = bd.buoy_dataset(
        gld["elist"].astype(np.int64), gld["tlist"].astype(np.int64), gld["coords"], trajs)
    sc = SimplicialComplex(cx)
    D = sc.max_degree
    y = so.onehot_targets(choice, D)
    w = _rand_weights(so.weight_shapes(1, [(3, 16)] * 3, 1), 0.3, 13)
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    L_lo, L_up = so.scone_shifts(B1, B2)
    nb, _ = so.neighborhoods(cx.edges, cx.n_nodes)
    X = flows.todense().astype(np.float64)
    ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X, y, train_mask, 0.0)
    ref_out = so.scone_forward(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X)
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = te.scone_func(wt, *shifts, readout, last, flows)
    assert _maxdiff(out.detach().cpu().numpy(), ref_out) <= TOL
    m = torch.as_tensor(train_mask, device="cuda").bool()
    loss = -(out[m] * torch.as_tensor(y, dtype=torch.float32, device="cuda")[m]).sum() / m.sum()
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL
    n_nbrs = sc.n_nbrs(last)
    acc_ref = so.accuracy_from_preds(ref_out, y, test_mask, n_nbrs)
    acc = so.accuracy_from_preds(out.detach().cpu().numpy().astype(np.float64), y, test_mask, n_nbrs)
    assert acc == acc_ref


def test_isolated_last_node_and_empty_flow(cfg1, sc1):
    """Edge cases: a trajectory whose last node has no neighbours (all -1 padding -> uniform log-probs -log D) and an
    all-zero flow (every activation stays exactly zero: no bias terms, TE:145-149)."""
    from scone_gcn_amd import trajectory_experiments as te
    nb, D = so.neighborhoods(cfg1["edges"], cfg1["n_nodes"])
    iso = int(np.nonzero((nb >= 0).sum(1) == 0)[0][0])
    w = _rand_weights(so.weight_shapes(1, [(3, 16)] * 3, 1), 0.25, 17)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    X = np.zeros((2, cfg1["E"], 1))
    X[0] = cfg1["flows"][3]
    out = te.scone_func(w, *shifts, readout, np.array([iso, int(cfg1["last_nodes"][3])]), X).cpu().numpy()
    assert np.allclose(out[0, :, 0], -np.log(D), atol=1e-6)
    assert np.allclose(out[1, :, 0], -np.log(D), atol=1e-6)          # zero flow -> zero logits everywhere


def test_train_loop_two_epochs_matches_oracle_trainer(cfg1, sc1):
    """Scone_GCN.train (STM:264-357) end to end: same seed-1030 weight / batch-mask stream, two epochs of Adam steps,
    final weights, returned losses and accuracies against an oracle trainer; then test() and two_target_accuracy()."""
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    stm.reseed(1030)
    N, bs, epochs = 120, 30, 2
    sel = np.arange(N)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    inputs = [readout, cfg1["last_nodes"][sel], cfg1["flows"][sel]]
    y, train_mask, test_mask = cfg1["targets"][sel], cfg1["train_mask"][sel], cfg1["test_mask"][sel]
    n_nbrs = sc1.n_nbrs(inputs[1])
    net = stm.Scone_GCN(epochs, 1e-3, bs, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, train_mask, model_type="scone")
    res = net.train(inputs, y, train_mask, test_mask, n_nbrs)

    rs = np.random.RandomState(1030)
    w = [(0.01 * rs.randn(*s)).astype(np.float32).astype(np.float64) for s in so.weight_shapes(1, [(3, 16)] * 3, 1)]
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, "scone")
    adam = so.Adam(w, 1e-3)
    n_batches = int(train_mask.sum()) // bs
    for i in range(epochs * n_batches):
        bm = so.draw_batch_mask(rs, N, bs, train_mask)
        _, g = so.scone_loss_and_grad(adam.x, shifts_o[0], shifts_o[1], Bc, inputs[1], X, y, bm, 5e-5)
        adam.update(i, g)
    for a, b in zip(net.weights, adam.x):
        assert _maxdiff(a.cpu().numpy(), b) <= 5e-6
    out = so.scone_forward(adam.x, shifts_o[0], shifts_o[1], Bc, inputs[1], X)
    ref = (so.loss_from_preds(out, y, train_mask, adam.x, 5e-5), so.accuracy_from_preds(out, y, train_mask, n_nbrs),
           so.loss_from_preds(out, y, test_mask, adam.x, 5e-5), so.accuracy_from_preds(out, y, test_mask, n_nbrs))
    assert abs(res[0] - ref[0]) <= 1e-5 and abs(res[2] - ref[2]) <= 1e-5
    assert res[1] == ref[1] and res[3] == ref[3]
    loss, acc = net.test(inputs, y, test_mask, n_nbrs)
    assert abs(loss - ref[2]) <= 1e-5 and acc == ref[3]
    t2 = net.two_target_accuracy(shifts, inputs, y, test_mask, n_nbrs)
    assert 0.0 <= t2 <= 1.0


def test_graph_replayed_bunch_step_equals_the_eager_step(cfg1, sc1):
    """The same for -model bunch (BunchPlan: fold, fused layers, node readout inside the captured graph): resident micro-batch
    bitwise, host batches to summation order."""
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.synthetic_data_gen import SparseFlows
    shifts, nbrhoods, _ = te.setup_from_complex(sc1, "bunch")
    N = 1000
    flows = SparseFlows(cfg1["flow_ptr"].astype(np.int64), cfg1["flow_idx"].astype(np.int64), cfg1["flow_val"].astype(np.float32), cfg1["E"])
    inputs = [nbrhoods, cfg1["last_nodes"], flows]
    y = cfg1["targets"]

    def make(graph):
        stm.reseed(1030)
        net = stm.Scone_GCN(1, 1e-2, 60, 5e-5, verbose=False)
        net.use_graph = graph
        net.setup(te.bunch_func, [(7, 32)] * 3, shifts, inputs, y, None, cfg1["train_mask"], model_type="bunch")
        with torch.no_grad():
            for w in net.weights:
                w.mul_(30.0)
        return net
    snap = lambda net, loss: (float(loss), net._flat_g.cpu().numpy().copy(), net._flat_w.cpu().numpy().copy())
    res = {}
    for graph in (False, True):
        net = make(graph)
        staged = net.stage(inputs, y, np.arange(10, 58))
        out = [snap(net, net.grad_step_staged(inputs, staged, 48)) for _ in range(3)]
        rs = np.random.RandomState(3)
        for step in range(3):
            m = np.zeros(N, int)
            m[rs.choice(N, 60 - 9 * step, replace=False)] = 1
            net._step = 3 + step
            out.append(snap(net, net.grad_step(inputs, y, m)))
        assert (len(net._graphs) > 0) == graph
        res[graph] = out
    for k, ((la, ga, wa), (lb, gb, wb)) in enumerate(zip(res[False], res[True])):
        if k < 3:
            assert la == lb and np.array_equal(ga, gb) and np.array_equal(wa, wb), k
        else:
            assert abs(la - lb) <= 1e-6 * max(1.0, abs(la)) and np.abs(ga - gb).max() <= 2e-6 * np.abs(ga).max() and np.abs(wa - wb).max() <= 5e-6
    assert np.abs(res[True][0][1]).max() > 1e-6


def test_graph_replayed_step_equals_the_eager_step(cfg1, sc1):
    """The launch-amortised step of small complexes (Scone_GCN._graph_accumulate: the device part of an optimiser step captured
    once into a HIP graph and replayed, host batches staged through fixed-address buffers) against the same steps launched
    eagerly.  Resident micro-batches (grad_step_staged) replay exactly the launches of the eager step: losses, gradients and
    weights over three Adam steps must be IDENTICAL.  Host batches (grad_step) run on the staging buffers' fixed slab count
    (unused trajectories: zero flow, zero target), so the weight-gradient partial sums are cut differently: equal to fp32
    summation order (<= 2e-6 of the largest gradient entry) over four steps on four different batch sizes."""
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.synthetic_data_gen import SparseFlows
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    N = 1000
    flows = SparseFlows(cfg1["flow_ptr"].astype(np.int64), cfg1["flow_idx"].astype(np.int64), cfg1["flow_val"].astype(np.float32), cfg1["E"])
    inputs = [readout, cfg1["last_nodes"], flows]
    y = cfg1["targets"]

    def make(graph):
        stm.reseed(1030)
        net = stm.Scone_GCN(1, 1e-2, 100, 5e-5, verbose=False)
        net.use_graph = graph
        net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, cfg1["train_mask"], model_type="scone")
        with torch.no_grad():
            for w in net.weights:
                w.mul_(20.0)
        return net
    snap = lambda net, loss: (float(loss), net._flat_g.cpu().numpy().copy(), net._flat_w.cpu().numpy().copy())
    res = {}
    for graph in (False, True):                                # resident micro-batch: first call eager + capture, then replays
        net = make(graph)
        staged = net.stage(inputs, y, np.arange(40, 104))
        res[graph] = [snap(net, net.grad_step_staged(inputs, staged, 64)) for _ in range(3)]
        assert (len(net._graphs) > 0) == graph
    for (la, ga, wa), (lb, gb, wb) in zip(res[False], res[True]):
        assert la == lb and np.array_equal(ga, gb) and np.array_equal(wa, wb)
    assert np.abs(res[True][0][1]).max() > 1e-5
    for graph in (False, True):                                # host batches of 100, 93, 86, 79 trajectories: padding slabs too
        net = make(graph)
        rs = np.random.RandomState(5)
        out = []
        for step in range(4):
            m = np.zeros(N, int)
            m[rs.choice(N, 100 - 7 * step, replace=False)] = 1
            net._step = step
            out.append(snap(net, net.grad_step(inputs, y, m)))
        assert (len(net._graphs) > 0) == graph
        res[graph] = out
    for (la, ga, wa), (lb, gb, wb) in zip(res[False], res[True]):
        gmax = np.abs(ga).max()
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la)) and np.abs(ga - gb).max() <= 2e-6 * gmax and np.abs(wa - wb).max() <= 2e-6


def test_train_loop_with_empty_batches(cfg1, sc1):
    """Batch masks that select no training sample (batch size 1 and a drawn test sample: steps 0, 1, 6 and 12 of this stream).
    The reference divides by the empty mask's sum there (STM:54) and poisons the weights with NaN; the mirror skips that gradient
    step and nothing else: the Adam step index stays the loop index (STM:306-310), the epoch-end evaluation still runs."""
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    stm.reseed(1030)
    N, bs, epochs = 12, 1, 2
    sel = np.arange(N)
    shifts, readout, _ = te.setup_from_complex(sc1, "scone")
    inputs = [readout, cfg1["last_nodes"][sel], cfg1["flows"][sel]]
    y, train_mask, test_mask = cfg1["targets"][sel], cfg1["train_mask"][sel], cfg1["test_mask"][sel]
    n_nbrs = sc1.n_nbrs(inputs[1])
    net = stm.Scone_GCN(epochs, 1e-3, bs, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, train_mask, model_type="scone")
    res = net.train(inputs, y, train_mask, test_mask, n_nbrs)

    rs = np.random.RandomState(1030)
    w = [(0.01 * rs.randn(*s)).astype(np.float32).astype(np.float64) for s in so.weight_shapes(1, [(3, 16)] * 3, 1)]
    shifts_o, Bc, X, act = _oracle_scone(cfg1, w, sel, "scone")
    adam = so.Adam(w, 1e-3)
    n_batches = int(train_mask.sum()) // bs
    empties = []
    for i in range(epochs * n_batches):
        bm = so.draw_batch_mask(rs, N, bs, train_mask)
        if int(bm.sum()) == 0:
            empties.append(i)
            continue
        _, g = so.scone_loss_and_grad(adam.x, shifts_o[0], shifts_o[1], Bc, inputs[1], X, y, bm, 5e-5)
        adam.update(i, g)
    assert empties and empties[0] == 0 and (epochs * n_batches - 1) not in empties      # the case this test is about
    for a, b in zip(net.weights, adam.x):
        assert np.isfinite(a.cpu().numpy()).all() and _maxdiff(a.cpu().numpy(), b) <= 5e-6
    out = so.scone_forward(adam.x, shifts_o[0], shifts_o[1], Bc, inputs[1], X)
    assert all(r is not None and np.isfinite(r) for r in res)
    assert abs(res[0] - so.loss_from_preds(out, y, train_mask, adam.x, 5e-5)) <= 1e-5
    assert abs(res[2] - so.loss_from_preds(out, y, test_mask, adam.x, 5e-5)) <= 1e-5


@pytest.mark.parametrize("hidden", [32, 16])             # 16: the slab-pair form of the kernels
@pytest.mark.parametrize("mode", ["dense", "zeros"])
def test_fused_first_layer_gradient_equals_the_separate_kernels(mode, hidden):
    """scn_conv_backward_fused_first (layer 1's backward contracts its input gradient with the shifted input y in registers and
    never writes it) against scn_conv_backward + scn_conv_dw_first on the same step: loss and all ten weight gradients."""
    _need_gpu()
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(6000)
    sc = SimplicialComplex(cx)
    N = 22
    paths = g.generate_random_walks(cx, m=N, seed=4, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=5)
    y = so.onehot_targets(choice, sc.max_degree)
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    inputs = [readout, last, flows]
    res = {}
    keep = ops.FUSE_FIRST
    try:
        for fused in (True, False):
            ops.FUSE_FIRST = fused
            stm.reseed(1030)
            net = stm.Scone_GCN(1, 1e-3, N, 0.0, verbose=False, skip_mode=mode)
            net.setup(te.scone_func, [(3, hidden)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
            with torch.no_grad():
                for w in net.weights:
                    w.mul_(15.0)
            with ops.KernelTimer() as kt:
                loss = float(net.grad_step_staged(inputs, net.stage(inputs, y, np.arange(N)), N, apply=False))
            assert any("dW_first" in k for k in kt.summary()) == fused          # the fused entry point really ran
            res[fused] = (loss, [t.detach().cpu().numpy().astype(np.float64) for t in net._grads])
    finally:
        ops.FUSE_FIRST = keep
    assert abs(res[True][0] - res[False][0]) <= 1e-7 * max(1.0, abs(res[False][0]))
    gmax = max(np.abs(b).max() for b in res[False][1])
    for a, b in zip(res[True][1], res[False][1]):
        assert np.abs(a - b).max() <= 2e-6 * gmax


@pytest.mark.parametrize("drop", [None, "zero_nodes_in", "no_faces_out"])
def test_fused_bunch_layer_operator_matches_scipy(drop):
    """scn_terms_forward / scn_terms_backward (the seven Bunch shifts as one operator on the concatenated row space, blocks =
    patches across the three levels) on random slabs against scipy: out_l = relu(sum_j (S_{j->l} x_j) W[l][j]) for the three
    levels, and on the transposed operator dx_l = (sum_j (S^T dz_j) W[l][j]^T) relu'(aux_l), dW[l][j] = aux_l^T (S^T dz_j) --
    including a level that is identically zero on input and a level that is not wanted on output."""
    _need_gpu()
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(2500)
    sc = SimplicialComplex(cx)
    shifts, nbr, _ = te.setup_from_complex(sc, "bunch")
    plan = ops.get_bunch_plan(shifts, nbr, ops.default_device())
    fwd, bwd = plan._terms_ops()
    assert fwd.plan_info()[0] > 0 and fwd.plan_info()[1] < 4.0           # patches: few staged sources per row
    S, sizes = 3, plan.sizes
    rs = np.random.RandomState(2)
    dev = [s.device_csr().astype(np.float64) for s in shifts]
    SRC, DST = ops.BUNCH_SRC, ops.BUNCH_DST
    xs = [rs.randn(S, n, 4, 32).astype(np.float32) for n in sizes]
    Wk = [(0.2 * rs.randn(32, 32)).astype(np.float32) for _ in range(7)]
    live_in = [drop != "zero_nodes_in", True, True]
    want = [True, True, drop != "no_faces_out"]

    def shift(m, x):                                                     # (rows_dst x rows_src) @ [S, rows_src, 4, 32]
        Sx, R = x.shape[0], x.shape[1]
        y = m @ x.transpose(1, 0, 2, 3).reshape(R, -1).astype(np.float64)
        return y.reshape(m.shape[0], Sx, 4, 32).transpose(1, 0, 2, 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xt = [t(x) if live_in[l] else None for l, x in enumerate(xs)]
    Ws = [[None] * 3 for _ in range(3)]
    for k in range(7):
        if live_in[SRC[k]]:
            Ws[DST[k]][SRC[k]] = t(Wk[k])
    fop = plan._terms_fwd_for(want)
    if drop == "no_faces_out":                       # its own plan: face rows belong to no block, node / edge rows fill them
        assert fop is not fwd and fop.plan_info()[0] < fwd.plan_info()[0] and fop.bins == (16, 48, 0)
    else:
        assert fop is fwd
    outs = fop.forward(xt, Ws, "relu", want)
    for l in range(3):
        if not want[l]:
            assert outs[l] is None
            continue
        ref = sum(shift(dev[k], xs[SRC[k]]) @ Wk[k].astype(np.float64) for k in range(7) if DST[k] == l and live_in[SRC[k]])
        ref = np.maximum(ref, 0)
        assert _maxdiff(outs[l].cpu().numpy(), ref) <= 2e-5
    # backward on the transposed operator: rows = input rows of level a, terms = target levels b
    dzs = [rs.randn(S, n, 4, 32).astype(np.float32) if want[l] else None for l, n in enumerate(sizes)]
    auxs = [np.maximum(rs.randn(S, n, 4, 32), 0).astype(np.float32) if live_in[l] else None for l, n in enumerate(sizes)]
    Wb = [[None] * 3 for _ in range(3)]
    dWb = [[None] * 3 for _ in range(3)]
    for k in range(7):
        a, b = SRC[k], DST[k]
        if auxs[a] is not None and dzs[b] is not None:
            Wb[a][b], dWb[a][b] = t(Wk[k]), torch.full((32, 32), 0.5, device="cuda")
    wantdx = [auxs[l] is not None and any(w is not None for w in Wb[l]) for l in range(3)]
    dxs = ops._terms_backward(bwd, [t(d) if d is not None else None for d in dzs], Wb,
                              [t(a) if a is not None else None for a in auxs], "relu", wantdx, dWb)
    for a in range(3):
        if not wantdx[a]:
            assert dxs[a] is None
            continue
        gk = {k: shift(dev[k].T.tocsr(), dzs[DST[k]]) for k in range(7) if SRC[k] == a and dzs[DST[k]] is not None}
        ref = sum(gk[k] @ Wk[k].astype(np.float64).T for k in gk) * (auxs[a] > 0)
        assert _maxdiff(dxs[a].cpu().numpy(), ref) <= 2e-5
        for k in gk:
            refw = np.einsum("srnc,srnd->cd", auxs[a].astype(np.float64), gk[k]) + 0.5
            assert _maxdiff(dWb[a][DST[k]].cpu().numpy(), refw) <= 2e-5 * max(1.0, np.abs(refw).max())


@pytest.mark.parametrize("model", ["scone", "bunch"])
def test_readout_on_a_hub_node_wider_than_one_wave(model):
    """A node of degree 90: neighbourhoods (TE:279) are 90 wide, more than the 64 lanes the readout kernels give one slot each
    (they returned SCN_ERR_UNSUPPORTED until round 4).  Trajectories that END AT THE HUB (90 live slots) and next to it (the hub
    among the neighbours, 80-odd padding rows inside the logsumexp, TE:151-152): loss and every weight gradient against the oracle."""
    _need_gpu()
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex
    from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
    base = g.random_SC_graph(500)
    hub = base.n_nodes
    near = np.argsort(np.linalg.norm(base.coords - [0.5, 0.5], axis=1))[:90]
    inset = np.zeros(hub + 1, bool)
    inset[near] = True
    spokes = np.stack([near, np.full(len(near), hub)], axis=1)
    rim = base.edges[inset[base.edges[:, 0]] & inset[base.edges[:, 1]]]
    cones = np.concatenate([rim, np.full((len(rim), 1), hub)], axis=1)
    cx = g.Complex(n_nodes=hub + 1, edges=np.unique(np.concatenate([base.edges, spokes]), axis=0),
                   faces=np.unique(np.concatenate([base.faces, cones]), axis=0),
                   coords=np.concatenate([base.coords, [[0.5, 0.5]]]))
    sc = SimplicialComplex(cx)
    D = sc.max_degree
    assert D == 90
    paths = g.generate_random_walks(base, m=10, seed=5)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=2)
    n = len(paths)
    last = np.asarray(last[:n]).copy()
    choice = np.asarray(choice[:n]).copy()
    last[:3] = hub                                                   # end at the hub: all 90 slots live
    choice[:3] = [0, 57, 89]
    last[3:5] = near[:2]                                             # end next to it
    choice[3:5] = 0
    X = flows.todense()[:n].astype(np.float64)
    y = so.onehot_targets(choice, D)
    B1, B2 = g.incidence_matrices(cx)
    if model == "scone":
        w = _rand_weights(so.weight_shapes(1, [(3, 16)] * 3, 1), 0.15, 3)
        L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
        import scipy.sparse as sp
        B1x = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()
        Bc = lambda v: B1x[sc.nbrhoods[v]].toarray()
        ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last, X, y, np.ones(n, int), 0.0)
        shifts, readout, _ = te.setup_from_complex(sc, "scone")
        args = (*shifts, readout, last, X)
        fn = te.scone_func
    else:
        w = _rand_weights(so.weight_shapes(1, [(7, 8)] * 2, 1, "bunch"), 0.3, 5)
        S = [m.tocsr() for m in compute_shift_matrices(B1, B2)]
        ref_loss, ref_g = so.bunch_loss_and_grad(w, S, sc.nbrhoods, last, X, y, np.ones(n, int), 0.0)
        shifts, nbrhoods, _ = te.setup_from_complex(sc, "bunch")
        args = (*shifts, nbrhoods, last, X)
        fn = te.bunch_func
    wt = [torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True) for a in w]
    out = fn(wt, *args)
    assert tuple(out.shape) == (n, D, 1)
    loss = -(out * torch.as_tensor(y, dtype=torch.float32, device="cuda")).sum() / n
    loss.backward()
    assert abs(float(loss.detach()) - ref_loss) <= TOL * max(1.0, abs(ref_loss))
    for k in range(len(w)):
        assert _maxdiff(wt[k].grad.cpu().numpy(), ref_g[k]) <= TOL, "weight %d" % k
