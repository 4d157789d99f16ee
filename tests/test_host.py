"""CPU tests of the host side: sparse generator / operator builder against the golden fixtures and the dense oracle,
layouts, flow containers, dataset folder IO, flag parser.  No GPU, no HIP calls."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import scone_oracle as so
from scone_gcn_amd import synthetic_data_gen as g
from scone_gcn_amd.bunch_model_matrices import compute_shift_matrices
from scone_gcn_amd.complex import Layout, SimplicialComplex, hilbert_index, union_pattern

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cx():
    return g.random_SC_graph(400)


def test_generator_reproduces_reference_complex_bit_for_bit(cx, cfg1):
    assert cx.n_edges == 1001 and cx.n_faces == 649
    assert np.array_equal(cx.edges, cfg1["edges"]) and np.array_equal(cx.faces, cfg1["faces"])
    assert np.array_equal(cx.coords, cfg1["coords"])
    B1, B2 = g.incidence_matrices(cx)
    assert np.array_equal(B1.toarray(), cfg1["B1"]) and np.array_equal(B2.toarray(), cfg1["B2"])
    e2, f2 = g.complex_from_incidence(cfg1["B1"], cfg1["B2"])
    assert np.array_equal(e2, cx.edges) and np.array_equal(f2, cx.faces)


def test_sparse_operators_equal_dense_oracle(cx, cfg1):
    sc = SimplicialComplex(cx)
    L_lo, L_up = so.scone_shifts(cfg1["B1"], cfg1["B2"])
    s = sc.scone_shifts()
    assert np.array_equal(s[0].toarray(), L_lo) and np.array_equal(s[1].toarray(), L_up)
    e = sc.ebli_shifts()
    E1, E2 = so.ebli_shifts(cfg1["B1"], cfg1["B2"])
    assert np.array_equal(e[0].toarray(), E1) and np.array_equal(e[1].toarray(), E2)
    flips = sc.flip_vector(1)
    F = so.flip_matrix(cx.n_edges)
    assert np.array_equal(np.diag(F), flips)
    sf = sc.scone_shifts(flips)
    Lf = so.scone_shifts(cfg1["B1"], cfg1["B2"], F)
    assert np.array_equal(sf[0].toarray(), Lf[0]) and np.array_equal(sf[1].toarray(), Lf[1])
    assert s[0].is_symmetric() and s[1].is_symmetric()


def test_closed_form_bunch_shifts_match_reference(cx):
    gold = np.load(os.path.join(GOLDEN, "cfg1_bunch.npz"))
    B1, B2 = g.incidence_matrices(cx)
    S = compute_shift_matrices(B1, B2)
    for name, M in zip(["S_00", "S_10", "S_01", "S_11", "S_21", "S_12", "S_22"], S):
        ref = so.dense_from_coo(gold[name + "_row"], gold[name + "_col"], gold[name + "_val"], gold[name + "_shape"])
        assert sp.issparse(M) and np.abs(M.toarray() - ref).max() <= 1e-12, name
    sc = SimplicialComplex(cx)
    shifts = sc.bunch_shifts()
    assert [s.shape for s in shifts] == [(400, 400), (400, 1001), (1001, 400), (1001, 1001), (1001, 649), (649, 1001), (649, 649)]
    assert not shifts[3].is_symmetric()              # S_11 is not symmetric


def test_layout_is_a_permutation_and_device_csr_is_conjugated(cx):
    sc = SimplicialComplex(cx)
    lay = sc.layout
    for lvl in range(3):
        assert np.array_equal(np.sort(lay.order[lvl]), np.arange(lay.sizes[lvl]))
        assert np.array_equal(lay.perm[lvl][lay.order[lvl]], np.arange(lay.sizes[lvl]))
    assert not lay.is_identity(0) and not lay.is_identity(1) and not lay.is_identity(2)
    S = sc.scone_shifts()[0]
    D = S.device_csr().toarray()
    P = lay.perm[1]
    assert np.array_equal(D[np.ix_(P, P)], S.toarray())       # D = P S P^T
    # locality: a block of the reordered operator (the layout's own cut points) needs far fewer distinct source rows per
    # row than 64-row windows of the input order; inside a block the rows are sorted by entry count
    def ratio(M, starts):
        M = sp.csr_matrix(M)
        b = np.flatnonzero(starts) if starts is not None else np.arange(0, M.shape[0], 64)
        e = np.append(b[1:], M.shape[0])
        return np.mean([len(np.unique(M.indices[M.indptr[i]:M.indptr[j]])) / (j - i) for i, j in zip(b, e)])
    starts = lay.block_starts[1]
    assert starts is not None and starts[0] == 1 and np.diff(np.flatnonzero(starts)).max() <= 64
    assert ratio(S.device_csr(), starts) < 0.8 * ratio(S.csr, None)
    def padded(nnz, b):      # lane-entries the 8-row groups walk, relative to the stored entries
        tot = 0
        for i, j in zip(b, np.append(b[1:], len(nnz))):
            for k in range(i, j, 8):
                tot += ((nnz[k:min(k + 8, j)].max() + 1) & ~1) * 8
        return tot / nnz.sum()
    nat = np.diff(S.csr.indptr)
    assert padded(np.diff(S.device_csr().indptr), np.flatnonzero(starts)) < 0.92 * padded(nat, np.arange(0, len(nat), 64))
    # no coordinates -> reverse Cuthill-McKee fallback still yields a valid layout
    sc2 = SimplicialComplex(g.Complex(cx.n_nodes, cx.edges, cx.faces, None, cx.valid_idxs))
    assert np.array_equal(np.sort(sc2.layout.order[1]), np.arange(cx.n_edges))


def test_block_layout_removes_most_lds_bank_conflicts_of_the_gather(cx):
    """scn_plan_gather_stats (host-only): with the plan's slot colours and entry order a lane-group read of the gather costs
    close to one LDS cycle; with slots in row order and entries in CSR order it costs close to two."""
    import ctypes
    from scone_gcn_amd import _lib
    sc = SimplicialComplex(g.random_SC_graph(3000))
    order, starts = sc.layout.order[1], sc.layout.block_starts[1]
    lower, upper = abs(sc.B1.T @ sc.B1).tocsr(), abs(sc.B2 @ sc.B2.T).tocsr()
    pat = (lower + upper).tocsr()[order][:, order].tocsr()
    pat.sort_indices()
    val1 = np.asarray(upper[order][:, order].tocsr()[pat.nonzero()]).ravel().astype(np.float32)
    rowptr, col = np.ascontiguousarray(pat.indptr, np.int32), np.ascontiguousarray(pat.indices, np.int32)
    as_p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    lib = _lib.load()
    for v1, st in ((val1, starts), (None, None)):
        out = np.zeros(4, np.int64)
        _lib.check(lib.scn_plan_gather_stats(pat.shape[0], as_p(rowptr), as_p(col), None if v1 is None else v1.ctypes.data, 1,
                                             None if st is None else st.ctypes.data, out.ctypes.data), "scn_plan_gather_stats")
        reads, legacy, planned, blocks = (int(v) for v in out)
        assert blocks >= pat.shape[0] // 64 and reads > 0
        assert legacy > 1.5 * reads                       # the conflicts this layout is there to remove (status 0: every block's
                                                          # layout also passed the library's own entry-by-entry check)
        assert reads <= planned < 1.2 * reads
    bad = col.copy()
    bad[0] = pat.shape[0]
    assert lib.scn_plan_gather_stats(pat.shape[0], as_p(rowptr), as_p(bad), None, 1, None, out.ctypes.data) != 0


def test_refine_order_deals_the_sorted_row_groups_evenly_to_the_simds():
    """scn_plan_refine_order: inside a block the rows are sorted by entry count, and the eight 8-row groups then go in the order
    (width ranks) 3 2 1 0 4 5 6 7, so that the two groups a SIMD hosts (wave i on SIMD i mod 4) pair widest with narrowest."""
    import ctypes
    from scone_gcn_amd import _lib
    n = 64                                               # one block (64 rows x <= 16 entries fit its ELL tile): row i has 2 + (i * 7) % 15 entries
    rows = [sorted({(i + 1 + 3 * k) % n for k in range(2 + (i * 7) % 15)} - {i}) for i in range(n)]
    rowptr = np.zeros(n + 1, np.int32)
    rowptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    order, starts = np.empty(n, np.int32), np.zeros(n, np.uint8)
    as_p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    _lib.check(_lib.load().scn_plan_refine_order(n, as_p(rowptr), as_p(col), 1, as_p(order), starts.ctypes.data), "refine")
    assert np.array_equal(np.sort(order), np.arange(n)) and starts[0] == 1 and starts[1:].sum() == 0
    width = np.diff(rowptr)[order].reshape(8, 8)
    assert all((np.diff(g) <= 0).all() for g in width)   # rows sorted inside every group
    gmax = width.max(axis=1)
    assert list(np.argsort(-gmax, kind="stable")) == [3, 2, 1, 0, 4, 5, 6, 7]
    pair = gmax[:4] + gmax[4:]                            # what the SIMDs of the 8-wave kernels carry
    assert pair.max() - pair.min() <= (gmax.max() - gmax.min()) // 2 + 2


def test_hilbert_index_is_a_bijection_on_a_grid():
    xs, ys = np.meshgrid(np.arange(16), np.arange(16))
    d = hilbert_index(xs.ravel(), ys.ravel(), order=4)
    assert np.array_equal(np.sort(d), np.arange(256))
    o = np.argsort(d)
    steps = np.abs(np.diff(xs.ravel()[o])) + np.abs(np.diff(ys.ravel()[o]))
    assert np.all(steps == 1)                                  # consecutive curve points are grid neighbours


def test_union_pattern_values(cx):
    sc = SimplicialComplex(cx)
    lo, up = (s.device_csr() for s in sc.scone_shifts())
    rowptr, cols, (v0, v1) = union_pattern([lo, up])
    U = sp.csr_matrix((v0, cols, rowptr), shape=lo.shape)
    V = sp.csr_matrix((v1, cols, rowptr), shape=lo.shape)
    assert np.array_equal(U.toarray(), lo.toarray()) and np.array_equal(V.toarray(), up.toarray())
    assert len(cols) == lo.nnz                                 # pattern(L_upper) inside pattern(L_lower)
    for r in range(0, lo.shape[0], 97):
        c = cols[rowptr[r]:rowptr[r + 1]]
        assert np.all(np.diff(c) > 0)


def test_sparse_flows_roundtrip_and_reference_flow_values(cx, cfg1):
    X = cfg1["flows"][:40]
    sf = g.SparseFlows.fromdense(X)
    assert np.array_equal(sf.todense(), X.astype(np.float32))
    sub = sf.select(np.array([3, 7, 7, 0]))
    assert np.array_equal(sub.todense(), X[[3, 7, 7, 0]].astype(np.float32))
    # path_to_flow semantics (SDG:327-344): +1 along increasing node number, -1 against it
    path = [int(cx.edges[10, 1]), int(cx.edges[10, 0])]
    f = g.paths_to_flows(cx, [path]).todense()[0, :, 0]
    assert f[10] == -1 and np.abs(f).sum() == 1


def test_walks_and_dataset_shapes(cx):
    paths = g.generate_random_walks(cx, m=30, seed=5)
    assert all(len(p) == len(set(p)) for p in paths)           # simple paths (SDG:239)
    adj = g.adjacency(cx)
    for p in paths[:5]:
        assert all(adj[p[i], p[i + 1]] == 1 for i in range(len(p) - 1))
    flows, choice, last, tnodes, prefixes = g.path_dataset(cx, paths, seed=1)
    nbr, deg = g.neighborhood_table(cx)
    assert len(flows) == 30 and np.all(nbr[last, choice] == tnodes)
    assert all(pre[-1] == l for pre, l in zip(prefixes, last))
    assert g.calibrate_n_points(1_000_000) == 369004


def test_dataset_folder_roundtrip(tmp_path, monkeypatch):
    from scone_gcn_amd import dataset_io
    monkeypatch.chdir(tmp_path)
    cx = dataset_io.generate_dataset(150, 40, folder="t", holes=True)
    X, (B1, B2), y, train_mask, test_mask, coords, last, tnodes = dataset_io.load_dataset("trajectory_data_1hop_t")
    assert X.shape == (40, cx.n_edges, 1) and y.shape[0] == 40 and y.shape[2] == 1
    assert train_mask.sum() == 32 and np.array_equal(train_mask + test_mask, np.ones(40))
    sc = SimplicialComplex.from_incidence(B1, B2, coords=coords)
    assert np.array_equal(sc.cx.edges, cx.edges) and np.array_equal(sc.cx.faces, cx.faces)
    assert np.all(y.sum(axis=(1, 2)) == 1)
    X2, _, y2, *_ = dataset_io.load_dataset("trajectory_data_2hop_t")
    assert np.all(np.abs(X2).sum(axis=(1, 2)) == np.abs(X).sum(axis=(1, 2)) + 1)   # 2-hop prefix has one more edge
    rev, rt, rl = dataset_io.load_reverse("trajectory_data_1hop_t")
    assert rev.shape == X.shape and len(rl) == 40
    # sparse storage path
    monkeypatch.setattr(dataset_io, "DENSE_LIMIT", 10)
    dataset_io.generate_dataset(150, 12, folder="s", holes=True)
    Xs, (B1s, B2s), *_ = dataset_io.load_dataset("trajectory_data_1hop_s")
    assert isinstance(Xs, g.SparseFlows) and sp.issparse(B1s) and os.path.exists("trajectory_data_1hop_s/B1.npz")


def test_hyperparams_parser_matches_reference_flags():
    from scone_gcn_amd.trajectory_experiments import hyperparams
    d = hyperparams(["prog"])
    assert d["model"] == "scone" and d["epochs"] == 1000 and d["learning_rate"] == 0.001 and d["weight_decay"] == 0.00005
    assert d["batch_size"] == 100 and d["hidden_layers"] == [(3, 16)] * 3 and d["data_folder_suffix"] == "working"
    d = hyperparams(["prog", "-model", "bunch", "-hidden_layers", "7_32_7_32_7_32", "-epochs", "5", "-flip_edges", "1",
                     "-model_name", "m1"])
    assert d["model"] == "bunch" and d["hidden_layers"] == [(7, 32)] * 3 and d["epochs"] == 5.0
    assert d["flip_edges"] == 1.0 and d["model_name"] == "m1"


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "scone_gcn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").lower() or f == "__init__.py" or \
                    all("import" not in line for line in src.splitlines() if "oracle" in line.lower()), f


def _buoy():
    from scone_gcn_amd import buoy_data as bd
    gld = np.load(os.path.join(GOLDEN, "buoy.npz"))
    trajs = [gld["traj_nodes"][gld["traj_ptr"][i]:gld["traj_ptr"][i + 1]].astype(int).tolist()
             for i in range(len(gld["traj_ptr"]) - 1)]
    return bd.buoy_dataset(gld["elist"].astype(np.int64), gld["tlist"].astype(np.int64), gld["coords"], trajs)


def test_ocean_drifter_dataset_counts():
    """Config 3 (BASELINE.json): the numbers SURVEY.md section 2 row 10 measured with the reference's converter rules."""
    cx, paths, flows, choice, last, tnodes, train_mask, test_mask = _buoy()
    assert (cx.n_nodes, cx.n_edges, cx.n_faces) == (133, 320, 186)
    assert len(paths) == 200 and train_mask.sum() == 160 and test_mask.sum() == 40
    assert all(5 - 0 <= len(p) <= 10 for p in paths)
    nbr, deg = g.neighborhood_table(cx)
    assert nbr.shape[1] == 6 and abs(deg.mean() - 4.8) < 0.05
    assert np.all(nbr[last, choice] == tnodes)
    B1, B2 = g.incidence_matrices(cx)
    assert abs(B1 @ B2).max() == 0


def test_jld2_reader_on_reference_file_if_present():
    path = "/root/reference/ocean_drifters_data/dataBuoys.jld2"
    if not os.path.exists(path):
        pytest.skip("reference data file not on this machine")
    from scone_gcn_amd import buoy_data as bd
    elist, tlist, coords, trajs = bd.read_buoy_file(path)
    gld = np.load(os.path.join(GOLDEN, "buoy.npz"))
    assert np.array_equal(elist, gld["elist"]) and np.array_equal(tlist, gld["tlist"]) and len(trajs) == 339
    assert bd.strip_paths([[1, 2, 1, 3, 4, 3, 5]]) == [[1, 3, 5]]


def test_work_list_layout_and_square_detection(cx):
    """Host logic of the zero-skipping mode: the (block, slab) list structure handed to the kernels, and the S_upper == S_lower^2
    test that routes Ebli to the composed plan."""
    import scipy.sparse as sp
    import torch
    from scone_gcn_amd import ops
    A = sp.csr_matrix(np.array([[0, 1, 0, 1, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 0]], np.int32))     # [slab, block]
    wl = ops.WorkList(A, torch.device("cpu"))
    assert wl.n_work == 2 and wl.items == 3
    assert wl.block[:2].tolist() == [1, 3] and wl.ptr[:3].tolist() == [0, 1, 3] and wl.slab[:3].tolist() == [0, 0, 1]
    empty = ops.WorkList(sp.csr_matrix((3, 5), dtype=np.int32), torch.device("cpu"))
    assert empty.n_work == 0 and empty.items == 0 and empty.block.data_ptr() != 0
    sc = SimplicialComplex(cx)
    L1, L1sq = sc.ebli_shifts()
    assert ops._is_square_of(L1sq, L1) and not ops._is_square_of(L1, L1sq)
    lo, up = sc.scone_shifts()
    assert not ops._is_square_of(up, lo)


def test_data_setup_returns_the_references_eleven_values(tmp_path, monkeypatch):
    """TE:311: inputs_all, y_all, train_mask, test_mask, shifts, G_undir, E_lookup, nbrhoods, n_nbrs, target_nodes_all,
    prefixes; under -flip_edges the flips come from the trainer's global stream reseeded with 1 (TE:214-219), the inputs are
    X F (TE:292-296) and the prefixes still describe the stored (unflipped) flows."""
    from scone_gcn_amd import dataset_io, scone_trajectory_model as stm, trajectory_experiments as te
    monkeypatch.chdir(tmp_path)
    cx = dataset_io.generate_dataset(150, 30, folder="d", holes=True)
    hp = te.hyperparams(["prog", "-model", "scone"])
    out = te.data_setup(hops=(1, 2), load=True, folder_suffix="d", hp=hp)
    assert len(out) == 11
    inputs_all, y_all, train_mask, test_mask, shifts, G, E_lookup, nbrhoods, n_nbrs, tn_all, prefixes = out
    assert len(inputs_all) == 2 and len(shifts) == 2 and len(prefixes) == 30
    assert len(G.nodes) == cx.n_nodes and len(G.edges) == cx.n_edges and G.degree[3] == len(G[3])
    assert max(G.degree, key=lambda x: x[1])[1] == nbrhoods.shape[1]                       # TE:273
    assert E_lookup[tuple(cx.edges[5])] == 5
    last = inputs_all[0][1]
    assert np.array_equal(n_nbrs, [len(G[n]) for n in last])
    X = inputs_all[0][2]
    for i in (0, 7, 29):                                                                   # prefixes <-> flows (SDG:299-344)
        assert prefixes[i][-1] == last[i]
        assert np.array_equal(g.paths_to_flows(cx, [prefixes[i]]).todense()[0], np.asarray(X)[i])
    # Bconds object answers like the closure of TE:298-303
    Bc = inputs_all[0][0]
    B1 = g.incidence_matrices(cx)[0].toarray()
    n0 = int(last[0])
    ref = np.concatenate([B1, np.zeros((1, B1.shape[1]))])[nbrhoods[n0]]
    assert np.array_equal(Bc(n0), ref)
    # prefixes reconstructed from the flows when the folder has no prefixes file
    os.remove("trajectory_data_1hop_d/prefixes.npz")
    assert te.data_setup(hops=(1,), load=True, folder_suffix="d", hp=hp)[10] == prefixes
    # -flip_edges: global stream reseeded with 1, flips drawn from it, weights continue it
    hpf = te.hyperparams(["prog", "-flip_edges", "1"])
    outf = te.data_setup(hops=(1,), load=True, folder_suffix="d", hp=hpf)
    rs = np.random.RandomState(1)
    flips = rs.choice([1, -1], size=cx.n_edges, replace=True, p=[0.8, 0.2])
    assert np.array_equal(np.asarray(outf[0][0][2])[:, :, 0], np.asarray(X)[:, :, 0] * flips[None, :])
    assert stm._RNG.randn() == rs.randn()
    assert outf[10] == prefixes
    stm.reseed(1030)


def test_probed_bcond_closure_yields_the_readout_tables_of_the_native_object(cx):
    """A plain closure n -> B1_jax[nbrhoods[n]] (TE:298-303) is probed by calling it; the tables built from the answers give the
    same logits as the native Bconds tables (NumPy evaluation of the readout formula, TE:151)."""
    from scone_gcn_amd.complex import ProbedBconds, adopt_bconds, adopt_shift, identity_layout
    sc = SimplicialComplex(cx, reorder=False)
    B1 = g.incidence_matrices(cx)[0].toarray()
    B1x = np.concatenate([B1, np.zeros((1, B1.shape[1]))])
    nb = sc.nbrhoods
    fn = lambda n: B1x[nb[n]]
    lay = identity_layout((1, cx.n_edges, 1))
    pb = adopt_bconds(fn, cx.n_edges, lay)
    assert adopt_bconds(fn, cx.n_edges, lay) is pb and isinstance(pb, ProbedBconds)
    last = np.array([5, 17, 5, 200, 17])
    rows = pb.prepare(last)
    assert rows.tolist() == [0, 1, 0, 2, 1] and pb.version == 1
    assert pb.prepare(np.array([17])).tolist() == [1] and pb.version == 1                  # nothing new: no rebuild
    ptr, edge, sign, edge_nodes = pb.incidence_tables()
    rs = np.random.RandomState(0)
    h = rs.randn(cx.n_edges)
    for r, n in zip(rows, last):
        got = np.array([0.0 if v < 0 else float(sign[ptr[v]:ptr[v + 1]] @ h[edge[ptr[v]:ptr[v + 1]]]) for v in pb.nbrhoods[r]])
        assert np.allclose(got, fn(n) @ h, atol=1e-6)
    for e in range(cx.n_edges):                                                            # endpoints: -2 = not probed
        ks = [k for k in edge_nodes[e] if k >= 0]
        for k in ks:
            assert e in edge[ptr[k]:ptr[k + 1]]
    bad = np.zeros((2, cx.n_edges))
    bad[0, :2] = 1.0                                    # edge 0 in two different rows with the SAME sign: not an incidence matrix
    bad[1, 0] = 1.0
    pbad = ProbedBconds(lambda n: bad, cx.n_edges, lay)
    pbad.prepare([0])
    with pytest.raises(TypeError):
        pbad.incidence_tables()
    with pytest.raises(TypeError):
        ProbedBconds(lambda n: np.ones((3, 5)), cx.n_edges, lay).prepare([0])            # wrong row length
    # dense shifts are wrapped once, in the caller's order
    L = (g.incidence_matrices(cx)[0].T @ g.incidence_matrices(cx)[0]).toarray()
    sh = adopt_shift(L, lay, 1, 1)
    assert adopt_shift(L, lay, 1, 1) is sh and np.array_equal(sh.toarray(), L) and np.array_equal(sh.device_csr().toarray(), L)


def test_bench_refuses_a_rank_count_it_cannot_have():
    """`bench.py --gpus N` never runs a silent single-GPU job: without N visible GPUs it exits non-zero before touching
    anything, and a launcher environment whose WORLD_SIZE differs from --gpus is an error too."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() < 64:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True)
        assert r.returncode == 2 and "GPU(s) are visible" in r.stderr and r.stdout.strip() == ""
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "2", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_hidden_width_promotion_pads_and_cuts_back():
    """ops.promoted_width / promote_weights / demote_grads (host logic of the mixed-width path): which stacks are padded, to
    what, and that padding a matrix and cutting its gradient back are inverse on the real entries."""
    import torch
    from scone_gcn_amd import ops
    assert ops.promoted_width([16, 16, 16]) is None and ops.promoted_width([32, 32]) is None
    assert ops.promoted_width([32, 16]) == 32 and ops.promoted_width([16, 32]) == 32
    assert ops.promoted_width([8, 8]) == 16 and ops.promoted_width([16, 24, 8]) == 32
    assert ops.promoted_width([40, 40]) is None and ops.promoted_width([32, 64]) is None
    rs = np.random.RandomState(0)
    shapes = [(1, 32)] * 3 + [(32, 16)] * 3 + [(16, 1)]
    w = [torch.tensor(rs.randn(*s_), dtype=torch.float32) for s_ in shapes]
    wp = ops.promote_weights(w, 3, 1, 32)
    assert [tuple(t.shape) for t in wp] == [(1, 32)] * 3 + [(32, 32)] * 3 + [(32, 1)]
    assert all(wp[i] is w[i] for i in range(3))                         # nothing to pad: the caller's tensor itself
    for a, b in zip(w, wp):
        assert torch.equal(b[:a.shape[0], :a.shape[1]], a)
        assert int((b != 0).sum()) == int((a != 0).sum())               # the padding is exact zeros
    # a padded stack computes the same function: dense check of one layer  act(X W) with zero-padded channels
    X = torch.tensor(rs.randn(5, 32), dtype=torch.float32)
    h = torch.tanh(X @ w[3])
    hp = torch.tanh(X @ wp[3])
    assert torch.equal(hp[:, :16], h) and float(hp[:, 16:].abs().max()) == 0.0
    assert torch.allclose(hp @ wp[6], h @ w[6], atol=1e-6)
    g = [torch.zeros_like(t) for t in w]
    gp = [torch.full_like(t, 2.0) if t is not t0 else g0 for t, t0, g0 in zip(wp, w, g)]
    ops.demote_grads(g, gp)
    assert all(float(g[i].sum()) == 0.0 for i in range(3)) and all(bool((g[i] == 2.0).all()) for i in range(3, 7))


def test_bench_roofline_is_per_kernel_family_and_cites_traffic_only_for_the_measured_sources(tmp_path, monkeypatch):
    """bench.py's bookkeeping (no GPU): the dominant kernel is picked per FAMILY (the plain and the fused-first backward are
    one template), every variant keeps its own fraction, and committed PMC traffic is cited only while the kernel sources
    hash to what the profile was measured on."""
    import json
    import bench
    from scone_gcn_amd import _lib
    table = {"conv_fwd c32->32": {"launches": 64, "avg_ms": 8.0, "alg_bytes": 32.8e9, "GB/s": 4100.0},
             "conv_bwd c32->32": {"launches": 32, "avg_ms": 11.7, "alg_bytes": 49.0e9, "GB/s": 4188.0},
             "conv_bwd c32->32 + dW_first": {"launches": 32, "avg_ms": 11.2, "alg_bytes": 34.7e9, "GB/s": 3098.0},
             "conv_fwd c1->32": {"launches": 32, "avg_ms": 3.1, "alg_bytes": 16.9e9, "GB/s": 5450.0},
             "readout": {"launches": 32, "avg_ms": 0.1, "alg_bytes": None, "GB/s": None}}
    r = bench.roofline_of(table, 128)
    assert r["kernel"] == "conv_bwd c32->32" and r["launches_per_step"] == 64
    assert set(r["variants"]) == {"conv_bwd c32->32", "conv_bwd c32->32 + dW_first"}
    want = (32 * 49.0e9 + 32 * 34.7e9) / ((32 * 11.7 + 32 * 11.2) * 1e-3) / 8.0e12
    assert abs(r["frac"] - want) < 1e-12
    assert abs(r["variants"]["conv_bwd c32->32 + dW_first"]["frac"] - 34.7e9 / 11.2e-3 / 8.0e12) < 1e-12
    assert 0.5 < r["share_of_step_kernel_time"] < 0.6
    prof = tmp_path / "profiles"
    prof.mkdir()
    doc = {"main_bench": {"launch": "test", "kernels": {
        "scn::bwd<a>": {"hbm_bytes_per_launch": 55.0e9, "launches_seen": 8, "timer_keys": ["conv_bwd c32->32"]},
        "scn::bwd<b>": {"hbm_bytes_per_launch": 42.0e9, "launches_seen": 8, "timer_keys": ["conv_bwd c32->32 + dW_first"]}}},
        "kernel_sources_sha": "stale"}
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, src = bench.measured_traffic("main_bench", list(r["variants"]))
    assert t is None and "not cited" in src
    doc["kernel_sources_sha"] = _lib.sources_sha()
    (prof / "r09_pmc_traffic.json").write_text(json.dumps(doc))
    t, src = bench.measured_traffic("main_bench", list(r["variants"]))
    assert t == 48.5e9 and "r09_pmc_traffic.json" in src


def _synthetic_full_record(n_configs=12, pad=400):
    """A bench record as large as the largest the script can produce (every optional object present, long strings)."""
    var = {"launches": 32, "launch_ms": 11.663105249404907, "algorithmic_bytes_per_launch": 49129194308.0, "frac": 0.5265449601265788}
    roof = {"bound": "hbm", "kernel": "conv_bwd c32->32", "achieved": 3652.8798765176844, "peak": 8000.0, "unit": "GB/s",
            "frac": 0.4566099845647106, "traffic": 48065196984.0, "launch_ms": 11.493759229779243, "launches_per_step": 64,
            "algorithmic_bytes_per_launch": 41985321796.0, "units_per_launch": 128, "share_of_step_kernel_time": 0.5431770577135748,
            "variants": {"conv_bwd c32->32": var, "conv_bwd c32->32 + dW_first": var},
            "variant0": "conv_bwd c32->32: 0.527 of peak, 11.663 ms x 32", "variant1": "conv_bwd c32->32 + dW_first: 0.385 of peak, 11.324 ms x 32",
            "traffic_source": "profiles/r04_pmc_traffic.json [main_bench] " + "x" * pad, "measured_copy_GBps": 4601.970897069156,
            "spmm_dual_frac": 0.6503029651545592, "spmm_dual_GBps": 5202.423721236473, "spmm_dual_ms": 9.44352035522461,
            "spmm_dual_algorithmic_bytes": 49129194308.0, "spmm_dual_traffic": 53684055637.333336, "spmm_dual_x": "[32, 996634, 128]",
            "dense_random": {"conv_fwd c32->32": {"ms": 10.2, "GB/s": 3208.1, "frac": 0.401}},
            "dense_random_fwd_frac": 0.4010171178198363, "dense_random_fwd_ms": 10.224109331766764,
            "dense_random_bwd_frac": 0.3987552937851131, "dense_random_bwd_ms": 15.400796890258789}
    kernels = {"conv_fwd c%d->32" % i: {"launches": 32, "avg_ms": 3.3347, "GB/s": 5000.1} for i in range(8)}
    side = {"workload": "w" * pad, "model": "scone", "value": 5486.308846631601, "ms_per_step": 23.330804659053683, "unit": "trajectories/s",
            "roofline": dict(roof), "kernels": kernels, "step_model": {"a": 1.0}, "layer_kernels_step": {"ms_per_step": 0.119, "value": 1.0}}
    return {"metric": "trajectories/sec fwd+bwd, 3-layer SCoNe |E|~1M batch=4096; SpMM HBM GB/s", "value": 3035.612345678, "unit": "trajectories/s",
            "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1349.3123456, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic complex " + "y" * pad, "edges": 996634, "nodes": 369004, "faces": 664069, "hidden": 32,
                       "global_batch": 4096, "per_gpu_batch": 4096, "micro_batch": 128, "parallelism": "dp1", "collective": "nccl " + "c" * pad,
                       "batch_seed": 1030, "nnz_lower": 11554450, "nnz_upper": 4981048, "nnz_pattern": 11554450},
            "roofline": roof,
            "cpu_baseline": {"value": 0.27999366080219046, "unit": "trajectories/s", "cores": 128, "kind": "port", "dtype": "f32",
                             "host_cpus": 256, "sample": "s" * (3 * pad), "dense_faithful_configs0": {"optimiser_steps_per_s": 0.29, "sample": "t" * pad}},
            "replicas_identical": True, "loss": 2.817912345678, "validation": {"loss": 2.8, "replicas_identical": True,
            "max_abs_weight_deviation_from_rank0": 0.0, "weights_sum": -0.1, "weights_l2": 1.0, "optimiser_steps_taken": 25, "note": "n" * pad},
            "step_model": {"kernel_algorithmic_bytes_per_trajectory": 1.3e9, "kernel_model_frac_of_hbm_peak_whole_step": 0.49, "note": "n" * pad},
            "kernels": kernels, "setup_s": 100.0, "weak_scaling": {"per_gpu_batch": 512, "global_batch": 512, "value": 3043.9, "ms_per_step": 168.2},
            "zero_skipping": {m: {"value": 41115.0, "ms_per_step": 12.4, "note": "z" * pad} for m in ("zeros", "field")},
            "spmm_dual": {"GB/s": 5202.4}, "parity": {"n": 4, "max_err": 3.4e-7, "tol": 1e-5, "pass": True, "oracle": "o" * pad},
            "configs": {"config %d %s" % (i, "k" * 20): dict(side) for i in range(n_configs)}}


def test_bench_prints_a_compact_last_line_the_driver_can_keep(capsys, tmp_path, monkeypatch):
    """The driver keeps an 8 KB tail of stdout and parses the last line (round 4's 21.5 KB single line came back `parsed: null`).
    bench.emit prints the full record on a `DETAIL {...}` line, writes it to bench_full.json / --out, and ends with a compact line of at
    most bench.COMPACT_LIMIT characters that still carries the headline, `config`, a flat `roofline` and `cpu_baseline`."""
    import json
    import bench
    full = _synthetic_full_record()
    assert len(json.dumps(full)) > 40000                      # far beyond what the driver keeps
    for rec in (full, _synthetic_full_record(n_configs=60, pad=2000)):
        c = bench.compact_line(rec)
        text = json.dumps(c, separators=(",", ":"))
        assert len(text) <= bench.COMPACT_LIMIT < 8000, len(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "loss", "replicas_identical"):
            assert k in c, k
        assert c["value"] == rec["value"] and c["ms_per_step"] == rec["ms_per_step"] and c["loss"] == rec["loss"]
        assert c["config"]["workload"] and c["config"]["edges"] == 996634 and "model" not in c["config"]
        r = c["roofline"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launches_per_step",
                  "algorithmic_bytes_per_launch", "units_per_launch", "variant0", "variant1", "spmm_dual_frac", "spmm_dual_ms",
                  "spmm_dual_traffic", "dense_random_fwd_frac", "dense_random_bwd_frac"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and all(not isinstance(v, (dict, list)) for v in r.values())
        b = c["cpu_baseline"]
        assert b["kind"] == "port" and b["cores"] == 128 and b["value"] > 0 and b["sample"]
    c = bench.compact_line(full)                              # the ordinary record keeps its side configurations
    assert len(c["configs"]) == 12 and set(next(iter(c["configs"].values()))) >= {"value", "ms_per_step", "kernel", "frac"}
    # a record without the optional objects (N > 1 ranks, --extras 0) compacts too
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config", "loss", "replicas_identical")}
    bare.update(roofline=None, cpu_baseline=None)
    assert bench.compact_line(bare)["roofline"] is None
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full, str(tmp_path / "sub" / "copy.json"))
    out = capsys.readouterr().out.splitlines()
    assert out[-2].startswith("DETAIL {") and json.loads(out[-2][7:]) == full
    assert json.loads(out[-1]) == bench.compact_line(full) and len(out[-1]) <= bench.COMPACT_LIMIT
    assert json.load(open(tmp_path / "bench_full.json")) == full and json.load(open(tmp_path / "sub" / "copy.json")) == full


def test_one_launch_step_is_taken_where_it_was_measured_to_pay():
    """ops.small_step_pays: the size / batch rule of the one-launch small-complex step against the measurements it encodes
    (profiles/r04_small_step_ab.txt, the table above ops.SMALL_STEP)."""
    from scone_gcn_amd import ops
    pays = ops.small_step_pays
    assert pays(319, 160) and pays(319, 512) and pays(319, 1000) and not pays(319, 2000)
    assert pays(639, 100) and pays(639, 512) and not pays(639, 1000)
    assert pays(822, 100) and pays(926, 100) and not pays(926, 300)
    assert pays(1001, 200) and pays(1001, 256) and not pays(1001, 512)
    # two workgroups per trajectory (|E| > 384, 2 x trajectories <= CUs): the fastest form at every size it was measured on
    # (profiles/r05_small_pair_ab.txt), the reference's own batch (|E| = 1001, 100 trajectories) included
    assert pays(1001, 100) and pays(1001, 128) and pays(1001, 32) and pays(1106, 100) and pays(498, 64)
    assert pays(1001, 160)                                # (past that form: the one-workgroup rule of round 4)
    keep = ops.SMALL_STEP_MAX_EDGES
    try:
        ops.SMALL_STEP_MAX_EDGES = 1 << 30               # SCN_SMALL_STEP=force
        assert pays(1001, 100) and pays(1001, 512)
    finally:
        ops.SMALL_STEP_MAX_EDGES = keep


def test_native_batch_assembler_fills_the_staging_words_like_the_numpy_reference():
    """scn_host_stage_batch (host-only entry point of the library) against a NumPy assembly of the same staging words: ragged flows
    of a batch as (trajectory slot, device edge row, value) triples, last nodes, targets / total, zeros elsewhere; a batch that does
    not fit is refused."""
    import ctypes
    from scone_gcn_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(4)
    N, D, E = 40, 7, 90
    lens = rs.randint(0, 9, size=N)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    edge = rs.randint(0, E, size=ptr[-1]).astype(np.int32)
    val = rs.randn(ptr[-1]).astype(np.float32)
    last = rs.randint(0, 30, size=N).astype(np.int32)
    y = rs.rand(N, D).astype(np.float32)
    e_cap, n_cap, total = 64, 12, 9.0
    n_words = 3 * e_cap + n_cap + n_cap * D
    for traj in ([3, 17, 4, 4, 39], [], [0]):
        traj = np.asarray(traj, np.int32)
        out = np.full(n_words, -1, np.int32)
        got = lib.scn_host_stage_batch(len(traj), traj.ctypes.data, N, ptr.ctypes.data, edge.ctypes.data, val.ctypes.data, last.ctypes.data,
                                       y.ctypes.data, D, total, e_cap, n_cap, out.ctypes.data)
        ref = np.zeros(n_words, np.int32)
        k = 0
        for j, n in enumerate(traj):
            for t in range(ptr[n], ptr[n + 1]):
                ref[k], ref[e_cap + k] = j, edge[t]
                ref[2 * e_cap + k] = val[t:t + 1].view(np.int32)[0]
                k += 1
            ref[3 * e_cap + j] = last[n]
            ref[3 * e_cap + n_cap + j * D:3 * e_cap + n_cap + (j + 1) * D] = (y[n].astype(np.float64) / total).astype(np.float32).view(np.int32)
        assert got == k and np.array_equal(out, ref)
    big = np.arange(N, dtype=np.int32)
    out = np.zeros(n_words, np.int32)
    assert lib.scn_host_stage_batch(N, big.ctypes.data, N, ptr.ctypes.data, edge.ctypes.data, val.ctypes.data, last.ctypes.data, y.ctypes.data,
                                    D, total, e_cap, n_cap, out.ctypes.data) == _lib.SCN_ERR_UNSUPPORTED
    for bad in ([3, N], [-1], [2, 5, 1 << 20]):                 # an index outside the data set is refused before anything is read through it
        traj = np.asarray(bad, np.int32)
        assert lib.scn_host_stage_batch(len(traj), traj.ctypes.data, N, ptr.ctypes.data, edge.ctypes.data, val.ctypes.data, last.ctypes.data,
                                        y.ctypes.data, D, total, e_cap, n_cap, out.ctypes.data) == _lib.SCN_ERR_BAD_ARG
