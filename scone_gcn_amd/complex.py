"""Host-side simplicial complex, shift operators and readout tables (NumPy / scipy.sparse only).

Replaces the dense operator construction of the reference's data_setup (TE:240-257: L1_lower = B1^T B1,
L1_upper = B2 B2^T, optional F L F flip, Ebli [L1, L1^2], Bunch S_00..S_22) and the readout inputs
(TE:270-303: nbrhoods, n_nbrs, B1_jax, Bconds_func) with sparse equivalents that scale to |E| ~ 1M.

Row order.  Device tensors keep each level's rows (nodes / edges / faces) in a locality order (Hilbert
curve over the simplex centroids when coordinates are known, reverse Cuthill-McKee otherwise) so that the
LDS-blocked kernels find a block's gather sources close together.  The permutation is internal: every
boundary object (Shift, Bconds, flows, nbrhoods) speaks the caller's original indices.
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

from .bunch_model_matrices import compute_shift_matrices
from .synthetic_data_gen import Complex, incidence_matrices, neighborhood_table, complex_from_incidence


def hilbert_index(x, y, order=16):
    """Hilbert-curve index of integer grid points (vectorised)."""
    x = np.asarray(x, np.int64).copy()
    y = np.asarray(y, np.int64).copy()
    d = np.zeros_like(x)
    s = 1 << (order - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - x, x)
        y = np.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        s >>= 1
    return d


def _perm_from_order(order):
    """order[new] = old  ->  perm[old] = new."""
    perm = np.empty(len(order), np.int64)
    perm[order] = np.arange(len(order))
    return perm


def _hilbert_order(pts):
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    q = ((pts - lo) / np.maximum(hi - lo, 1e-30) * 65535.0).astype(np.int64)
    return np.argsort(hilbert_index(q[:, 0], q[:, 1]), kind="stable")


def _refine_order(pattern, order):
    """(order[new] = old, block_start[new]) with rows re-sorted by entry count inside the device plan's blocks
    (scn_plan_refine_order; host-side code of the library, no GPU involved)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    m = sp.csr_matrix(pattern)[order][:, order].tocsr()
    m.sort_indices()
    n = m.shape[0]
    rowptr = np.ascontiguousarray(m.indptr, np.int32)
    col = np.ascontiguousarray(m.indices, np.int32)
    out = np.empty(n, np.int32)
    starts = np.zeros(n, np.uint8)
    as_p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    _lib.check(lib.scn_plan_refine_order(n, as_p(rowptr), as_p(col), 1, as_p(out), starts.ctypes.data),
               "scn_plan_refine_order")
    return np.asarray(order)[out], starts


class Layout:
    """Row permutation of the three levels: perm[level][old] = new, order[level][new] = old; block_starts[level] is the
    optional cut hint for the LDS-blocked plan (1 where a block begins) that goes with the order."""

    def __init__(self, sizes, orders=None, block_starts=None, merged=None):
        self.sizes = tuple(int(s) for s in sizes)
        self.merged = merged            # optional uint8 [sum(sizes)]: level of the i-th simplex along ONE curve through all levels
        self.order = []
        self.perm = []
        self.block_starts = list(block_starts) if block_starts is not None else [None, None, None]
        for lvl, n in enumerate(self.sizes):
            o = np.arange(n, dtype=np.int64) if orders is None or orders[lvl] is None else np.asarray(orders[lvl], np.int64)
            assert len(o) == n and np.array_equal(np.sort(o), np.arange(n))
            self.order.append(o)
            self.perm.append(_perm_from_order(o))

    def is_identity(self, level):
        return np.array_equal(self.order[level], np.arange(self.sizes[level]))


class Shift:
    """A shift operator (rows = target level, columns = source level) in the caller's index order.

    Drop-in for the dense matrices the reference passes as S_lower / S_upper / S_ab (TE:137, 155, 173):
    supports .shape, .T, @ on NumPy arrays and .toarray(); the HIP path consumes .device_csr()."""

    def __init__(self, matrix, layout, row_level, col_level):
        m = sp.csr_matrix(matrix, dtype=np.float64)
        m.sum_duplicates()
        m.sort_indices()
        self.csr = m
        self.layout = layout
        self.row_level, self.col_level = row_level, col_level
        assert m.shape == (layout.sizes[row_level], layout.sizes[col_level]), "shift shape does not match the complex"
        self._T = None
        self._dev = None
        self._cache = {}

    @property
    def shape(self):
        return self.csr.shape

    @property
    def T(self):
        if self._T is None:
            self._T = Shift(self.csr.T.tocsr(), self.layout, self.col_level, self.row_level)
            self._T._T = self
        return self._T

    def toarray(self):
        return self.csr.toarray()

    def __matmul__(self, other):
        return self.csr @ other

    def is_symmetric(self):
        if self.shape[0] != self.shape[1]:
            return False
        d = self.csr - self.csr.T
        return d.nnz == 0 or float(abs(d).max()) == 0.0

    def device_csr(self):
        """P_row M P_col^T with sorted column indices: the operator in device row order."""
        if self._dev is None:
            coo = self.csr.tocoo()
            r = self.layout.perm[self.row_level][coo.row]
            c = self.layout.perm[self.col_level][coo.col]
            m = sp.csr_matrix((coo.data, (r, c)), shape=self.shape)
            m.sort_indices()
            self._dev = m
        return self._dev


def union_pattern(mats):
    """Shared CSR pattern of several same-shaped matrices and each one's values on it (zeros where absent)."""
    shape = mats[0].shape
    ncols = shape[1]
    keys = []
    for m in mats:
        coo = m.tocoo()
        keys.append(coo.row.astype(np.int64) * ncols + coo.col.astype(np.int64))
    ukeys = np.unique(np.concatenate(keys)) if keys else np.zeros(0, np.int64)
    rows = (ukeys // ncols).astype(np.int64)
    cols = (ukeys % ncols).astype(np.int32)
    rowptr = np.zeros(shape[0] + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    vals = []
    for m, k in zip(mats, keys):
        coo = m.tocoo()
        v = np.zeros(len(ukeys), np.float32)
        v[np.searchsorted(ukeys, k)] = coo.data.astype(np.float32)
        vals.append(v)
    return rowptr, cols, vals


class Bconds:
    """The readout operand: callable like the reference's Bconds_func(n) = B1_jax[nbrhoods[n]] (TE:298-303) and
    carrier of the sparse tables the HIP readout uses (node-major incidence CSR, padded neighbour table)."""

    def __init__(self, cx, layout, nbrhoods, flips=None):
        self.cx, self.layout = cx, layout
        self.nbrhoods = nbrhoods                              # (V, D) original node ids, -1 padded
        self.flips = None if flips is None else np.asarray(flips, np.float64)
        self._dev = {}

    def __call__(self, n):
        B1, _ = incidence_matrices(self.cx)
        if self.flips is not None:
            B1 = B1 @ sp.diags(self.flips)                    # TE:291
        B1 = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()   # zero row for index -1 (TE:288)
        return B1[np.asarray(self.nbrhoods[n])].toarray()

    def incidence_tables(self):
        """Node-major CSR of (flipped) B1 with edges in DEVICE order; endpoints per device edge."""
        E = self.cx.n_edges
        pe = self.layout.perm[1]
        tail, head = self.cx.edges[:, 0], self.cx.edges[:, 1]
        f = np.ones(E) if self.flips is None else self.flips
        node = np.concatenate([tail, head])
        edge = np.concatenate([pe, pe])
        sign = np.concatenate([-f, f])
        order = np.lexsort((edge, node))
        node, edge, sign = node[order], edge[order], sign[order]
        ptr = np.zeros(self.cx.n_nodes + 1, np.int64)
        np.add.at(ptr, node + 1, 1)
        edge_nodes = np.zeros((E, 2), np.int32)
        edge_nodes[pe, 0] = tail
        edge_nodes[pe, 1] = head
        return np.cumsum(ptr).astype(np.int32), edge.astype(np.int32), sign.astype(np.float32), edge_nodes


# ----------------------------------------------------------------------------------------------
# adoption of reference-style arguments: a caller that built its operands the way the reference's
# data_setup does -- dense (E, E) ndarrays for the shifts (TE:240-257), a closure for Bconds_func (TE:298-303) -- hands
# them to scone_func / ebli_func / bunch_func unchanged; they are wrapped ONCE (cached on the object) in the native
# types.  Meant for the small complexes where dense operands exist at all; no locality reordering is attempted.
# ----------------------------------------------------------------------------------------------
_IDENTITY_LAYOUTS = {}
_ADOPTED = {}            # id(object) -> (object kept alive, wrapper)


def identity_layout(sizes):
    sizes = tuple(int(x) for x in sizes)
    if sizes not in _IDENTITY_LAYOUTS:
        _IDENTITY_LAYOUTS[sizes] = Layout(sizes)
    return _IDENTITY_LAYOUTS[sizes]


def _to_csr(S):
    if sp.issparse(S):
        return S.tocsr().astype(np.float64)
    if hasattr(S, "detach"):                                  # torch tensor
        S = S.detach().cpu().numpy()
    a = np.asarray(S, dtype=np.float64)
    if a.ndim != 2:
        raise TypeError("a shift operator must be a 2-D matrix (dense, scipy.sparse) or a Shift")
    return sp.csr_matrix(a)


def adopt_shift(S, layout, row_level, col_level):
    """Shift for a dense / scipy / torch matrix in the caller's index order (cached on the object's identity)."""
    if isinstance(S, Shift):
        return S
    hit = _ADOPTED.get(id(S))
    if hit is not None and hit[0] is S and hit[1].layout is layout:
        return hit[1]
    sh = Shift(_to_csr(S), layout, row_level, col_level)
    _ADOPTED[id(S)] = (S, sh)
    return sh


class ProbedBconds:
    """Readout operand for a plain callable Bcond_func(n) -> (D, E) array of B1 rows (TE:298-303): the rows are read off
    by CALLING it for the last nodes that actually occur, and the sparse tables the HIP readout needs are rebuilt from
    what came back.  A row is taken to be the incidence row of one node (entries +-1 up to a per-edge flip, every edge in
    at most two rows with opposite signs); an all-zero row is a padding slot (the appended zero row, TE:288)."""

    def __init__(self, fn, n_edges, layout):
        self.fn, self.n_edges, self.layout = fn, int(n_edges), layout
        self.version = 0
        self._slot = {}            # last node -> row of the table
        self._node = {}            # bytes of an incidence row -> pseudo node id
        self._rows = []            # pseudo node id -> (edge idx, sign)
        self._table = []           # per probed last node: list of pseudo ids (-1 = zero row)
        self.nbrhoods = np.zeros((0, 1), np.int64)
        self.flips = None

    def __call__(self, n):
        return self.fn(n)

    def prepare(self, last_nodes):
        """Probe the nodes not seen yet; returns last_nodes remapped to rows of this object's tables."""
        ln = np.asarray(last_nodes.cpu() if hasattr(last_nodes, "cpu") else last_nodes).astype(np.int64).ravel()
        grew = False
        for u in np.unique(ln):
            if int(u) in self._slot:
                continue
            M = self.fn(int(u))
            M = np.asarray(M.detach().cpu().numpy() if hasattr(M, "detach") else M, dtype=np.float64)
            if M.ndim != 2 or M.shape[1] != self.n_edges:
                raise TypeError("Bcond_func(n) must return a (D, E) array of incidence rows (TE:298-303)")
            ids = []
            for row in M:
                e = np.flatnonzero(row)
                if len(e) == 0:
                    ids.append(-1)
                    continue
                key = (e.tobytes(), row[e].tobytes())
                if key not in self._node:
                    self._node[key] = len(self._rows)
                    self._rows.append((e.astype(np.int64), row[e].copy()))
                ids.append(self._node[key])
            self._slot[int(u)] = len(self._table)
            self._table.append(ids)
            grew = True
        if grew:
            D = max(len(t) for t in self._table)
            if any(len(t) != D for t in self._table):
                raise TypeError("Bcond_func must return the same number of rows for every node")
            self.nbrhoods = np.asarray(self._table, np.int64).reshape(len(self._table), D)
            self.version += 1
        return np.asarray([self._slot[int(u)] for u in ln], np.int64)

    def incidence_tables(self):
        """Same layout as Bconds.incidence_tables, over the pseudo nodes discovered so far (unseen endpoints: -2)."""
        pe = self.layout.perm[1]
        node = np.concatenate([np.full(len(e), k, np.int64) for k, (e, _) in enumerate(self._rows)] or [np.zeros(0, np.int64)])
        edge = np.concatenate([pe[e] for e, _ in self._rows] or [np.zeros(0, np.int64)])
        sign = np.concatenate([v for _, v in self._rows] or [np.zeros(0)])
        order = np.lexsort((edge, node))
        node, edge, sign = node[order], edge[order], sign[order]
        ptr = np.zeros(len(self._rows) + 1, np.int64)
        np.add.at(ptr, node + 1, 1)
        edge_nodes = np.full((self.n_edges, 2), -2, np.int32)
        fill = np.zeros(self.n_edges, np.int64)
        first_sign = np.zeros(self.n_edges)
        for k, e, v in zip(node, edge, sign):
            if fill[e] == 2 or (fill[e] == 1 and v * first_sign[e] > 0):
                raise TypeError("Bcond_func does not select incidence rows: an edge must appear in at most two rows, with "
                                "opposite signs")
            if fill[e] == 0:
                first_sign[e] = v
            edge_nodes[e, fill[e]] = k
            fill[e] += 1
        return np.cumsum(ptr).astype(np.int32), edge.astype(np.int32), sign.astype(np.float32), edge_nodes


def adopt_bconds(fn, n_edges, layout):
    if isinstance(fn, Bconds) or isinstance(fn, ProbedBconds):
        return fn
    if not callable(fn):
        raise TypeError("Bcond_func must be callable: the Bconds object of SimplicialComplex.bconds() or a function n -> (D, E) rows")
    hit = _ADOPTED.get(id(fn))
    if hit is not None and hit[0] is fn and hit[1].layout is layout:
        return hit[1]
    pb = ProbedBconds(fn, n_edges, layout)
    _ADOPTED[id(fn)] = (fn, pb)
    return pb


class _Degrees:
    def __init__(self, deg):
        self._deg = deg

    def __getitem__(self, n):
        return int(self._deg[n])

    def __iter__(self):
        return iter((i, int(d)) for i, d in enumerate(self._deg))

    def __len__(self):
        return len(self._deg)


class UndirGraph:
    """What the reference's data_setup hands back as G_undir (a networkx graph read from G_undir.pkl, SDG:436 -- gpickle
    no longer exists), reduced to the accesses the experiment driver makes (TE:273-279, 456-458, 500): .nodes, .edges,
    .degree (iterable of (node, degree) and indexable), G[node] -> neighbours; .complex is the SimplicialComplex."""

    def __init__(self, sc):
        self.complex = sc
        self.nodes = range(sc.cx.n_nodes)
        self.edges = [tuple(e) for e in sc.cx.edges.tolist()]
        self.degree = _Degrees(sc.degrees)

    def __getitem__(self, n):
        row = self.complex.nbrhoods[int(n)]
        return [int(v) for v in row[row >= 0]]

    def neighbors(self, n):
        return iter(self[n])


class SimplicialComplex:
    """Complex + layout + operator factory.  Build from arrays, from a Complex, or from B1/B2 (dense or sparse)."""

    def __init__(self, cx, reorder=True):
        self.cx = cx
        self.B1, self.B2 = incidence_matrices(cx)
        self.nbrhoods, self.degrees = neighborhood_table(cx)
        self.max_degree = int(self.nbrhoods.shape[1])
        orders = [None, None, None]
        if reorder and cx.n_edges > 1:
            if cx.coords is not None:
                orders[0] = _hilbert_order(cx.coords[:cx.n_nodes])
                mid = 0.5 * (cx.coords[cx.edges[:, 0]] + cx.coords[cx.edges[:, 1]])
                orders[1] = _hilbert_order(mid)
                if cx.n_faces > 1:
                    orders[2] = _hilbert_order(cx.coords[cx.faces].mean(axis=1))
            else:
                rcm = lambda M: np.asarray(reverse_cuthill_mckee(sp.csr_matrix(M), symmetric_mode=True), np.int64)
                orders[0] = rcm(self.B1 @ self.B1.T)
                orders[1] = rcm(self.B1.T @ self.B1)
                if cx.n_faces > 1:
                    orders[2] = rcm(self.B2.T @ self.B2)
        starts = [None, None, None]
        if reorder and cx.n_edges > 1:
            # second pass: inside the 64-row blocks of the device plan, sort rows by the entry count of the level's own
            # Laplacian (the lanes of a wave walk their rows' ELL lists in lock-step; see scn_plan_refine_order)
            pats = [self.B1 @ self.B1.T, self.B1.T @ self.B1, self.B2.T @ self.B2 if cx.n_faces > 1 else None]
            for lvl in range(3):
                if orders[lvl] is not None and pats[lvl] is not None:
                    orders[lvl], starts[lvl] = _refine_order(pats[lvl], orders[lvl])
        self.layout = Layout((cx.n_nodes, cx.n_edges, cx.n_faces), orders, starts)

    @classmethod
    def from_incidence(cls, B1, B2, coords=None, reorder=True):
        edges, faces = complex_from_incidence(B1, B2)
        order = np.lexsort((edges[:, 1], edges[:, 0]))
        if not np.array_equal(order, np.arange(len(order))):
            raise ValueError("B1 columns must list edges in sorted (tail, head) order, as the reference writes them")
        return cls(Complex(n_nodes=sp.csr_matrix(B1).shape[0], edges=edges, faces=faces, coords=coords), reorder=reorder)

    # ---- operators (TE:240-257) ----
    def flip_vector(self, seed=1):
        """flips ~ choice([1,-1], p=[.8,.2]) under seed 1 (TE:216-219)."""
        rs = np.random.RandomState(seed)
        return rs.choice([1, -1], size=self.cx.n_edges, replace=True, p=[0.8, 0.2]).astype(np.float64)

    def hodge_laplacians(self, flips=None):
        L_lo = (self.B1.T @ self.B1).tocsr()
        L_up = (self.B2 @ self.B2.T).tocsr()
        if flips is not None:
            F = sp.diags(flips)
            L_lo, L_up = (F @ L_lo @ F).tocsr(), (F @ L_up @ F).tocsr()        # TE:242-244
        return L_lo, L_up

    def scone_shifts(self, flips=None):
        L_lo, L_up = self.hodge_laplacians(flips)
        return [Shift(L_lo, self.layout, 1, 1), Shift(L_up, self.layout, 1, 1)]   # TE:247-248

    def ebli_shifts(self, flips=None):
        L_lo, L_up = self.hodge_laplacians(flips)
        L1 = (L_lo + L_up).tocsr()
        return [Shift(L1, self.layout, 1, 1), Shift((L1 @ L1).tocsr(), self.layout, 1, 1)]   # TE:251-253

    def bunch_layout(self):
        """Row orders for the Bunch model: every level along the SAME Hilbert curve (keys of the simplex centroids in one
        common frame, no per-block re-sorting), plus the merged order of the three levels along that curve -- the fused Bunch
        layer cuts its blocks as patches across the levels (scn_terms_create).  Without coordinates: the default orders, merged
        level by level (correct, but the patches then have no cross-level locality)."""
        if getattr(self, "_bunch_layout", None) is None:
            cx = self.cx
            sizes = (cx.n_nodes, cx.n_edges, cx.n_faces)
            if cx.coords is not None and cx.n_edges > 1 and cx.n_faces > 0:
                c = np.asarray(cx.coords, np.float64)
                pts = [c[:cx.n_nodes], 0.5 * (c[cx.edges[:, 0]] + c[cx.edges[:, 1]]), c[cx.faces].mean(axis=1)]
                lo, hi = c.min(axis=0), c.max(axis=0)
                orders, keys = [], []
                for p in pts:
                    q = ((p - lo) / np.maximum(hi - lo, 1e-30) * 65535.0).astype(np.int64)
                    k = hilbert_index(q[:, 0], q[:, 1])
                    o = np.argsort(k, kind="stable")
                    orders.append(o)
                    keys.append(k[o])
                lvl = np.concatenate([np.full(n, l, np.uint8) for l, n in enumerate(sizes)])
                merged = lvl[np.argsort(np.concatenate(keys), kind="stable")]
                self._bunch_layout = Layout(sizes, orders, None, merged=np.ascontiguousarray(merged, np.uint8))
            else:
                merged = np.concatenate([np.full(n, l, np.uint8) for l, n in enumerate(sizes)])
                self._bunch_layout = Layout(sizes, [self.layout.order[l] for l in range(3)], None, merged=merged)
        return self._bunch_layout

    def bunch_shifts(self):
        S = compute_shift_matrices(self.B1, self.B2)                                           # TE:255-257
        lv = [(0, 0), (0, 1), (1, 0), (1, 1), (1, 2), (2, 1), (2, 2)]   # (row level, col level) of S_00,S_10,S_01,S_11,S_21,S_12,S_22
        lay = self.bunch_layout()
        return [Shift(m, lay, r, c) for m, (r, c) in zip(S, lv)]

    def bconds(self, flips=None):
        return Bconds(self.cx, self.layout, self.nbrhoods, flips)

    def n_nbrs(self, last_nodes):
        return self.degrees[np.asarray(last_nodes)]                                          # TE:276
