"""Sparse, scalable synthetic complexes and trajectories (host side, NumPy/scipy).

Follows the recipe of the reference generator -- trajectory_analysis/synthetic_data_gen.py (SDG):
  random_SC_graph      SDG:82-137   points under seed 1, sorted along x+y, Delaunay, two holes
  incidence_matrices   SDG:139-161  B1 (-1 tail / +1 head, tail < head), B2 (+1,+1,-1 for (a,b),(b,c),(a,c))
  generate_random_walks SDG:178-243 BEGIN -> A_r -> B_r -> END by concatenated shortest paths, r = i % 3
  split_paths / path_to_flow / path_dataset  SDG:245-258, 327-373
but never materialises a dense V x E or E x F matrix, so it reaches |E| ~ 1M (the reference's dense
B1/B2 stop at ~1e4 edges).  For n = 400 the complex is bit-identical to the reference's
(tests/golden/cfg1_complex.npz); the walks use our own BFS tie-breaking, so they follow the reference's
distribution, not its exact node sequences (the exact ones are in tests/golden/cfg1_paths.npz).
"""
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order, dijkstra
from scipy.spatial import Delaunay


@dataclass
class Complex:
    """A 2-dimensional simplicial complex: nodes 0..n_nodes-1, sorted edges (a<b), sorted faces (a<b<c)."""
    n_nodes: int
    edges: np.ndarray          # (E, 2) int64, lexicographically sorted, a < b
    faces: np.ndarray          # (F, 3) int64, sorted rows, lexicographically sorted
    coords: np.ndarray = None  # (V, 2) float64 or None
    valid_idxs: np.ndarray = None

    @property
    def n_edges(self):
        return int(self.edges.shape[0])

    @property
    def n_faces(self):
        return int(self.faces.shape[0])

    def edge_index(self, a, b):
        """Index of edge (min,max) for arrays a, b (vectorised binary search on the sorted edge list)."""
        a, b = np.asarray(a, np.int64), np.asarray(b, np.int64)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = lo * self.n_nodes + hi
        keys = self.edges[:, 0] * self.n_nodes + self.edges[:, 1]
        idx = np.searchsorted(keys, key)
        if np.any(idx >= len(keys)) or np.any(keys[np.minimum(idx, len(keys) - 1)] != key):
            raise KeyError("edge not in complex")
        return idx


def random_SC_graph(n, holes=True):
    """SDG:82-137 without networkx.  Returns a Complex (edges/faces as arrays)."""
    rs = np.random.RandomState(1)                                        # SDG:98
    coords = rs.rand(n, 2)
    coords = coords[np.argsort(np.sum(coords, axis=1))]                  # SDG:102-104
    tri = Delaunay(coords)                                               # SDG:107 (qhull is deterministic)
    if holes:
        valid = (np.linalg.norm(coords - [1 / 4, 3 / 4], axis=1) > 1 / 8) \
            & (np.linalg.norm(coords - [3 / 4, 1 / 4], axis=1) > 1 / 8)  # SDG:109-110
    else:
        valid = np.ones(n, dtype=bool)
    simp = np.sort(tri.simplices.astype(np.int64), axis=1)
    simp = simp[valid[simp].all(axis=1)]                                 # SDG:114
    faces = np.unique(simp, axis=0)                                      # sorted, like sorted([...])
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [0, 2]]])   # SDG:120-124
    edges = np.unique(e, axis=0)                                         # SDG:127
    return Complex(n_nodes=n, edges=edges, faces=faces, coords=coords,
                   valid_idxs=np.nonzero(valid)[0])


def incidence_matrices(cx):
    """Sparse B1 (V x E) and B2 (E x F) with the reference's sign conventions (SDG:139-161)."""
    E, F = cx.n_edges, cx.n_faces
    ar = np.arange(E)
    B1 = sp.csr_matrix((np.concatenate([-np.ones(E), np.ones(E)]),
                        (np.concatenate([cx.edges[:, 0], cx.edges[:, 1]]), np.concatenate([ar, ar]))),
                       shape=(cx.n_nodes, E))
    if F:
        f = cx.faces
        e_ab = cx.edge_index(f[:, 0], f[:, 1])
        e_bc = cx.edge_index(f[:, 1], f[:, 2])
        e_ac = cx.edge_index(f[:, 0], f[:, 2])
        fr = np.arange(F)
        B2 = sp.csr_matrix((np.concatenate([np.ones(F), np.ones(F), -np.ones(F)]),
                            (np.concatenate([e_ab, e_bc, e_ac]), np.concatenate([fr, fr, fr]))),
                           shape=(E, F))
    else:
        B2 = sp.csr_matrix((E, 0))
    return B1, B2


def complex_from_incidence(B1, B2):
    """Recover (edges, faces) from incidence matrices given dense or sparse (the dataset folder format)."""
    B1 = sp.csc_matrix(B1)
    V, E = B1.shape
    edges = np.zeros((E, 2), np.int64)
    for e in range(E):
        rows = B1.indices[B1.indptr[e]:B1.indptr[e + 1]]
        vals = B1.data[B1.indptr[e]:B1.indptr[e + 1]]
        edges[e, 0] = rows[np.argmin(vals)]       # -1 = tail
        edges[e, 1] = rows[np.argmax(vals)]       # +1 = head
    B2c = sp.csc_matrix(B2)
    F = B2c.shape[1]
    faces = np.zeros((F, 3), np.int64)
    for f in range(F):
        es = B2c.indices[B2c.indptr[f]:B2c.indptr[f + 1]]
        faces[f] = np.unique(edges[es].ravel())
    return edges, faces


def adjacency(cx):
    a, b = cx.edges[:, 0], cx.edges[:, 1]
    n = cx.n_nodes
    return sp.csr_matrix((np.ones(2 * len(a)), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n))


def _tree_path(pred, root, v):
    """Nodes from root to v along a BFS predecessor tree (inclusive)."""
    out = [int(v)]
    while out[-1] != root:
        p = pred[out[-1]]
        if p < 0:
            return None
        out.append(int(p))
    return out[::-1]


def generate_random_walks(cx, m=1000, seed=1030, waypoint_pool=None, metric="hops"):
    """SDG:178-243.  `waypoint_pool=k` draws the A_r / B_r waypoints from k candidates per region so that
    only 6k shortest-path trees are built however many walks are requested (needed at |V| ~ 4e5); None = fresh
    draw per walk as the reference does.  metric="hops" is the reference's BFS shortest path; "euclid" weights
    edges by length (Dijkstra): on very large Delaunay complexes the hop-count paths all detour along the long
    convex-hull edges, overlap there and fail the reference's simple-path test (SDG:239) almost always."""
    rs = np.random.RandomState(seed)
    pts, valid = cx.coords, cx.valid_idxs
    s = np.sum(pts[valid], axis=1)
    BEGIN, END = valid[s < 1 / 4], valid[s > 7 / 4]                      # SDG:207-208
    A012 = valid[(s > 1 / 4) & (s < 1)]
    B012 = valid[(s < 7 / 4) & (s > 1)]

    def regions(X):
        d = pts[X, 1] - pts[X, 0]
        return [X[(d < 1 / 2) & (d > -1 / 2)], X[d > 1 / 2], X[d < -1 / 2]]   # SDG:211-218
    A, B = regions(A012), regions(B012)
    if waypoint_pool:
        A = [rs.choice(a, size=min(waypoint_pool, len(a)), replace=False) for a in A]
        B = [rs.choice(b, size=min(waypoint_pool, len(b)), replace=False) for b in B]
    G = adjacency(cx)
    if metric == "euclid":
        a, b = cx.edges[:, 0], cx.edges[:, 1]
        w = np.linalg.norm(pts[a] - pts[b], axis=1)
        G = sp.csr_matrix((np.concatenate([w, w]), (np.concatenate([a, b]), np.concatenate([b, a]))),
                          shape=(cx.n_nodes, cx.n_nodes))
    trees = {}

    def tree(v):
        if v not in trees:
            if metric == "euclid":
                trees[v] = dijkstra(G, directed=False, indices=v, return_predecessors=True)[1]
            else:
                trees[v] = breadth_first_order(G, v, directed=False, return_predecessors=True)[1]
        return trees[v]

    paths, i, tries = [], 0, 0
    while len(paths) < m:
        tries += 1
        if tries > 50 * m + 1000:
            raise RuntimeError("could not generate enough simple walks")
        r = i % 3
        v_begin = int(rs.choice(BEGIN))
        v_1, v_2 = int(rs.choice(A[r])), int(rs.choice(B[r]))
        v_end = int(rs.choice(END))
        t1, t2 = tree(v_1), tree(v_2)
        p_a = _tree_path(t1, v_1, v_begin)
        p_b = _tree_path(t1, v_1, v_2)
        p_c = _tree_path(t2, v_2, v_end)
        if p_a is None or p_b is None or p_c is None:
            continue
        path = p_a[::-1][:-1] + p_b[:-1] + p_c                            # SDG:236-238
        if len(path) == len(set(path)):                                   # SDG:239
            paths.append(path)
            i += 1
        if not waypoint_pool and len(trees) > 64:
            trees.clear()
    return paths


def split_paths(paths, rs, truncate_paths=True, suffix_size=2):
    """SDG:245-258."""
    if truncate_paths:
        paths = [p[:4 + rs.choice(range(2, len(p) - 4))] for p in paths]
    prefixes = [p[:-suffix_size] for p in paths]
    suffixes = [p[-suffix_size:] for p in paths]
    return prefixes, suffixes, [p[-1] for p in prefixes]


@dataclass
class SparseFlows:
    """Edge flows of N trajectories as ragged (edge index, value) lists: the sparse form of flows_in (N, E, 1)."""
    ptr: np.ndarray      # (N+1,) int64
    idx: np.ndarray      # (nnz,) int64 edge ids
    val: np.ndarray      # (nnz,) float32
    n_edges: int

    def __len__(self):
        return len(self.ptr) - 1

    def select(self, sel):
        sel = np.asarray(sel)
        lens = self.ptr[sel + 1] - self.ptr[sel]
        ptr = np.concatenate([[0], np.cumsum(lens)])
        take = np.concatenate([np.arange(self.ptr[i], self.ptr[i + 1]) for i in sel]) if len(sel) else np.zeros(0, np.int64)
        return SparseFlows(ptr.astype(np.int64), self.idx[take], self.val[take], self.n_edges)

    def todense(self):
        X = np.zeros((len(self), self.n_edges, 1), np.float32)
        rows = np.repeat(np.arange(len(self)), np.diff(self.ptr))
        np.add.at(X[:, :, 0], (rows, self.idx), self.val)
        return X

    @staticmethod
    def fromdense(X):
        X = np.asarray(X)
        X2 = X.reshape(X.shape[0], X.shape[1])
        rows, cols = np.nonzero(X2)
        ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=X2.shape[0]))])
        return SparseFlows(ptr.astype(np.int64), cols.astype(np.int64), X2[rows, cols].astype(np.float32), X2.shape[1])


def paths_to_flows(cx, paths):
    """path_to_flow (SDG:327-344) for a list of node paths -> SparseFlows (+1 along a<b, -1 against)."""
    ptr, idx, val = [0], [], []
    for p in paths:
        p = np.asarray(p, np.int64)
        if len(p) > 1:
            a, b = p[:-1], p[1:]
            e = cx.edge_index(a, b)
            sgn = np.where(a < b, 1.0, -1.0)
            ue, inv = np.unique(e, return_inverse=True)
            v = np.zeros(len(ue))
            np.add.at(v, inv, sgn)
            idx.append(ue)
            val.append(v)
            ptr.append(ptr[-1] + len(ue))
        else:
            ptr.append(ptr[-1])
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    val = np.concatenate(val) if val else np.zeros(0)
    return SparseFlows(np.asarray(ptr, np.int64), idx.astype(np.int64), val.astype(np.float32), cx.n_edges)


def flow_to_path(flow, edges, last_node):
    """Node path of a +-1 edge flow that ends in last_node (SDG:299-325): walk backwards along the oriented edges."""
    flow = np.asarray(flow).reshape(-1)
    into = {}
    for i in np.flatnonzero(flow):
        a, b = (int(edges[i][0]), int(edges[i][1])) if flow[i] > 0 else (int(edges[i][1]), int(edges[i][0]))
        into.setdefault(b, []).append(a)                       # oriented edge a -> b
    path, cur, left = [int(last_node)], int(last_node), int(np.count_nonzero(flow))
    while left:
        if not into.get(cur):
            raise ValueError("flow is not a path ending in last_node")
        cur = into[cur].pop()
        path.append(cur)
        left -= 1
    return path[::-1]


def neighborhood_table(cx):
    """nbrhoods (V, D): sorted neighbours, -1 padded (TE:273-279); also degrees."""
    G = adjacency(cx)
    deg = np.diff(G.indptr)
    D = int(deg.max())
    tab = -np.ones((cx.n_nodes, D), np.int64)
    G.sort_indices()
    rows = np.repeat(np.arange(cx.n_nodes), deg)
    pos = np.arange(len(G.indices)) - np.repeat(G.indptr[:-1], deg)
    tab[rows, pos] = G.indices
    return tab, deg


def path_dataset(cx, paths, seed=None, truncate_paths=True, rs=None):
    """1-hop part of SDG:346-361: prefix flows, one-hot target index among the sorted neighbours, last nodes."""
    rs = rs if rs is not None else np.random.RandomState(seed)
    prefixes, suffixes, last_nodes = split_paths(paths, rs, truncate_paths)
    nbr, _ = neighborhood_table(cx)
    target_nodes = np.asarray([s[0] for s in suffixes], np.int64)
    last_nodes = np.asarray(last_nodes, np.int64)
    choice = np.argmax(nbr[last_nodes] == target_nodes[:, None], axis=1)
    return paths_to_flows(cx, prefixes), choice.astype(np.int64), last_nodes, target_nodes, prefixes


def calibrate_n_points(target_edges, holes=True):
    """Number of points whose complex has ~target_edges edges (E ~ 2.7 V with the two holes cut out)."""
    return max(16, int(round(target_edges / (2.71 if holes else 2.98))))
