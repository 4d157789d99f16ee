#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's NumPy-only side.

Run ONLY in the authoring container (needs /root/reference, which never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python /root/repo/tests/golden/make_golden.py

What is imported from the reference (read-only, unmodified):
  * trajectory_analysis/synthetic_data_gen.py : random_SC_graph, incidence_matrices,
    generate_random_walks, path_dataset           (SDG:82-161, 178-258, 327-373)
  * trajectory_analysis/bunch_model_matrices.py : compute_shift_matrices   (BMM:118-135)
The reference's JAX side (scone_func & co.) cannot be imported here (jax is not installed), so these
fixtures pin INPUTS and OPERATORS, not forward/backward values.

Two names removed from current networkx / NumPy are aliased before import, as SURVEY.md §8(c)
records: nx.OrderedDiGraph (SDG:117) and np.float (SDG:294).

Outputs (small, committed):
  cfg1_complex.npz  : edges, faces, coords, valid_idxs, B1/B2 as COO
  cfg1_paths.npz    : 1000 walks -> flows (edge idx, sign), targets, last/target nodes, masks (fwd + reversed)
  cfg1_bunch.npz    : the 7 Bunch shift matrices as COO (fp64)
  tiny4_complex.npz : the 4-node known-answer graph of projection_model.test_dataset (PM:128-151), B1/B2
"""
import os
import sys

import numpy as np
import networkx as nx

REF = "/root/reference/trajectory_analysis"
OUT = os.path.dirname(os.path.abspath(__file__))

# compatibility aliases for names the installed networkx/NumPy dropped
if not hasattr(nx, "OrderedDiGraph"):
    nx.OrderedDiGraph = nx.DiGraph
if not hasattr(np, "float"):
    np.float = float

sys.path.insert(0, REF)
sys.dont_write_bytecode = True
import synthetic_data_gen as sdg            # noqa: E402
import bunch_model_matrices as bmm          # noqa: E402


def coo(M):
    r, c = np.nonzero(M)
    return r.astype(np.int32), c.astype(np.int32), np.asarray(M)[r, c].astype(np.float64)


def flows_sparse(flows):
    """(N, E, 1) dense -> ragged (ptr, edge idx, value)."""
    N = flows.shape[0]
    ptr = [0]
    idx, val = [], []
    for i in range(N):
        nz = np.nonzero(flows[i, :, 0])[0]
        idx.append(nz)
        val.append(flows[i, nz, 0])
        ptr.append(ptr[-1] + len(nz))
    return (np.asarray(ptr, np.int32), np.concatenate(idx).astype(np.int32),
            np.concatenate(val).astype(np.float64))


def main():
    # --- cfg 1: same call order as generate_dataset(400, 1000, ...) SDG:375-411 (plot and gpickle skipped)
    G, V, E, faces, edge_to_idx, coords, valid_idxs = sdg.random_SC_graph(400, holes=True)
    B1, B2 = sdg.incidence_matrices(G, V, E, faces, edge_to_idx)
    G_undir, paths = sdg.generate_random_walks(G, coords, valid_idxs, m=1000)
    rev_paths = [p[::-1] for p in paths]
    train_mask = np.asarray([1] * int(len(paths) * 0.8) + [0] * int(len(paths) * 0.2))
    np.random.shuffle(train_mask)
    test_mask = 1 - train_mask
    max_degree = int(np.max([deg for n, deg in G_undir.degree()]))
    fw = sdg.path_dataset(G_undir, E, edge_to_idx, paths, max_degree)
    rv = sdg.path_dataset(G_undir, E, edge_to_idx, rev_paths, max_degree)

    b1r, b1c, b1v = coo(B1)
    b2r, b2c, b2v = coo(B2)
    np.savez_compressed(os.path.join(OUT, "cfg1_complex.npz"),
                        edges=np.asarray(E, np.int32), faces=np.asarray(faces, np.int32),
                        coords=coords, valid_idxs=valid_idxs.astype(np.int32),
                        n_nodes=np.int32(len(V)),
                        B1_row=b1r, B1_col=b1c, B1_val=b1v, B1_shape=np.asarray(B1.shape, np.int32),
                        B2_row=b2r, B2_col=b2c, B2_val=b2v, B2_shape=np.asarray(B2.shape, np.int32),
                        max_degree=np.int32(max_degree))

    def pack(prefix, d):
        flows1, targ1, last1, suf1, flows2, targ2, last2, suf2 = d
        p1, i1, v1 = flows_sparse(flows1)
        p2, i2, v2 = flows_sparse(flows2)
        return {prefix + "flow1_ptr": p1, prefix + "flow1_idx": i1, prefix + "flow1_val": v1,
                prefix + "targets1": np.asarray(targ1, np.float64)[:, :, 0].argmax(1).astype(np.int32),
                prefix + "targets1_sum": np.asarray(targ1).sum(axis=(1, 2)),
                prefix + "last1": np.asarray(last1, np.int32), prefix + "tnode1": np.asarray(suf1, np.int32),
                prefix + "flow2_ptr": p2, prefix + "flow2_idx": i2, prefix + "flow2_val": v2,
                prefix + "targets2": np.asarray(targ2, np.float64)[:, :, 0].argmax(1).astype(np.int32),
                prefix + "last2": np.asarray(last2, np.int32), prefix + "tnode2": np.asarray(suf2, np.int32)}

    d = {"train_mask": train_mask.astype(np.int8), "test_mask": test_mask.astype(np.int8),
         "n_edges": np.int32(len(E)), "max_degree": np.int32(max_degree),
         "path_ptr": np.cumsum([0] + [len(p) for p in paths]).astype(np.int32),
         "path_nodes": np.concatenate([np.asarray(p) for p in paths]).astype(np.int32)}
    d.update(pack("", fw))
    d.update(pack("rev_", rv))
    np.savez_compressed(os.path.join(OUT, "cfg1_paths.npz"), **d)

    # --- Bunch shift matrices from the reference's dense arithmetic (BMM:71-135)
    S = bmm.compute_shift_matrices(B1, B2)
    names = ["S_00", "S_10", "S_01", "S_11", "S_21", "S_12", "S_22"]
    d = {}
    for n, M in zip(names, S):
        r, c, v = coo(np.where(np.abs(M) > 1e-14, M, 0.0))
        d[n + "_row"], d[n + "_col"], d[n + "_val"] = r, c, v
        d[n + "_shape"] = np.asarray(M.shape, np.int32)
    np.savez_compressed(os.path.join(OUT, "cfg1_bunch.npz"), **d)

    # --- tiny 4-node known-answer graph (PM:128-151); nx.from_numpy_matrix is gone, build it by hand
    A = np.array([[0, 1, 1, 1], [1, 0, 1, 0], [1, 1, 0, 1], [1, 0, 1, 0]])
    G4 = nx.DiGraph()
    G4.add_nodes_from(range(4))
    E4 = sorted((i, j) for i in range(4) for j in range(i + 1, 4) if A[i, j])
    G4.add_edges_from(E4)
    faces4 = bmm.get_faces(G4.to_undirected())
    e2i = {e: i for i, e in enumerate(E4)}
    B1_4, B2_4 = sdg.incidence_matrices(G4, sorted(G4.nodes), E4, faces4, e2i)
    S4 = bmm.compute_shift_matrices(B1_4, B2_4)
    d = {"edges": np.asarray(E4, np.int32), "faces": np.asarray(faces4, np.int32),
         "B1": B1_4.astype(np.float64), "B2": B2_4.astype(np.float64)}
    for n, M in zip(names, S4):
        d[n] = np.asarray(M, np.float64)
    np.savez_compressed(os.path.join(OUT, "tiny4_complex.npz"), **d)

    print("edges", len(E), "faces", len(faces), "max_degree", max_degree,
          "B1", B1.shape, "B2", B2.shape, "paths", len(paths))
    for n, M in zip(names, S):
        print(n, M.shape, "nnz", int((np.abs(M) > 1e-14).sum()))


if __name__ == "__main__":
    main()
