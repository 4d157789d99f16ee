"""Ocean-drifter (Madagascar buoys) dataset: Schaub's JLD2 file -> SCoNe complex + trajectories.

Follows the reference converter ocean_drifters_data/buoy_data.py (BD): edge list `elist`, triangle list `tlist`,
hexagon centres `HexcentersXY`, trajectories `TrajectoriesNodes` (all 1-indexed, BD:17-33); backtracking steps are
stripped (strip_paths, SDG:43-61), paths with >= 5 nodes keep their last 10 nodes (BD:56-57), no truncation, suffix of
2 nodes, 80/20 split under seed 1 (BD:70-73).  Reads the file with the pure-Python jld2_reader (no h5py here).
"""
import numpy as np

from .jld2_reader import JLD2File
from .synthetic_data_gen import Complex, neighborhood_table, paths_to_flows


def strip_paths(paths):
    """Remove immediate back-tracking (a, b, a -> a), SDG:43-61."""
    out = []
    for path in paths:
        res = []
        for node in path:
            if len(res) < 2:
                res.append(node)
            elif node == res[-2]:
                res.pop()
            else:
                res.append(node)
        out.append(res)
    return out


def read_buoy_file(path):
    f = JLD2File(path)
    elist = np.asarray(f["elist"][0], np.int64) - 1                   # (2, 320)   BD:17-18
    tlist = np.asarray(f["tlist"][0], np.int64) - 1                   # (3, 186)   BD:20-21
    raw, _ = f["HexcentersXY"]
    coords = np.frombuffer(raw, dtype="<f8").reshape(-1, 2).copy()    # BD:29-30
    trajs = []
    for ref in f["TrajectoriesNodes"][0]:                             # BD:32-33: refs -> arrays of refs -> boxed Int64
        inner, cls = f.read(int(ref))
        if cls == 7:
            trajs.append([int(f.read(int(r))[0]) - 1 for r in inner])
        else:
            trajs.append([int(v) - 1 for v in np.asarray(inner).ravel()])
    return elist, tlist, coords, trajs


def buoy_complex(elist, tlist, coords=None):
    """BD:38-47: sorted node-sorted edges, sorted faces."""
    e = np.sort(elist.T, axis=1)
    edges = np.unique(e, axis=0)
    faces = np.unique(np.sort(tlist.T, axis=1), axis=0)
    n_nodes = int(edges.max()) + 1
    c = None if coords is None or len(coords) < n_nodes else coords[:n_nodes]
    return Complex(n_nodes=n_nodes, edges=edges, faces=faces, coords=c, valid_idxs=np.arange(n_nodes))


def buoy_dataset(elist, tlist, coords, trajs, seed=1):
    """Complex, kept paths, 1-hop flows / targets / last nodes and the train / test masks (BD:49-88)."""
    cx = buoy_complex(elist, tlist, coords)
    paths = [p[-10:] for p in strip_paths(trajs) if len(p) >= 5]      # BD:56-57
    rs = np.random.RandomState(seed)                                   # BD:70
    train_mask = np.asarray([1] * round(len(paths) * 0.8) + [0] * round(len(paths) * 0.2))
    rs.shuffle(train_mask)
    prefixes = [p[:-2] for p in paths]                                 # truncate_paths=False, suffix 2 (SDG:254-256)
    last = np.asarray([p[-1] for p in prefixes], np.int64)
    tnodes = np.asarray([p[-2] for p in paths], np.int64)
    nbr, _ = neighborhood_table(cx)
    choice = np.argmax(nbr[last] == tnodes[:, None], axis=1)
    flows = paths_to_flows(cx, prefixes)
    return cx, paths, flows, choice, last, tnodes, train_mask, 1 - train_mask


def load_buoy_dataset(path, seed=1):
    return buoy_dataset(*read_buoy_file(path), seed=seed)
