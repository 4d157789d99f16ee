"""Reader / writer for the reference's dataset folder format (SDG:11-31, 421-447), pickle-free.

    trajectory_data_{1,2}hop_<suffix>/{flows_in,B1,B2,targets,train_mask,test_mask,last_nodes,target_nodes,
                                       rev_flows_in,rev_targets,rev_last_nodes,rev_target_nodes,coords}.npy
Differences from the reference, all backwards compatible on read:
  * B1 / B2 / flows_in may be stored sparse (`B1.npz`, `B2.npz`, `flows_in.npz`; scipy / ragged format) because the
    dense arrays stop at ~1e4 edges; dense `.npy` files written by the reference load unchanged.
  * the graph is recovered from B1 instead of `G_undir.pkl` (networkx gpickle no longer exists).
"""
import os

import numpy as np
import scipy.sparse as sp

from . import synthetic_data_gen as sdg
from .synthetic_data_gen import SparseFlows

DENSE_LIMIT = 50_000_000   # write dense .npy only below this many elements


def _save_matrix(folder, name, M):
    M = sp.csr_matrix(M)
    if M.shape[0] * M.shape[1] <= DENSE_LIMIT:
        np.save(os.path.join(folder, name + '.npy'), M.toarray())
    else:
        sp.save_npz(os.path.join(folder, name + '.npz'), M)


def _load_matrix(folder, name):
    p = os.path.join(folder, name + '.npy')
    if os.path.exists(p):
        return sp.csr_matrix(np.load(p))
    return sp.load_npz(os.path.join(folder, name + '.npz')).tocsr()


def _save_flows(folder, name, flows):
    if len(flows) * flows.n_edges <= DENSE_LIMIT:
        np.save(os.path.join(folder, name + '.npy'), flows.todense().astype(np.float64))
    else:
        np.savez(os.path.join(folder, name + '.npz'), ptr=flows.ptr, idx=flows.idx, val=flows.val,
                 n_edges=np.int64(flows.n_edges))


def _load_flows(folder, name):
    p = os.path.join(folder, name + '.npy')
    if os.path.exists(p):
        return np.load(p)
    d = np.load(os.path.join(folder, name + '.npz'))
    return SparseFlows(d['ptr'], d['idx'], d['val'], int(d['n_edges']))


def _onehot(choice, D):
    y = np.zeros((len(choice), D, 1))
    y[np.arange(len(choice)), choice, 0] = 1.0
    return y


def generate_dataset(n, m, folder, holes=True, seed=1030):
    """SDG:375-428 with the sparse generator: writes the 1-hop and 2-hop folders."""
    cx = sdg.random_SC_graph(n, holes=holes)
    B1, B2 = sdg.incidence_matrices(cx)
    rs = np.random.RandomState(seed)
    paths = sdg.generate_random_walks(cx, m=m, seed=int(rs.randint(1 << 30)))
    rev_paths = [p[::-1] for p in paths]
    train_mask = np.asarray([1] * int(len(paths) * 0.8) + [0] * int(len(paths) * 0.2))   # SDG:392-394
    rs.shuffle(train_mask)
    test_mask = 1 - train_mask
    nbr, deg = sdg.neighborhood_table(cx)
    D = nbr.shape[1]

    kept = []

    def both_hops(ps):
        prefixes, suffixes, last1 = sdg.split_paths(ps, rs, True, 2)
        kept.append(prefixes)
        f1 = sdg.paths_to_flows(cx, prefixes)
        t1 = np.asarray([s[0] for s in suffixes])
        c1 = np.argmax(nbr[np.asarray(last1)] == t1[:, None], axis=1)
        prefixes2 = [list(p) + [s[0]] for p, s in zip(prefixes, suffixes)]               # SDG:364-371
        f2 = sdg.paths_to_flows(cx, prefixes2)
        t2 = np.asarray([s[1] for s in suffixes])
        c2 = np.argmax(nbr[t1] == t2[:, None], axis=1)
        return (f1, _onehot(c1, D), np.asarray(last1), t1), (f2, _onehot(c2, D), t1, t2)
    fw, rv = both_hops(paths), both_hops(rev_paths)
    for h in (0, 1):
        fol = 'trajectory_data_%dhop_%s' % (h + 1, folder)
        os.makedirs(fol, exist_ok=True)
        _save_flows(fol, 'flows_in', fw[h][0]); _save_matrix(fol, 'B1', B1); _save_matrix(fol, 'B2', B2)
        np.save(os.path.join(fol, 'targets.npy'), fw[h][1])
        np.save(os.path.join(fol, 'train_mask.npy'), train_mask); np.save(os.path.join(fol, 'test_mask.npy'), test_mask)
        np.save(os.path.join(fol, 'coords.npy'), cx.coords)
        np.save(os.path.join(fol, 'last_nodes.npy'), fw[h][2]); np.save(os.path.join(fol, 'target_nodes.npy'), fw[h][3])
        _save_flows(fol, 'rev_flows_in', rv[h][0]); np.save(os.path.join(fol, 'rev_targets.npy'), rv[h][1])
        np.save(os.path.join(fol, 'rev_last_nodes.npy'), rv[h][2]); np.save(os.path.join(fol, 'rev_target_nodes.npy'), rv[h][3])
        if h == 0:                                            # the optional prefixes file (TE:282), ragged instead of pickled
            np.savez(os.path.join(fol, 'prefixes.npz'), ptr=np.concatenate([[0], np.cumsum([len(p) for p in kept[0]])]),
                     nodes=np.concatenate([np.asarray(p, np.int64) for p in kept[0]]))
    return cx


def load_prefixes(folder):
    """prefixes of the 1-hop folder as a list of node lists, or None when the folder has none (TE:281-284)."""
    p = os.path.join(folder, 'prefixes.npz')
    if os.path.exists(p):
        d = np.load(p)
        return [d['nodes'][d['ptr'][i]:d['ptr'][i + 1]].astype(int).tolist() for i in range(len(d['ptr']) - 1)]
    p = os.path.join(folder, 'prefixes.npy')
    if os.path.exists(p):
        try:
            return [list(map(int, q)) for q in np.load(p, allow_pickle=False)]
        except ValueError:
            return None                                       # a pickled object array: not read (no pickle at this boundary)
    return None


def load_dataset(folder):
    """SDG:430-447: X, [B1, B2], y, train_mask, test_mask, coords (instead of G_undir), last_nodes, target_nodes."""
    X = _load_flows(folder, 'flows_in')
    B1, B2 = _load_matrix(folder, 'B1'), _load_matrix(folder, 'B2')
    ld = lambda n: np.load(os.path.join(folder, n + '.npy'))
    coords = ld('coords') if os.path.exists(os.path.join(folder, 'coords.npy')) else None
    return X, [B1, B2], ld('targets'), ld('train_mask'), ld('test_mask'), coords, ld('last_nodes'), ld('target_nodes')


def load_reverse(folder):
    return _load_flows(folder, 'rev_flows_in'), np.load(os.path.join(folder, 'rev_targets.npy')), \
        np.load(os.path.join(folder, 'rev_last_nodes.npy'))
