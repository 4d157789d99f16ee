"""Data-parallel path on CPU: world_size-2 gloo.  The compute inside each rank is the ORACLE gradient (tests may
use it); what is under test is scone_gcn_amd/distributed.py -- sharding, normalisation by the GLOBAL batch size,
the single all-reduce of the flat gradient buffer, and identical replicas afterwards."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scone_oracle as so
from scone_gcn_amd import distributed as dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from scone_gcn_amd import synthetic_data_gen as g
    cx = g.random_SC_graph(60, holes=False)
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    paths = g.generate_random_walks(cx, m=10, seed=2)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=3)
    nb, D = so.neighborhoods(cx.edges, cx.n_nodes)
    y = so.onehot_targets(choice, D)
    w = so.generate_weights(1, [(3, 4)] * 2, 1)
    w = [10 * a for a in w]
    return B1, B2, flows.todense().astype(np.float64), y, last, nb, w


def _flat(gs):
    return np.concatenate([g.ravel() for g in gs])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B1, B2, X, y, last, nb, w = _problem()
    L_lo, L_up = so.scone_shifts(B1, B2)
    Bc = so.make_Bconds(B1, nb)
    idx = np.array([0, 2, 3, 5, 6, 7, 9])            # the masked batch (odd size: uneven shards)
    flat = torch.zeros(sum(a.size for a in w), dtype=torch.float64)

    def grad_fn(local, total):
        m = np.zeros(len(X), int)
        m[local] = 1
        _, g = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last, X, y, m, 0.0)
        flat.add_(torch.from_numpy(_flat(g)) * (len(local) / total))   # oracle normalises by the local count
    local = dp.data_parallel_grad(idx, grad_fn, flat)
    out[rank] = (flat.numpy().copy(), local.copy())
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_equals_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    B1, B2, X, y, last, nb, w = _problem()
    L_lo, L_up = so.scone_shifts(B1, B2)
    m = np.zeros(len(X), int)
    m[[0, 2, 3, 5, 6, 7, 9]] = 1
    _, g = so.scone_loss_and_grad(w, L_lo, L_up, so.make_Bconds(B1, nb), last, X, y, m, 0.0)
    ref = _flat(g)
    assert np.abs(out[0][0] - ref).max() < 1e-12 and np.array_equal(out[0][0], out[1][0])   # identical replicas
    assert sorted(np.concatenate([out[0][1], out[1][1]]).tolist()) == [0, 2, 3, 5, 6, 7, 9]
    assert abs(len(out[0][1]) - len(out[1][1])) <= 1


def test_shard_indices_partition():
    idx = np.arange(100)[::3]
    for world in (1, 2, 3, 8):
        parts = [dp.shard_indices(idx, r, world) for r in range(world)]
        assert np.array_equal(np.concatenate(parts), idx)
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert dp.world() == (0, 1)
