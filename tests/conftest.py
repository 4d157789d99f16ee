import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cfg1():
    """Config-1 complex + trajectories from the committed golden fixtures (dense fp64 arrays)."""
    from oracle import scone_oracle as so
    c = np.load(os.path.join(GOLDEN, "cfg1_complex.npz"))
    p = np.load(os.path.join(GOLDEN, "cfg1_paths.npz"))
    B1 = so.dense_from_coo(c["B1_row"], c["B1_col"], c["B1_val"], c["B1_shape"])
    B2 = so.dense_from_coo(c["B2_row"], c["B2_col"], c["B2_val"], c["B2_shape"])
    E = int(p["n_edges"])
    D = int(p["max_degree"])
    return {
        "edges": c["edges"], "faces": c["faces"], "coords": c["coords"], "n_nodes": int(c["n_nodes"]),
        "B1": B1, "B2": B2, "E": E, "D": D,
        "flows": so.flows_from_ragged(p["flow1_ptr"], p["flow1_idx"], p["flow1_val"], E),
        "flow_ptr": p["flow1_ptr"], "flow_idx": p["flow1_idx"], "flow_val": p["flow1_val"],
        "targets": so.onehot_targets(p["targets1"], D), "target_choice": p["targets1"],
        "last_nodes": p["last1"].astype(np.int64), "target_nodes": p["tnode1"],
        "train_mask": p["train_mask"].astype(np.int64), "test_mask": p["test_mask"].astype(np.int64),
        "rev_flows": so.flows_from_ragged(p["rev_flow1_ptr"], p["rev_flow1_idx"], p["rev_flow1_val"], E),
        "rev_targets": so.onehot_targets(p["rev_targets1"], D), "rev_last_nodes": p["rev_last1"].astype(np.int64),
    }


@pytest.fixture(scope="session")
def big_complex():
    """The benchmark complex (|E| = 996 634, BASELINE configs[3] / [4]) and its SimplicialComplex: built once per session,
    GPU tests only (the layout pass calls into the library, the tests that use it need the device anyway)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd.complex import SimplicialComplex
    cx = g.random_SC_graph(g.calibrate_n_points(1_000_000))
    assert abs(cx.n_edges - 1_000_000) < 20_000
    return cx, SimplicialComplex(cx)
