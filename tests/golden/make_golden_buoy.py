#!/usr/bin/env python3
"""Config-3 fixture: the reference's data file ocean_drifters_data/dataBuoys.jld2 (3.7 MB, HDF5) reduced to the
arrays SCoNe needs (edge list, triangle list, hexagon centres, 339 node trajectories) -> tests/golden/buoy.npz.

Run in the authoring container only:   python tests/golden/make_golden_buoy.py
The file is parsed with scone_gcn_amd/jld2_reader.py (h5py is not installed); the conversion rules of the reference's
buoy_data.py are restated in scone_gcn_amd/buoy_data.py and tested against the counts SURVEY.md section 2 records.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scone_gcn_amd.buoy_data import read_buoy_file   # noqa: E402

elist, tlist, coords, trajs = read_buoy_file("/root/reference/ocean_drifters_data/dataBuoys.jld2")
ptr = np.cumsum([0] + [len(t) for t in trajs]).astype(np.int32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "buoy.npz"), elist=elist.astype(np.int16),
                    tlist=tlist.astype(np.int16), coords=coords, traj_ptr=ptr,
                    traj_nodes=np.concatenate([np.asarray(t) for t in trajs]).astype(np.int16))
print("edges", elist.shape, "triangles", tlist.shape, "hexagons", coords.shape, "trajectories", len(trajs))
