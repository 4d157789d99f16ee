#!/bin/bash
# usage (on the GPU box): tools/small_pair_timeout.sh [outdir] -- the paired one-launch step when a partner never raises its flag: a
# diagnostic build in which the second workgroup of every pair skips one hand-over flag (-DSM_AB_DROP_POST; the poll bound cut to
# 20000 rounds so that the run takes milliseconds).  The launch must END and must say so: loss and every weight gradient NaN.
set -euo pipefail
export TMPDIR=/tmp
O=${1:-gpurun_out/pair_timeout}; mkdir -p $O
bash tools/ab_build.sh droppost "-DSM_AB_DROP_POST -DSM_SPIN_LIMIT=20000" > $O/build.txt 2>&1
SCN_SMALL_STEP=force SCN_LIB_PATH=tools/ab/lib_droppost.so timeout -k 10 120 python3 - > $O/timeout.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from scone_gcn_amd import synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(400); sc = SimplicialComplex(cx)
N = 100
paths = g.generate_random_walks(cx, m=N, seed=1)
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=1)
y = np.zeros((N, sc.max_degree, 1)); y[np.arange(N), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
stm.reseed(1030)
net = stm.Scone_GCN(1, 1e-3, N, 5e-5, verbose=False)
net.use_graph = False
net.setup(te.scone_func, [(3, 16)] * 3, shifts, inputs, y, None, np.ones(N, int), model_type="scone")
staged = net.stage(inputs, y, np.arange(N))
t0 = time.perf_counter()
loss = float(net.grad_step_staged(inputs, staged, N, apply=False))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gflat = net._flat_g.cpu().numpy()
print("|E| = %d, %d trajectories, one flag of every pair withheld: the launch ended after %.1f ms; loss = %r; %d of %d gradient entries NaN"
      % (cx.n_edges, N, dt * 1e3, loss, int(np.isnan(gflat).sum()), gflat.size))
assert np.isnan(loss) and np.isnan(gflat).all()
print("ok: the failure is visible in every output")
PY
grep -v amdgpu.ids $O/timeout.txt
