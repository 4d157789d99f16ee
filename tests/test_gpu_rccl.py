"""RCCL on the hardware that exists: ONE fresh child process with backend "nccl" (= RCCL on ROCm) and world size 1 runs the
data-parallel step with the gradient all-reduce FORCED (distributed.all_reduce_sum_(force=True) on the device-resident flat
gradient buffer) and bench.py's max-over-ranks reduction (distributed.all_max on a CUDA tensor).  This proves librccl loads
under torch on the box, the communicator initialises, and the collective path of SURVEY section 8e runs on CUDA tensors; the
multi-GPU curve itself needs more than one GPU.  The child is started BEFORE anything in it touches the GPU and is never
re-exec'd."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
from scone_gcn_amd import distributed as dp, synthetic_data_gen as g, trajectory_experiments as te, scone_trajectory_model as stm
from scone_gcn_amd.complex import SimplicialComplex
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
cx = g.random_SC_graph(1500)
sc = SimplicialComplex(cx)
paths = g.generate_random_walks(cx, m=37, seed=5, waypoint_pool=8, metric="euclid")
flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=6)
y = np.zeros((37, sc.max_degree, 1)); y[np.arange(37), choice, 0] = 1.0
shifts, readout, _ = te.setup_from_complex(sc, "scone")
inputs = [readout, last, flows]
res = {}
for forced in (False, True):
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-2, 30, 5e-5, verbose=False)
    net.collective_always = forced
    net.setup(te.scone_func, [(3, 32)] * 3, shifts, inputs, y, None, np.ones(37, int), model_type="scone")
    with torch.no_grad():
        for w in net.weights:
            w.mul_(12.0)
    mask = np.ones(37, int); mask[[1, 8, 30]] = 0
    part = float(net.grad_step(inputs, y, mask, apply=False))
    res["grad%d" % forced] = net._flat_g.cpu().numpy().copy()
    res["part%d" % forced] = part
    for _ in range(2):
        net.grad_step(inputs, y, mask)
    res["w%d" % forced] = net._flat_w.cpu().numpy().copy()
# the collective itself on a device tensor of the gradient buffer's size, and the bench's max-over-ranks
flat = torch.arange(6272, device="cuda", dtype=torch.float32)
ref = flat.clone()
assert dp.all_reduce_sum_(flat, force=True) is flat
torch.cuda.synchronize()
res["sum_ok"] = bool(torch.equal(flat, ref))
res["max"] = dp.all_max(1.25, device="cuda", force=True)
res["scalar"] = dp.all_reduce_scalar(2.5, "cuda")
dist.barrier()
torch.cuda.synchronize()
np.savez(out, **res)
dist.destroy_process_group()
'''


def test_rccl_one_rank_process_group_runs_the_gradient_all_reduce_on_the_device(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "rccl_rank.py"
    script.write_text(_SCRIPT)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / "rccl.npz")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(tmp_path / "rccl.npz")
    assert bool(d["sum_ok"]) and float(d["max"]) == 1.25 and float(d["scalar"]) == 2.5
    # a one-rank sum is the identity: the forced-collective step equals the plain one bit for bit
    assert np.abs(d["grad1"]).max() > 1e-4
    assert np.array_equal(d["grad0"], d["grad1"]) and np.array_equal(d["w0"], d["w1"])
    assert float(d["part0"]) == float(d["part1"])
