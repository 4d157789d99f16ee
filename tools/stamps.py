#!/usr/bin/env python3
"""Diagnostic: run fwd_c32 from the -DSCN_STAMPS build and print where a wave's cycles go per slab iteration."""
import ctypes, os, sys
os.environ["SCN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libscone_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scone_gcn_amd import _lib, ops, synthetic_data_gen as g, trajectory_experiments as te
from scone_gcn_amd.complex import SimplicialComplex
lib = _lib.load()
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "scone")
plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
E, C, S = cx.n_edges, 32, 32
W = [torch.randn(C, C, device="cuda") * 0.1 for _ in range(3)]
x = torch.randn(S, E, 4, C, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if "f32" in sys.argv:
    os.environ["SCN_F32_MFMA"] = "1"   # forward stamps live in the fp32-MFMA kernel; backward: bf16 kernel unless "f32" is given
def run():
    if which.startswith("fwd"):
        plan.conv.forward([x], W, C, "tanh")
    else:
        plan.conv.backward([x], W, x, "tanh", True, [torch.zeros_like(w) for w in W])
run(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
lib.scn_debug_stamps = ctypes.CDLL(os.environ["SCN_LIB_PATH"]).scn_debug_stamps
lib.scn_debug_stamps(buf, 1)
run(); torch.cuda.synchronize()
lib.scn_debug_stamps(buf, 0)
names = ["wait vmcnt(0)", "barrier", "gather", "mfma+epilogue", "-", "-"] if (which == "fwd" and "f32" not in sys.argv) else ["wait vmcnt(0)", "barrier", "dma issue", "stores", "gather", "mfma+epilogue"] if which == "fwd" else ["wait vmcnt(0)", "barrier", "gather", "dgrad chain", "epilogue+dW", "-"]
waves = buf[7]; tot = sum(buf[i] for i in range(6))
wpw = 16 if (which == "fwd" and "f32" not in sys.argv) else 8
iters = plan.conv.plan_info()[0] * 32 * wpw / max(waves, 1)
print("waves", waves, "slab iterations per wave %.0f" % iters)
for i, n in enumerate(names):
    print("%-14s %6.1f %%   %8.0f cycles per iteration" % (n, 100.0 * buf[i] / tot, buf[i] / waves / iters))
print("total per iteration %.0f cycles (memtime ticks)" % (tot / waves / iters))
print("busiest wave / mean wave (stamped time over the whole launch): %.3f" % (buf[6] / (tot / waves)))
