#!/usr/bin/env python3
"""Diagnostic: run fwd_c32 / bwd_c32 from the -DSCN_STAMPS build (tools/build_stamps.sh) and print where a wave's cycles go per
slab iteration -- over all waves, and per wave INDEX of the workgroup (who waits at the slab barrier, who arrives last).

    python tools/stamps.py [fwd] [bwd] [bwdf] [dense] [sparse] [zeros]
"""
import ctypes, os, sys
os.environ["SCN_LIB_PATH"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libscone_hip_stamps.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scone_gcn_amd import _lib, ops, synthetic_data_gen as g, trajectory_experiments as te
from scone_gcn_amd.complex import SimplicialComplex
lib = _lib.load()
dbg = ctypes.CDLL(os.environ["SCN_LIB_PATH"])
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "scone")
plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
E, C, S = cx.n_edges, 32, 32
W = [torch.randn(C, C, device="cuda") * 0.1 for _ in range(3)]
W1 = [torch.randn(1, C, device="cuda") * 0.1 for _ in range(3)]
kinds = [a for a in sys.argv[1:] if a in ("fwd", "bwd", "bwdf")] or ["fwd"]
datas = [a for a in sys.argv[1:] if a in ("dense", "sparse", "zeros")] or ["dense"]
for data in datas:
    x = torch.randn(S, E, 4, C, device="cuda")
    if data == "sparse":      # like the benchmark's activations: ~5 % of the 64-row groups of a slab carry values, the rest are exact zeros
        keep = (torch.rand(S, (E + 63) // 64, device="cuda") < 0.05).repeat_interleave(64, dim=1)[:, :E]
        x *= keep[:, :, None, None]
    if data == "zeros":       # nearer to a real batch still: 0.2 % of the groups
        keep = (torch.rand(S, (E + 63) // 64, device="cuda") < 0.002).repeat_interleave(64, dim=1)[:, :E]
        x *= keep[:, :, None, None]
    aux = torch.tanh(x)
    for which in kinds:
        def run():
            if which == "fwd":
                plan.conv.forward([x], W, C, "tanh")
            elif which == "bwd":
                plan.conv.backward([x], W, aux, "tanh", True, [torch.zeros_like(w) for w in W])
            else:
                y = torch.randn(S, E, 4, 4, device="cuda")
                plan.conv.backward_fused_first(x, W, aux, "tanh", y, [torch.zeros_like(w) for w in W], [torch.zeros_like(w) for w in W1])
        run(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        wbuf = (ctypes.c_ulonglong * 160)()
        dbg.scn_debug_stamps(buf, 1)
        dbg.scn_debug_stamps_waves(wbuf, 1)
        run(); torch.cuda.synchronize()
        dbg.scn_debug_stamps(buf, 0)
        dbg.scn_debug_stamps_waves(wbuf, 0)
        names = ["wait vmcnt(0)", "barrier", "gather", "mfma+epilogue", "-", "-"] if which == "fwd" else ["wait vmcnt(0)", "barrier", "gather", "dgrad chain", "epilogue+dW", "-"]
        waves = buf[7]; tot = sum(buf[i] for i in range(6))
        wpw = 16 if which == "fwd" else 8
        iters = plan.conv.plan_info()[0] * 32 * wpw / max(waves, 1)
        print("==== %s, %s data: waves %d, slab iterations per wave %.0f" % (which, data, waves, iters))
        for i, n in enumerate(names):
            print("%-14s %6.1f %%   %8.0f cycles per iteration" % (n, 100.0 * buf[i] / tot, buf[i] / waves / iters))
        print("total per iteration %.0f cycles (memtime ticks)" % (tot / waves / iters))
        print("busiest wave / mean wave (stamped time over the whole launch): %.3f" % (buf[6] / (tot / waves)))
        print("per wave index (cycles per barrier interval): wait, barrier skew (rms), segments 2-4 | mean SIMD id")
        for w in range(wpw):
            r = [wbuf[w * 10 + i] for i in range(10)]
            n = max(r[7], 1)
            print("wave %2d  wait %6.0f  skew %6.0f (rms %6.0f)  %6.0f %6.0f %6.0f | simd %.2f  intervals %d"
                  % (w, r[0] / n, r[1] / n, (r[6] * 1024.0 / n) ** 0.5, r[2] / n, r[3] / n, r[4] / n, r[8] / max(r[9], 1), r[7]))
