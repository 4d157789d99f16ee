#!/usr/bin/env python3
"""Does the first layer's kernel (fwd_c1: one input channel in, 16 GB out) depend on WHERE its tensors sit?  Its mean time moves
between 3.0 and 3.8 ms from process to process while every launch inside a process takes the same time.
Here: one process, the output placed at different byte offsets of one oversized buffer, and in freshly allocated buffers after
other allocations of different sizes; 5 launches each (hip events).  usage: python tools/fwd1_placement.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scone_gcn_amd import ops, synthetic_data_gen as g, trajectory_experiments as te
from scone_gcn_amd.complex import SimplicialComplex
cx = g.random_SC_graph(g.calibrate_n_points(1_000_000)); sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "scone")
plan = ops.get_scone_plan(shifts[0], shifts[1], readout, "tanh", ops.default_device())
E, C, S = cx.n_edges, 32, 32
dev = "cuda"
torch.manual_seed(0)
W1 = [torch.randn(1, C, device=dev) * 0.1 for _ in range(3)]
x = torch.zeros(S, E, 4, 1, device=dev)
x[:, ::977] = 1.0                                   # a few non-zero flow entries, as a batch of paths has
n_out, n_y = S * E * 4 * C, S * E * 4 * 4


def timed(out, y, reps=5):
    plan.conv.forward_first(x, W1, C, "tanh", out=out, y=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        plan.conv.forward_first(x, W1, C, "tanh", out=out, y=y)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


big = torch.empty(n_out + (1 << 26), device=dev)
ybuf = torch.empty(n_y + (1 << 22), device=dev)
print("output at byte offsets of one buffer (base %#x):" % big.data_ptr(), flush=True)
for off in (0, 64, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096, 1 << 21, 3 << 20, 1 << 24, (1 << 26) - 256):
    o = big[off // 4: off // 4 + n_out].view(S, E, 4, C)
    print("   +%-10d %.3f ms" % (off, timed(o, ybuf[:n_y].view(S, E, 4, 4))), flush=True)
del big
print("fresh allocations after fillers of different sizes:", flush=True)
keep = []
for filler in (0, 1 << 20, 3 << 20, 1 << 28, 1 << 30, 5 << 30, 17 << 30, 1 << 29, 7 << 30, 1 << 27, 9 << 30, 3 << 29):
    if filler:
        keep.append(torch.empty(filler, device=dev, dtype=torch.uint8))
    o = torch.empty(S, E, 4, C, device=dev)
    t_k = timed(o, ybuf[:n_y].view(S, E, 4, 4))
    o.zero_(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        o.zero_()
    b.record(); torch.cuda.synchronize()
    print("   filler %-12d out at %#x: %.3f ms   (torch zero_ of the same tensor: %.3f ms)" % (filler, o.data_ptr(), t_k, a.elapsed_time(b) / 5),
          flush=True)
    del o
    torch.cuda.empty_cache()
