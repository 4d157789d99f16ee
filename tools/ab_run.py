#!/usr/bin/env python3
"""Same-process, same-box A/B of library builds: ONE complex, ONE set of tensors, a plan per library build (tools/ab_build.sh),
and the timed kernels ALTERNATING between the builds round by round -- box-to-box and run-to-run drift cancel.

    python tools/ab_run.py --libs base=scone_gcn_amd/libscone_hip.so,x=tools/ab/lib_x.so --which fwd,bwd,bwdf,spmm \
                           --data dense,sparse --rounds 3 --reps 3

Prints per (data, kernel, build) the mean launch time over all rounds, per-round values, and output checksums (builds of one
kernel must agree unless the variant changes summation order)."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scone_gcn_amd import _lib, ops, synthetic_data_gen as g, trajectory_experiments as te   # noqa: E402
from scone_gcn_amd.complex import SimplicialComplex                                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", required=True, help="name=path,... (paths relative to the repo root)")
ap.add_argument("--edges", type=int, default=1_000_000)
ap.add_argument("--slabs", type=int, default=32)
ap.add_argument("--hidden", type=int, default=32)
ap.add_argument("--which", default="fwd,bwd,bwdf")
ap.add_argument("--data", default="dense,sparse")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

cx = g.random_SC_graph(g.calibrate_n_points(a.edges))
sc = SimplicialComplex(cx)
shifts, readout, _ = te.setup_from_complex(sc, "scone")
dev = ops.default_device()
bunch = any(k.startswith("t") for k in a.which.split(","))          # tfwd / tbwd: the fused Bunch layer (the model's third-layer launches)
if bunch:
    bshifts, bnbr, _ = te.setup_from_complex(sc, "bunch")
builds = {}
_lib.AB_OLDER_LIBRARY = True          # a "before" library built from an older revision may lack the newest entry points
for spec in a.libs.split(","):
    name, path = spec.split("=")
    path = path if os.path.isabs(path) else os.path.join(ROOT, path)
    _lib._lib = None
    _lib.LIB_PATH = path
    lib = _lib.load()
    if bunch:
        builds[name] = (lib, ops.BunchPlan(bshifts, bnbr, dev))
        print("build %s: %s (bunch)" % (name, path), flush=True)
    else:
        builds[name] = (lib, ops.SconePlan(shifts[0], shifts[1], readout, "tanh", dev))
        print("build %s: %s, plan blocks %s" % (name, path, builds[name][1].conv.plan_info()), flush=True)

E, C, S = cx.n_edges, a.hidden, a.slabs
torch.manual_seed(0)
W = [torch.randn(C, C, device=dev) * 0.1 for _ in range(3)]
W1 = [torch.randn(1, C, device=dev) * 0.1 for _ in range(3)]
which = a.which.split(",")


def call(plan, k, x, aux, yrec):
    if k in ("tfwd", "tbwd"):
        want = [True, True, False]
        if k == "tfwd":
            Wn = [[BW[l][j] if l != 2 else None for j in range(3)] for l in range(3)]
            return [o for o in plan._terms_fwd_for(want).forward(bxs, Wn, "relu", want) if o is not None]
        dzs = [bxs[0], bxs[1], None]
        Wn = [[BWb[a_][b_] if b_ != 2 else None for b_ in range(3)] for a_ in range(3)]
        dWn = [[torch.zeros(32, 32, device=dev) if (b_ != 2 and BWb[a_][b_] is not None) else None for b_ in range(3)] for a_ in range(3)]
        out = ops._terms_backward(plan._terms_bwd_for([True, True, False]), dzs, Wn, bauxs, "relu", [True] * 3, dWn)
        return [o for o in out if o is not None] + [d for row in dWn for d in row if d is not None]
    if k == "spmm":
        return plan.conv.spmm_dual(x.view(S, E, 4 * C))
    if k == "fwd1":                                  # the first layer: one input channel (x[..., :1]) -> C
        return plan.conv.forward_first(x1, W1, C, "tanh", out=out1, y=y1)
    if k == "fwd":
        return plan.conv.forward([x], W, C, "tanh")
    if k == "bwd":
        dWs = [torch.zeros_like(w) for w in W]
        return plan.conv.backward([x], W, aux, "tanh", True, dWs), dWs
    if k == "bwdf":
        dWs, dW1 = [torch.zeros_like(w) for w in W], [torch.zeros_like(w) for w in W1]
        assert plan.conv.backward_fused_first(x, W, aux, "tanh", yrec, dWs, dW1)
        return dWs, dW1
    raise ValueError(k)


def checksum(o):
    if torch.is_tensor(o):
        return "%.9e" % float(o.double().abs().sum())
    return " ".join(checksum(t) for t in o)


if bunch:
    sizes = next(iter(builds.values()))[1].sizes
    BS = 16
    bxs = [torch.randn((BS, n_, 4, 32), device=dev) for n_ in sizes]
    bauxs = [torch.relu(torch.randn((BS, n_, 4, 32), device=dev)) for n_ in sizes]
    BW = [[None] * 3 for _ in range(3)]
    BWb = [[None] * 3 for _ in range(3)]
    for k_ in range(7):
        w_ = torch.randn(32, 32, device=dev) * 0.1
        BW[ops.BUNCH_DST[k_]][ops.BUNCH_SRC[k_]] = w_
        BWb[ops.BUNCH_SRC[k_]][ops.BUNCH_DST[k_]] = w_
for data in a.data.split(","):
    x = torch.randn(S, E, 4, C, device=dev)
    if data == "sparse":      # like the benchmark's activations: ~5 % of the 64-row groups of a slab carry values, the rest exact zeros
        keep = (torch.rand(S, (E + 63) // 64, device=dev) < 0.05).repeat_interleave(64, dim=1)[:, :E]
        x *= keep[:, :, None, None]
    aux = torch.tanh(torch.randn(S, E, 4, C, device=dev))
    yrec = torch.randn(S, E, 4, 4, device=dev)
    if "fwd1" in which:
        x1 = x[..., :1].contiguous()
        out1 = torch.empty(S, E, 4, C, device=dev)
        y1 = torch.empty(S, E, 4, 4, device=dev)
    times = {(k, n): [] for k in which for n in builds}
    for k in which:
        for n, (lib, plan) in builds.items():
            _lib._lib = lib
            print("checksum %-6s %-5s %-10s %s" % (data, k, n, checksum(call(plan, k, x, aux, yrec))), flush=True)
    torch.cuda.synchronize()
    names = list(builds)
    for r in range(a.rounds):
        for k in which:
            for n in names[r % len(names):] + names[:r % len(names)]:      # rotate: no build always runs behind the same one
                lib, plan = builds[n]
                _lib._lib = lib
                with ops.KernelTimer() as kt:
                    for _ in range(a.reps):
                        call(plan, k, x, aux, yrec)
                (_, (cnt, ms)), = kt.summary().items()
                times[(k, n)].append(ms)
    for k in which:
        base = None
        for n in builds:
            t = times[(k, n)]
            mean = sum(t) / len(t)
            base = mean if base is None else base
            print("%-6s %-5s %-10s %8.3f ms  (%+.2f %% vs first)  rounds: %s" % (data, k, n, mean, 100.0 * (mean / base - 1.0),
                                                                           " ".join("%.3f" % v for v in t)), flush=True)
    del x, aux, yrec
    torch.cuda.empty_cache()
