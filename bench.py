#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec, forward + backward (+ gradient all-reduce + Adam), 3-layer SCoNe.

    python bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no launcher environment the script starts N fresh ranks itself through torch.distributed.run
(before anything touches the GPU) and exits with their code; launched by `python -m torch.distributed.run ... bench.py
--gpus N` it is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).  A rank count that
differs from --gpus is an error, never a silent single-GPU run.

Workload (BASELINE.json metric; configs[3]): synthetic complex with |E| ~ 1M (370 000 points, the reference
generator's recipe), hidden = 32, fp32, GLOBAL batch 4096 sharded over the ranks (4096 / N trajectories per GPU ->
strong scaling; at N = 1 one GPU runs all 4096 as 32 micro-batches of 128).  A step = one optimiser step on a batch
that is already resident in HBM: forward, loss, backward, RCCL all-reduce of the flat weight-gradient buffer, fused
ridge + Adam.  Rank 0 prints the full record as `DETAIL {...}` (also written to bench_full.json / --out) and then, as the
LAST stdout line, ONE compact JSON line of at most 6000 characters (compact_line): the driver keeps an 8 KB tail.

Extra objects in the full record (N = 1 only, unless noted; the compact line carries their flat summaries):
  roofline      the kernel FAMILY (template variants summed: the plain and the fused-first backward are one family) that takes
                the most time in the step, timed live with events on the launch stream: algorithmic bytes of its launches /
                their duration vs the 8 TB/s HBM peak, every variant's own fraction in `variants`; `traffic` = HBM bytes per
                launch from the committed rocprofv3 --pmc passes, cited only while the kernel sources are the ones the passes
                ran on; `spmm_dual_*` = the SpMM half of the metric (dense random X); `dense_random_*` repeats the two fused
                C=32 kernels on dense random tensors (the benchmark's activations are mostly exact zeros).
  validation    (every N) loss of the batch after the steps, summed over ranks, and the replicas' weights against rank 0's:
                the global batch is drawn once with seed 1030 on every rank and sharded by index, so N = 1 and N > 1 agree.
  cpu_baseline  the reference formulation on this box's host cores: B2 = fp32 torch.sparse_csr forward + autograd
                backward on a bounded sample of the SAME workload (oracle/torch_sparse.py), B1 = the dense-faithful fp32
                torch restatement (dense shifts, full-N forward then mask, autograd, Adam; oracle/torch_dense.py) on
                configs[0]; threads = torch.get_num_threads().
  parity        loss + all ten weight gradients of the HIP step against the fp64 scipy-CSR oracle on trajectories of the
                full-size complex.
  configs       BASELINE.json configs[1] (|E|~50k, hidden 16, batch 1024), configs[2] (ocean drifters, full batch),
                configs[4] (Bunch, |E|~1M, hidden 32, batch 1024) and Ebli at |E|~1M, each with live kernel timing.
  weak_scaling  the same step at 512 trajectories per GPU (every N).
  zero_skipping the same step with exact zero-skipping work lists (reported BESIDE the dense value, never as it).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--edges", type=int, default=1_000_000, help="target |E| of the synthetic complex")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--global-batch", type=int, default=4096, help="trajectories per optimiser step over ALL ranks")
    ap.add_argument("--per-gpu-batch", type=int, default=0,
                    help="> 0: weak scaling instead -- this many trajectories per GPU (global batch = N x this)")
    ap.add_argument("--cpu-sample", type=int, default=4, help="trajectories in the sparse CPU-baseline sample (0 = skip)")
    ap.add_argument("--parity-sample", type=int, default=4, help="trajectories of the full-size oracle comparison (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="0: headline only (no other configs, no SpMM, no skipping modes)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal: ranks may share one GPU)")
    ap.add_argument("--skip-modes", default="zeros,field",
                    help="also time the 512-per-GPU step in these zero-skipping modes ('' = none)")
    ap.add_argument("--out", default="", help="also write the FULL record to this file (e.g. profiles/rNN_bench_1gpu.json)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh ranks (children) and wait.  Nothing here touches the GPU:
    torch.cuda.device_count() does not initialise it on this image, and the children are new processes."""
    import torch
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible\n" % (args.gpus, ndev))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): the oracle side, timed -- never part of `value`
# ----------------------------------------------------------------------------------------------------------------

def cpu_baseline_sparse(cx, sc, flows, choice, last, hidden, n_sample):
    """B2: fp32 torch.sparse_csr restatement (multi-threaded), fwd + autograd bwd on n_sample trajectories of the workload."""
    import warnings
    import torch
    from oracle import scone_oracle as so
    from oracle import torch_sparse as ts
    from scone_gcn_amd import synthetic_data_gen as g
    warnings.filterwarnings("ignore", message="Sparse CSR tensor support is in beta")
    sel = np.arange(n_sample)
    B1, B2 = g.incidence_matrices(cx)
    L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
    Sl, Su = ts.csr_tensor(L_lo), ts.csr_tensor(L_up)
    X = torch.from_numpy(flows.select(sel).todense())
    y = torch.tensor(so.onehot_targets(choice[sel], sc.max_degree), dtype=torch.float32)
    w = [torch.tensor(a, dtype=torch.float32) for a in so.generate_weights(1, [(3, hidden)] * 3, 1)]
    rows = ts.make_inc_rows(B1, sc.nbrhoods)
    ts.loss_and_grad(w, Sl, Su, Sl, Su, rows, last[sel[:1]], X[:1], y[:1], 5e-5)      # warm-up: thread pool, sparse kernels, autograd
    passes = 2
    t0 = time.perf_counter()
    for _ in range(passes):
        ts.loss_and_grad(w, Sl, Su, Sl, Su, rows, last[sel], X, y, 5e-5)
    dt = (time.perf_counter() - t0) / passes
    return {"value": n_sample / dt, "unit": "trajectories/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "dtype": "f32", "host_cpus": os.cpu_count(),
            "sample": "%d trajectories of the same |E|=%d complex, hidden %d: this repo's CPU restatement of the reference "
                      "formulation (TE:137-152, STM:42-56; oracle/torch_sparse.py -- the reference's JAX code cannot run here) "
                      "in fp32 with torch.sparse_csr shifts (dense shifts do not exist at this size), forward + autograd "
                      "backward, no optimiser, %d torch threads; one warm-up trajectory, then the mean of %d passes, %.1f s each"
                      % (n_sample, cx.n_edges, hidden, torch.get_num_threads(), passes, dt)}


def cpu_baseline_dense_cfg1(steps=2):
    """B1: the reference's own formulation on its own configuration (configs[0]: 400-point complex, 1000 trajectories,
    hidden 16, batch 100): dense (E, E) fp32 shifts, forward over ALL N then mask (STM:46), autograd, Adam (STM:306-326)."""
    import torch
    from oracle import scone_oracle as so
    from oracle import torch_dense as td
    from scone_gcn_amd import synthetic_data_gen as g
    cx = g.random_SC_graph(400)
    paths = g.generate_random_walks(cx, m=1000, seed=1030)
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7)
    B1, B2 = (m.toarray() for m in g.incidence_matrices(cx))
    L_lo, L_up = (torch.tensor(m, dtype=torch.float32) for m in so.scone_shifts(B1, B2))
    nb, D = so.neighborhoods(cx.edges, cx.n_nodes)
    B1x = torch.tensor(np.concatenate([B1, np.zeros((1, B1.shape[1]))]), dtype=torch.float32)
    nbt = torch.as_tensor(nb)
    X = torch.from_numpy(flows.todense())
    y = torch.tensor(so.onehot_targets(choice, D), dtype=torch.float32)
    ws = [torch.tensor(a, dtype=torch.float32, requires_grad=True) for a in so.generate_weights(1, [(3, 16)] * 3, 1)]
    opt = torch.optim.Adam(ws, lr=1e-3, eps=1e-8)
    rs = np.random.RandomState(1030)
    train_mask = np.array([1] * 800 + [0] * 200)
    lastt = torch.as_tensor(last)
    times = []
    for _ in range(steps + 1):
        bm = torch.as_tensor(so.draw_batch_mask(rs, 1000, 100, train_mask))
        t0 = time.perf_counter()
        opt.zero_grad()
        out = td.scone_func(ws, L_lo, L_up, B1x, nbt, lastt, X)
        td.loss_fn(out, y, bm, ws, 5e-5).backward()
        opt.step()
        times.append((time.perf_counter() - t0, int(bm.sum())))
    dt = sum(t for t, _ in times[1:]) / steps
    nb_mean = sum(n for _, n in times[1:]) / steps
    return {"optimiser_steps_per_s": 1.0 / dt, "batch_trajectories_per_s": nb_mean / dt, "forward_trajectories_per_s": 1000 / dt,
            "cores": int(torch.get_num_threads()), "dtype": "f32",
            "sample": "configs[0]: |E|=%d, 1000 trajectories, hidden 16, batch 100 (~%d after the train mask): dense fp32 "
                      "shifts, full-N forward then mask, autograd, Adam; %d steps after one warm-up, %.2f s/step"
                      % (cx.n_edges, round(nb_mean), steps, dt)}


def oracle_parity(cx, sc, net, inputs, flows, choice, last, hidden, n_sample):
    """loss + every weight gradient of the HIP step vs the fp64 scipy-CSR oracle on n_sample full-size trajectories."""
    import scipy.sparse as sp
    import torch
    from oracle import scone_oracle as so
    from scone_gcn_amd import synthetic_data_gen as g
    sel = np.arange(n_sample)
    B1, B2 = g.incidence_matrices(cx)
    L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
    B1x = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()
    Bc = lambda n: B1x[sc.nbrhoods[n]].toarray()
    X = flows.select(sel).todense().astype(np.float64)
    y = so.onehot_targets(choice[sel], sc.max_degree)
    w = [t.detach().cpu().numpy().astype(np.float64) for t in net.weights]
    t0 = time.perf_counter()
    ref_loss, ref_g = so.scone_loss_and_grad(w, L_lo, L_up, Bc, last[sel], X, y, np.ones(n_sample, int), 0.0)
    dt = time.perf_counter() - t0
    staged = net.stage(inputs, y, sel, skip="dense")
    loss = float(net.grad_step_staged(inputs, staged, n_sample, apply=False))
    got = [t.detach().cpu().numpy().astype(np.float64) for t in net._grads]
    gmax = max(float(np.abs(r).max()) for r in ref_g)
    err = max(float(np.abs(a - b).max()) for a, b in zip(got, ref_g))
    tol = 1e-5
    return {"n": n_sample, "max_err": max(err, abs(loss - ref_loss)), "loss_err": abs(loss - ref_loss), "grad_max_abs_err": err,
            "grad_err_rel_to_max_grad": err / gmax, "max_abs_grad": gmax, "tol": tol, "oracle_s": dt,
            "pass": bool(max(err, abs(loss - ref_loss)) <= tol),
            "oracle": "oracle/scone_oracle.py, fp64 NumPy + scipy CSR, same trajectories / weights, |E|=%d, hidden %d" % (cx.n_edges, hidden)}


# ----------------------------------------------------------------------------------------------------------------
# one timed configuration
# ----------------------------------------------------------------------------------------------------------------

def time_config(net, inputs, staged, total, steps, warmup, sync, all_max):
    """W untimed + K timed optimiser steps on resident micro-batches; returns (seconds over ranks (max), kernel table)."""
    from scone_gcn_amd import ops
    for _ in range(warmup):
        net.grad_step_staged(inputs, staged, total)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.grad_step_staged(inputs, staged, total)
    sync()
    dt = all_max(time.perf_counter() - t0)
    with ops.KernelTimer() as kt:                      # separate pass: the events do not sit in the timed region
        net.grad_step_staged(inputs, staged, total)
    return dt, kt.table()


def family(key):
    """Timer key -> kernel family: the template variants of one kernel (the fused-first backward `+ dW_first`, the
    `(dW only)` form) are one family; different kernels (first layer c1->C vs the fused C->C) are not."""
    return key.replace(" + dW_first", "").replace(" (dW only)", "").replace(" + partial", "")


def dominant(table):
    """(family, [keys]) of the kernel family with the largest sum of launches x mean time among those with a byte model."""
    fams = {}
    for k, r in table.items():
        if r["alg_bytes"]:
            fams.setdefault(family(k), []).append(k)
    if not fams:
        return None, None
    f = max(fams, key=lambda f: sum(table[k]["launches"] * table[k]["avg_ms"] for k in fams[f]))
    return f, sorted(fams[f])


def roofline_of(table, units_per_launch):
    """The dominant kernel FAMILY of the timed step: achieved = algorithmic bytes of all its launches / their summed duration
    (HIP events on the launch stream, ops.KernelTimer); every template variant's own fraction beside it."""
    fam, keys = dominant(table)
    if fam is None:
        return None
    n = sum(table[k]["launches"] for k in keys)
    ms = sum(table[k]["launches"] * table[k]["avg_ms"] for k in keys)
    nb = sum(table[k]["launches"] * table[k]["alg_bytes"] for k in keys)
    step_ms = sum(r["launches"] * r["avg_ms"] for r in table.values())
    ach = nb / (ms * 1e-3)
    out = {"bound": "hbm", "kernel": fam, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
           "traffic": None, "launch_ms": ms / n, "launches_per_step": n, "algorithmic_bytes_per_launch": nb / n,
           "units_per_launch": units_per_launch, "share_of_step_kernel_time": ms / step_ms if step_ms else None,
           "variants": {k: {"launches": table[k]["launches"], "launch_ms": table[k]["avg_ms"],
                            "algorithmic_bytes_per_launch": table[k]["alg_bytes"],
                            "frac": table[k]["alg_bytes"] / (table[k]["avg_ms"] * 1e-3) / HBM_PEAK} for k in keys}}
    for i, k in enumerate(keys):                        # flat copies: a parser that drops nested objects still sees them
        out["variant%d" % i] = "%s: %.3f of peak, %.3f ms x %d" % (k, out["variants"][k]["frac"], table[k]["avg_ms"], table[k]["launches"])
    return out


def traffic_file():
    """The newest committed profiles/rNN_pmc_traffic.json (rocprofv3 --pmc passes merged by tools/pmc_merge.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return files[-1] if files else None


def measured_traffic(section, keys):
    """(HBM bytes per launch, source) of a timed kernel family from the committed rocprofv3 --pmc passes: 2 * FETCH_SIZE +
    WRITE_SIZE (MI355X_MICROARCH.md section HBM), launch-weighted over the family's variants.  The file carries the hash of the
    kernel sources it was measured on (scone_gcn_amd._lib.sources_sha); when the checkout's differs the bytes are NOT cited."""
    from scone_gcn_amd import _lib
    path = traffic_file()
    if path is None:
        return None, "no profiles/rNN_pmc_traffic.json"
    doc = json.load(open(path))
    rel = os.path.relpath(path, ROOT)
    if doc.get("kernel_sources_sha") != _lib.sources_sha():
        return None, "%s was measured on other kernel sources (%s, checkout %s): not cited" % (
            rel, str(doc.get("kernel_sources_sha"))[:12], _lib.sources_sha()[:12])
    sec = doc.get(section, {})
    tot = n = 0.0
    names = []
    for name, row in sec.get("kernels", {}).items():
        if set(row.get("timer_keys", [])) & set(keys) and row.get("hbm_bytes_per_launch"):
            tot += row["hbm_bytes_per_launch"] * row.get("launches_seen", 1)
            n += row.get("launches_seen", 1)
            names.append(name)
    if not n:
        return None, "%s [%s] has no row for %s" % (rel, section, keys)
    return tot / n, "%s [%s] %s (%s)" % (rel, section, ", ".join(names), sec.get("launch", ""))


def slim(table):
    return {k: {"launches": r["launches"], "avg_ms": round(r["avg_ms"], 4),
                "GB/s": None if r["GB/s"] is None else round(r["GB/s"], 1)} for k, r in table.items()}


def make_net(model, hidden, sc, flows, last, y, B, skip="dense", layers=None):
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import trajectory_experiments as te
    shifts, readout, _ = te.setup_from_complex(sc, model)
    inputs = [readout, last, flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False, skip_mode=skip)
    net.setup(te.MODEL_FUNCS[model], layers or [(7 if model == "bunch" else 3, hidden)] * 3, shifts, inputs, y, None, np.ones(B, int),
              model_type=model)
    return net, inputs


def dataset(cx, sc, n, seed):
    from scone_gcn_amd import synthetic_data_gen as g
    paths = g.generate_random_walks(cx, m=n, seed=seed, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=seed + 7)
    y = np.zeros((n, sc.max_degree, 1))
    y[np.arange(n), choice, 0] = 1.0
    return flows, choice, last, y


def side_config(name, model, cx, sc, hidden, batch, steps, sync, seed=1030, data=None, layers=None, traffic_section=None):
    """One more BASELINE configuration on this GPU: step time, trajectories/s, its own dominant-kernel roofline."""
    import torch
    t0 = time.perf_counter()
    flows, choice, last, y = data if data is not None else dataset(cx, sc, batch, seed)
    net, inputs = make_net(model, hidden, sc, flows, last, y, batch, layers=layers)
    staged = net.stage(inputs, y, np.arange(batch))
    mb = staged[0][0].shape[0] * 4
    torch.cuda.synchronize()
    setup = time.perf_counter() - t0
    dt, table = time_config(net, inputs, staged, batch, steps, 1, sync, lambda x: x)
    # A step replayed from a HIP graph is tens of microseconds: an event pair around a 40 us kernel costs about as much as the kernel
    # (round 4's line had launch_ms > ms_per_step for the drifter complex).  There the kernels' durations are the replayed step's
    # wall time shared out in proportion to their event times -- an upper bound per kernel, consistent with ms_per_step.
    timing = "hip events on the launch stream"
    ev_ms = sum(r["launches"] * r["avg_ms"] for r in table.values())
    step_ms = dt / steps * 1e3
    if net._graphs and ev_ms > step_ms > 0:
        for r in table.values():
            r["avg_ms"] *= step_ms / ev_ms
            r["GB/s"] = (r["alg_bytes"] / (r["avg_ms"] * 1e-3) / 1e9) if (r["alg_bytes"] and r["avg_ms"] > 0) else None
        timing = "graph-replayed step: wall time of the replay shared out by the kernels' event times (events cost as much as these kernels)"
    out = {"workload": name, "model": model, "edges": cx.n_edges, "nodes": cx.n_nodes, "faces": cx.n_faces, "hidden": hidden,
           "batch": batch, "micro_batch": mb, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "value": batch * steps / dt, "unit": "trajectories/s", "plan": type(net._plan(inputs)).__name__,
           "roofline": roofline_of(table, mb), "kernels": slim(table), "setup_s": round(setup, 1)}
    if layers:
        out["hidden_layers"] = [list(l) for l in layers]
    if out["roofline"]:
        out["roofline"]["timing"] = timing
    if out["roofline"] and traffic_section:
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(traffic_section, list(out["roofline"]["variants"]))
    out["graph_replayed_step"] = bool(net._graphs)
    alg = sum(r["alg_bytes"] * r["launches"] for r in table.values() if r["alg_bytes"])
    out["step_model"] = {"kernel_algorithmic_bytes_per_trajectory": alg / batch,
                         "frac_of_hbm_peak_whole_step": out["value"] * alg / batch / HBM_PEAK}
    del net, staged
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------------------------------
# the printed record: the driver keeps an 8 KB tail of stdout, so the LAST line is a compact form (<= COMPACT_LIMIT
# characters) and the full record goes to an earlier `DETAIL {...}` line and to bench_full.json next to this script
# ----------------------------------------------------------------------------------------------------------------

COMPACT_LIMIT = 6000


def _sig(x, digits=5):
    """Floats to `digits` significant digits (the compact line is for reading and parsing, the full record keeps everything)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _cut(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(line, limit=COMPACT_LIMIT):
    """The record the driver parses: the headline keys, `config`, a FLAT `roofline` (family fraction, traffic, its variants as
    strings, the SpMM half of the metric, the dense-random fractions), `cpu_baseline`, the self-validation and one
    {value, ms_per_step, kernel, frac} entry per side configuration.  Optional parts are dropped, least important first, until
    the serialised line fits `limit` characters; the headline keys, `roofline` and `cpu_baseline` are never dropped."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: line.get(k) for k in head}
    cfg = dict(line.get("config") or {})
    cfg["workload"] = _cut(cfg.get("workload", ""), 220)
    if cfg.get("arithmetic"):
        cfg["arithmetic"] = _cut(cfg["arithmetic"], 140)
    if cfg.get("collective"):
        cfg["collective"] = _cut(cfg["collective"], 120)
    out["config"] = cfg
    rf = line.get("roofline")
    if rf:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "launches_per_step",
                "algorithmic_bytes_per_launch", "units_per_launch", "share_of_step_kernel_time", "timing", "variant0", "variant1",
                "variant2", "spmm_dual_frac", "spmm_dual_GBps", "spmm_dual_ms", "spmm_dual_traffic", "spmm_dual_algorithmic_bytes",
                "dense_random_fwd_frac", "dense_random_fwd_ms", "dense_random_bwd_frac", "dense_random_bwd_ms", "measured_copy_GBps")
        out["roofline"] = {k: rf[k] for k in keep if k in rf}
        if rf.get("traffic_source"):
            out["roofline"]["traffic_source"] = _cut(rf["traffic_source"], 90)
    else:
        out["roofline"] = None
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "dtype") if k in cb}
        out["cpu_baseline"]["sample"] = _cut(cb.get("sample", ""), 260)
        d0 = cb.get("dense_faithful_configs0")
        if d0:
            out["cpu_baseline"]["configs0_dense_faithful_steps_per_s"] = d0.get("optimiser_steps_per_s")
    else:
        out["cpu_baseline"] = None
    out["loss"] = line.get("loss")
    out["replicas_identical"] = line.get("replicas_identical")
    par = line.get("parity")
    if par:
        out["parity"] = {"max_err": par.get("max_err"), "tol": par.get("tol"), "pass": par.get("pass"), "n": par.get("n")}
    optional = []                                             # (key, value) in order of importance; dropped from the END
    cf = {}
    for name, c in (line.get("configs") or {}).items():
        r = c.get("roofline") or {}
        cf[name] = {"value": c.get("value"), "ms_per_step": c.get("ms_per_step"), "kernel": r.get("kernel"), "frac": r.get("frac")}
        if r.get("traffic"):
            cf[name]["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes_per_launch"]
        if "layer_kernels_step" in c:
            cf[name]["layer_kernels_ms_per_step"] = c["layer_kernels_step"].get("ms_per_step")
    if cf:
        optional.append(("configs", cf))
    sm = line.get("step_model")
    if sm:
        optional.append(("step_model", {"kernel_algorithmic_bytes_per_trajectory": sm.get("kernel_algorithmic_bytes_per_trajectory"),
                                        "kernel_model_frac_of_hbm_peak_whole_step": sm.get("kernel_model_frac_of_hbm_peak_whole_step")}))
    ws = line.get("weak_scaling")
    if ws:
        optional.append(("weak_scaling", {k: ws.get(k) for k in ("per_gpu_batch", "global_batch", "value", "ms_per_step")}))
    zs = line.get("zero_skipping")
    if zs:
        optional.append(("zero_skipping", {m: {"value": v.get("value"), "ms_per_step": v.get("ms_per_step")} for m, v in zs.items()}))
    if line.get("kernels"):
        optional.append(("kernels", {k: [r["launches"], r["avg_ms"]] for k, r in line["kernels"].items()}))
    val = line.get("validation")
    if val:
        optional.append(("validation", {k: val.get(k) for k in ("max_abs_weight_deviation_from_rank0", "weights_sum", "weights_l2",
                                                                "optimiser_steps_taken")}))
    out["detail"] = "full record: the preceding 'DETAIL {...}' stdout line and bench_full.json"
    for k, v in optional:
        out[k] = v
    out = _sig(out)
    for k in ("value", "ms_per_step", "loss"):                # the numbers other records are checked against keep every digit
        out[k] = line.get(k)
    while len(json.dumps(out, separators=(",", ":"))) > limit and optional:
        k, _ = optional.pop()
        out.pop(k, None)
    if len(json.dumps(out, separators=(",", ":"))) > limit:      # still too long: strings go next, numbers stay
        out["config"] = {k: v for k, v in out["config"].items() if not isinstance(v, str) or k == "parallelism"} | {
            "workload": _cut(out["config"].get("workload", ""), 80)}
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = _cut(out["cpu_baseline"]["sample"], 80)
        if out.get("roofline"):
            out["roofline"].pop("traffic_source", None)
    return out


def emit(line, out_path=None):
    """Print the full record on a prefixed line, write it to bench_full.json (and `out_path`), then print the compact record
    as the LAST line of stdout."""
    full = json.dumps(line)
    for path in (os.path.join(ROOT, "bench_full.json"), out_path):
        if path:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
                with open(path, "w") as f:
                    f.write(full + "\n")
            except OSError as e:                              # a read-only checkout must not cost the measurement
                sys.stderr.write("bench.py: could not write %s: %s\n" % (path, e))
    print("DETAIL " + full)
    sys.stdout.flush()
    print(json.dumps(compact_line(line), separators=(",", ":")))
    sys.stdout.flush()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU\n" % (args.gpus, world))
        sys.exit(2)
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    from scone_gcn_amd import distributed as dp
    from scone_gcn_amd import ops
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd.complex import SimplicialComplex

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(dt):
        return dp.all_max(dt, device="cuda" if args.backend == "nccl" else "cpu")

    weak = args.per_gpu_batch > 0
    if not weak and args.global_batch % world:
        sys.stderr.write("bench.py: --global-batch must be a multiple of the rank count\n")
        sys.exit(2)
    B = args.per_gpu_batch if weak else args.global_batch // world         # this rank's trajectories per step
    total = B * world

    t_setup = time.perf_counter()
    cx = g.random_SC_graph(g.calibrate_n_points(args.edges))
    sc = SimplicialComplex(cx)
    # the GLOBAL batch is drawn ONCE with a fixed seed -- every rank draws the same `total` trajectories -- and
    # sharded by index (distributed.shard_indices, exactly as Scone_GCN.grad_step shards a masked batch): the batch, hence the
    # loss and the weights after K steps, do not depend on the rank count (up to the summation order of the all-reduce).
    flows_all, choice_all, last_all, y_all = dataset(cx, sc, total, 1030)
    mine = dp.shard_indices(np.arange(total), rank, world)
    assert len(mine) == B
    flows, choice, last, y = flows_all.select(mine), choice_all[mine], last_all[mine], y_all[mine]
    del flows_all, y_all
    net, inputs = make_net("scone", args.hidden, sc, flows, last, y, B)
    staged = net.stage(inputs, y, np.arange(B))
    plan = net._plan(inputs)
    E, C = cx.n_edges, args.hidden
    mb = staged[0][0].shape[0] * ops.NS                     # trajectories per launch (micro-batch)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    dt, table = time_config(net, inputs, staged, total, args.steps, args.warmup, sync, all_max)
    value = total * args.steps / dt
    roofline = roofline_of(table, mb)
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (same |E|, hidden and launch size only; dropped when the
    # kernel sources have changed since the passes were taken)
    if roofline and E == 996634 and mb == 128 and C == 32:
        for section in ("main_bench", "main"):          # on the benchmark's own trajectories (what launch_ms is measured on), else dense random
            roofline["traffic"], roofline["traffic_source"] = measured_traffic(section, list(roofline["variants"]))
            if roofline["traffic"]:
                break
    # ---- self-validation of the data-parallel step (every N): the loss of one more step on the same batch, summed over the
    # ranks, and the replicas' weights against rank 0's (identical Adam on identical reduced gradients => bitwise equal)
    part = net.grad_step_staged(inputs, staged, total, apply=False)        # this rank's share of the data term, weights untouched
    red_dev = "cuda" if (world == 1 or args.backend == "nccl") else "cpu"
    loss_t = part.detach().double().reshape(1).to(red_dev)
    w0 = net._flat_w.detach().clone().to(red_dev)
    if world > 1:
        dist.all_reduce(loss_t, op=dist.ReduceOp.SUM)
        dist.broadcast(w0, src=0)
    dev_w = (net._flat_w.detach().to(red_dev) - w0).abs().max().double().reshape(1)
    if world > 1:
        dist.all_reduce(dev_w, op=dist.ReduceOp.MAX)
    wf = net._flat_w.detach().double()
    validation = {"loss": float(loss_t.item()), "replicas_identical": bool(float(dev_w.item()) == 0.0),
                  "max_abs_weight_deviation_from_rank0": float(dev_w.item()),
                  "weights_sum": float(wf.sum().item()), "weights_l2": float(wf.norm().item()),
                  "optimiser_steps_taken": args.steps + args.warmup,
                  "note": "loss = batch cross-entropy (data term, STM:54) after the warm-up and timed steps, summed over ranks; the global "
                          "batch is seed-1030's for every N, so N = 1 and N > 1 lines agree to summation order"}

    alg_step = sum(r["alg_bytes"] * r["launches"] for r in table.values() if r["alg_bytes"])
    survey_bytes = 4.0 * E * (15 * C + 2)
    step_model = {"survey_model_bytes_per_trajectory": survey_bytes,
                  "survey_model_bytes_rate_not_achieved_bandwidth_frac_of_hbm_peak": value / world * survey_bytes / HBM_PEAK,
                  "kernel_algorithmic_bytes_per_trajectory": alg_step / B,
                  "kernel_model_frac_of_hbm_peak_whole_step": value / world * (alg_step / B) / HBM_PEAK,
                  "note": "survey model = SURVEY.md section 8d (4*E*(15*C+2): every activation tensor once per pass) -- the fused "
                          "kernels move FEWER bytes than that model, so its rate is a model-bytes rate, not achieved bandwidth; "
                          "kernel model = sum of the timed kernels' own algorithmic bytes (what `roofline` and `kernels` price)"}
    collective = None
    if world > 1:
        ver = ""
        if args.backend == "nccl":
            try:
                ver = " (RCCL %s)" % ".".join(map(str, torch.cuda.nccl.version()))
            except Exception as e:                          # version query only; the collective itself has already run
                ver = " (RCCL version query failed: %s)" % type(e).__name__
        collective = "%s%s, dist.get_world_size() = %d: one all-reduce (sum) of %d fp32 weight gradients per step" % (
            dist.get_backend(), ver, dist.get_world_size(), net._flat_g.numel())

    line = {
        "metric": "trajectories/sec fwd+bwd, 3-layer SCoNe |E|~1M batch=%d; SpMM HBM GB/s" % total, "value": value,
        "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if weak else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic complex |E|=%d (V=%d, F=%d), 3-layer SCoNe hidden=%d, global batch %d = %d "
                               "trajectories/GPU x %d GPU(s), micro-batch %d, fwd+bwd+allreduce+Adam"
                               % (E, cx.n_nodes, cx.n_faces, C, total, B, world, mb),
                   "edges": E, "nodes": cx.n_nodes, "faces": cx.n_faces, "hidden": C, "global_batch": total,
                   "per_gpu_batch": B, "micro_batch": mb, "parallelism": "dp%d" % world,
                   "collective": collective, "batch_seed": 1030,
                   "arithmetic": "fp32 tensors and accumulation; dense products on the f16 MFMA as hi+lo splits of both operands under "
                                 "power-of-two row scales (3 of 4 cross products, dot-product error <= ~3*2^-22*sum|ab|, DESIGN.md 3.2)",
                   "nnz_lower": plan.nnz_lower, "nnz_upper": plan.nnz_upper, "nnz_pattern": plan.nnz_pattern},
        "roofline": roofline, "cpu_baseline": None, "replicas_identical": validation["replicas_identical"],
        "loss": validation["loss"], "validation": validation, "step_model": step_model, "kernels": slim(table), "setup_s": t_setup,
    }

    # ---- weak-scaling companion: 512 trajectories per GPU (every N), + the zero-skipping modes on the same batch
    if args.extras and not weak:
        nb = min(512, B)
        sel = np.arange(nb)
        st512 = net.stage(inputs, y, sel)
        dtw, _ = time_config(net, inputs, st512, nb * world, max(2, min(args.steps, 5)), 1, sync, all_max)
        ksteps = max(2, min(args.steps, 5))
        line["weak_scaling"] = {"per_gpu_batch": nb, "global_batch": nb * world, "value": nb * world * ksteps / dtw,
                                "unit": "trajectories/s", "ms_per_step": dtw / ksteps * 1e3, "steps": ksteps}
        skipping = {}
        for mode in [m for m in args.skip_modes.split(",") if m]:
            st_m = net.stage(inputs, y, sel, skip=mode)
            if st_m[0][3] is None:
                continue
            net.grad_step_staged(inputs, st_m, nb * world)                  # allocates the pooled zero buffers
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                net.grad_step_staged(inputs, st_m, nb * world)
            sync()
            dtm = all_max(time.perf_counter() - t1)
            af = st_m[0][3]["active_fraction"]
            skipping[mode] = {"value": nb * world * args.steps / dtm, "unit": "trajectories/s", "per_gpu_batch": nb,
                              "ms_per_step": dtm / args.steps * 1e3,
                              "active_fraction_of_block_slab_items": {k: [round(v, 4) for v in af[k]] for k in af},
                              "note": "same step, same results; work items whose values are exactly zero"
                                      + (" or that the loss cannot see" if mode == "field" else "") + " are not computed"}
            import glob
            tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_skip_traffic.json")))   # rocprofv3 --pmc passes of tools/pmc_skip.sh
            if tfs and E == 996634 and C == 32:
                tf = tfs[-1]
                meas = json.load(open(tf))
                skipping[mode]["hbm_bytes_per_trajectory"] = {
                    "dense_model": survey_bytes, "measured_dense": meas["dense"]["hbm_bytes_per_trajectory"],
                    "measured_this_mode": meas[mode]["hbm_bytes_per_trajectory"],
                    "source": "%s (2*FETCH_SIZE+WRITE_SIZE, steady state; measured in the round the file names)" % os.path.relpath(tf, ROOT)}
            del st_m
        if skipping:
            line["zero_skipping"] = skipping
        del st512
    del staged
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and args.extras:
        # measured device-copy ceiling: 4 GiB read + 4 GiB write
        src = torch.empty(1 << 30, device="cuda", dtype=torch.float32)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        if roofline:
            roofline["measured_copy_GBps"] = copy_gbps
        del src, dst
        # the SpMM metric and the two fused C=32 kernels on DENSE RANDOM tensors (no zeros to favour clocks or caches)
        S = 32
        xr = torch.randn((S, E, 4, C), device="cuda", dtype=torch.float32)
        plan.conv.spmm_dual(xr.view(S, E, 4 * C))
        with ops.KernelTimer() as kt2:
            for _ in range(5):
                plan.conv.spmm_dual(xr.view(S, E, 4 * C))
        (_, r2), = kt2.table().items()
        sp = {"GB/s": r2["GB/s"], "frac_of_8TBps": r2["GB/s"] * 1e9 / HBM_PEAK, "ms": r2["avg_ms"],
              "algorithmic_bytes": r2["alg_bytes"], "x": "[%d, %d, %d] dense random fp32" % (S, E, 4 * C)}
        if E == 996634 and C == 32:
            sp["traffic"], sp["traffic_source"] = measured_traffic("main", ["spmm_dual k%d" % (4 * C)])
        line["spmm_dual"] = sp
        if roofline:                                      # the SpMM half of BASELINE's metric, inside `roofline` (flat scalars)
            roofline["spmm_dual_frac"] = sp["frac_of_8TBps"]
            roofline["spmm_dual_GBps"] = sp["GB/s"]
            roofline["spmm_dual_ms"] = sp["ms"]
            roofline["spmm_dual_algorithmic_bytes"] = sp["algorithmic_bytes"]
            roofline["spmm_dual_traffic"] = sp.get("traffic")
            roofline["spmm_dual_x"] = sp["x"]
        if C == 32:
            Wr = [torch.randn(C, C, device="cuda") * 0.1 for _ in range(3)]
            aux = torch.tanh(torch.randn((S, E, 4, C), device="cuda"))
            dWs = [torch.zeros_like(w) for w in Wr]
            plan.conv.forward([xr], Wr, C, "tanh")
            plan.conv.backward([xr], Wr, aux, "tanh", True, dWs)
            with ops.KernelTimer() as kt3:
                for _ in range(3):
                    plan.conv.forward([xr], Wr, C, "tanh")
                    plan.conv.backward([xr], Wr, aux, "tanh", True, dWs)
            dense_random = {k: {"ms": r["avg_ms"], "GB/s": r["GB/s"], "frac": r["GB/s"] * 1e9 / HBM_PEAK}
                            for k, r in kt3.table().items()}
            if roofline:
                roofline["dense_random"] = dense_random
                for k, r in dense_random.items():         # flat copies
                    tag = "fwd" if k.startswith("conv_fwd") else "bwd"
                    roofline["dense_random_%s_frac" % tag] = r["frac"]
                    roofline["dense_random_%s_ms" % tag] = r["ms"]
            del Wr, aux, dWs
        del xr
        torch.cuda.empty_cache()

        if args.parity_sample > 0:
            line["parity"] = oracle_parity(cx, sc, net, inputs, flows, choice, last, C, args.parity_sample)
        if args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline_sparse(cx, sc, flows, choice, last, C, args.cpu_sample)
            line["cpu_baseline"]["dense_faithful_configs0"] = cpu_baseline_dense_cfg1()
        del net
        torch.cuda.empty_cache()

        configs = {}
        # configs[4]: Bunch (SCCONV), same complex, hidden 32, batch 1024; Ebli (SNN) beside it
        configs["configs[4] bunch"] = side_config("BASELINE configs[4]: -model bunch, |E|~1M, hidden 32, batch 1024", "bunch",
                                                  cx, sc, 32, 1024, 2, sync, traffic_section="bunch")
        configs["ebli"] = side_config("-model ebli (SNN), |E|~1M, hidden 32, batch 128", "ebli", cx, sc, 32, 128, 3, sync,
                                      traffic_section="ebli")
        configs["ebli hidden 16"] = side_config("-model ebli (SNN), |E|~1M, hidden 16, batch 128 (ring SpMM + the power kernels on slab pairs)",
                                                "ebli", cx, sc, 16, 128, 3, sync)
        # the reference's documented mixed-width stack (TE:51) on the same complex: runs on the hidden-32 kernels (promotion)
        configs["mixed widths"] = side_config("-hidden_layers [(3,32),(3,16)] (TE:51), |E|~1M, batch 128", "scone", cx, sc, 32, 128, 3,
                                              sync, layers=[(3, 32), (3, 16)])
        configs["uniform 2x32"] = side_config("-hidden_layers [(3,32),(3,32)], |E|~1M, batch 128 (reference point for the mixed stack)",
                                              "scone", cx, sc, 32, 128, 3, sync, layers=[(3, 32), (3, 32)])
        # a hidden width above 32 (-hidden_layers takes any, TE:103-110): 32-channel blocks on the same fused kernels (ops.SconePlan._wide_stack)
        configs["hidden 64"] = side_config("-hidden_layers [(3,64)]*3, |E|~1M, batch 128 (32-channel blocks on the hidden-32 kernels)", "scone",
                                           cx, sc, 64, 128, 2, sync)
        del plan, inputs
        # configs[1]: |E| ~ 50k, hidden 16, batch 1024
        cx2 = g.random_SC_graph(g.calibrate_n_points(50_000))
        sc2 = SimplicialComplex(cx2)
        configs["configs[1]"] = side_config("BASELINE configs[1]: synthetic |E|~50k, hidden 16, batch 1024", "scone", cx2, sc2,
                                            16, 1024, 10, sync, traffic_section="configs[1]")
        # configs[0]: the reference's own case (TE:86-90): 400-point complex (|E| = 1001), hidden 16, batch 100, resident on the device
        cx0 = g.random_SC_graph(400)
        configs["configs[0]"] = side_config("BASELINE configs[0]: synthetic_data_gen.py 400-point complex, 3-layer SCoNe hidden 16, batch 100 "
                                            "(the reference's own problem size; its dense-faithful CPU restatement is cpu_baseline.dense_faithful_configs0)",
                                            "scone", cx0, SimplicialComplex(cx0), 16, 100, 1000, sync)
        # the same case on the layer-by-layer kernels (the default is the one-launch step with two workgroups per trajectory; a step is
        # ~80 us, so a thousand of them: the queue's start-up and the final synchronisation are ~100 us)
        keep = ops.SMALL_STEP
        ops.SMALL_STEP = False
        try:
            layers_ = side_config("configs[0] on the layer-by-layer kernels (ops.SMALL_STEP = False)", "scone", cx0, SimplicialComplex(cx0),
                                  16, 100, 1000, sync)
            configs["configs[0]"]["layer_kernels_step"] = {"ms_per_step": layers_["ms_per_step"], "value": layers_["value"],
                                                           "unit": layers_["unit"], "kernels": layers_["kernels"]}
        finally:
            ops.SMALL_STEP = keep
        # configs[2]: ocean drifters, full training batch (the trajectories of tests/golden/buoy.npz)
        bpath = os.path.join(ROOT, "tests", "golden", "buoy.npz")
        if os.path.exists(bpath):
            from scone_gcn_amd import buoy_data as bd
            gld = np.load(bpath)
            trajs = [gld["traj_nodes"][gld["traj_ptr"][i]:gld["traj_ptr"][i + 1]].astype(int).tolist()
                     for i in range(len(gld["traj_ptr"]) - 1)]
            cx3, _, fl3, ch3, last3, _, train_mask, _ = bd.buoy_dataset(gld["elist"].astype(np.int64), gld["tlist"].astype(np.int64),
                                                                       gld["coords"], trajs)
            sc3 = SimplicialComplex(cx3)
            tr = np.nonzero(train_mask)[0]
            y3 = np.zeros((len(tr), sc3.max_degree, 1))
            y3[np.arange(len(tr)), ch3[tr], 0] = 1.0
            configs["configs[2]"] = side_config("BASELINE configs[2]: ocean drifters, 3-layer SCoNe hidden 16, full training batch",
                                                "scone", cx3, sc3, 16, len(tr), 1000, sync,
                                                data=(fl3.select(tr), ch3[tr], last3[tr], y3))
        line["configs"] = configs

    if rank == 0:
        emit(line, args.out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
