#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec, forward + backward (+ gradient all-reduce + Adam), 3-layer SCoNe.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[3]): synthetic complex with |E| ~ 1M (370 000 points, the reference generator's
recipe), hidden = 32, fp32, 512 trajectories per GPU per step (global batch 4096 at 8 GPUs -> weak scaling).
A step = one optimiser step on one batch that is already resident in HBM: forward, loss, backward, RCCL all-reduce
of the flat weight-gradient buffer, fused ridge + Adam.  Prints ONE JSON line on rank 0.

Extra objects in the line:
  roofline     -- the kernel family that takes the most time in the step, timed live with events on the launch
                  stream: algorithmic bytes per launch / mean launch duration vs the 8 TB/s HBM peak.
  cpu_baseline -- the CPU oracle (oracle/scone_oracle.py with scipy CSR shifts: a "port" of the reference
                  formulation) timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--edges", type=int, default=1_000_000, help="target |E| of the synthetic complex")
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--per-gpu-batch", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=4, help="trajectories in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--spmm", type=int, default=1, help="also time the standalone dual SpMM (reported as extra)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal: ranks may share one GPU)")
    ap.add_argument("--skip-modes", default="zeros,field",
                    help="also time the same step in these zero-skipping modes (reported beside the dense value; '' = none)")
    return ap.parse_args()


def cpu_baseline(cx, sc, flows, choice, last, hidden, n_sample):
    """fwd + bwd of the reference formulation on the host: fp64 NumPy oracle with scipy CSR shifts."""
    import scipy.sparse as sp
    from oracle import scone_oracle as so
    from scone_gcn_amd import synthetic_data_gen as g
    sel = np.arange(n_sample)
    B1, B2 = g.incidence_matrices(cx)
    L_lo, L_up = (B1.T @ B1).tocsr(), (B2 @ B2.T).tocsr()
    B1x = sp.vstack([B1, sp.csr_matrix((1, B1.shape[1]))]).tocsr()
    Bc = lambda n: B1x[sc.nbrhoods[n]].toarray()
    X = flows.select(sel).todense().astype(np.float64)
    y = so.onehot_targets(choice[sel], sc.max_degree)
    w = so.generate_weights(1, [(3, hidden)] * 3, 1)
    t0 = time.perf_counter()
    so.scone_loss_and_grad(w, L_lo, L_up, Bc, last[sel], X, y, np.ones(n_sample, int), 5e-5)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    return {"value": n_sample / dt, "unit": "trajectories/s", "cores": int(threads), "kind": "port",
            "sample": "%d trajectories of the same |E|=%d complex, hidden %d, fp64 NumPy + scipy CSR oracle, "
                      "fwd+bwd (no optimiser), %.1f s" % (n_sample, cx.n_edges, hidden, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    from scone_gcn_amd import ops
    from scone_gcn_amd import scone_trajectory_model as stm
    from scone_gcn_amd import synthetic_data_gen as g
    from scone_gcn_amd import trajectory_experiments as te
    from scone_gcn_amd.complex import SimplicialComplex

    t_setup = time.perf_counter()
    n_points = g.calibrate_n_points(args.edges)
    cx = g.random_SC_graph(n_points)
    sc = SimplicialComplex(cx)
    B = args.per_gpu_batch
    paths = g.generate_random_walks(cx, m=B, seed=1030 + rank, waypoint_pool=8, metric="euclid")
    flows, choice, last, _, _ = g.path_dataset(cx, paths, seed=7 + rank)
    D = sc.max_degree
    y = np.zeros((B, D, 1))
    y[np.arange(B), choice, 0] = 1.0
    shifts, readout, _ = te.setup_from_complex(sc, "scone")
    inputs = [readout, last, flows]
    stm.reseed(1030)
    net = stm.Scone_GCN(1, 1e-3, B, 5e-5, verbose=False)
    net.setup(te.scone_func, [(3, args.hidden)] * 3, shifts, inputs, y, None, np.ones(B, int), model_type="scone")
    staged = net.stage(inputs, y, np.arange(B))
    total = B * world
    plan = net._plan(inputs)
    E, C = cx.n_edges, args.hidden
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        net.grad_step_staged(inputs, staged, total)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.grad_step_staged(inputs, staged, total)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # --- live per-kernel timing of the same step (separate pass so the events do not sit in the timed region)
    with ops.KernelTimer() as kt:
        net.grad_step_staged(inputs, staged, total)
    ksum = kt.summary()
    mb = staged[0][0].shape[0] * ops.NS                     # trajectories per launch (micro-batch)
    nnz_pat, nnz_lo, nnz_up = plan.nnz_pattern, plan.nnz_lower, plan.nnz_upper
    csr_bytes = 4 * (nnz_lo + nnz_up) + 4 * nnz_pat + 4 * (E + 1)

    def alg_bytes(key):
        # activation tensors read once + written once per launch, CSR once
        if key.startswith("conv_fwd"):
            cin = int(key.split("c")[2].split("->")[0])
            cout = int(key.split("->")[1])
            return 4.0 * E * mb * (cin + cout) + csr_bytes
        if key.startswith("conv_bwd"):
            cdz = int(key.split("c")[2].split("->")[0])
            caux = int(key.split("->")[1].split()[0])
            wr = 0 if "dW only" in key else caux
            return 4.0 * E * mb * (cdz + caux + wr) + csr_bytes
        if key.startswith("conv_dw_first"):                      # same tensors as the dW-only backward: dz and x, once
            return 4.0 * E * mb * (int(key.split("c")[2]) + 1) + csr_bytes
        return None
    tot = {k: n * ms for k, (n, ms) in ksum.items()}
    dom = max((k for k in tot if alg_bytes(k) is not None), key=lambda k: tot[k])
    n_launch, ms = ksum[dom]
    achieved = alg_bytes(dom) / (ms * 1e-3)
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/; same |E|, hidden and launch size only)
    traffic = None
    kmap = {"conv_bwd c32->32": "scn::bwd_c32_bf16_kernel", "conv_fwd c32->32": "scn::fwd_c32_w16_kernel",
            "conv_fwd c1->32": "scn::fwd_c1_kernel", "conv_bwd c32->1 (dW only)": "scn::bwd_c1_kernel"}
    tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tfile) and dom in kmap and E == 996634 and mb == 128:
        traffic = json.load(open(tfile)).get(kmap[dom], {}).get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic, "launch_ms": ms, "launches_per_step": n_launch,
                "algorithmic_bytes_per_launch": alg_bytes(dom), "units_per_launch": mb,
                "traffic_source": "profiles/r01_pmc_traffic.json (2*FETCH_SIZE+WRITE_SIZE per launch)" if traffic else None}
    kernels = {k: {"launches": n, "avg_ms": ms_, "GB/s": (alg_bytes(k) / (ms_ * 1e-3) / 1e9) if alg_bytes(k) else None}
               for k, (n, ms_) in ksum.items()}

    extra = {}
    if rank == 0:                                    # measured device-copy ceiling: 4 GiB read + 4 GiB write
        src = torch.empty(1 << 30, device="cuda", dtype=torch.float32)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        roofline["measured_copy_GBps"] = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    if args.spmm and rank == 0:
        K = 128
        S = 32
        xr = torch.randn((S, E, K), device="cuda", dtype=torch.float32)      # dense random: no zero skipping
        plan.conv.spmm_dual(xr)
        with ops.KernelTimer() as kt2:
            for _ in range(5):
                plan.conv.spmm_dual(xr)
        (nk, ms2), = kt2.summary().values()
        b = 12.0 * E * K * S + csr_bytes
        extra["spmm_dual"] = {"GB/s": b / (ms2 * 1e-3) / 1e9, "frac_of_8TBps": b / (ms2 * 1e-3) / HBM_PEAK,
                              "ms": ms2, "x": "[%d, %d, %d] dense random fp32" % (S, E, K)}
        del xr

    # --- the same step with exact zero-skipping (work lists): identical results, reported BESIDE the dense headline value
    skipping = {}
    for mode in [m for m in args.skip_modes.split(",") if m]:
        st_m = net.stage(inputs, y, np.arange(B), skip=mode)
        if st_m[0][3] is None:
            continue
        net.grad_step_staged(inputs, st_m, total)                           # warm-up (allocates the pooled zero buffers)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            net.grad_step_staged(inputs, st_m, total)
        sync()
        dtm = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dtm], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t.item())
        af = st_m[0][3]["active_fraction"]
        skipping[mode] = {"value": total * args.steps / dtm, "unit": "trajectories/s", "ms_per_step": dtm / args.steps * 1e3,
                          "active_fraction_of_block_slab_items": {k: [round(v, 4) for v in af[k]] for k in af},
                          "note": "same step, same results; work items whose values are exactly zero"
                                  + (" or that the loss cannot see" if mode == "field" else "") + " are not computed"}
        tf = os.path.join(ROOT, "profiles", "r01_skip_traffic.json")        # rocprofv3 --pmc passes of tools/pmc_skip.sh
        if os.path.exists(tf) and E == 996634 and C == 32:
            meas = json.load(open(tf))
            skipping[mode]["hbm_bytes_per_trajectory"] = {
                "dense_model": 4.0 * E * (15 * C + 2), "measured_dense": meas["dense"]["hbm_bytes_per_trajectory"],
                "measured_this_mode": meas[mode]["hbm_bytes_per_trajectory"],
                "source": "profiles/r01_skip_traffic.json (2*FETCH_SIZE+WRITE_SIZE, steady state)"}
        del st_m

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        cpu = cpu_baseline(cx, sc, flows, choice, last, args.hidden, args.cpu_sample)

    if rank == 0:
        value = total * args.steps / dt
        bytes_per_traj = 4.0 * E * (15 * C + 2)
        line = {
            "metric": "trajectories/sec fwd+bwd, 3-layer SCoNe |E|~1M batch=4096", "value": value,
            "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic complex |E|=%d (V=%d, F=%d), 3-layer SCoNe hidden=%d, %d trajectories/GPU "
                                   "(global batch %d), fwd+bwd+allreduce+Adam" % (E, cx.n_nodes, cx.n_faces, C, B, total),
                       "edges": E, "nodes": cx.n_nodes, "faces": cx.n_faces, "hidden": C, "global_batch": total,
                       "micro_batch": mb, "parallelism": "dp%d" % world,
                       "nnz_lower": nnz_lo, "nnz_upper": nnz_up, "nnz_pattern": nnz_pat},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "step_model": {"algorithmic_bytes_per_trajectory": bytes_per_traj,
                           "frac_of_hbm_peak_whole_step": value / world * bytes_per_traj / HBM_PEAK},
            "kernels": kernels, "setup_s": t_setup,
        }
        line.update(extra)
        if skipping:
            line["zero_skipping"] = skipping
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
