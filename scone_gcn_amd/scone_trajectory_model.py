"""Host-side mirror of the reference trainer class (trajectory_analysis/scone_trajectory_model.py = STM).

Scone_GCN keeps the reference's constructor, attributes and methods
    Scone_GCN(epochs, step_size, batch_size, weight_decay, verbose=True)           STM:18
    .setup(model, hidden_layers, shifts, inputs, y, in_axes, train_mask, model_type='scone')   STM:245
    .loss(weights, inputs, y, mask)   .accuracy(shifts, inputs, y, mask, n_nbrs)    STM:42, 59
    .train(inputs, y, train_mask, test_mask, n_nbrs) -> (train_loss, train_acc, test_loss, test_acc)   STM:264
    .test(inputs, y, mask, n_nbrs) -> (loss, acc)    .two_target_accuracy(...)      STM:359, 73
    .generate_weights(in_channels, hidden_layers, out_channels)    .weights         STM:215
but runs the hot path on the GPU: forward/backward through libscone_hip.so on only the trajectories a mask
selects (the reference runs all N and masks afterwards, STM:46 -- same value), the whole weight list in one flat
fp32 buffer so that the gradient all-reduce and the fused Adam/ridge kernel see a single array, and the batch
sharded over ranks when torch.distributed is initialised (scone_gcn_amd/distributed.py).
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops
from . import distributed as dp
from .synthetic_data_gen import SparseFlows
from .trajectory_experiments import MODEL_ACT, MODEL_FUNCS, resolve_operands

# module-level legacy RNG seeded like the reference's `onp.random.seed(1030)` at import (STM:15): weights
# (STM:237), batch-mask shuffles (STM:320) and random targets (STM:79, 91) are drawn from it in call order.
_RNG = np.random.RandomState(1030)


def reseed(seed=1030):
    global _RNG
    _RNG = np.random.RandomState(seed)


def _select(inputs, idx):
    """Rows idx of the per-trajectory inputs [readout operand, last_nodes, X]."""
    X = inputs[-1]
    if isinstance(X, SparseFlows):
        Xs = X.select(idx)
    elif torch.is_tensor(X):
        Xs = X[torch.as_tensor(idx, device=X.device)]
    else:
        Xs = np.asarray(X)[idx]
    return [inputs[0], np.asarray(inputs[1])[idx], Xs]


def _n_samples(X):
    return len(X) if isinstance(X, SparseFlows) else X.shape[0]


class _StaticStage:
    """Device buffers with FIXED addresses for one host batch (what a captured graph replays on): the flow entries of up to
    `cap_traj` trajectories as (trajectory, edge, value) triples, their last nodes and their targets -- filled through ONE pinned
    host buffer and ONE host-to-device copy per step.  Unused entries carry value 0 (the scatter adds nothing), unused trajectories
    have an all-zero flow and an all-zero target row (no loss, no gradient).  The targets are pre-scaled by 1 / (global batch
    count), so the captured cross-entropy kernel's scale argument is the constant -1."""

    def __init__(self, plan, X, D, cap_traj, device):
        self.plan, self.X, self.D = plan, X, D
        self.S = ops.pad_count(cap_traj) // ops.NS
        self.n_cap = self.S * ops.NS
        lens = np.sort(np.diff(X.ptr))[::-1]
        self.e_cap = max(int(lens[:self.n_cap].sum()), 1)                    # entries of the longest n_cap trajectories
        self.perm = plan.layout.perm[1].astype(np.int32)
        E = plan.n_edges
        n_words = 3 * self.e_cap + self.n_cap + self.n_cap * D
        self.host = [torch.zeros(n_words, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.done = [None, None]
        self.turn = 0
        self.dev = torch.zeros(n_words, dtype=torch.int32, device=device)
        e, n = self.e_cap, self.n_cap
        self.sample_d, self.edge_d = self.dev[0:e], self.dev[e:2 * e]
        self.val_d = self.dev[2 * e:3 * e].view(torch.float32)
        self.last_d = self.dev[3 * e:3 * e + n]
        self.y_d = self.dev[3 * e + n:].view(torch.float32).view(n, D)
        self.x = torch.zeros((self.S, E, ops.NS, 1), device=device, dtype=torch.float32)
        self.staged = (self.x, self.last_d, self.y_d, None)
        # the data set as the native assembler takes it (scn_host_stage_batch): entries with their device row index, per trajectory
        self._ptr = np.ascontiguousarray(X.ptr, np.int64)
        self._edge = np.ascontiguousarray(self.perm[X.idx], np.int32)
        self._val = np.ascontiguousarray(X.val, np.float32)
        self._last_src = self._last = self._y_src = self._y = None

    def load(self, idx, last_nodes, y, total):
        X, e, n = self.X, self.e_cap, self.n_cap
        idx = np.asarray(idx)
        m = len(idx)
        if m and (int(idx.min()) < 0 or int(idx.max()) >= len(X.ptr) - 1):      # (the native assembler refuses them too)
            raise IndexError("trajectory index outside the data set of %d trajectories" % (len(X.ptr) - 1))
        starts = X.ptr[idx]
        lens = X.ptr[idx + 1] - starts
        k = int(lens.sum())
        if m > n or k > e:
            return False
        t = self.turn
        self.turn ^= 1
        if self.done[t] is not None:
            self.done[t].synchronize()                                   # the copy that last read this host buffer (two steps ago)
        if self._last_src is not last_nodes:                             # the data set's last nodes / targets, converted once per object
            self._last_src, self._last = last_nodes, np.ascontiguousarray(np.asarray(last_nodes), np.int32)
        if self._y_src is not y:
            self._y_src, self._y = y, np.ascontiguousarray(np.asarray(y, np.float32).reshape(len(y), -1))
        traj = np.ascontiguousarray(idx, np.int32)
        got = _lib.load().scn_host_stage_batch(m, traj.ctypes.data, len(self._ptr) - 1, self._ptr.ctypes.data, self._edge.ctypes.data,
                                               self._val.ctypes.data, self._last.ctypes.data, self._y.ctypes.data, self.D,
                                               float(total), e, n, self.host[t].data_ptr())
        if got != k:
            raise RuntimeError("scn_host_stage_batch: %d (expected %d flow entries)" % (got, k))
        self.dev.copy_(self.host[t], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[t] = ev
        return True

    def scatter(self):
        """x = the staged flows as slabs (captured into the graph: scn_scatter_flows zeroes x, then adds every entry)."""
        _lib.check(_lib.load().scn_scatter_flows(self.S, ops.NS, self.plan.n_edges, self.e_cap, ops._dev(self.sample_d, torch.int32),
                                                 ops._dev(self.edge_d, torch.int32), ops._dev(self.val_d), ops._dev(self.x),
                                                 ops._stream()), "scn_scatter_flows")


class Scone_GCN():
    def __init__(self, epochs, step_size, batch_size, weight_decay, verbose=True, process_group=None, skip_mode="dense"):
        # skip_mode: "dense" (every block of every slab, the reference's formulation), "zeros" (skip work items whose
        # values are exactly zero) or "field" (also skip what the loss cannot see); same results, see SconePlan.activity
        assert skip_mode in ("dense", "zeros", "field")
        self.skip_mode = skip_mode
        self.random_targets = None
        self.trained = False
        self.model = None
        self.model_single = None
        self.shifts = None
        self.weights = None
        self.epochs = int(epochs)                      # STM:35-38
        self.step_size = step_size
        self.batch_size = int(batch_size)
        self.weight_decay = weight_decay
        self.verbose = verbose
        self.process_group = process_group
        self.collective_always = False                 # True: run the gradient all-reduce even in a one-rank process group
        self.model_type = 'scone'
        self._flat_w = self._flat_g = self._m = self._v = None
        self._step = 0
        # launch-amortised step for small complexes (hipGraph replay, see _graph_accumulate): on by default, only ever used when
        # the whole batch is one micro-batch of at most GRAPH_MAX_ELEMS activation elements
        self.use_graph = True
        self._graphs = {}
        self._static = {}
        # evaluation cache (see _eval_all): the epoch-end loss / accuracy passes of train() (STM:328-337) ask for the SAME data set's
        # log-probabilities four times under the same weights
        self.use_eval_cache = True
        self._eval = None
        self._wver = 0                                 # bumped by the raw-pointer Adam launch (torch's own version counter sees the rest)

    # ------------------------------------------------------------------ weights
    def generate_weights(self, in_channels, hidden_layers, out_channels):
        """Same shapes, order and 0.01 * randn values as STM:215-242 (fp32 on the device)."""
        weight_shapes = []
        if len(hidden_layers) > 0:
            weight_shapes += [(in_channels, hidden_layers[0][1])] * hidden_layers[0][0]
            for i in range(len(hidden_layers) - 1):
                for _ in range(hidden_layers[i + 1][0]):
                    weight_shapes += [(hidden_layers[i][1], hidden_layers[i + 1][1])]
            if self.model_type == 'bunch':
                weight_shapes += [(hidden_layers[-1][1], out_channels)] * hidden_layers[-1][0]
            else:
                weight_shapes += [(hidden_layers[-1][1], out_channels)]
            host = [0.01 * _RNG.randn(*s) for s in weight_shapes]
        else:
            raise ValueError("at least one hidden layer is required")
        self._install(host)
        if self.verbose:
            print('# of parameters: {}'.format(int(np.sum([np.prod(s) for s in weight_shapes]))))

    def _install(self, host_weights):
        device = ops.default_device()
        sizes = [int(np.prod(w.shape)) for w in host_weights]
        flat = np.concatenate([np.asarray(w, np.float64).ravel() for w in host_weights]).astype(np.float32)
        self._flat_w = torch.from_numpy(flat).to(device)
        self._flat_g = torch.zeros_like(self._flat_w)
        self._m = torch.zeros_like(self._flat_w)
        self._v = torch.zeros_like(self._flat_w)
        self._step_dev = torch.zeros(2, device=device, dtype=torch.int32)      # [0] the step index where scn_adam_step_dev reads it ([1]: scratch)
        self._step_dev_value = 0                                              # ... and what it holds (host mirror)
        self._offsets = np.concatenate([[0], np.cumsum(sizes)])
        self._shapes = [tuple(w.shape) for w in host_weights]
        self.weights = self._views(self._flat_w)
        self._grads = self._views(self._flat_g)
        self._drop_graphs()                             # captured steps point at the previous buffers
        self._eval = None                               # ... and cached log-probabilities belong to the previous weights

    def _drop_graphs(self, keep=0):
        """Forget captured steps (oldest first) down to `keep`; a graph may still be replaying, so the device is drained first."""
        if len(getattr(self, "_graphs", {})) > keep:
            torch.cuda.synchronize()
            while len(self._graphs) > keep:
                self._graphs.pop(next(iter(self._graphs)))

    def _views(self, flat):
        return [flat[self._offsets[k]:self._offsets[k + 1]].view(self._shapes[k]) for k in range(len(self._shapes))]

    def save_weights(self, path):
        """Pickle-free replacement of onp.save('models/<name>', weights) (TE:482-486)."""
        np.savez(path, **{"w%d" % k: w.detach().cpu().numpy() for k, w in enumerate(self.weights)})

    def load_weights(self, path):
        d = np.load(path)
        self._install([d["w%d" % k] for k in range(len(d.files))])

    # ------------------------------------------------------------------ setup
    def setup(self, model, hidden_layers, shifts, inputs, y, in_axes, train_mask, model_type='scone'):
        """Set up model for training / calling (STM:245-262).  `in_axes` is accepted for signature parity; the
        model functions are batched natively."""
        self.model_type = model_type
        self.shifts = shifts
        self.model = model
        self.model_single = model
        X = inputs[-1]
        in_channels = 1 if isinstance(X, SparseFlows) else int(X.shape[-1])
        out_channels = int(np.asarray(y).shape[-1])
        self.generate_weights(in_channels, hidden_layers, out_channels)

    def _plan(self, inputs):
        if self.model is MODEL_FUNCS.get(self.model_type):
            dev = ops.default_device()
            shifts, readout = resolve_operands(self.model_type, self.shifts, inputs[0])   # dense / closure operands: wrapped once
            if self.model_type == 'bunch':
                return ops.get_bunch_plan(shifts, readout, dev)
            return ops.get_scone_plan(shifts[0], shifts[1], readout, MODEL_ACT[self.model_type], dev)
        return None

    # ------------------------------------------------------------------ loss / metrics
    def _ridge(self, weights):
        if weights is self.weights and self._flat_w is not None:          # one reduction and one host synchronisation, not one per matrix
            return self.weight_decay * float((self._flat_w.double() ** 2).sum())
        return self.weight_decay * sum(float((w.double() ** 2).sum()) for w in weights)   # STM:54-56

    # On the reference's own sizes an epoch of train() is 8 optimiser steps (~0.2 ms each) and four evaluation passes -- train loss, train
    # accuracy, test loss, test accuracy (STM:328-337) -- each of which staged its trajectories from the host and ran its own forward:
    # 3 of the epoch's 5.3 ms (tools/cfg1_step_time.py).  All four read log-probabilities of one data set under one set of weights: the
    # whole set is staged ONCE (kept on the device for as long as the same input objects come back) and forwarded ONCE per weight
    # version; the four metrics index the result.  Same launches as before per trajectory, so the same values.
    EVAL_CACHE_ELEMS = 1 << 26          # rows x trajectories x widest layer up to which the whole data set is evaluated at once

    def _weights_version(self):
        return (int(self._flat_w._version), self._wver)

    def invalidate_eval_cache(self):
        """Forget the cached evaluation and the staged copies of y / last_nodes.  The caches key on OBJECT IDENTITY and on the two weight
        version counters: data arrays handed to train() / test() / grad_step() are treated as immutable while the net holds them, and a
        weight write that neither torch's version counter nor _adam sees (`.data` assignment, a raw-pointer kernel of the caller's) must
        be followed by this call -- as must an in-place edit of y / last_nodes / the flows."""
        self._eval = None
        self._wver += 1
        for st in self._static.values():
            st._last_src = st._last = st._y_src = st._y = None

    def _eval_all(self, plan, inputs):
        """log-probabilities (N, D, 1) of ALL trajectories of `inputs` under self.weights (device tensor, do not modify)."""
        N = _n_samples(inputs[-1])
        ver = self._weights_version()
        c = self._eval
        same = (c is not None and c["flows"] is inputs[-1] and c["last"] is inputs[1] and c["readout"] is inputs[0]
                and c["plan"] is plan and c["n"] == N)
        if same and c["ver"] == ver:
            return c["logp"]
        staged = c["staged"] if same else self.stage(inputs, np.zeros((N, plan.max_deg, 1), np.float32), np.arange(N), skip="dense")
        outs = []
        with torch.no_grad():
            for x, last_dev, _, _ in staged:
                logp, saved = plan.forward(x, last_dev, self.weights)
                outs.append(logp)
                del saved
        logp = (outs[0] if len(outs) == 1 else torch.cat(outs))[:N].unsqueeze(-1)
        self._eval = {"flows": inputs[-1], "last": inputs[1], "readout": inputs[0], "plan": plan, "n": N, "ver": ver,
                      "staged": staged, "logp": logp}
        return logp

    def _eval_cache_ok(self, plan, inputs):
        if not self.use_eval_cache or plan is None or self.skip_mode != "dense" or ops.KernelTimer._stack:
            return False
        widest = max(max(sh) for sh in self._shapes)
        rows = sum(plan.sizes) if type(plan) is ops.BunchPlan else plan.n_edges
        return rows * ops.pad_count(_n_samples(inputs[-1])) * (plan.promotion(self.weights) or widest) <= self.EVAL_CACHE_ELEMS

    def _predict(self, weights, inputs, idx=None):
        """log-probabilities (n, D, 1) for trajectories idx (all when None), no autograd."""
        if weights is self.weights and self.skip_mode == "dense":
            plan = self._plan(inputs)
            if self._eval_cache_ok(plan, inputs):
                logp = self._eval_all(plan, inputs)
                if idx is None:
                    return logp.clone()
                return logp[torch.as_tensor(np.asarray(idx), device=logp.device, dtype=torch.long)]
        plan = self._plan(inputs) if (self.skip_mode != "dense" and self.model_type != 'bunch' and weights is self.weights) else None
        if plan is not None:                            # zero-skipping forward (same log-probabilities, see SconePlan.activity)
            idx_all = np.arange(_n_samples(inputs[-1])) if idx is None else np.asarray(idx)
            staged = self.stage(inputs, np.zeros((_n_samples(inputs[-1]), plan.max_deg, 1), np.float32), idx_all)
            if all(st[3] is not None for st in staged):
                outs = []
                for x, last_dev, _, activity in staged:
                    logp, saved = plan.forward(x, last_dev, self.weights, activity)
                    outs.append(logp.clone())
                    plan.release(saved)
                return torch.cat(outs)[:len(idx_all)].unsqueeze(-1)
        sub = inputs if idx is None else _select(inputs, idx)
        with torch.no_grad():
            return self.model(weights, *self.shifts, *sub)

    def loss(self, weights, inputs, y, mask):
        """Cross-entropy per masked flow + ridge (STM:42-56).  Returns a Python float."""
        idx = np.nonzero(np.asarray(mask) == 1)[0]
        preds = self._predict(weights, inputs, idx)
        yt = torch.as_tensor(np.asarray(y)[idx], device=preds.device, dtype=torch.float32)
        return float(-(preds.double() * yt.double()).sum() / len(idx)) + self._ridge(weights)

    def accuracy(self, shifts, inputs, y, mask, n_nbrs):
        """Ratio of correct predictions among the true neighbours (STM:59-71)."""
        idx = np.nonzero(np.asarray(mask) == 1)[0]
        target_choice = np.argmax(np.asarray(y)[idx], axis=1)
        preds = self._predict(self.weights, inputs, idx).cpu().numpy().astype(np.float64)
        nn = np.asarray(n_nbrs)[idx]
        preds[np.arange(preds.shape[1])[None, :] >= nn[:, None]] = -100        # preds[i, n_nbrs[i]:] = -100 (STM:66-67)
        pred_choice = np.argmax(preds, axis=1)
        return float(np.mean(pred_choice == target_choice))

    def two_target_accuracy(self, shifts, inputs, y, mask, n_nbrs):
        """STM:73-108, including its quirk of re-drawing against the PREDICTED choice."""
        N = _n_samples(inputs[-1])
        n_nbrs = np.asarray(n_nbrs)
        if type(self.random_targets) != np.ndarray:
            self.random_targets = _RNG.randint(0, high=n_nbrs, size=N)
        preds = self._predict(self.weights, inputs).cpu().numpy().astype(np.float64)
        for i in range(len(preds)):
            preds[i, n_nbrs[i]:] = -100
        m = np.asarray(mask) == 1
        pred_choice = np.argmax(preds[m], axis=1).reshape(-1)
        for i in range(N):
            # pred_choice is a jax array in the reference: an index past its end is clamped to the last element (STM:90)
            pc = pred_choice[min(i, len(pred_choice) - 1)]
            while n_nbrs[i] > 1 and self.random_targets[i] == pc:     # (n_nbrs == 1 would never terminate: keep the draw)
                self.random_targets[i] = _RNG.randint(0, high=n_nbrs[i])
        rows = np.arange(N)
        random_probs = preds[rows, self.random_targets, 0]
        true_choice = np.argmax(np.asarray(y), axis=1).reshape((N,))
        true_probs = preds[rows, true_choice, 0]
        t, r = true_probs[m], random_probs[m]
        return float((np.sum(t > r) + 0.5 * np.sum(t == r)) / np.sum(m))

    # ------------------------------------------------------------------ gradient step (STM:306-310)
    def stage(self, inputs, y, idx, skip=None):
        """Move trajectories idx to the device as micro-batches of flow slabs: list of (x, last_nodes, y, activity) with
        x [S, E, ns, 1] fp32, last_nodes [S*ns] int32, y [S*ns, D] fp32 (zero rows for padding) and the work lists of the
        zero-skipping mode (None = dense; skip in {"dense", "zeros", "field"}, default self.skip_mode)."""
        skip = self.skip_mode if skip is None else skip
        plan = self._plan(inputs)
        device = self._flat_w.device
        k = 7 if self.model_type == 'bunch' else 3
        widths = [1] + [self._shapes[k * i][1] for i in range(len(self._shapes) // k)]
        P = plan.promotion(self.weights)                # hidden widths the kernels do not take are zero-padded (ops.promote_weights)
        if P:
            widths = [w if w == 1 else P for w in widths]
        rows = sum(plan.sizes) if self.model_type == 'bunch' else plan.n_edges
        budget = None
        if self.model_type == 'bunch' and ops.FOLD_BUNCH and len(widths) > 3 and widths[1] == widths[2] == 32:
            # the first hidden layer is never materialised (rank-one fold, DESIGN.md section 3.1 / profiles/HISTORY.md section 3.1) and levels the loss cannot see are not
            # computed: measured 0.93 GB per trajectory at |E| = 1M where the generic estimate says 1.3 -- 128 trajectories per launch
            # (119 GB at peak) instead of 64, +3.4 % on configs[4] (tools/mb_sweep.py)
            widths = [widths[0]] + widths[2:]
            budget = 0.5 * torch.cuda.mem_get_info(device)[1]
        mb = ops.micro_batch_size(rows, widths, len(idx), budget_bytes=budget, device=device)
        staged = []
        for c0 in range(0, len(idx), mb):
            sel = idx[c0:c0 + mb]
            sub = _select(inputs, sel)
            x, n = ops.flows_to_slabs(sub[2], plan.layout, device)
            last_dev = ops._last_nodes_dev(ops.remap_last_nodes(plan, sub[1]), x.shape[0] * ops.NS, device)
            D = np.asarray(y).shape[1]
            yt = torch.zeros((x.shape[0] * ops.NS, D), device=device, dtype=torch.float32)
            yt[:n] = torch.as_tensor(np.asarray(y)[sel], dtype=torch.float32).reshape(n, -1).to(device)
            activity = None
            if skip not in (None, "dense") and self.model_type != 'bunch':
                activity = plan.activity(inputs[2], inputs[1], len(self._shapes) // 3, self._shapes[0][1], skip, sel=sel)
            staged.append((x, last_dev, yt, activity))
        return staged

    def _accumulate_staged(self, plan, staged, total, fresh=False, adam=False):
        """flat_g += d/dW of  -sum_n <logp_n, y_n> / total over the staged micro-batches (fresh: flat_g = ..., i.e. zeroed first).
        Returns that partial loss as a 0-dim device tensor (no host synchronisation inside the step).
        adam (callers: _adam_in_graph): the optimiser step follows at once -- when the batch is one micro-batch on the one-launch
        step, its summing launch applies it (self._adam_applied says whether it did)."""
        lib = _lib.load()
        self._adam_applied = False
        # The loss accumulator and (fresh) the gradient buffer are initialised BY the first launch that writes them -- the one-launch
        # step's overwrite form, or the first loss launch of the layer path (scn_masked_ce_begin) -- not by fill launches of their own:
        # on the reference's own sizes a step is ~15 launches of 5-15 us each.
        part = torch.empty((1,), device=self._flat_w.device, dtype=torch.float64)
        part_set, zero_g = False, bool(fresh)
        for x, last_dev, yt, activity in staged:
            if activity is None and type(plan) is ops.SconePlan:
                if not part_set and not zero_g:
                    part.zero_()                                  # (accumulating call whose first micro-batch takes the one-launch step)
                    part_set = True
                fuse = bool(adam) and len(staged) == 1 and not part_set
                if plan.small_step(x, last_dev, yt, -1.0 / total, self.weights, self._grads, part, overwrite=not part_set,
                                   adam=(self._flat_w, self._m, self._v, self.step_size, self.weight_decay, self._step_dev)
                                   if fuse else None):
                    part_set, zero_g = True, False
                    self._adam_applied = fuse
                    continue
            logp, saved = plan.forward(x, last_dev, self.weights, activity) if activity else plan.forward(x, last_dev, self.weights)
            d_logp = torch.empty_like(logp)
            _lib.check(lib.scn_masked_ce_begin(logp.numel(), ops._dev(logp), ops._dev(yt), -1.0 / total, ops._dev(d_logp),
                                               ctypes.c_void_p(part.data_ptr()), 0 if part_set else 1,
                                               ops._dev(self._flat_g) if zero_g else None, self._flat_g.numel() if zero_g else 0,
                                               ops._stream()), "scn_masked_ce_begin")
            part_set, zero_g = True, False
            plan.backward(saved, logp, d_logp, last_dev, self.weights, self._grads)
            del saved
        if not part_set:
            part.zero_()
        if zero_g:
            self._flat_g.zero_()
        return part[0]

    def _accumulate_grad(self, plan, inputs, y, idx, total):
        return self._accumulate_staged(plan, self.stage(inputs, y, idx), total)

    # ------------------------------------------------------------------ launch-amortised step (hipGraph)
    # On the reference's own problem sizes (TE:86-90: |E| = 1001, batch 100) one optimiser step is ~25 kernel launches of a few
    # microseconds each: the step is bound by launch overhead, not by the kernels.  The device part of a step whose batch is a
    # single micro-batch -- zero the gradient buffer, (scatter the input flows,) forward, readout, cross-entropy, backward -- is
    # therefore captured ONCE into a HIP graph (torch.cuda.CUDAGraph: the C-ABI launches go to the capturing stream) and replayed;
    # the gradient all-reduce and the fused ridge + Adam kernel (whose step index changes) stay ordinary launches behind it.
    GRAPH_MAX_ELEMS = 1 << 22          # rows x trajectories x widest layer: beyond this the kernels dominate the step anyway
    GRAPH_CACHE = 6

    def _graph_ok(self, plan, n_traj):
        if not self.use_graph or self.skip_mode != "dense" or ops.KernelTimer._stack:
            return False
        if type(plan) not in (ops.SconePlan, ops.BunchPlan) or getattr(plan, "_probed", False):
            return False
        widest = max(max(sh) for sh in self._shapes)
        rows = sum(plan.sizes) if type(plan) is ops.BunchPlan else plan.n_edges
        return rows * ops.pad_count(n_traj) * (plan.promotion(self.weights) or widest) <= self.GRAPH_MAX_ELEMS

    def _adam_in_graph(self, apply):
        """The optimiser step rides inside the captured graph when nothing has to happen between the gradient and the update (one
        rank: no all-reduce) and the step index can come from device memory (scn_adam_step_dev)."""
        return (bool(apply) and self._adam_on_device() and not self.collective_always
                and dp.world(self.process_group)[1] == 1)

    def _graph_accumulate(self, plan, staged, total, prologue=None, key_extra=(), adam=False):
        """flat_g = gradient of the staged micro-batch (as _accumulate_staged after zeroing flat_g), through a captured graph:
        the first call with a given (plan, buffers, total, weight buffer) runs eagerly -- which also warms every kernel up --
        and captures; later calls replay.  The buffers of `staged` must keep their addresses (they are kept alive here).
        adam: the graph ends with the optimiser step (_adam_in_graph; the caller does the host bookkeeping, _adam_done)."""
        x, last_dev, yt, _ = staged[0]
        key = (id(plan), x.data_ptr(), last_dev.data_ptr(), yt.data_ptr(), tuple(x.shape), float(total), self._flat_w.data_ptr(),
               self._flat_g.data_ptr(), tuple(self._shapes), bool(adam)) + tuple(key_extra)
        if adam:
            self._sync_step_dev()                        # (outside the graph: a fill launch, only when the host moved the index)
        hit = self._graphs.get(key)
        if hit is not None:
            hit[0].replay()
            return hit[1]

        def body():
            if prologue is not None:
                prologue()
            part = self._accumulate_staged(plan, staged, total, fresh=True, adam=adam)
            if adam and not self._adam_applied:
                self._adam_launch()
            return part
        loss = body()                                    # eager: this call's result, and the warm-up of the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            part = body()
        self._drop_graphs(self.GRAPH_CACHE - 1)
        self._graphs[key] = (g, part, staged, plan)
        return loss

    def grad_step_staged(self, inputs, staged, total, apply=True):
        """grad_step on micro-batches already resident on the device (this rank's shard); `total` is the GLOBAL
        number of trajectories in the batch (all ranks)."""
        plan = self._plan(inputs)
        if len(staged) == 1 and staged[0][3] is None and self._graph_ok(plan, staged[0][0].shape[0] * ops.NS):
            fused = self._adam_in_graph(apply)
            loss = self._graph_accumulate(plan, staged, total, adam=fused)
            if fused:
                self._adam_done()
                return loss
        else:
            loss = self._accumulate_staged(plan, staged, total, fresh=True)
        dp.all_reduce_sum_(self._flat_g, self.process_group, force=self.collective_always)
        if apply:
            self._adam()
        return loss

    def _static_stage(self, plan, inputs, y):
        """Fixed-address staging buffers for host batches of up to batch_size trajectories (the graph replays on them)."""
        X = inputs[-1]
        D = int(np.asarray(y).shape[1])
        key = (id(plan), id(X), D, self.batch_size)
        st = self._static.get(key)
        if st is None:
            if len(self._static) >= 4:
                self._static.pop(next(iter(self._static)))
            st = self._static[key] = _StaticStage(plan, X, D, self.batch_size, self._flat_w.device)
        return st

    def grad_step(self, inputs, y, batch_mask, apply=True):
        """One optimiser step on the masked batch: gradient of self.loss (STM:307) + Adam update (STM:310, 326).
        Returns this rank's share of the data term of the loss as a 0-dim device tensor (sum over ranks = the
        batch cross-entropy; the ridge term is not included)."""
        idx = np.nonzero(np.asarray(batch_mask) == 1)[0]
        plan = self._plan(inputs)
        if plan is None:
            return self._grad_step_autograd(inputs, y, idx, apply)
        if isinstance(inputs[-1], SparseFlows) and len(idx) and self._graph_ok(plan, self.batch_size):
            rank, ws = dp.world(self.process_group)
            local = dp.shard_indices(idx, rank, ws)
            st = self._static_stage(plan, inputs, y)
            if st.load(local, inputs[1], y, len(idx)):        # False: more trajectories / flow entries than the buffers hold
                fused = self._adam_in_graph(apply)
                loss = self._graph_accumulate(plan, [st.staged], 1.0, prologue=st.scatter, key_extra=("static",), adam=fused)
                if fused:
                    self._adam_done()
                    return loss
                dp.all_reduce_sum_(self._flat_g, self.process_group, force=self.collective_always)
                if apply:
                    self._adam()
                return loss
        acc = {}

        def grad_fn(local, total):
            acc["loss"] = self._accumulate_grad(plan, inputs, y, local, total)
        dp.data_parallel_grad(idx, grad_fn, self._flat_g, self.process_group, force=self.collective_always)
        if apply:
            self._adam()
        return acc.get("loss", torch.zeros((), device=self._flat_w.device, dtype=torch.float64))

    def _grad_step_autograd(self, inputs, y, idx, apply):
        """Fallback for a user-supplied differentiable model callable (still GPU-only)."""
        ws = [w.detach().requires_grad_(True) for w in self.weights]
        sub = _select(inputs, idx)
        preds = self.model(ws, *self.shifts, *sub)
        yt = torch.as_tensor(np.asarray(y)[idx], device=preds.device, dtype=torch.float32)
        data = -(preds * yt).sum() / len(idx)
        gs = torch.autograd.grad(data, ws)
        self._flat_g.zero_()
        for g, v in zip(gs, self._grads):
            v.add_(g)
        dp.all_reduce_sum_(self._flat_g, self.process_group, force=self.collective_always)
        if apply:
            self._adam()
        return data.detach().double()

    ADAM_DEV_MAX = 65536                                 # SCN_ADAM_DEV_MAX (include/scone_hip.h): one workgroup's worth of parameters

    def _adam_on_device(self):
        return self._flat_w.numel() <= self.ADAM_DEV_MAX

    def _sync_step_dev(self):
        """The device copy of the step index follows self._step: the launch itself leaves i + 1 there, so this writes only after the
        host moved the index (train() sets the LOOP index, STM:310; a skipped empty batch, a restart)."""
        if self._step_dev_value != int(self._step):
            self._step_dev[:1].fill_(int(self._step))
            self._step_dev_value = int(self._step)

    def _adam_launch(self):
        """The launch alone (capturable: its arguments do not change from step to step)."""
        _lib.check(_lib.load().scn_adam_step_dev(self._flat_w.numel(), ops._dev(self._flat_w), ops._dev(self._flat_g),
                                                 ops._dev(self._m), ops._dev(self._v), float(self.step_size), 0.9, 0.999, 1e-8,
                                                 ctypes.c_void_p(self._step_dev.data_ptr()), float(self.weight_decay), 1.0,
                                                 ops._stream()), "scn_adam_step_dev")

    def _adam_done(self):
        """Host bookkeeping of one optimiser step taken on the device."""
        self._step += 1
        self._step_dev_value += 1
        self._wver += 1                                 # (a raw-pointer write: invisible to torch's version counter)

    def _adam(self):
        """Fused ridge + Adam on the flat buffer: g + 2*wd*w, b1=.9, b2=.999, eps=1e-8 outside the sqrt, (i+1) bias
        correction -- jax.experimental.optimizers.adam as driven by STM:300-326.  The step index travels in device memory
        (scn_adam_step_dev; the same bits as scn_adam_step) so that the graph-replayed step of small complexes can end with this very
        launch; flat buffers beyond one workgroup's worth take the host-index form."""
        if self._adam_on_device():
            self._sync_step_dev()
            self._adam_launch()
            self._adam_done()
            return
        lib = _lib.load()
        _lib.check(lib.scn_adam_step(self._flat_w.numel(), ops._dev(self._flat_w), ops._dev(self._flat_g),
                                     ops._dev(self._m), ops._dev(self._v), float(self.step_size), 0.9, 0.999, 1e-8,
                                     int(self._step), float(self.weight_decay), 1.0, ops._stream()), "scn_adam_step")
        self._step += 1
        self._wver += 1                                 # (a raw-pointer write: invisible to torch's version counter)

    # ------------------------------------------------------------------ train / test
    def train(self, inputs, y, train_mask, test_mask, n_nbrs):
        """Trains a batched model to predict y (STM:264-357)."""
        N = _n_samples(inputs[-1])
        train_mask, test_mask = np.asarray(train_mask), np.asarray(test_mask)
        n_train_samples = int(np.sum(train_mask))
        n_batches = n_train_samples // self.batch_size
        self._m.zero_(); self._v.zero_(); self._step = 0          # adam state re-initialised (STM:312)
        unshuffled_batch_mask = np.array([1] * self.batch_size + [0] * (N - self.batch_size))
        train_loss = train_acc = test_loss = test_acc = None
        for i in range(self.epochs * n_batches):                  # STM:318
            batch_mask = np.array(unshuffled_batch_mask)
            _RNG.shuffle(batch_mask)
            batch_mask = np.logical_and(batch_mask, train_mask)
            if batch_mask.sum() > 0:
                self._step = i                                    # update_fun(i, ...): bias correction with the LOOP index (STM:310)
                self.grad_step(inputs, y, batch_mask)
            # (an empty batch has no gradient: the reference would divide by zero there; only the step is skipped)
            if i % n_batches == n_batches - 1:                    # STM:328-337
                train_loss = self.loss(self.weights, inputs, y, train_mask)
                train_acc = self.accuracy(self.shifts, inputs, y, train_mask, n_nbrs)
                test_loss = self.loss(self.weights, inputs, y, test_mask)
                test_acc = self.accuracy(self.shifts, inputs, y, test_mask, n_nbrs)
                if self.verbose and dp.world(self.process_group)[0] == 0:
                    print('Epoch {} -- train loss: {:.6f} -- train acc {:.3f} -- test loss {:.6f} -- test acc {:.3f}'
                          .format(i // n_batches, train_loss, train_acc, test_loss, test_acc))
        if self.verbose and dp.world(self.process_group)[0] == 0:
            print("Epochs: {}, learning rate: {}, batch size: {}, model: {}".format(
                self.epochs, self.step_size, self.batch_size, getattr(self.model, '__name__', str(self.model))))
        self.trained = True
        return train_loss, train_acc, test_loss, test_acc

    def test(self, test_inputs, y, test_mask, n_nbrs):
        """Return the loss and accuracy for the given inputs (STM:359-368)."""
        loss = self.loss(self.weights, test_inputs, y, test_mask)
        acc = self.accuracy(self.shifts, test_inputs, y, test_mask, n_nbrs)
        if self.verbose:
            print("Test loss: {:.6f}, Test acc: {:.3f}".format(loss, acc))
        return loss, acc
