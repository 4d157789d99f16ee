"""Device-side operators: thin PyTorch-ROCm plumbing over the C-ABI (device memory, streams) -- no math here.

ConvOp wraps a scn_conv_t; SconePlan / BunchPlan hold everything one model needs on the device and
expose forward / backward over "flow slabs" ([n_slabs, rows, ns, C] fp32, see include/scone_hip.h).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ACT, GroupDesc, WorkListDesc, check, i32_array, ptr_array
from .complex import Shift, union_pattern
from .synthetic_data_gen import SparseFlows

NS = 4   # trajectories per slab
Y_STRIDE = 4   # floats per point of the shifted first-layer input y = (x, S_lo x, S_up x, 0)   (include/scone_hip.h)


class KernelTimer:
    """Optional per-call timing with events on the launch stream (torch's current stream), used by bench.py to get
    the average duration of one kernel family live, together with the ALGORITHMIC bytes of every timed call (each operand
    tensor read once, each result written once, the operator's CSR arrays once -- SURVEY.md section 8d).  Disabled by
    default: no events, no overhead.  Timers nest (a stack): every active timer sees the calls made while it is open."""
    _stack = []

    def __init__(self):
        self.records = {}

    def __enter__(self):
        KernelTimer._stack.append(self)
        return self

    def __exit__(self, *a):
        KernelTimer._stack.remove(self)

    def summary(self):
        """key -> (launches, mean ms)."""
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v) / max(len(v), 1)) for k, v in self.records.items()}

    def table(self):
        """key -> {"launches", "avg_ms", "alg_bytes" (mean per launch, None where no model is stated), "GB/s"}."""
        torch.cuda.synchronize()
        out = {}
        for k, v in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in v) / max(len(v), 1)
            nb = [n for _, _, n in v if n is not None]
            alg = sum(nb) / len(nb) if nb else None
            out[k] = {"launches": len(v), "avg_ms": ms, "alg_bytes": alg,
                      "GB/s": (alg / (ms * 1e-3) / 1e9) if (alg and ms > 0) else None}
        return out


class _timed:
    def __init__(self, key, nbytes=None):
        self.key, self.nbytes = key, nbytes

    def __enter__(self):
        if KernelTimer._stack:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *a):
        if KernelTimer._stack:
            self.b.record()
            for t in KernelTimer._stack:
                t.records.setdefault(self.key, []).append((self.a, self.b, self.nbytes))


def _nbytes(*tensors):
    return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))


# Module switches for tests and same-box A/B runs (set them from Python; the one environment variable read here is SCN_SMALL_STEP, once,
# at import):
FUSE_FIRST = True      # False: separate scn_conv_backward + scn_conv_dw_first instead of the fused-first backward
# Small complexes: the whole gradient step of a micro-batch in one launch (SconePlan.small_step, csrc/scn_small.hip), one workgroup per
# trajectory.  SCN_SMALL_STEP=0 turns it off, =force lifts the rule of small_step_pays() below.  Measured per graph-replayed optimiser
# step, one launch against the layer-by-layer kernels, ms (tools/small_step.py, profiles/r04_small_step_ab.txt):
#   |E| =  319:  160 trajectories 0.053 / 0.110   256: 0.061 / 0.118   512: 0.109 / 0.148   1000: 0.197 / 0.198
#   |E| =  639:  100 trajectories 0.086 / 0.110   256: 0.091 / 0.140   512: 0.163 / 0.197
#   |E| =  822:  100 trajectories 0.097 / 0.115          |E| = 926:  100 trajectories 0.107 / 0.115
#   |E| = 1001:  100 trajectories 0.119 / 0.120   128: 0.121 / 0.122   160: 0.123 / 0.144   256: 0.134 / 0.150   512: 0.242 / 0.193
# One workgroup's chain does not shorten with the batch, so the launch costs (rounds of 256 workgroups) x (chain of this |E|); the layer
# kernels grow with the work.  At |E| = 1001 the chain is eight row tiles per wave and layer: worth it for a full round, not beyond.
# Round 5: above 384 edges and up to CUs / 2 trajectories the library gives every trajectory TWO workgroups (half the row tiles each,
# rows handed over through memory after every layer; include/scone_hip.h), paired / one workgroup each / layer by layer, ms
# (tools/small_pair_ab.sh, profiles/r05_small_pair_ab.txt; the layer kernels themselves are round 5's):
#   |E| =  498:   64 trajectories 0.054 / 0.066 / 0.089   128: 0.057 / 0.068 / 0.100
#   |E| =  639:   64: 0.065 / 0.078 / 0.094   100: 0.067 / 0.079 / 0.101   128: 0.069 / 0.079 / 0.103
#   |E| =  822:   64: 0.069 / 0.091 / 0.097   100: 0.072 / 0.093 / 0.105   128: 0.073 / 0.093 / 0.107
#   |E| = 1001:   32: 0.076 / 0.106 / 0.089    64: 0.079 / 0.107 / 0.102   100: 0.081 / 0.110 / 0.109   128: 0.083 / 0.111 / 0.111
#   |E| = 1106:   64: 0.088 / 0.118 / 0.099   100: 0.093 / 0.119 / 0.121   128: 0.094 / 0.121 / 0.125
# -- wherever that form applies it is the fastest of the three.
SMALL_STEP = os.environ.get("SCN_SMALL_STEP", "1") != "0"
SMALL_STEP_MAX_EDGES = (1 << 30) if os.environ.get("SCN_SMALL_STEP") == "force" else 960     # up to here for any batch of one round


_N_CUS = {}


def device_cus(device=None):
    """Compute units of the device the step runs on (256 on a whole MI355X; fewer on a partitioned one), asked once per device."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if idx not in _N_CUS:
        _N_CUS[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _N_CUS[idx]


def small_step_pays(n_edges, n_traj, n_cus=256):
    """The size rule for the one-launch step (see the table above); n_traj counts the padded trajectories = workgroups, n_cus the
    device's compute units (callers pass device_cus(); the table was measured on 256)."""
    rounds = -(-n_traj // n_cus)
    if n_edges > 384 and 2 * n_traj <= n_cus:          # two workgroups per trajectory (the library's own condition, small_paired())
        return True
    if n_edges <= min(384, SMALL_STEP_MAX_EDGES):
        return rounds <= 4
    if n_edges <= min(768, SMALL_STEP_MAX_EDGES):
        return rounds <= 2
    if n_edges <= SMALL_STEP_MAX_EDGES:
        return rounds == 1 or (SMALL_STEP_MAX_EDGES >= (1 << 30) and rounds <= 2)
    return rounds == 1 and 2 * n_traj >= n_cus
FUSE_BUNCH = True      # False: per-shift SpMMs + dense-term kernels for every Bunch layer instead of the fused three-level kernels
FOLD_BUNCH = True      # False: the first two Bunch layers as ordinary layers instead of the rank-one fold (BunchPlan._fold_forward)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), "expected a contiguous CUDA tensor"
    return ctypes.c_void_p(t.data_ptr())


class WorkList:
    """Device copy of a (block, slab) work list of the zero-skipping mode (scn_work_list).  Built from a boolean
    scipy matrix A[slab, block]; `items` counts the (block, slab) pairs."""

    def __init__(self, A, device):
        import scipy.sparse as sp
        At = sp.csr_matrix(A).T.tocsr()                     # rows = blocks, column indices = slabs (ascending)
        At.sort_indices()
        nz = np.flatnonzero(np.diff(At.indptr) > 0)
        ptr = np.concatenate([[0], np.cumsum(np.diff(At.indptr)[nz])]).astype(np.int32)
        # (an empty list still needs valid pointers: pad the arrays with one unused element)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(np.append(a, 0), np.int32)).to(device)
        self.block, self.ptr, self.slab = to(nz), to(ptr), to(At.indices)
        self.n_work, self.items = int(len(nz)), int(At.nnz)
        self.desc = WorkListDesc(self.n_work, self.block.data_ptr(), self.ptr.data_ptr(), self.slab.data_ptr())

    def ref(self):
        return ctypes.byref(self.desc)


class ConvOp:
    """One shift-convolution operator (scn_conv_t).  groups: list of dicts
    {"mats": [scipy csr in DEVICE order] (0..2, same shape), "identity": bool, "n_cols": int}."""

    def __init__(self, n_rows, groups, block_start=None):
        lib = _lib.load()
        self.n_rows = int(n_rows)
        self.group_cols = []
        self.slot_group = []
        descs = (GroupDesc * len(groups))()
        keep = []
        for gi, g in enumerate(groups):
            mats = g["mats"]
            n_cols = int(g["n_cols"])
            if mats:
                rowptr, cols, vals = union_pattern(mats)
            else:
                rowptr, cols, vals = np.zeros(n_rows + 1, np.int32), np.zeros(0, np.int32), []
            rowptr = np.ascontiguousarray(rowptr, np.int32)
            cols = np.ascontiguousarray(cols, np.int32)
            vals = [np.ascontiguousarray(v, np.float32) for v in vals]
            keep += [rowptr, cols] + vals
            d = descs[gi]
            d.n_cols, d.identity, d.n_vals, d.nnz = n_cols, int(bool(g.get("identity"))), len(vals), len(cols)
            d.rowptr = rowptr.ctypes.data
            d.col = cols.ctypes.data if len(cols) else None
            d.val0 = vals[0].ctypes.data if len(vals) > 0 and len(cols) else None
            d.val1 = vals[1].ctypes.data if len(vals) > 1 and len(cols) else None
            self.group_cols.append(n_cols)
            self.slot_group += [gi] * (d.identity + d.n_vals)
        h = ctypes.c_void_p()
        if block_start is not None:                # layout hint that goes with the row order (Layout.block_starts)
            block_start = np.ascontiguousarray(block_start, np.uint8)
            assert len(block_start) == self.n_rows
            keep.append(block_start)
        check(lib.scn_conv_create_blocked(self.n_rows, len(groups), descs,
                                          block_start.ctypes.data if block_start is not None else None, ctypes.byref(h)),
              "scn_conv_create")
        self.handle = h
        self.n_groups = len(groups)
        self.n_slots = len(self.slot_group)
        self.nnz = [int(descs[g].nnz) for g in range(len(groups))]
        self.n_vals = [int(descs[g].n_vals) for g in range(len(groups))]
        # column indices + every value array + row pointers, once per launch
        self.csr_bytes = float(sum(4 * self.nnz[g] * (1 + self.n_vals[g]) + 4 * (self.n_rows + 1) for g in range(len(groups))))
        del keep

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().scn_conv_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def plan_info(self):
        nb, ms = ctypes.c_int32(0), ctypes.c_float(0)
        check(_lib.load().scn_conv_plan_info(self.handle, ctypes.byref(nb), ctypes.byref(ms)), "scn_conv_plan_info")
        return nb.value, ms.value

    def forward(self, srcs, Ws, c_out, act, out=None, wl=None, partial=None):
        """wl: WorkList (zero-skipping): only the listed items of `out` are written; `out` must be given, all-zero.
        partial: tensor of the output's shape added to the pre-activation (scn_conv_forward_accumulate; out defaults to it, in place)."""
        lib = _lib.load()
        assert wl is None or out is not None
        if partial is not None:
            assert wl is None and tuple(partial.shape) == (srcs[0].shape[0], self.n_rows, srcs[0].shape[2], c_out)
            out = partial if out is None else out
        S, ns = srcs[0].shape[0], srcs[0].shape[2]
        assert len(srcs) == self.n_groups and len(Ws) == self.n_slots
        c_in = []
        for g, x in enumerate(srcs):
            assert x.shape[0] == S and x.shape[1] == self.group_cols[g] and x.shape[2] == ns, "source slab shape"
            c_in.append(x.shape[3])
        for s, w in enumerate(Ws):
            assert tuple(w.shape) == (c_in[self.slot_group[s]], c_out), "weight shape"
        if out is None:
            out = torch.empty((S, self.n_rows, ns, c_out), device=srcs[0].device, dtype=torch.float32)
        if partial is not None:
            with _timed("conv_fwd c%s->%d + partial" % ("+".join(map(str, c_in)), c_out), _nbytes(out, partial, *srcs) + self.csr_bytes):
                check(lib.scn_conv_forward_accumulate(self.handle, S, ns, ptr_array([_dev(x).value for x in srcs]), i32_array(c_in),
                                                      ptr_array([_dev(w).value for w in Ws]), c_out, ACT[act], _dev(partial), _dev(out),
                                                      _stream()), "scn_conv_forward_accumulate")
            return out
        with _timed("conv_fwd c%s->%d" % ("+".join(map(str, c_in)), c_out), None if wl is not None else _nbytes(out, *srcs) + self.csr_bytes):
            check(lib.scn_conv_forward_list(self.handle, S, ns, ptr_array([_dev(x).value for x in srcs]), i32_array(c_in),
                                            ptr_array([_dev(w).value for w in Ws]), c_out, ACT[act], _dev(out),
                                            wl.ref() if wl is not None else None, _stream()), "scn_conv_forward")
        return out

    def backward(self, dzs, Ws, aux, act, need_dx, dWs, dx=None, wl=None, dx_partial=None):
        """dWs: list of tensors ACCUMULATED into.  Returns dx or None.  wl: WorkList (zero-skipping): only the listed
        items of `dx` are written (it must be given, all-zero) and only they contribute to dWs.
        dx_partial: tensor of dx's shape ADDED to the input gradient (scn_conv_backward_accumulate; dx defaults to it, in place)."""
        lib = _lib.load()
        assert wl is None or (dx is not None or not need_dx)
        if dx_partial is not None:
            assert wl is None and need_dx and tuple(dx_partial.shape) == tuple(aux.shape)
            dx = dx_partial if dx is None else dx
        S, ns, c_aux = aux.shape[0], aux.shape[2], aux.shape[3]
        assert aux.shape[1] == self.n_rows and len(dzs) == self.n_groups
        c_dz = []
        for g, x in enumerate(dzs):
            assert x.shape[0] == S and x.shape[1] == self.group_cols[g] and x.shape[2] == ns, "dz slab shape"
            c_dz.append(x.shape[3])
        for s, (w, dw) in enumerate(zip(Ws, dWs)):
            assert tuple(w.shape) == (c_aux, c_dz[self.slot_group[s]]) and tuple(dw.shape) == tuple(w.shape)
        cdz = i32_array(c_dz)
        nbytes = lib.scn_conv_backward_workspace(self.handle, S, ns, cdz, c_aux)
        ws = torch.empty(max(int(nbytes), 256), device=aux.device, dtype=torch.uint8)
        if need_dx and dx is None:
            dx = torch.empty_like(aux)
        if dx_partial is not None:
            with _timed("conv_bwd c%s->%d + partial" % ("+".join(map(str, c_dz)), c_aux), _nbytes(aux, dx, dx_partial, *dzs) + self.csr_bytes):
                check(lib.scn_conv_backward_accumulate(self.handle, S, ns, ptr_array([_dev(x).value for x in dzs]), cdz,
                                                       ptr_array([_dev(w).value for w in Ws]), _dev(aux), c_aux, ACT[act],
                                                       _dev(dx_partial), _dev(dx), ptr_array([_dev(d).value for d in dWs]),
                                                       ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
                      "scn_conv_backward_accumulate")
            return dx
        with _timed("conv_bwd c%s->%d%s" % ("+".join(map(str, c_dz)), c_aux, "" if need_dx else " (dW only)"),
                    None if wl is not None else _nbytes(aux, dx if need_dx else None, *dzs) + self.csr_bytes):
            check(lib.scn_conv_backward_list(self.handle, S, ns, ptr_array([_dev(x).value for x in dzs]), cdz,
                                             ptr_array([_dev(w).value for w in Ws]), _dev(aux), c_aux, ACT[act],
                                             _dev(dx) if need_dx else None, ptr_array([_dev(d).value for d in dWs]),
                                             ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                             wl.ref() if wl is not None else None, _stream()), "scn_conv_backward")
        return dx if need_dx else None

    def forward_first(self, x, Ws, c_out, act, out=None, y=None, wl=None):
        """First layer (one 1-channel input): returns (out, y) with y = (x, S_lo x, S_up x, 0) per point, or None when the
        shape is not served (scn_conv_forward_first).  wl: as in forward (out and y given, all-zero)."""
        lib = _lib.load()
        S, rows, ns, c_in = x.shape
        if c_in != 1 or self.n_groups != 1 or self.n_slots != 3:
            return None
        assert wl is None or (out is not None and y is not None)
        if out is None:
            out = torch.empty((S, self.n_rows, ns, c_out), device=x.device, dtype=torch.float32)
        if y is None:
            y = torch.empty((S, self.n_rows, ns, Y_STRIDE), device=x.device, dtype=torch.float32)
        with _timed("conv_fwd c1->%d" % c_out, None if wl is not None else _nbytes(x, out) + self.csr_bytes):
            st = lib.scn_conv_forward_first(self.handle, S, ns, _dev(x), ptr_array([_dev(w).value for w in Ws]), c_out,
                                            ACT[act], _dev(out), _dev(y), wl.ref() if wl is not None else None, _stream())
        if st == _lib.SCN_ERR_UNSUPPORTED:
            return None
        check(st, "scn_conv_forward_first")
        return out, y

    def forward_power(self, x0, x, Ws, act):
        """out = act(x0 W0 + x W1 + (S x) W2) for an operator with identity + one value array (scn_conv_forward_power);
        None when the shape is not served."""
        lib = _lib.load()
        S, rows, ns, c = x.shape
        out = torch.empty_like(x)
        with _timed("conv_fwd_power c%d" % c, _nbytes(x0, x, out) + self.csr_bytes):
            st = lib.scn_conv_forward_power(self.handle, S, ns, _dev(x0), _dev(x), ptr_array([_dev(w).value for w in Ws]), c,
                                            ACT[act], _dev(out), _stream())
        if st == _lib.SCN_ERR_UNSUPPORTED:
            return None
        check(st, "scn_conv_forward_power")
        return out

    def backward_power(self, dz, g1, Ws, aux, act, need_dx, dWs):
        """Backward of forward_power given g1 = S^T dz (scn_conv_backward_power); returns (served, dx)."""
        lib = _lib.load()
        S, rows, ns, c = dz.shape
        nbytes = int(lib.scn_conv_backward_power_workspace(self.handle, S, ns, c))
        if nbytes == 0:
            return False, None
        ws = torch.empty(nbytes, device=dz.device, dtype=torch.uint8)
        dx = torch.empty_like(aux) if need_dx else None
        with _timed("conv_bwd_power c%d" % c, _nbytes(dz, g1, aux, dx) + self.csr_bytes):
            check(lib.scn_conv_backward_power(self.handle, S, ns, _dev(dz), _dev(g1), ptr_array([_dev(w).value for w in Ws]),
                                              _dev(aux), c, ACT[act], _dev(dx) if need_dx else None,
                                              ptr_array([_dev(d).value for d in dWs]), ctypes.c_void_p(ws.data_ptr()),
                                              ws.numel(), _stream()), "scn_conv_backward_power")
        return True, dx

    def backward_fused_first(self, dz, Ws, aux, act, y, dWs, dWs_first, wl=None):
        """Backward of the layer after the first one, fused with the first layer's weight gradient: dWs (this layer) and
        dWs_first are accumulated, the input gradient is never written (scn_conv_backward_fused_first).  False when the shape
        is not served."""
        lib = _lib.load()
        S, rows, ns, c = dz.shape
        if tuple(aux.shape) != (S, rows, ns, c) or tuple(y.shape) != (S, rows, ns, Y_STRIDE) or self.n_groups != 1:
            return False
        nbytes = int(lib.scn_conv_backward_fused_first_workspace(self.handle, S, ns, c))
        if nbytes == 0:
            return False
        ws = torch.empty(nbytes, device=dz.device, dtype=torch.uint8)
        with _timed("conv_bwd c%d->%d + dW_first" % (c, c), None if wl is not None else _nbytes(dz, aux, y) + self.csr_bytes):
            check(lib.scn_conv_backward_fused_first(self.handle, S, ns, _dev(dz), ptr_array([_dev(w).value for w in Ws]), _dev(aux),
                                                    c, ACT[act], _dev(y), ptr_array([_dev(d).value for d in dWs]),
                                                    ptr_array([_dev(d).value for d in dWs_first]),
                                                    ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                                    wl.ref() if wl is not None else None, _stream()), "scn_conv_backward_fused_first")
        return True

    def clear(self, t, wl):
        """Zero the listed items of a [S, rows, ns, C] tensor (scn_clear_list)."""
        check(_lib.load().scn_clear_list(self.handle, t.shape[2], t.shape[3], _dev(t), wl.ref(), _stream()), "scn_clear_list")

    def plan_blocks(self):
        nb = self.plan_info()[0]
        row0 = np.zeros(nb + 1, np.int32)
        check(_lib.load().scn_conv_plan_blocks(self.handle, row0.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
              "scn_conv_plan_blocks")
        return row0

    def dw_first_served(self, dz):
        """Whether dw_first takes this gradient tensor's shape (scn_conv_dw_first_workspace > 0)."""
        S, rows, ns, c = dz.shape
        return rows == self.n_rows and int(_lib.load().scn_conv_dw_first_workspace(self.handle, S, ns, c)) > 0

    def dw_first(self, x, y, dz, dWs, wl=None):
        """First-layer weight gradient through the forward operator (scn_conv_dw_first); y: the shifted input saved by
        forward_first (or None: recomputed from x).  False if the shape is not served."""
        lib = _lib.load()
        S, rows, ns, c = dz.shape
        assert rows == self.n_rows and (y is None or tuple(y.shape) == (S, rows, ns, Y_STRIDE))
        nbytes = int(lib.scn_conv_dw_first_workspace(self.handle, S, ns, c))
        if nbytes == 0:
            return False
        ws = torch.empty(nbytes, device=dz.device, dtype=torch.uint8)
        with _timed("conv_dw_first c%d" % c, None if wl is not None else _nbytes(dz) + 4.0 * S * rows * ns + self.csr_bytes):
            check(lib.scn_conv_dw_first(self.handle, S, ns, _dev(x) if x is not None else None,
                                        _dev(y) if y is not None else None, _dev(dz), c,
                                        ptr_array([_dev(d).value for d in dWs]), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                        wl.ref() if wl is not None else None, _stream()), "scn_conv_dw_first")
        return True

    def spmm_dual(self, x, dual=True):
        lib = _lib.load()
        S, rows, k = x.shape
        assert rows == self.group_cols[0]
        ya = torch.empty((S, self.n_rows, k), device=x.device, dtype=torch.float32)
        yb = torch.empty_like(ya) if dual else None
        with _timed("spmm_dual k%d" % k if dual else "spmm k%d" % k, _nbytes(x, ya, yb) + self.csr_bytes):
            check(lib.scn_spmm_dual(self.handle, S, k, _dev(x), _dev(ya), _dev(yb) if dual else None, _stream()),
                  "scn_spmm_dual")
        return ya, yb


class TermsOp:
    """The seven Bunch shifts of one direction as ONE operator on the concatenated row space [nodes | edges | faces]
    (scn_terms_create): blocks[(l, j)] = the shift (device order) from source level j to target level l, or absent."""

    def __init__(self, sizes, blocks, merged, bins, rows_per_wave):
        import scipy.sparse as sp
        lib = _lib.load()
        self.sizes = tuple(int(x) for x in sizes)
        self.bins = tuple(int(b) for b in bins)
        off = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        grid = [[blocks.get((l, j)) for j in range(3)] for l in range(3)]
        for l in range(3):
            for j in range(3):
                if grid[l][j] is None:
                    grid[l][j] = sp.csr_matrix((self.sizes[l], self.sizes[j]))
        M = sp.bmat(grid, format="csr")
        M.sort_indices()
        R = int(off[3])
        rowptr = np.ascontiguousarray(M.indptr, np.int32)
        col = np.ascontiguousarray(M.indices, np.int32)
        val = np.ascontiguousarray(M.data, np.float32)
        term = ((col >= off[1]).astype(np.uint8) + (col >= off[2]).astype(np.uint8)).astype(np.uint8)
        lvl = np.ascontiguousarray(off, np.int32)
        if merged is None:                                     # no common curve known: level by level
            merged = np.concatenate([np.full(n, l, np.uint8) for l, n in enumerate(self.sizes)])
        merged = np.ascontiguousarray(merged, np.uint8)
        assert len(merged) == R
        binsa = np.ascontiguousarray(self.bins, np.int32)
        h = ctypes.c_void_p()
        check(lib.scn_terms_create(R, rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, term.ctypes.data, lvl.ctypes.data,
                                   merged.ctypes.data, binsa.ctypes.data, int(rows_per_wave), ctypes.byref(h)), "scn_terms_create")
        self.handle = h
        self.nnz = int(M.nnz)
        self.csr_bytes = 4.0 * self.nnz * 2 + 4.0 * (R + 1)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().scn_conv_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def plan_info(self):
        nb, ms = ctypes.c_int32(0), ctypes.c_float(0)
        check(_lib.load().scn_conv_plan_info(self.handle, ctypes.byref(nb), ctypes.byref(ms)), "scn_conv_plan_info")
        return nb.value, ms.value

    def forward(self, xs, Ws, act, want):
        """xs[j]: level tensors [S, rows_j, ns, 32] or None (identically zero); Ws[l][j]: (32, 32) tensors or None; want[l]:
        compute level l.  Returns the list of outputs (None where not wanted)."""
        lib = _lib.load()
        ref = next(x for x in xs if x is not None)
        S, ns = ref.shape[0], ref.shape[2]
        outs = [torch.empty((S, self.sizes[l], ns, 32), device=ref.device, dtype=torch.float32) if want[l] else None
                for l in range(3)]
        flatW = [Ws[l][j] for l in range(3) for j in range(3)]
        nb = _nbytes(*[x for x in xs if x is not None], *[o for o in outs if o is not None]) + self.csr_bytes
        with _timed("terms_fwd c32", nb):
            check(lib.scn_terms_forward(self.handle, S, ns, ptr_array([_dev(x).value if x is not None else None for x in xs]),
                                        ptr_array([_dev(w).value if w is not None else None for w in flatW]), 32, ACT[act],
                                        ptr_array([_dev(o).value if o is not None else None for o in outs]), _stream()),
                  "scn_terms_forward")
        return outs


def _terms_backward(op, dzs, Ws, auxs, act, want_dx, dWs):
    """TermsOp.backward (kept a function so the class above stays forward-only readable): dzs[j] / auxs[l] level tensors or
    None, Ws[l][j] / dWs[l][j] (32, 32) tensors or None, want_dx[l].  Returns the list of input gradients (None where not wanted)."""
    lib = _lib.load()
    ref = next(x for x in dzs if x is not None)
    S, ns = ref.shape[0], ref.shape[2]
    dxs = [torch.empty((S, op.sizes[l], ns, 32), device=ref.device, dtype=torch.float32) if want_dx[l] else None for l in range(3)]
    nbytes = int(lib.scn_terms_backward_workspace(op.handle, S, ns, 32))
    assert nbytes > 0, "terms backward not served"
    ws = torch.empty(nbytes, device=ref.device, dtype=torch.uint8)
    flat = lambda M: ptr_array([_dev(M[l][j]).value if M[l][j] is not None else None for l in range(3) for j in range(3)])
    p3 = lambda L: ptr_array([_dev(t).value if t is not None else None for t in L])
    nb = _nbytes(*[x for x in dzs if x is not None], *[x for x in auxs if x is not None], *[x for x in dxs if x is not None]) + op.csr_bytes
    with _timed("terms_bwd c32", nb):
        check(lib.scn_terms_backward(op.handle, S, ns, p3(dzs), flat(Ws), p3(auxs), 32, ACT[act], p3(dxs), flat(dWs),
                                     ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()), "scn_terms_backward")
    return dxs


def _terms_backward_first(op, dzs, Ws, auxs, act, ys, dWs, dWs_first):
    """_terms_backward for the layer that follows the 1-channel first layer: no input gradients come back -- they are contracted
    with the first layer's shifted input ys[l] (S, rows_l, ns, 1) into dWs_first[l] (1, 32) inside the kernel
    (scn_terms_backward_fused_first)."""
    lib = _lib.load()
    ref = next(x for x in dzs if x is not None)
    S, ns = ref.shape[0], ref.shape[2]
    nbytes = int(lib.scn_terms_backward_workspace(op.handle, S, ns, 32))
    assert nbytes > 0, "terms backward not served"
    ws = torch.empty(nbytes, device=ref.device, dtype=torch.uint8)
    flat = lambda M: ptr_array([_dev(M[l][j]).value if M[l][j] is not None else None for l in range(3) for j in range(3)])
    p3 = lambda L: ptr_array([_dev(t).value if t is not None else None for t in L])
    nb = _nbytes(*[x for x in dzs if x is not None], *[x for x in auxs if x is not None], *[x for x in ys if x is not None]) + op.csr_bytes
    with _timed("terms_bwd c32 + dW_first", nb):
        check(lib.scn_terms_backward_fused_first(op.handle, S, ns, p3(dzs), flat(Ws), p3(auxs), 32, ACT[act], p3(ys), flat(dWs),
                                                 p3(dWs_first), ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
              "scn_terms_backward_fused_first")


def _pairable(widths, ns):
    """All widths 16: two consecutive points are one 32-channel point under block-diagonal weights diag(W, W), which is the
    shape the MFMA dense-term kernels take (the same trick as the paired C=16 convolution kernels)."""
    return ns % 2 == 0 and all(int(c) == 16 for c in widths)


DENSE_FWD_LDS = 64 * 1024      # scn_dense_terms_forward: bytes of LDS its weight matrices may take
DENSE_BWD_WW = 1024            # scn_dense_terms_backward: c_aux * c_k it serves (one 32 x 32 weight-gradient tile per term)


def dense_terms_forward(Gs, Ws, c_out, act):
    """out[p,:] = act(sum_k G_k[p,:] @ W_k) over slab tensors [S, rows, ns, c_k] (scn_dense_terms_forward)."""
    lib = _lib.load()
    S, R, ns = Gs[0].shape[:3]
    if _pairable([g.shape[3] for g in Gs] + [c_out], ns):
        out = dense_terms_forward([g.view(S, R, ns // 2, 32) for g in Gs], [torch.block_diag(w, w) for w in Ws], 32, act)
        return out.view(S, R, ns, 16)
    cins = [g.shape[3] for g in Gs]
    if max(cins + [c_out]) > 32 and c_out % 32 == 0 and all(c % 32 == 0 for c in cins):
        # widths above 32 (multiples of 32: promotion pads to them) as 32 x 32 blocks on the MFMA kernel -- the partial
        # pre-activations of an output block add up in scn_sum_act; the VALU kernel took 85 ms for one 64-wide launch at |E| = 1M
        nb_in = max(cins) // 32
        blocks = [[g if c == 32 else g[..., 32 * b:32 * b + 32].contiguous() for b in range(c // 32)] for g, c in zip(Gs, cins)]
        outs = []
        for j in range(c_out // 32):
            parts = []
            for b in range(nb_in):
                sel = [k for k, c in enumerate(cins) if c > 32 * b]
                parts.append(dense_terms_forward([blocks[k][b] for k in sel],
                                                 [Ws[k][32 * b:32 * b + 32, 32 * j:32 * j + 32].contiguous() for k in sel], 32,
                                                 act if nb_in == 1 else "none"))
            outs.append(parts[0] if len(parts) == 1 else sum_act(parts, act))
        return torch.cat(outs, dim=3)
    if 4 * c_out * sum(cins) > DENSE_FWD_LDS and c_out > 32:
        # the kernel keeps every W_k in LDS (64 KB): wider outputs run as column blocks of 32 (independent: no sums)
        return torch.cat([dense_terms_forward(Gs, [w[:, c0:c0 + 32].contiguous() for w in Ws], min(32, c_out - c0), act)
                          for c0 in range(0, c_out, 32)], dim=3)
    out = torch.empty((S, R, ns, c_out), device=Gs[0].device, dtype=torch.float32)
    with _timed("dense_fwd x%d ->%d" % (len(Gs), c_out), _nbytes(out, *Gs)):
        check(lib.scn_dense_terms_forward(S * R * ns, len(Gs), ptr_array([_dev(g).value for g in Gs]),
                                          i32_array([g.shape[3] for g in Gs]), ptr_array([_dev(w).value for w in Ws]),
                                          c_out, ACT[act], _dev(out), _stream()), "scn_dense_terms_forward")
    return out


def dense_terms_backward(Gs, Ws, aux, act, need_dx, dWs):
    lib = _lib.load()
    S, R, ns, c_aux = aux.shape
    if _pairable([g.shape[3] for g in Gs] + [c_aux], ns):
        wide = [torch.zeros((32, 32), device=aux.device, dtype=torch.float32) for _ in Gs]
        dx = dense_terms_backward([g.view(S, R, ns // 2, 32) for g in Gs], [torch.block_diag(w, w) for w in Ws],
                                  aux.view(S, R, ns // 2, 32), act, need_dx, wide)
        for d, w in zip(dWs, wide):                       # the two diagonal blocks of the virtual 32x32 gradient
            d.add_(w[:16, :16] + w[16:, 16:])
        return dx.view(S, R, ns, 16) if need_dx else None
    if any(c_aux * g.shape[3] > DENSE_BWD_WW for g in Gs) and max(c_aux, max(g.shape[3] for g in Gs)) > 1:
        return _dense_terms_backward_blocks(Gs, Ws, aux, act, need_dx, dWs)
    cs = i32_array([g.shape[3] for g in Gs])
    n_points = S * R * ns
    nbytes = lib.scn_dense_terms_backward_workspace(n_points, len(Gs), cs, c_aux)
    ws = torch.empty(max(int(nbytes), 256), device=aux.device, dtype=torch.uint8)
    dx = torch.empty_like(aux) if need_dx else None
    with _timed("dense_bwd x%d" % len(Gs), _nbytes(aux, dx, *Gs)):
        check(lib.scn_dense_terms_backward(n_points, len(Gs), ptr_array([_dev(g).value for g in Gs]), cs,
                                           ptr_array([_dev(w).value for w in Ws]), _dev(aux), c_aux, ACT[act],
                                           _dev(dx) if need_dx else None, ptr_array([_dev(d).value for d in dWs]),
                                           ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
              "scn_dense_terms_backward")
    return dx


# ----------------------------------------------------------------------------------------------
# slabs
# ----------------------------------------------------------------------------------------------

def pad_count(n, ns=NS):
    return (n + ns - 1) // ns * ns


def flows_to_slabs(flow, layout, device, ns=NS):
    """(N, E, 1) dense array/tensor or SparseFlows (caller's edge order) -> [S, E, ns, 1] in device row order."""
    lib = _lib.load()
    perm = layout.perm[1]
    if isinstance(flow, SparseFlows):
        N, E = len(flow), flow.n_edges
        S = pad_count(N, ns) // ns
        x = torch.empty((S, E, ns, 1), device=device, dtype=torch.float32)
        sample = np.repeat(np.arange(N, dtype=np.int32), np.diff(flow.ptr))
        sample_d = torch.from_numpy(sample).to(device)
        edge_d = torch.from_numpy(perm[flow.idx].astype(np.int32)).to(device)
        val_d = torch.from_numpy(np.ascontiguousarray(flow.val, np.float32)).to(device)
        check(lib.scn_scatter_flows(S, ns, E, len(sample), _dev(sample_d, torch.int32), _dev(edge_d, torch.int32),
                                    _dev(val_d), _dev(x), _stream()), "scn_scatter_flows")
        return x, N
    t = torch.as_tensor(flow)
    N, E = t.shape[0], t.shape[1]
    assert t.numel() == N * E, "flows must have one input channel (STM:261)"
    t = t.reshape(N, E).to(device=device, dtype=torch.float32)
    Np = pad_count(N, ns)
    if Np != N:
        t = torch.cat([t, t.new_zeros((Np - N, E))])
    order = torch.from_numpy(layout.order[1]).to(device)
    x = t.index_select(1, order).reshape(Np // ns, ns, E).permute(0, 2, 1).contiguous().unsqueeze(-1)
    return x, N


def slabs_to_batch(x, layout, level, n):
    """[S, rows, ns, C] (device order) -> (n, rows, C) in the caller's order (debug / tests)."""
    S, R, ns, C = x.shape
    perm = torch.from_numpy(layout.perm[level]).to(x.device)
    return x.permute(0, 2, 1, 3).reshape(S * ns, R, C)[:n].index_select(1, perm)


def batch_to_slabs(t, layout, level, ns=NS):
    """(N, rows, C) caller order -> [S, rows, ns, C] device order."""
    N, R, C = t.shape
    Np = pad_count(N, ns)
    if Np != N:
        t = torch.cat([t, t.new_zeros((Np - N, R, C))])
    order = torch.from_numpy(layout.order[level]).to(t.device)
    return t.index_select(1, order).reshape(Np // ns, ns, R, C).permute(0, 2, 1, 3).contiguous()


def remap_last_nodes(plan, last_nodes):
    """Last nodes as the plan's readout tables index them: the caller's node ids for a Bconds object; for a probed
    Bcond_func closure (complex.ProbedBconds) the rows of its table, probing nodes it has not seen yet."""
    bc = getattr(plan, "bconds", None)
    if bc is not None and hasattr(bc, "prepare"):
        ln = bc.prepare(last_nodes)
        plan.sync_readout()
        return ln
    return last_nodes


def _last_nodes_dev(last_nodes, n_pad, device):
    ln = np.zeros(n_pad, np.int32)
    a = last_nodes.detach().cpu().numpy() if torch.is_tensor(last_nodes) else np.asarray(last_nodes)
    ln[:len(a)] = a
    return torch.from_numpy(ln).to(device)


def _dense_terms_backward_blocks(Gs, Ws, aux, act, need_dx, dWs):
    """dense_terms_backward beyond the kernel's widths (c_aux * c_k > 1024): aux in channel blocks of <= 32 (independent: the
    block's rows of every W_k and dW_k), and where a term is wider than 32 its channels in blocks of 32 as well -- the partial
    input gradients of an aux block, each already times act'(aux), add up (scn_sum_act).  W_k: (c_aux, c_k)."""
    c_aux = aux.shape[3]
    cmax = max(g.shape[3] for g in Gs)
    kb = -(-cmax // 32) if cmax > 32 else 1             # channel blocks of the widest term
    dxs = []
    for a0 in range(0, c_aux, 32):
        a1 = min(c_aux, a0 + 32)
        aux_a = aux[..., a0:a1].contiguous()
        parts = []
        for j in range(kb):
            sel = [(k, 32 * j, min(g.shape[3], 32 * j + 32)) for k, g in enumerate(Gs) if g.shape[3] > 32 * j] if kb > 1 else \
                  [(k, 0, g.shape[3]) for k, g in enumerate(Gs)]
            if not sel:
                continue
            Gj = [Gs[k] if (c0 == 0 and c1 == Gs[k].shape[3]) else Gs[k][..., c0:c1].contiguous() for k, c0, c1 in sel]
            Wj = [Ws[k][a0:a1, c0:c1].contiguous() for k, c0, c1 in sel]
            dWj = [torch.zeros_like(w) for w in Wj]
            parts.append(dense_terms_backward(Gj, Wj, aux_a, act, need_dx, dWj))
            for (k, c0, c1), d in zip(sel, dWj):
                dWs[k][a0:a1, c0:c1] += d
        if need_dx:
            dxs.append(parts[0] if len(parts) == 1 else sum_act(parts, "none"))
    return torch.cat(dxs, dim=3) if need_dx else None


def spmm_chunked(op, x):
    """y = S x for a slab tensor x (S, R, ns, c) of ANY width: the LDS-blocked SpMM takes ns * c <= 128 columns per launch, wider
    operands go through it in channel blocks (the shift acts on every channel alike)."""
    S, R, ns, c = x.shape
    step = max(1, 128 // ns)
    if c <= step:
        y, _ = op.spmm_dual(x.view(S, R, ns * c), dual=False)
        return y.view(S, op.n_rows, ns, c)
    outs = []
    for c0 in range(0, c, step):
        w = min(step, c - c0)
        y, _ = op.spmm_dual(x[..., c0:c0 + w].contiguous().view(S, R, ns * w), dual=False)
        outs.append(y.view(S, op.n_rows, ns, w))
    return torch.cat(outs, dim=3)


# ----------------------------------------------------------------------------------------------
# hidden-width promotion
# ----------------------------------------------------------------------------------------------
# The MFMA kernels serve hidden widths 16 (on slab pairs) and 32.  Any other stack the reference documents -- mixed widths such
# as `-hidden_layers [(3, 32), (3, 16)]` (TE:51) or 3_8_3_8 (TE:82) -- runs on the SAME kernels with every hidden width
# zero-padded to one promoted width: no layer of this path has a bias and act(0) = 0 for every activation it uses, so the padded
# channels are exactly zero in every activation and every gradient, the real channels see the same sums (the extra terms
# are exact zeros), and the padded rows / columns of the weight gradients are dropped again on the way out.

WIDE_MAX = 128        # widest hidden layer served by the 32-channel-block decomposition (SconePlan._wide_stack)


def promoted_width(widths, wide=False):
    """Width every hidden layer is padded to, or None when the stack runs as it is (uniform 16 or 32; wider than 32 unless
    `wide`: then the next multiple of 32 up to WIDE_MAX -- such a stack runs in 32-channel blocks)."""
    ws = {int(c) for c in widths}
    if len(ws) == 1 and ws <= {16, 32}:
        return None
    m = max(ws)
    if m > 32:
        return -(-m // 32) * 32 if (wide and m <= WIDE_MAX) else None
    return 16 if m <= 16 else 32


def sum_act(terms, act, out=None):
    """out = act(terms[0] + terms[1] + ...) elementwise (scn_sum_act; out defaults to terms[0], in place)."""
    out = terms[0] if out is None else out
    check(_lib.load().scn_sum_act(out.numel(), len(terms), ptr_array([_dev(t).value for t in terms]), ACT[act], _dev(out), _stream()),
          "scn_sum_act")
    return out


def promote_weights(weights, n_first, n_last, P):
    """Zero-padded copies: rows of every matrix but the first layer's n_first, columns of every matrix but the last n_last."""
    out, n = [], len(weights)
    for i, w in enumerate(weights):
        r = w.shape[0] if i < n_first else P
        c = w.shape[1] if i >= n - n_last else P
        out.append(w if (r, c) == tuple(w.shape) else torch.nn.functional.pad(w, (0, c - w.shape[1], 0, r - w.shape[0])).contiguous())
    return out


def demote_grads(grads, padded):
    for g, gp in zip(grads, padded):
        if gp is not g:
            g.add_(gp[:g.shape[0], :g.shape[1]])


# ----------------------------------------------------------------------------------------------
# scone / ebli
# ----------------------------------------------------------------------------------------------

class SconePlan:
    """Device state of a scone/ebli model: the fused conv operator (identity + S_lower + S_upper on their shared
    pattern), its transpose when the shifts are not symmetric, and the readout tables."""

    def __init__(self, S_lower, S_upper, bconds, act, device):
        assert isinstance(S_lower, Shift) and isinstance(S_upper, Shift), "shifts must come from SimplicialComplex"
        self.layout = S_lower.layout
        self.act = act
        self.device = device
        E = S_lower.shape[0]
        self.n_edges = E
        self._init_operators(S_lower, S_upper)
        self.bconds = bconds
        self._dz_zero = {}                              # all-zero readout-gradient buffers, by shape (see backward)
        self._zero_pool = {}                            # zero-skipping mode: all-zero activation / gradient buffers
        self._blocks = None                             # (block of row, block adjacency), built on first use
        self._upload_readout()

    def _upload_readout(self):
        """Device copies of the readout tables (TE:279, 288, 298-303).  A ProbedBconds (plain Bcond_func closure) grows as
        new last nodes are probed: sync_readout() re-uploads when its version has moved."""
        bconds, device = self.bconds, self.device
        ptr, edge, sign, edge_nodes = bconds.incidence_tables()
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.inc_ptr, self.inc_edge, self.inc_sign, self.edge_nodes = to(ptr), to(edge), to(sign), to(edge_nodes)
        nbr = np.asarray(bconds.nbrhoods)
        if nbr.shape[0] == 0:
            nbr = -np.ones((1, 1), np.int64)
        self.nbr = to(nbr.astype(np.int32))
        self.n_nodes, self.max_deg = nbr.shape
        self._h_nbr, self._h_inc_ptr, self._h_inc_edge = nbr, np.asarray(ptr), np.asarray(edge)
        inc_deg = np.diff(np.asarray(ptr)).astype(np.int64)             # readout items of a last node: incident edges of its neighbours
        self.max_items = int(np.where(nbr >= 0, inc_deg[np.maximum(nbr, 0)], 0).sum(axis=1).max()) if inc_deg.size else 0
        self._ro_rows = None
        self._readout_version = getattr(bconds, "version", 0)
        self._probed = hasattr(bconds, "prepare")

    def sync_readout(self):
        if getattr(self.bconds, "version", 0) != self._readout_version:
            self._upload_readout()

    def _init_operators(self, S_lower, S_upper):
        E = self.n_edges
        lo, up = S_lower.device_csr(), S_upper.device_csr()
        hint = self.layout.block_starts[S_lower.row_level]
        self._pattern = (abs(lo) + abs(up)).tocsr()
        self.conv = ConvOp(E, [{"mats": [lo, up], "identity": True, "n_cols": E}], hint)
        if S_lower.is_symmetric() and S_upper.is_symmetric():
            self.conv_T = self.conv
        else:
            self.conv_T = ConvOp(E, [{"mats": [lo.T.tocsr(), up.T.tocsr()], "identity": True, "n_cols": E}], hint)
        self.nnz_pattern = self.conv.nnz[0]
        self.nnz_lower, self.nnz_upper = int(lo.nnz), int(up.nnz)

    # -- raw forward/backward over slabs (no autograd): used by the autograd Function and by the trainer
    def conv_stack(self, x, weights, activity=None):
        n_layers = (len(weights) - 1) / 3
        assert n_layers % 1 == 0, "wrong number of weights"                    # TE:141-142
        hs, y0 = [x], None
        S, E, ns, _ = x.shape
        for i in range(int(n_layers)):
            w = weights[3 * i:3 * i + 3]
            c_out = w[0].shape[1]
            wl = activity["fwd"][i] if activity else None
            out = self._zeros((S, E, ns, c_out)) if activity else None
            first = None
            if i == 0:
                first = self.conv.forward_first(x, w, c_out, self.act, out=out,
                                                y=self._zeros((S, E, ns, Y_STRIDE)) if activity else None, wl=wl)
                assert first is not None or not activity
            if first is not None:                       # 1-channel input: keep the shifted input for the weight gradient
                hs.append(first[0])
                y0 = first[1]
            else:
                hs.append(self.conv.forward([hs[-1]], w, c_out, self.act, out=out, wl=wl))
        return hs, y0

    # -- zero-skipping mode ---------------------------------------------------------------------------------
    def _block_graph(self):
        """(block of row, row adjacency incl. self and transposes, rows -> blocks incidence, n_blocks)."""
        if self._blocks is None:
            import scipy.sparse as sp
            row0 = self.conv.plan_blocks()
            nb = len(row0) - 1
            E = self.n_edges
            blk_of = (np.searchsorted(row0, np.arange(E), side="right") - 1).astype(np.int64)
            pat = (self._pattern != 0).astype(np.int32)
            radj = ((pat + pat.T + sp.identity(E, dtype=np.int32, format="csr")) > 0).astype(np.int32).tocsr()
            to_blk = sp.csr_matrix((np.ones(E, np.int32), (np.arange(E), blk_of)), shape=(E, nb))
            self._blocks = (blk_of, radj, to_blk, nb)
        return self._blocks

    def _readout_rows(self):
        """(n_nodes x n_edges) 0/1: device rows of the edges incident to a neighbour of the node (what Bcond(node) reads)."""
        if getattr(self, "_ro_rows", None) is None:
            import scipy.sparse as sp
            V, D = self._h_nbr.shape
            r, c = np.nonzero(self._h_nbr >= 0)
            nbr = sp.csr_matrix((np.ones(len(r), np.int32), (r, self._h_nbr[r, c])), shape=(V, V))
            deg = np.diff(self._h_inc_ptr)
            inc = sp.csr_matrix((np.ones(len(self._h_inc_edge), np.int32),
                                 (np.repeat(np.arange(V), deg), self._h_inc_edge)), shape=(V, self.n_edges))
            self._ro_rows = ((nbr @ inc) > 0).astype(np.int32).tocsr()
        return self._ro_rows

    def _trajectory_supports(self, flow, last_nodes, n_layers):
        """Per TRAJECTORY and layer l = 1..L, as 0/1 matrices (N x n_blocks): Z[l] blocks holding a row where H_l can be
        non-zero, D[l] blocks holding a row the loss can see (support of the gradient of layer l's pre-activation), F[l]
        blocks holding a row that is both.  Computed once per dataset (cached on a hash of the inputs' content): supports are
        tracked per ROW -- one hop = the operator's own pattern -- and only then mapped to plan blocks."""
        import scipy.sparse as sp
        import hashlib
        ln = np.asarray(last_nodes)
        h = hashlib.blake2b(digest_size=16)                  # keyed on CONTENT: an in-place edit of the inputs must miss
        if isinstance(flow, SparseFlows):
            for a in (flow.ptr, flow.idx, flow.val):
                h.update(np.ascontiguousarray(a).view(np.uint8))
        else:
            fa = flow.detach().cpu().numpy() if torch.is_tensor(flow) else np.asarray(flow)
            h.update(str(fa.shape).encode())
            h.update(np.ascontiguousarray(fa).view(np.uint8))
        h.update(np.ascontiguousarray(ln, np.int64).view(np.uint8))
        key = (n_layers, h.hexdigest())
        c = getattr(self, "_act_cache", None)
        if c is not None and c["key"] == key:
            return c
        blk_of, radj, to_blk, nb = self._block_graph()
        perm = self.layout.perm[1]
        fl = flow if isinstance(flow, SparseFlows) else SparseFlows.fromdense(np.asarray(flow))
        N, E = len(fl), self.n_edges

        def ones(M):                                         # 0/1 pattern of a product, in place
            M = M.tocsr()
            M.data[:] = 1
            return M
        hop = lambda M: ones(M @ radj)
        blocks = lambda M: ones(M @ to_blk)
        traj = np.repeat(np.arange(N), np.diff(fl.ptr))
        rows = [ones(sp.csr_matrix((np.ones(len(traj), np.int32), (traj, perm[fl.idx])), shape=(N, E)))]
        for _ in range(n_layers):
            rows.append(hop(rows[-1]))                       # row support of H_1 .. H_L
        # rows the readout touches: edges incident to the neighbours of the last node (Bcond(last), TE:298-303)
        last = np.asarray(last_nodes)[:N]
        sel = sp.csr_matrix((np.ones(N, np.int32), (np.arange(N), last)), shape=(N, self.n_nodes))
        need = [None] * (n_layers + 1)
        need[n_layers] = ones(sel @ self._readout_rows())
        for l in range(n_layers - 1, 0, -1):
            need[l] = hop(need[l + 1])
        c = {"key": key, "N": N, "nb": nb,
             "Z": [None] + [blocks(rows[l]) for l in range(1, n_layers + 1)],
             "D": [None] + [blocks(need[l]) for l in range(1, n_layers + 1)],
             "F": [None] + [blocks(rows[l].multiply(need[l])) for l in range(1, n_layers + 1)]}
        self._act_cache = c
        return c

    def activity(self, flow, last_nodes, n_layers, hidden, mode, sel=None):
        """Work lists of one micro-batch = trajectories `sel` (default: all) of the dataset (flow, last_nodes).
        mode "zeros": every (block, slab) item that can hold a non-zero value (a layer's output is exactly zero outside the
        one-hop closure of its input's support); mode "field": additionally only what the loss can see (the readout reads H_L
        on the edges around the last nodes; each layer below needs one more hop).  None when the shape is not served by the
        work-list kernels."""
        import scipy.sparse as sp
        if mode in (None, "dense") or hidden not in (16, 32) or not self.conv.plan_info()[0] or self._probed:
            return None
        c = self._trajectory_supports(flow, last_nodes, n_layers)
        sel = np.arange(c["N"]) if sel is None else np.asarray(sel)
        n = len(sel)
        S = pad_count(n, NS) // NS
        agg = sp.csr_matrix((np.ones(n, np.int32), (np.arange(n) // NS, sel)), shape=(S, c["N"]))   # slab <- its trajectories
        dev = self.device
        fsrc = c["F"] if mode == "field" else c["Z"]
        fwd = [WorkList(agg @ fsrc[l], dev) for l in range(1, n_layers + 1)]
        bwd = [None] + [WorkList(agg @ c["D"][l], dev) for l in range(1, n_layers + 1)]
        total = S * c["nb"]
        return {"fwd": fwd, "bwd": bwd, "mode": mode,
                "active_fraction": {"fwd": [w.items / total for w in fwd], "bwd": [w.items / total for w in bwd[1:]]}}

    def _zeros(self, shape):
        pool = self._zero_pool.setdefault(tuple(shape), [])
        return pool.pop() if pool else torch.zeros(tuple(shape), device=self.device, dtype=torch.float32)

    def release(self, saved):
        """Forward-only use of the zero-skipping mode (prediction): hand the forward's pooled buffers back, all-zero again."""
        hs, bh, y0, activity = saved[:4]
        if activity:
            for l in range(1, len(hs)):
                self._give_back(hs[l], activity["fwd"][l - 1])
            self._give_back(y0, activity["fwd"][0])

    def _give_back(self, t, wl):
        self.conv.clear(t, wl)                              # all-zero again
        self._zero_pool.setdefault(tuple(t.shape), []).append(t)

    def readout(self, H, w_last, last_dev):
        lib = _lib.load()
        S, E, ns, C = H.shape
        N = S * ns
        assert tuple(w_last.shape) == (C, 1), "readout weight must be (C, 1) (STM:233, out_channels = 1)"
        bh = torch.empty((N, self.max_deg, C), device=H.device, dtype=torch.float32)
        logits = torch.empty((N, self.max_deg), device=H.device, dtype=torch.float32)
        logp = torch.empty_like(logits)
        check(lib.scn_readout_forward(S, ns, E, C, _dev(H), _dev(w_last), _dev(self.nbr, torch.int32), self.n_nodes,
                                      self.max_deg, _dev(last_dev, torch.int32), _dev(self.inc_ptr, torch.int32),
                                      _dev(self.inc_edge, torch.int32), _dev(self.inc_sign), _dev(bh), _dev(logits),
                                      _dev(logp), _stream()), "scn_readout_forward")
        return logp, bh, logits

    def _blocked(self):
        op = self.conv if self.conv is not None else self.op
        return op.plan_info()[0] > 0

    # -- small complexes: the whole gradient step of a micro-batch in one launch (scn_small_step, csrc/scn_small.hip)
    def small_step(self, x, last_dev, yt, scale, weights, grads, loss, overwrite=False, adam=None):
        """grads[k] += d/dW[k] of scale * sum_n <logp_n, yt_n> and loss[0] += that sum, with one workgroup per trajectory keeping
        the activations in LDS through all layers and both directions (the reference's own sizes, TE:86-90).  False when the
        shape is not served (hidden width other than 16, a complex too large for the LDS, ...): the caller then runs
        forward / scn_masked_ce / backward.  overwrite: grads and loss are SET instead of accumulated into.
        adam (with overwrite): (flat_w, m, v, lr, weight_decay, step_dev) -- the launch that sums the gradient also applies the
        optimiser step to the flat weight buffer `weights` are views of (scn_small_step_adam)."""
        if not SMALL_STEP or self.conv is None or self.conv.n_groups != 1 or len(weights) < 7 or (len(weights) - 1) % 3:
            return False
        L = (len(weights) - 1) // 3
        S, E, ns, c_in = x.shape
        if not small_step_pays(E, S * ns, device_cus(x.device)):
            return False
        hidden = weights[0].shape[1]
        shapes = [(1, hidden)] * 3 + [(hidden, hidden)] * (3 * (L - 1)) + [(hidden, 1)]
        if c_in != 1 or [tuple(w.shape) for w in weights] != shapes or tuple(yt.shape) != (S * ns, self.max_deg):
            return False
        lib = _lib.load()
        if not lib.scn_small_step_supported(self.conv.handle, L, hidden, self.max_deg, self.max_items):
            return False
        ws = torch.empty(int(lib.scn_small_step_workspace(E, S * ns, L)), device=x.device, dtype=torch.uint8)
        # bytes the launch has to move at least: the input flows, the saved activations written and read back, the operator
        nb = x.numel() * 4 + 2 * (L - 1) * S * ns * E * hidden * 4 + self.conv.csr_bytes
        if adam is not None:
            assert overwrite, "the fused optimiser step takes the whole gradient of the batch"
            flat_w, m, v, lr, wd, step_dev = adam
            with _timed("small_step L%d c%d" % (L, hidden), nb):
                check(lib.scn_small_step_adam(self.conv.handle, self.conv_T.handle, S, ns, L, hidden, _dev(x), _dev(last_dev, torch.int32),
                                              _dev(yt), float(scale), _dev(self.nbr, torch.int32), self.n_nodes, self.max_deg,
                                              self.max_items, _dev(self.inc_ptr, torch.int32), _dev(self.inc_edge, torch.int32),
                                              _dev(self.inc_sign), ptr_array([_dev(w).value for w in weights]), ACT[self.act],
                                              ptr_array([_dev(g).value for g in grads]), ctypes.c_void_p(loss.data_ptr()),
                                              ctypes.c_void_p(ws.data_ptr()), ws.numel(), _dev(flat_w), _dev(m), _dev(v), float(lr),
                                              0.9, 0.999, 1e-8, ctypes.c_void_p(step_dev.data_ptr()), float(wd), _stream()),
                      "scn_small_step_adam")
            return True
        with _timed("small_step L%d c%d" % (L, hidden), nb):
            check(lib.scn_small_step(self.conv.handle, self.conv_T.handle, S, ns, L, hidden, _dev(x), _dev(last_dev, torch.int32),
                                     _dev(yt), float(scale), _dev(self.nbr, torch.int32), self.n_nodes, self.max_deg,
                                     self.max_items, _dev(self.inc_ptr, torch.int32), _dev(self.inc_edge, torch.int32),
                                     _dev(self.inc_sign), ptr_array([_dev(w).value for w in weights]), ACT[self.act],
                                     ptr_array([_dev(g).value for g in grads]), ctypes.c_void_p(loss.data_ptr()),
                                     1 if overwrite else 0, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
                  "scn_small_step")
        return True

    def promotion(self, weights):
        """Promoted hidden width of this weight list on this plan (None: runs as it is)."""
        if len(weights) < 4 or (len(weights) - 1) % 3 or not self._blocked():
            return None
        # widths above 32 are padded to a multiple of 32 on both plans: the fused plan runs them in 32-channel blocks (_wide_stack), the
        # composed Ebli plan (PowerPlan) keeps its per-shift structure and runs its dense-term kernels as 32 x 32 MFMA blocks
        return promoted_width([w.shape[1] for w in weights[:-1]], wide=weights[0].shape[0] == 1)

    def forward(self, x, last_dev, weights, activity=None):
        P = self.promotion(weights)
        wp = promote_weights(weights, 3, 1, P) if P else None
        w = wp if P else weights
        if P and P > 32 and type(self) is SconePlan:
            hs, y0 = self._wide_stack(x, w, P)
            # the readout is linear in H: block by block with the block's rows of W_last, the logits added and normalised in one launch
            bhs, parts = [], []
            for j, Hj in enumerate(hs[-1]):
                _, bh, lg = self.readout(Hj, w[-1][32 * j:32 * j + 32], last_dev)
                bhs.append(bh)
                parts.append(lg)
            logp = torch.empty_like(parts[0])
            check(_lib.load().scn_logits_sum_log_softmax(logp.shape[0], self.max_deg, len(parts), ptr_array([_dev(t).value for t in parts]),
                                                         _dev(parts[0]), _dev(logp), _stream()), "scn_logits_sum_log_softmax")
            return logp, (hs, bhs, y0, None, wp, "wide")
        hs, y0 = self.conv_stack(x, w, activity)
        logp, bh, _ = self.readout(hs[-1], w[-1], last_dev)
        return logp, (hs, bh, y0, activity, wp)

    # -- hidden widths above 32 (TE:103-110 accepts any): every activation is P / 32 separate 32-channel tensors and a layer is
    # its (input block i, output block j) pairs on the fused C = 32 kernels -- act(sum_i sum_s (S_s H_i) W_s[i, j]) (TE:143-149):
    # the partial pre-activations of an output block are added and activated by scn_sum_act; in the backward the partial input
    # gradients of an input block add up likewise (act' of the layer below is a common factor of the partial results).  The
    # same products in the same association order per 32 x 32 weight block; ~4x the hidden-32 step at hidden 64 instead of the
    # generic one-row-per-workgroup kernels (measured 236x, profiles/HISTORY.md section 3.3).
    @staticmethod
    def _wblock(W, i, j):
        r = slice(None) if W.shape[0] == 1 else slice(32 * i, 32 * i + 32)
        return W[r, 32 * j:32 * j + 32].contiguous()

    def _wide_stack(self, x, w, P):
        k, L = P // 32, (len(w) - 1) // 3
        hs, y0 = [[x]], None
        for l in range(L):
            W, prev, outs = w[3 * l:3 * l + 3], hs[-1], []
            for j in range(k):
                if l == 0:
                    first = self.conv.forward_first(x, [self._wblock(Ws, 0, j) for Ws in W], 32, self.act)
                    assert first is not None, "wide hidden layers need the first-layer fast path (1-channel flows)"
                    outs.append(first[0])
                    y0 = first[1] if y0 is None else y0
                else:
                    # input block by input block INTO the output block: the partial pre-activation of the blocks before is added
                    # inside the next launch (scn_conv_forward_accumulate, in place), the last launch applies the activation
                    acc = None
                    for i in range(len(prev)):
                        last = i == len(prev) - 1
                        acc = self.conv.forward([prev[i]], [self._wblock(Ws, i, j) for Ws in W], 32, self.act if last else "none",
                                                partial=acc)
                    outs.append(acc)
            hs.append(outs)
        return hs, y0

    def _wide_backward(self, saved, logp, d_logp, last_dev, w, grads):
        hs, bhs, y0, _, _, _ = saved
        k, L = len(hs[-1]), len(hs) - 1
        # the readout's gradient block by block (d_logits depends on the summed logits only), each into its own pooled all-zero buffer
        tops = [self._readout_grad(hs[-1][j], bhs[j], logp, d_logp, last_dev, [w[-1][32 * j:32 * j + 32]], [grads[-1][32 * j:32 * j + 32]])
                for j in range(k)]
        dzs = [t[0] for t in tops]
        for l in reversed(range(L)):
            W, G = w[3 * l:3 * l + 3], grads[3 * l:3 * l + 3]
            if l == 0:                                  # dW_s[0, j] = sum_p (S_s x)[p] dz_j[p]: one stream over dz per block
                for j in range(k):
                    gj = [torch.zeros((1, 32), device=self.device, dtype=torch.float32) for _ in range(3)]
                    assert self.conv.dw_first(hs[0][0], y0, dzs[j], gj)
                    for Gs, g in zip(G, gj):
                        Gs[:, 32 * j:32 * j + 32] += g
                if L == 1:                              # (a single wide layer: the readout gradients were this layer's dz)
                    for dz_top, key in tops:
                        self._release_top(dz_top, key, last_dev)
                break
            new = []
            for i in range(len(hs[l])):
                acc = None                              # output block by output block INTO the input block's gradient (in place)
                for j in range(k):
                    gij = [torch.zeros((32, 32), device=self.device, dtype=torch.float32) for _ in range(3)]
                    acc = self.conv_T.backward([dzs[j]], [self._wblock(Ws, i, j) for Ws in W], hs[l][i], self.act, True, gij,
                                               dx_partial=acc)
                    for Gs, g in zip(G, gij):
                        Gs[32 * i:32 * i + 32, 32 * j:32 * j + 32] += g
                new.append(acc)
            if l == L - 1:                              # the top layer is done with the readout gradients: wipe and keep them
                for dz_top, key in tops:
                    self._release_top(dz_top, key, last_dev)
            dzs = new
        return grads

    def _readout_grad(self, H, bh, logp, d_logp, last_dev, weights, grads):
        """Gradient of the readout w.r.t. the last layer's pre-activation, into a pooled all-zero buffer (it is zero except
        on the edges around the last nodes; _release_top wipes exactly those rows again -- no 16 GB memset)."""
        lib = _lib.load()
        S, E, ns, C = H.shape
        key = (S, E, ns, C)
        pool = self._dz_zero.get(key)
        # small complexes: the launch zeroes the buffer itself, every trajectory's wave its own column (dz_is_zero = 2) -- no fill launch;
        # large ones: a pooled all-zero buffer whose touched rows are wiped after use (_release_top)
        small = H.numel() * 4 <= self.SMALL_DZ_BYTES
        dz_top = pool.pop() if pool else (torch.empty_like(H) if small else torch.zeros_like(H))
        d_logits = torch.empty_like(logp)
        d_logp = d_logp.contiguous()
        check(lib.scn_readout_backward(S, ns, E, C, _dev(H), _dev(weights[-1]), _dev(self.nbr, torch.int32),
                                       self.n_nodes, self.max_deg, _dev(last_dev, torch.int32),
                                       _dev(self.inc_ptr, torch.int32), _dev(self.inc_edge, torch.int32),
                                       _dev(self.inc_sign), _dev(self.edge_nodes, torch.int32), _dev(bh),
                                       _dev(d_logp), _dev(logp), ACT[self.act], _dev(d_logits), _dev(dz_top), 2 if small else 1,
                                       _dev(grads[-1]), _stream()), "scn_readout_backward")
        return dz_top, key

    SMALL_DZ_BYTES = 8 << 20      # below this the readout's backward zeroes the whole buffer itself instead of keeping it all-zero (small complexes)

    def _release_top(self, dz_top, key, last_dev):
        S, E, ns, C = key
        if dz_top.numel() * 4 <= self.SMALL_DZ_BYTES:
            self._dz_zero.setdefault(key, []).append(dz_top)     # (may hold anything: see _readout_grad)
            return
        check(_lib.load().scn_readout_clear_dz(S, ns, E, C, _dev(self.nbr, torch.int32), self.n_nodes, self.max_deg,
                                               _dev(last_dev, torch.int32), _dev(self.inc_ptr, torch.int32),
                                               _dev(self.inc_edge, torch.int32), _dev(self.edge_nodes, torch.int32),
                                               _dev(dz_top), _stream()), "scn_readout_clear_dz")
        self._dz_zero.setdefault(key, []).append(dz_top)

    def backward(self, saved, logp, d_logp, last_dev, weights, grads):
        """grads: list of tensors (same shapes as weights) accumulated into."""
        wp = saved[4]
        run = self._wide_backward if len(saved) == 6 else self._backward
        if wp is not None:                              # promoted widths: gradients of the padded matrices, cut back afterwards
            gp = [torch.zeros_like(w) if w is not w0 else g for w, w0, g in zip(wp, weights, grads)]
            run(saved, logp, d_logp, last_dev, wp, gp)
            demote_grads(grads, gp)
            return grads
        return run(saved, logp, d_logp, last_dev, weights, grads)

    def _backward(self, saved, logp, d_logp, last_dev, weights, grads):
        hs, bh, y0, activity, _ = saved
        dz_top, key = self._readout_grad(hs[-1], bh, logp, d_logp, last_dev, weights, grads)
        S, E, ns, C = hs[-1].shape
        L = len(hs) - 1
        dz = dz_top
        fused_first = False
        for i in reversed(range(L)):
            if i == 0 and fused_first:
                break                                   # the first layer's weight gradient came out of layer 1's backward
            wl_out = activity["bwd"][i] if (activity and i > 0) else None      # items of this layer's input gradient
            wl_in = activity["bwd"][i + 1] if activity else None                # support of dz
            dz_in = dz
            if i == 1 and hs[0].shape[3] == 1 and y0 is not None and FUSE_FIRST and \
                    self.conv_T.backward_fused_first(dz, weights[3:6], hs[1], self.act, y0, grads[3:6], grads[0:3], wl=wl_out):
                fused_first = True                      # dx of this layer only feeds dW_first: contracted in registers, never written
                dz = None
            elif i == 0 and hs[0].shape[3] == 1 and self.conv.dw_first(hs[0], y0, dz, grads[0:3], wl=wl_in):
                dz = None                               # first layer: shifted 1-channel input x one stream over dz
            else:
                assert not activity or i > 0, "zero-skipping needs the first-layer fast path"
                dx = self._zeros(hs[i].shape) if activity else None
                dz = self.conv_T.backward([dz], weights[3 * i:3 * i + 3], hs[i], self.act, i > 0, grads[3 * i:3 * i + 3],
                                          dx=dx, wl=wl_out)
            if i == L - 1:                              # the top layer is done with the readout gradient: wipe and keep it
                self._release_top(dz_top, key, last_dev)
            elif activity:
                self._give_back(dz_in, wl_in)
        if activity:                                    # the forward's buffers go back to the pool, all-zero again
            for l in range(1, L + 1):
                self._give_back(hs[l], activity["fwd"][l - 1])
            self._give_back(y0, activity["fwd"][0])
        return grads


class PowerPlan(SconePlan):
    """scone-like model whose second shift is the SQUARE of the first (Ebli / SNN: L1 and L1^2, TE:251-253), for complexes
    where the rows of the square no longer fit the LDS-blocked plan (> 128 distinct sources).  The square is never formed:
    per layer  G1 = S H,  G2 = S G1  on the LDS-blocked SpMM, then  act(H W0 + G1 W1 + G2 W2)  and its backward on the dense
    term kernels (scn_dense_terms_*) -- the same composition the Bunch plan uses."""

    def _init_operators(self, S_lower, S_upper):
        E = self.n_edges
        m = S_lower.device_csr()
        hint = self.layout.block_starts[S_lower.row_level]
        # identity slot on: the fused kernels read the staged tensor's own row as their middle term (the SpMM ignores it)
        self.op = ConvOp(E, [{"mats": [m], "identity": True, "n_cols": E}], hint)
        self.op_T = self.op if S_lower.is_symmetric() else ConvOp(E, [{"mats": [m.T.tocsr()], "identity": True, "n_cols": E}], hint)
        self.conv = self.conv_T = None
        self.nnz_pattern = self.nnz_lower = int(m.nnz)
        self.nnz_upper = int(S_upper.csr.nnz)

    @staticmethod
    def _shift(op, x):
        return spmm_chunked(op, x)

    def activity(self, *a, **k):
        return None

    def conv_stack(self, x, weights, activity=None):
        n_layers = (len(weights) - 1) / 3
        assert n_layers % 1 == 0, "wrong number of weights"                    # TE:159-160
        hs, y0 = [x], None
        for i in range(int(n_layers)):
            w = weights[3 * i:3 * i + 3]
            g1 = self._shift(self.op, hs[-1])
            out = self.op.forward_power(hs[-1], g1, w, self.act) if hs[-1].shape[3] == w[0].shape[1] else None
            if out is None:                            # widths the fused kernel does not take: shift again, dense terms
                g2 = self._shift(self.op, g1)
                out = dense_terms_forward([hs[-1], g1, g2], w, w[0].shape[1], self.act)
                if i == 0 and x.shape[3] == 1:
                    y0 = torch.cat([x, g1, g2, torch.zeros_like(x)], dim=3)  # the shifted input per point, for the first layer's weight gradient
            hs.append(out)
        return hs, y0

    def _backward(self, saved, logp, d_logp, last_dev, weights, grads):
        hs, bh, y0, _, _ = saved
        dz_top, key = self._readout_grad(hs[-1], bh, logp, d_logp, last_dev, weights, grads)
        L = len(hs) - 1
        dz = dz_top
        for i in reversed(range(L)):
            g1 = self._shift(self.op_T, dz) if not (i == 0 and y0 is not None) else None
            served, dx = (self.op_T.backward_power(dz, g1, weights[3 * i:3 * i + 3], hs[i], self.act, i > 0,
                                                   grads[3 * i:3 * i + 3])
                          if (g1 is not None and hs[i].shape[3] == dz.shape[3]) else (False, None))
            if not served and i == 0 and hs[0].shape[3] == 1 and y0 is not None \
                    and self.op.dw_first(None, y0, dz, grads[0:3]):
                dx = None                               # first layer: dW_k = sum_p (S^k x)[p] dz[p], one stream over dz
            elif not served:
                g1 = self._shift(self.op_T, dz) if g1 is None else g1
                g2 = self._shift(self.op_T, g1)
                dx = dense_terms_backward([dz, g1, g2], weights[3 * i:3 * i + 3], hs[i], self.act, i > 0,
                                          grads[3 * i:3 * i + 3])
            if i == L - 1:
                self._release_top(dz_top, key, last_dev)
            dz = dx
        return grads


class _SconeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x, last_dev, *weights):
        logp, saved = plan.forward(x, last_dev, weights)
        ctx.plan, ctx.saved, ctx.last_dev = plan, saved, last_dev
        ctx.save_for_backward(logp, *weights)
        return logp

    @staticmethod
    def backward(ctx, d_logp):
        logp, *weights = ctx.saved_tensors
        grads = [torch.zeros_like(w) for w in weights]
        ctx.plan.backward(ctx.saved, logp, d_logp, ctx.last_dev, weights, grads)
        ctx.saved = None
        return (None, None, None, *grads)


# ----------------------------------------------------------------------------------------------
# bunch
# ----------------------------------------------------------------------------------------------

BUNCH_SRC = [0, 1, 0, 1, 2, 1, 2]      # input level of weight slot k   (TE:184-192)
BUNCH_DST = [0, 0, 1, 1, 1, 2, 2]      # output level of weight slot k


class BunchPlan:
    """Device state of the Bunch (SCCONV) model.  Each of the seven shifts S_k (and its transpose for the backward) is a
    single-operator ConvOp served by the LDS-blocked SpMM; the per-level sum over shifts, the weights and the relu are
    one dense kernel (scn_dense_terms_*).  Shapes the blocked SpMM does not take (ns*C > 128) fall back to the generic
    multi-group kernels (one operator per output / input level)."""

    def __init__(self, shifts, nbrhoods, device):
        assert len(shifts) == 7 and all(isinstance(s, Shift) for s in shifts)
        self.layout = shifts[0].layout
        self.device = device
        self.sizes = self.layout.sizes
        self.n_edges = int(self.sizes[1])
        dev = [s.device_csr() for s in shifts]
        hints = self.layout.block_starts
        # a level without simplices (a complex without faces: B2 has no columns, BMM:71-135 still yields the seven shapes) carries
        # nothing: its shifts are empty operators, its tensors have no rows -- the shifts that touch it are left out altogether
        self._live = [self.sizes[BUNCH_SRC[k]] > 0 and self.sizes[BUNCH_DST[k]] > 0 for k in range(7)]
        self.term_fwd = [ConvOp(self.sizes[BUNCH_DST[k]], [{"mats": [dev[k]], "identity": False,
                                                           "n_cols": self.sizes[BUNCH_SRC[k]]}], hints[BUNCH_DST[k]])
                         if self._live[k] else None for k in range(7)]
        self.term_bwd = [ConvOp(self.sizes[BUNCH_SRC[k]], [{"mats": [dev[k].T.tocsr()], "identity": False,
                                                           "n_cols": self.sizes[BUNCH_DST[k]]}], hints[BUNCH_SRC[k]])
                         if self._live[k] else None for k in range(7)]
        self.fwd_slots = [[k for k in range(7) if BUNCH_DST[k] == lvl and self._live[k]] for lvl in range(3)]
        self.bwd_slots = [[k for k in range(7) if BUNCH_SRC[k] == lvl and self._live[k]] for lvl in range(3)]
        self._dev_csr = dev
        self._generic = None
        self._terms = None                              # fused-layer operators (forward, transposed), built on first use
        self._terms_nf = None                           # forward operator of a layer whose face output nothing reads
        nb = np.asarray(nbrhoods)
        pn = self.layout.perm[0]
        # padding index -1 wraps to the LAST node of the caller's numbering (TE:201); resolve it here, in device order
        nbd = np.where(nb >= 0, pn[np.maximum(nb, 0)], pn[self.sizes[0] - 1])
        self.nbr = torch.from_numpy(np.ascontiguousarray(nbd, np.int32)).to(device)
        self.max_deg = nb.shape[1]

    def _generic_ops(self):
        if self._generic is None:
            dev = self._dev_csr
            fwd = [ConvOp(self.sizes[lvl], [{"mats": [dev[k]], "identity": False, "n_cols": self.sizes[BUNCH_SRC[k]]}
                                            for k in self.fwd_slots[lvl]]) for lvl in range(3)]
            bwd = [ConvOp(self.sizes[lvl], [{"mats": [dev[k].T.tocsr()], "identity": False,
                                             "n_cols": self.sizes[BUNCH_DST[k]]} for k in self.bwd_slots[lvl]])
                   for lvl in range(3)]
            self._generic = (fwd, bwd)
        return self._generic

    def _terms_ops(self):
        """The seven shifts as one operator on the concatenated row space, and its transpose (scn_terms_*): the fused layer.
        None when the plan builder cannot hold the complex (SCN_ERR_UNSUPPORTED: a single concatenated row with more than
        112 distinct sources, e.g. a hub node) -- the per-shift path then carries every layer, forward AND backward."""
        if self._terms is None and not all(self._live):
            self._terms = False                          # an empty level: the per-shift path (nothing to fuse across three levels)
        if self._terms is None:
            dev = self._dev_csr
            try:
                # bins: rows of (nodes, edges, faces) a block holds = the natural 0.37 : 1 : 0.67 proportions in wave units
                fwd = TermsOp(self.sizes, {(BUNCH_DST[k], BUNCH_SRC[k]): dev[k] for k in range(7)}, self.layout.merged,
                              (12, 32, 20), 4)
                bwd = TermsOp(self.sizes, {(BUNCH_SRC[k], BUNCH_DST[k]): dev[k].T.tocsr() for k in range(7)}, self.layout.merged,
                              (16, 32, 16), 8)
                self._terms = (fwd, bwd)
            except _lib.SconeHipError as e:
                if e.status != _lib.SCN_ERR_UNSUPPORTED:
                    raise
                self._terms = False
        return self._terms or None

    def _terms_fwd_for(self, want):
        """Forward fused-layer operator for the wanted output levels: when the faces are not wanted (the layer before the last
        one: TE:198-201 reads only the nodes of the last layer, which the faces do not feed) a plan whose blocks hold node and
        edge rows only -- the face rows' share of every block goes to them (fewer blocks, fewer staged sources per output row)."""
        fwd = self._terms_ops()[0]
        if want[2] or not (want[0] and want[1]):
            return fwd
        if self._terms_nf is None:
            dev = self._dev_csr
            blocks = {(BUNCH_DST[k], BUNCH_SRC[k]): dev[k] for k in range(7) if BUNCH_DST[k] != 2}
            try:
                self._terms_nf = TermsOp(self.sizes, blocks, self.layout.merged, (16, 48, 0), 4)
            except _lib.SconeHipError as e:
                if e.status != _lib.SCN_ERR_UNSUPPORTED:
                    raise
                self._terms_nf = False
        return self._terms_nf or fwd

    def _terms_bwd_for(self, dz_present):
        """Transposed fused-layer operator for the levels whose gradient exists: when the faces carry none (the layer before the
        last one: its face output was never computed) the two shifts INTO the faces are left out of the operator, so no block
        spends staged sources on rows that are never staged."""
        bwd = self._terms_ops()[1]
        if dz_present[2] or not (dz_present[0] and dz_present[1]):
            return bwd
        if getattr(self, "_terms_bwd_nf", None) is None:
            dev = self._dev_csr
            blocks = {(BUNCH_SRC[k], BUNCH_DST[k]): dev[k].T.tocsr() for k in range(7) if BUNCH_DST[k] != 2}
            try:
                self._terms_bwd_nf = TermsOp(self.sizes, blocks, self.layout.merged, (16, 32, 16), 8)
            except _lib.SconeHipError as e:
                if e.status != _lib.SCN_ERR_UNSUPPORTED:
                    raise
                self._terms_bwd_nf = False
        return self._terms_bwd_nf or bwd

    def _fused_ok(self, ns, widths_out, widths_in):
        return (FUSE_BUNCH and ns == NS and set(widths_out) == {32} and set(widths_in) == {32} and self._terms_ops() is not None)

    @staticmethod
    def _slot(dst, src):
        return next((k for k in range(7) if BUNCH_DST[k] == dst and BUNCH_SRC[k] == src), None)

    @staticmethod
    def _blocked_ok(ns, c):
        return (ns * c) % 4 == 0                        # (any width: operands wider than 128 columns go in channel blocks)

    def _spmm(self, op, x):
        return spmm_chunked(op, x)

    def conv_stack(self, x, weights):
        n_layers = len(weights) / 7
        assert n_layers % 1 == 0, "wrong number of weights"                    # TE:177-178
        S, E, ns, _ = x.shape
        cur = [torch.zeros((S, self.sizes[0], ns, 1), device=x.device), x,
               torch.zeros((S, self.sizes[2], ns, 1), device=x.device)]         # TE:179
        zero = [True, False, True]          # levels known to be identically zero (no bias terms: zeros stay zeros)
        # only the node level of the last layer is read (TE:198-201): walk the seven shifts backwards to find which level
        # outputs can reach it, and leave the others uncomputed (their gradient is identically zero as well)
        L = int(n_layers)
        need = [[False] * 3 for _ in range(L + 1)]
        need[L][0] = True
        for i in range(L - 1, 0, -1):
            for k in range(7):
                if self._live[k] and need[i + 1][BUNCH_DST[k]]:
                    need[i][BUNCH_SRC[k]] = True
        states, zeros = [cur], [zero]
        first_g = {}
        self._fold = None
        if self._fold_ok(x, weights, L, ns):
            cur, zero = self._fold_forward(x, weights, need[2])
            states += [[None] * 3, cur]                     # the first layer's output is never materialised
            zeros += [[False] * 3, zero]
        for i in range(2 if self._fold is not None else 0, L):
            nxt, nzero = [], []
            c_outs = {weights[7 * i + k].shape[1] for k in range(7)}
            c_ins = {cur[l].shape[3] for l in range(3) if not zero[l] and cur[l] is not None}
            if self._fused_ok(ns, c_outs, c_ins):
                # fused layer: one launch for the three levels (scn_terms_forward)
                xs = [None if (zero[l] or cur[l] is None) else cur[l] for l in range(3)]
                Ws = [[None] * 3 for _ in range(3)]
                for k in range(7):
                    if xs[BUNCH_SRC[k]] is not None:
                        Ws[BUNCH_DST[k]][BUNCH_SRC[k]] = weights[7 * i + k]
                outs = self._terms_fwd_for(need[i + 1]).forward(xs, Ws, "relu", need[i + 1])
                cur, zero = outs, [o is None for o in outs]
                states.append(cur)
                zeros.append(zero)
                continue
            for lvl in range(3):
                if not need[i + 1][lvl]:
                    nxt.append(None)
                    nzero.append(True)          # nothing downstream that matters reads it: treated like a zero level
                    continue
                ks = [k for k in self.fwd_slots[lvl] if not zero[BUNCH_SRC[k]]]
                c_out = weights[7 * i + self.fwd_slots[lvl][0]].shape[1]
                if not ks:
                    nxt.append(torch.zeros((S, self.sizes[lvl], ns, c_out), device=x.device))
                    nzero.append(True)
                    continue
                Ws = [weights[7 * i + k] for k in ks]
                if (FUSE_BUNCH and c_out == 1 and all(cur[BUNCH_SRC[k]].shape[3] > 1 for k in ks)
                        and self._blocked_ok(ns, 1)):
                    # one output channel (the last layer): project FIRST, then shift -- (S_k X_k) W_k = S_k (X_k W_k): every
                    # input level is read once and the shifts run on 1-channel tensors
                    ys = [dense_terms_forward([cur[BUNCH_SRC[k]]], [w], 1, "none") for k, w in zip(ks, Ws)]
                    Gs = [self._spmm(self.term_fwd[k], y) for k, y in zip(ks, ys)]
                    one = torch.ones((1, 1), device=x.device, dtype=torch.float32)
                    nxt.append(dense_terms_forward(Gs, [one] * len(Gs), 1, "relu"))
                    nzero.append(False)
                    continue
                if all(self._blocked_ok(ns, cur[BUNCH_SRC[k]].shape[3]) for k in ks):
                    Gs = [self._spmm(self.term_fwd[k], cur[BUNCH_SRC[k]]) for k in ks]
                    if i == 0 and x.shape[3] == 1:          # the shifted 1-channel input S_k x: all the first layer's weight
                        first_g.update(zip(ks, Gs))         # gradient needs (dW_k = sum_p (S_k x)[p] dz[p], one stream over dz)
                    nxt.append(dense_terms_forward(Gs, Ws, c_out, "relu"))
                else:
                    fwd, _ = self._generic_ops()
                    allk = self.fwd_slots[lvl]
                    nxt.append(fwd[lvl].forward([cur[BUNCH_SRC[k]] for k in allk], [weights[7 * i + k] for k in allk],
                                                c_out, "relu"))
                nzero.append(False)
            cur, zero = nxt, nzero
            states.append(cur)
            zeros.append(zero)
        self._zeros = zeros
        self._first_g = first_g
        return states

    # -- the first TWO layers without a 32-channel gather --------------------------------------------------------------
    # bunch_func starts from [0, flow, 0] (TE:179): after the first layer every level is relu of ONE rank-one term,
    # H1_j = relu(g_j (x) w_j) with g_j = S_k1 x one channel wide (k1 = the slot that feeds level j from the edges), and
    #     relu(g w) = max(g, 0) relu(w) + min(g, 0) min(w, 0)
    # makes the second layer's pre-activation a sum of rank-one terms of SHIFTED SCALARS:
    #     (S_k H1_j) W2_k = (S_k g_j^+) (x) (relu(w_j) W2_k)  +  (S_k g_j^-) (x) (min(w_j, 0) W2_k).
    # So H1 is never formed: seven shifts of two one-channel tensors each (scn_spmm_dual), one rank-one expansion per level
    # (scn_dense_terms_forward) -- and in the backward ONE stream over the second layer's pre-activation gradient per level
    # (u_k^+- = sum_p (S_k g^+-)[p] dZ2[p][:], scn_dense_terms_backward) from which both layers' weight gradients follow by
    # scn_fold1_backward.  Same sums as TE:183-195 in another association order; relu'(0) = 0 as everywhere here.
    def _fold_ok(self, x, weights, L, ns):
        if not (FOLD_BUNCH and FUSE_BUNCH and L >= 3 and x.shape[3] == 1 and self._blocked_ok(ns, 1) and all(self._live)):
            return False
        c1 = {weights[k].shape[1] for k in range(7)}
        c2 = {weights[7 + k].shape[1] for k in range(7)}
        return len(c1) == 1 and len(c2) == 1 and next(iter(c2)) in (16, 32, 64) and all(weights[k].shape[0] == 1 for k in range(7))

    def _fold_forward(self, x, weights, need2):
        lib = _lib.load()
        c1, c2 = weights[0].shape[1], weights[7].shape[1]
        pm = {}                                            # level j -> (g^+, g^-, first-layer slot k1)
        for k1 in range(7):
            if BUNCH_SRC[k1] != 1:
                continue                                   # the node and face levels start at zero: their slots contribute nothing
            g = self._spmm(self.term_fwd[k1], x)
            S = g.shape[0]
            gpm = torch.empty((2 * S,) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)   # [g^+ slabs | g^- slabs]
            check(lib.scn_split_sign(g.numel(), _dev(g), _dev(gpm[:S]), _dev(gpm[S:]), _stream()), "scn_split_sign")
            pm[BUNCH_DST[k1]] = (gpm, k1)
        terms, outs = {}, []
        for lvl in range(3):
            if not need2[lvl]:
                outs.append(None)
                continue
            Gs, Ws = [], []
            for k2 in self.fwd_slots[lvl]:
                gpm, k1 = pm[BUNCH_SRC[k2]]
                spm = self._spmm(self.term_fwd[k2], gpm)   # both parts in one launch: they are just more slabs of a one-channel tensor
                sp, sm = spm[:spm.shape[0] // 2], spm[spm.shape[0] // 2:]
                ap = torch.empty((1, c2), device=x.device, dtype=torch.float32)
                am = torch.empty_like(ap)
                check(lib.scn_fold1_forward(_dev(weights[k1]), _dev(weights[7 + k2]), c1, c2, _dev(ap), _dev(am), _stream()),
                      "scn_fold1_forward")
                terms[k2] = (sp, sm, k1)
                Gs += [sp, sm]
                Ws += [ap, am]
            outs.append(dense_terms_forward(Gs, Ws, c2, "relu"))
        self._fold = terms
        return outs, [o is None for o in outs]

    def _fold_backward(self, fold, dz, dzero, weights, grads):
        """Weight gradients of the first two layers from dz = the gradient of the second layer's pre-activation per level."""
        lib = _lib.load()
        c1, c2 = weights[0].shape[1], weights[7].shape[1]
        for lvl in range(3):
            if dzero[lvl] or dz[lvl] is None:
                continue
            ks = [k2 for k2 in self.fwd_slots[lvl] if k2 in fold]
            Gs, us = [], []
            for k2 in ks:
                Gs += [fold[k2][0], fold[k2][1]]
                us += [torch.zeros((c2, 1), device=dz[lvl].device, dtype=torch.float32) for _ in range(2)]
            dense_terms_backward(Gs, us, dz[lvl], "none", False, us)       # us[q][c] = sum_p Gs[q][p] dz[p][c]  (W unused: no dx)
            for q, k2 in enumerate(ks):
                k1 = fold[k2][2]
                check(lib.scn_fold1_backward(_dev(weights[k1]), _dev(weights[7 + k2]), _dev(us[2 * q]), _dev(us[2 * q + 1]), c1, c2,
                                             _dev(grads[7 + k2]), _dev(grads[k1]), _stream()), "scn_fold1_backward")

    def promotion(self, weights):
        if len(weights) < 14 or len(weights) % 7:
            return None
        return promoted_width([w.shape[1] for w in weights[:-7]], wide=True)   # above 32: multiples of 32 (32 x 32 MFMA blocks of the dense terms)

    def forward(self, x, last_dev, weights):
        lib = _lib.load()
        P = self.promotion(weights)
        wp = promote_weights(weights, 7, 7, P) if P else None
        states = self.conv_stack(x, wp if P else weights)
        nodes_out = states[-1][0]
        S, V, ns, C = nodes_out.shape
        assert C == 1, "bunch readout needs one output channel (TE:198-201)"
        logits = torch.empty((S * ns, self.max_deg), device=x.device, dtype=torch.float32)
        logp = torch.empty_like(logits)
        check(lib.scn_node_readout_forward(S, ns, V, _dev(nodes_out), _dev(self.nbr, torch.int32), self.max_deg,
                                           _dev(last_dev, torch.int32), _dev(logits), _dev(logp), _stream()),
              "scn_node_readout_forward")
        return logp, (states, self._zeros, self._first_g, wp, self._fold)

    def backward(self, saved, logp, d_logp, last_dev, weights, grads):
        wp = saved[3]
        if wp is not None:                              # promoted widths (see promote_weights)
            gp = [torch.zeros_like(w) if w is not w0 else g for w, w0, g in zip(wp, weights, grads)]
            self._backward(saved, logp, d_logp, last_dev, wp, gp)
            demote_grads(grads, gp)
            return grads
        return self._backward(saved, logp, d_logp, last_dev, weights, grads)

    def _backward(self, saved, logp, d_logp, last_dev, weights, grads):
        lib = _lib.load()
        states, zeros, first_g, _, fold = saved
        nodes_out = states[-1][0]
        S, V, ns, _ = nodes_out.shape
        dz = [torch.empty_like(nodes_out), None, None]
        dzero = [False, True, True]             # only the node level feeds the loss (TE:198)
        check(lib.scn_node_readout_backward(S, ns, V, _dev(nodes_out), _dev(self.nbr, torch.int32), self.max_deg,
                                            _dev(last_dev, torch.int32), _dev(d_logp.contiguous()), _dev(logp),
                                            ACT["relu"], _dev(dz[0]), _stream()), "scn_node_readout_backward")
        L = len(states) - 1
        for i in reversed(range(L)):
            if i == 1 and fold is not None:             # the first two layers: no input gradients, one stream over dz per level
                self._fold_backward(fold, dz, dzero, weights, grads)
                break
            x, xzero = states[i], zeros[i]
            new_dz, new_zero = [None, None, None], [True, True, True]
            c_dz = {dz[l].shape[3] for l in range(3) if not dzero[l]}
            c_x = {x[l].shape[3] for l in range(3) if not xzero[l] and x[l] is not None}
            if self._fused_ok(ns, c_dz, c_x):
                # fused layer backward on the transposed operator (scn_terms_backward): rows = this layer's input rows
                dzs = [None if dzero[l] else dz[l] for l in range(3)]
                auxs = [None if (xzero[l] or x[l] is None) else x[l] for l in range(3)]
                Ws = [[None] * 3 for _ in range(3)]
                dWs = [[None] * 3 for _ in range(3)]
                for k in range(7):
                    a, b = BUNCH_SRC[k], BUNCH_DST[k]
                    if auxs[a] is not None and dzs[b] is not None:
                        Ws[a][b], dWs[a][b] = weights[7 * i + k], grads[7 * i + k]
                want = [i > 0 and auxs[l] is not None and any(w is not None for w in Ws[l]) for l in range(3)]
                # the layer after the 1-channel first layer: its input gradients feed the first layer's weight gradient and nothing
                # else -- contracted with the shifted input S_k x inside the kernel, never written (dW_k[0][c] = sum_p (S_k x)[p] dx[p][c])
                k_of = {BUNCH_DST[k]: k for k in first_g}
                if (i == 1 and FUSE_FIRST and first_g and all(not want[l] or l in k_of for l in range(3))
                        and all(first_g[k].shape[3] == 1 and weights[k].shape == (1, 32) for k in first_g)):
                    ys = [first_g[k_of[l]].contiguous() if (want[l] and l in k_of) else None for l in range(3)]
                    dWf = [grads[k_of[l]] if ys[l] is not None else None for l in range(3)]
                    _terms_backward_first(self._terms_ops()[1], dzs, Ws, auxs, "relu", ys, dWs, dWf)
                    break
                dxs = _terms_backward(self._terms_bwd_for([d is not None for d in dzs]), dzs, Ws, auxs, "relu", want, dWs)
                dz, dzero = dxs, [d is None for d in dxs]
                continue
            if i == 0 and FUSE_BUNCH and first_g and all(x[l] is None or x[l].shape[3] == 1 for l in range(3)):
                # first layer (one 1-channel input level): the shift sits on the 1-channel side, dW_k[0][c] = sum_p (S_k x)[p] dz[p][c]
                # -- dz of every level streamed once, no transposed SpMM, no gathered 32-channel tensor (scn_conv_dw_first)
                live = [(k, g, dz[BUNCH_DST[k]]) for k, g in first_g.items()
                        if not dzero[BUNCH_DST[k]] and dz[BUNCH_DST[k]] is not None]
                # all or nothing: the per-level loop below accumulates EVERY shift's gradient, so this path may only touch
                # `grads` when it serves all of them (every shift's operator builds its block plan on its own)
                if all(self.term_fwd[k].dw_first_served(d) for k, _, d in live):
                    for k, g, d in live:
                        y = torch.zeros((g.shape[0], g.shape[1], g.shape[2], Y_STRIDE), device=g.device, dtype=torch.float32)
                        y[..., 0] = g[..., 0]
                        dummy = [torch.zeros_like(grads[k]) for _ in range(2)]
                        served = self.term_fwd[k].dw_first(None, y, d, [grads[k]] + dummy)
                        assert served
                    break
            for lvl in range(3):
                ks = [k for k in self.bwd_slots[lvl] if not dzero[BUNCH_DST[k]]]
                if not ks or xzero[lvl]:
                    # a zero input level receives no weight gradient; its own gradient is only needed below layer 0
                    if i > 0 and ks and xzero[lvl]:
                        raise AssertionError("zero level above the first layer")
                    continue
                Ws = [weights[7 * i + k] for k in ks]
                dWs = [grads[7 * i + k] for k in ks]
                if all(self._blocked_ok(ns, dz[BUNCH_DST[k]].shape[3]) for k in ks):
                    Gs = [self._spmm(self.term_bwd[k], dz[BUNCH_DST[k]]) for k in ks]
                    new_dz[lvl] = dense_terms_backward(Gs, Ws, x[lvl], "relu", i > 0, dWs)
                else:
                    _, bwd = self._generic_ops()
                    allk = self.bwd_slots[lvl]
                    dzs = [dz[BUNCH_DST[k]] if not dzero[BUNCH_DST[k]]
                           else torch.zeros((S, self.sizes[BUNCH_DST[k]], ns, weights[7 * i + k].shape[1]), device=x[lvl].device)
                           for k in allk]
                    new_dz[lvl] = bwd[lvl].backward(dzs, [weights[7 * i + k] for k in allk], x[lvl], "relu", i > 0,
                                                    [grads[7 * i + k] for k in allk])
                new_zero[lvl] = False
            dz, dzero = new_dz, new_zero
        return grads


class _BunchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x, last_dev, *weights):
        logp, states = plan.forward(x, last_dev, weights)
        ctx.plan, ctx.states, ctx.last_dev = plan, states, last_dev
        ctx.save_for_backward(logp, *weights)
        return logp

    @staticmethod
    def backward(ctx, d_logp):
        logp, *weights = ctx.saved_tensors
        grads = [torch.zeros_like(w) for w in weights]
        ctx.plan.backward(ctx.states, logp, d_logp, ctx.last_dev, weights, grads)
        ctx.states = None
        return (None, None, None, *grads)


# ----------------------------------------------------------------------------------------------
# plan cache + micro-batching
# ----------------------------------------------------------------------------------------------

def get_scone_plan(S_lower, S_upper, bconds, act, device):
    key = ("scone", id(S_upper), id(bconds), act, str(device))
    if key not in S_lower._cache:
        plan = None
        wide = S_upper.csr.nnz and int(np.diff(S_upper.csr.indptr).max()) >= 128      # such a row cannot enter a block
        if wide and _is_square_of(S_upper, S_lower):
            # rows of S^2 with more than 128 sources do not fit the LDS-blocked plan: compose S (S H) instead of fusing S^2
            plan = PowerPlan(S_lower, S_upper, bconds, act, device)
            if plan.op.plan_info()[0] == 0:
                plan = None
        S_lower._cache[key] = plan if plan is not None else SconePlan(S_lower, S_upper, bconds, act, device)
    return S_lower._cache[key]


def _is_square_of(S2, S):
    if S.shape[0] != S.shape[1] or S2.shape != S.shape:
        return False
    d = (S2.csr - S.csr @ S.csr).tocsr()
    scale = max(1.0, float(abs(S2.csr).max()) if S2.csr.nnz else 1.0)
    return d.nnz == 0 or float(abs(d).max()) <= 1e-9 * scale


def get_bunch_plan(shifts, nbrhoods, device):
    key = ("bunch",) + tuple(id(s) for s in shifts[1:]) + (str(device),)
    if key not in shifts[0]._cache:
        shifts[0]._cache[key] = BunchPlan(shifts, nbrhoods, device)
    return shifts[0]._cache[key]


def micro_batch_size(n_rows_total, widths, n, ns=NS, budget_bytes=None, device=None):
    """Trajectories per micro-batch so that saved activations + two gradient buffers fit the budget; the batch is
    cut into equal micro-batches (multiples of ns)."""
    if budget_bytes is None:
        free, total = torch.cuda.mem_get_info(device)
        budget_bytes = 0.35 * total
    per_sample = 4.0 * n_rows_total * (sum(widths) + 2 * max(widths))
    mb_max = int(budget_bytes // max(per_sample, 1.0))
    mb_max = max(ns, min(mb_max, 16384) // ns * ns)
    if mb_max >= 64:
        mb_max = mb_max // 64 * 64          # whole groups of 16 slabs: what the kernels' slab grouping and grid splits are tuned for
    n_pad = pad_count(n, ns)
    n_chunks = -(-n_pad // mb_max)
    return pad_count(-(-n_pad // n_chunks), ns)


def as_device_weights(weights, device):
    out = []
    for w in weights:
        t = w if torch.is_tensor(w) else torch.as_tensor(np.asarray(w))
        if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=device, dtype=torch.float32).contiguous()
        out.append(t)
    return out


def default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("scone_gcn_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
