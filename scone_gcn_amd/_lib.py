"""ctypes binding of libscone_hip.so (the C-ABI declared in include/scone_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, this raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or scone_gcn_amd/csrc/build.sh.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCN_LIB_PATH") or os.path.join(_HERE, "libscone_hip.so")   # override: diagnostic builds only

c_i32, c_i64, c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
c_void_p, c_size_t = ctypes.c_void_p, ctypes.c_size_t
P_i32 = ctypes.POINTER(c_i32)
P_f32 = ctypes.POINTER(c_f32)

SCN_MAX_GROUPS = 3
SCN_MAX_SLOTS = 4
ACT = {"none": 0, "tanh": 1, "relu": 2, "leaky_relu": 3}
SCN_ERR_BAD_ARG = -1              # include/scone_hip.h
SCN_ERR_UNSUPPORTED = -4


class WorkListDesc(ctypes.Structure):          # scn_work_list (device pointers)
    _fields_ = [("n_work", ctypes.c_int32), ("block", ctypes.c_void_p), ("ptr", ctypes.c_void_p), ("slab", ctypes.c_void_p)]


class GroupDesc(ctypes.Structure):
    _fields_ = [("n_cols", c_i32), ("identity", c_i32), ("n_vals", c_i32), ("reserved", c_i32),
                ("nnz", c_i64), ("rowptr", c_void_p), ("col", c_void_p), ("val0", c_void_p), ("val1", c_void_p)]


# name -> (restype, argtypes); every name here must be exported by the library and declared in the header
SIGNATURES = {
    "scn_version": (ctypes.c_int, []),
    "scn_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "scn_last_hip_error": (ctypes.c_char_p, []),
    "scn_conv_create": (ctypes.c_int, [c_i32, c_i32, ctypes.POINTER(GroupDesc), ctypes.POINTER(c_void_p)]),
    "scn_conv_create_blocked": (ctypes.c_int, [c_i32, c_i32, ctypes.POINTER(GroupDesc), c_void_p, ctypes.POINTER(c_void_p)]),
    "scn_conv_destroy": (ctypes.c_int, [c_void_p]),
    "scn_conv_n_slots": (ctypes.c_int, [c_void_p]),
    "scn_conv_plan_info": (ctypes.c_int, [c_void_p, P_i32, P_f32]),
    "scn_conv_forward": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                        ctypes.POINTER(c_void_p), c_i32, c_i32, c_void_p, c_void_p]),
    "scn_conv_backward_workspace": (c_size_t, [c_void_p, c_i32, c_i32, P_i32, c_i32]),
    "scn_conv_backward": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                         ctypes.POINTER(c_void_p), c_void_p, c_i32, c_i32, c_void_p,
                                         ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "scn_spmm_dual": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "scn_dense_terms_forward": (ctypes.c_int, [c_i64, c_i32, ctypes.POINTER(c_void_p), P_i32, ctypes.POINTER(c_void_p), c_i32,
                                               c_i32, c_void_p, c_void_p]),
    "scn_dense_terms_backward_workspace": (c_size_t, [c_i64, c_i32, P_i32, c_i32]),
    "scn_dense_terms_backward": (ctypes.c_int, [c_i64, c_i32, ctypes.POINTER(c_void_p), P_i32, ctypes.POINTER(c_void_p),
                                                c_void_p, c_i32, c_i32, c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                                c_size_t, c_void_p]),
    "scn_readout_forward": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "scn_readout_backward": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i32, c_i32,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_void_p]),
    "scn_readout_clear_dz": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32, c_void_p, c_i32, c_i32, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "scn_node_readout_forward": (ctypes.c_int, [c_i32, c_i32, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
                                                c_void_p, c_void_p]),
    "scn_node_readout_backward": (ctypes.c_int, [c_i32, c_i32, c_i32, c_void_p, c_void_p, c_i32, c_void_p, c_void_p,
                                                 c_void_p, c_i32, c_void_p, c_void_p]),
    "scn_logits_sum_log_softmax": (ctypes.c_int, [c_i32, c_i32, c_i32, ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    "scn_host_stage_batch": (c_i64, [c_i32, c_void_p, c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32, ctypes.c_double, c_i32,
                                     c_i32, c_void_p]),
    "scn_scatter_flows": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]),
    "scn_conv_dw_first_workspace": (c_size_t, [c_void_p, c_i32, c_i32, c_i32]),
    "scn_conv_dw_first": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_i32,
                                         ctypes.POINTER(c_void_p), c_void_p, c_size_t, ctypes.POINTER(WorkListDesc),
                                         c_void_p]),
    "scn_conv_forward_first": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, ctypes.POINTER(c_void_p), c_i32, c_i32,
                                              c_void_p, c_void_p, ctypes.POINTER(WorkListDesc), c_void_p]),
    "scn_conv_plan_blocks": (ctypes.c_int, [c_void_p, P_i32]),
    "scn_conv_forward_power": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_i32,
                                              c_i32, c_void_p, c_void_p]),
    "scn_conv_backward_power_workspace": (c_size_t, [c_void_p, c_i32, c_i32, c_i32]),
    "scn_conv_backward_power": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, c_void_p, ctypes.POINTER(c_void_p),
                                               c_void_p, c_i32, c_i32, c_void_p, ctypes.POINTER(c_void_p), c_void_p,
                                               c_size_t, c_void_p]),
    "scn_conv_forward_accumulate": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                                   ctypes.POINTER(c_void_p), c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "scn_conv_forward_list": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                             ctypes.POINTER(c_void_p), c_i32, c_i32, c_void_p,
                                             ctypes.POINTER(WorkListDesc), c_void_p]),
    "scn_conv_backward_accumulate": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                                    ctypes.POINTER(c_void_p), c_void_p, c_i32, c_i32, c_void_p, c_void_p,
                                                    ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "scn_conv_backward_list": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), P_i32,
                                              ctypes.POINTER(c_void_p), c_void_p, c_i32, c_i32, c_void_p,
                                              ctypes.POINTER(c_void_p), c_void_p, c_size_t,
                                              ctypes.POINTER(WorkListDesc), c_void_p]),
    "scn_clear_list": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, ctypes.POINTER(WorkListDesc), c_void_p]),
    "scn_plan_refine_order": (ctypes.c_int, [c_i32, P_i32, P_i32, c_i32, P_i32, c_void_p]),
    "scn_plan_gather_stats": (ctypes.c_int, [c_i32, P_i32, P_i32, c_void_p, c_i32, c_void_p, c_void_p]),
    "scn_conv_backward_fused_first_workspace": (c_size_t, [c_void_p, c_i32, c_i32, c_i32]),
    "scn_conv_backward_fused_first": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_void_p, ctypes.POINTER(c_void_p), c_void_p, c_i32,
                                                     c_i32, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p,
                                                     c_size_t, ctypes.POINTER(WorkListDesc), c_void_p]),
    "scn_terms_create": (ctypes.c_int, [c_i32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i32,
                                        ctypes.POINTER(c_void_p)]),
    "scn_terms_forward": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_i32, c_i32,
                                         ctypes.POINTER(c_void_p), c_void_p]),
    "scn_terms_backward_workspace": (c_size_t, [c_void_p, c_i32, c_i32, c_i32]),
    "scn_terms_backward": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                          ctypes.POINTER(c_void_p), c_i32, c_i32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                          c_void_p, c_size_t, c_void_p]),
    "scn_terms_backward_fused_first": (ctypes.c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                                      ctypes.POINTER(c_void_p), c_i32, c_i32, ctypes.POINTER(c_void_p),
                                                      ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "scn_split_sign": (ctypes.c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "scn_sum_act": (ctypes.c_int, [c_i64, c_i32, ctypes.POINTER(c_void_p), c_i32, c_void_p, c_void_p]),
    "scn_fold1_forward": (ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "scn_fold1_backward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p, c_void_p]),
    "scn_small_step_supported": (ctypes.c_int, [c_void_p, c_i32, c_i32, c_i32, c_i32]),
    "scn_small_step_workspace": (c_size_t, [c_i32, c_i32, c_i32]),
    "scn_small_step_pairing": (ctypes.c_int, [c_i32]),
    "scn_small_step": (ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_f32, c_void_p,
                                      c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_i32,
                                      ctypes.POINTER(c_void_p), c_void_p, c_i32, c_void_p, c_size_t, c_void_p]),
    "scn_small_step_adam": (ctypes.c_int, [c_void_p, c_void_p, c_i32, c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, c_f32, c_void_p,
                                           c_i32, c_i32, c_i32, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_i32,
                                           ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_size_t,
                                           c_void_p, c_void_p, c_void_p, c_f32, c_f32, c_f32, c_f32, c_void_p, c_f32, c_void_p]),
    "scn_masked_ce": (ctypes.c_int, [c_i64, c_void_p, c_void_p, c_f32, c_void_p, c_void_p, c_void_p]),
    "scn_masked_ce_begin": (ctypes.c_int, [c_i64, c_void_p, c_void_p, c_f32, c_void_p, c_void_p, c_i32, c_void_p, c_i64, c_void_p]),
    "scn_adam_step": (ctypes.c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_f32, c_f32, c_f32, c_f32,
                                     c_i32, c_f32, c_f32, c_void_p]),
    "scn_adam_step_dev": (ctypes.c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_f32, c_f32, c_f32, c_f32,
                                         c_void_p, c_f32, c_f32, c_void_p]),
}

_lib = None
AB_OLDER_LIBRARY = False              # set by tools/ab_run.py: tolerate entry points a build from an older revision does not have


class SconeHipError(RuntimeError):
    status = 0                        # the C-ABI status code (include/scone_hip.h)


def load():
    """Load the library (once).  Raises ImportError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: its bundled libamdhip64.so.7 then serves both torch and this library, so the
    # device pointers and streams torch hands us belong to the same HIP runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run scone_gcn_amd/csrc/build.sh "
            f"(or __graft_entry__.build()). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if AB_OLDER_LIBRARY and not hasattr(lib, name):
            continue                  # tools/ab_run.py only: an older build of the library as the "before" side of an A/B
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.scn_error_string(status).decode()
        hip = lib.scn_last_hip_error().decode()
        err = SconeHipError(f"{what} failed: {msg} (status {status})" + (f" [{hip}]" if status == -3 and hip else ""))
        err.status = status
        raise err


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


def i32_array(vals):
    arr = (c_i32 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


def sources_sha():
    """SHA-256 over the kernel sources (csrc/*.hip, *.inc, *.h, in name order): the stamp a committed counter profile carries
    (tools/pmc_merge.py) and bench.py compares before citing measured HBM bytes for a kernel."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.inc")) + glob.glob(os.path.join(src, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()
