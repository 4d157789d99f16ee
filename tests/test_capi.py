"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol include/*.h declares.
No compute call is made here (that is the -m gpu suite)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "scone_gcn_amd", "libscone_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["bash", os.path.join(ROOT, "scone_gcn_amd", "csrc", "build.sh")])
    return ctypes.CDLL(LIB)


def _declared():
    names = set()
    for f in os.listdir(os.path.join(ROOT, "include")):
        if f.endswith(".h"):
            src = open(os.path.join(ROOT, "include", f)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(scn_[a-z0-9_]+)\s*\(", src))
    return names


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 16
    for n in sorted(names):
        assert hasattr(lib, n), "missing export " + n


def test_python_binding_covers_the_header():
    from scone_gcn_amd import _lib
    assert set(_lib.SIGNATURES) == _declared()


def test_version_and_error_strings(lib):
    lib.scn_version.restype = ctypes.c_int
    lib.scn_error_string.restype = ctypes.c_char_p
    assert lib.scn_version() >= 100
    assert lib.scn_error_string(0) == b"ok"
    assert b"shape" in lib.scn_error_string(-2) and b"unsupported" in lib.scn_error_string(-4)


def test_bad_arguments_are_rejected_without_touching_the_gpu(lib):
    h = ctypes.c_void_p()
    assert lib.scn_conv_create(0, 1, None, ctypes.byref(h)) == -1      # SCN_ERR_BAD_ARG
    assert lib.scn_conv_forward(None, 1, 4, None, None, None, 16, 1, None, None) == -1
    assert lib.scn_adam_step(ctypes.c_int64(0), None, None, None, None, ctypes.c_float(1e-3), ctypes.c_float(.9),
                             ctypes.c_float(.999), ctypes.c_float(1e-8), 0, ctypes.c_float(0), ctypes.c_float(1),
                             None) == -2                                # SCN_ERR_BAD_SHAPE
    f = ctypes.c_float
    dev_args = (f(1e-3), f(.9), f(.999), f(1e-8))
    assert lib.scn_adam_step_dev(ctypes.c_int64(0), None, None, None, None, *dev_args, None, f(0), f(1), None) == -2
    assert lib.scn_adam_step_dev(ctypes.c_int64(8), None, None, None, None, *dev_args, None, f(0), f(1), None) == -1
    assert lib.scn_adam_step_dev(ctypes.c_int64(65537), None, None, None, None, *dev_args, None, f(0), f(1), None) == -4   # one workgroup's worth
    assert lib.scn_conv_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from scone_gcn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_model_functions_refuse_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from scone_gcn_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.default_device()


def test_header_is_plain_c(tmp_path):
    """include/scone_hip.h is the drop-in boundary: it has to compile as C99 without any C++ or HIP header."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "scone_hip.h"\nint main(void) { scn_work_list w = {0, 0, 0, 0}; (void)w; return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-c", str(src), "-o", str(tmp_path / "hdr.o")])


def test_compiler_never_touches_m0_in_kernels_with_inline_assembly_lds_dma(tmp_path):
    """The fused kernels issue their LDS-DMA as inline assembly that writes M0 (csrc/scn_blk_common.inc, lds_dma16) without
    telling the compiler (M0 is a reserved register: it cannot be listed as clobbered).  That is only sound while the compiler
    itself has no use for M0 in those kernels: compile the translation unit to device assembly and check every kernel that
    contains such a block for M0 outside of the blocks (kernels on the builtin LDS-DMA use M0 themselves and must have none)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "scn_blocked.s"
    csrc = os.path.join(ROOT, "scone_gcn_amd", "csrc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           "-S", "--cuda-device-only", "-o", str(out), os.path.join(csrc, "scn_blocked.hip")],
                          stderr=subprocess.DEVNULL)
    asm = out.read_text()
    kernels = re.split(r"\n(?=_ZN3scn[^\n]*:\s*; @)", asm)
    with_blocks = 0
    for k in kernels:
        m = re.match(r"(_ZN3scn\S+):", k)
        if not m:
            continue
        body = k.split("s_endpgm")[0]
        outside = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", body, flags=re.S)
        inside = len(re.findall(r"\bm0\b", body)) - len(re.findall(r"\bm0\b", outside))
        if inside:
            with_blocks += 1
            assert not re.search(r"\bm0\b", outside), m.group(1) + ": the compiler uses M0 next to an inline-assembly LDS-DMA"
    assert with_blocks >= 20          # forward, backward, fused Bunch and ring SpMM instantiations
