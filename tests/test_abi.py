"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "streamchat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hip_lib):
    names = declared_symbols()
    assert "sc_kmeans_fit" in names and "sc_abi_version" in names
    raw = ctypes.CDLL(os.path.join(ROOT, "streamchat_amd", "libstreamchat_hip.so"))
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, f"declared in include/streamchat_hip.h but not exported: {missing}"


def test_binding_covers_header(hip_lib):
    from streamchat_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_abi_version_and_error_text(hip_lib):
    assert hip_lib.sc_abi_version() == 8
    assert b"abi=8" in hip_lib.sc_build_info() and b"attention=" in hip_lib.sc_build_info()
    # argument validation happens before any device work, so it is callable without a GPU
    rc = hip_lib.sc_kmeans_fit(None, 0, 1, 8, 1, None, None, None, 0, 1, 1e-4, None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"null pointer" in hip_lib.sc_last_error()
    assert hip_lib.sc_kmeans_workspace_bytes(400, 576 * 3584, 5) > 0


def test_abi_version_matches_header_and_changelog_names_every_symbol_added_since_v2(hip_lib):
    src = open(os.path.join(ROOT, "include", "streamchat_hip.h")).read()
    ver = int(re.search(r"#define\s+SC_ABI_VERSION\s+(\d+)", src).group(1))
    assert hip_lib.sc_abi_version() == ver
    log = src[src.index("ABI changelog"):src.index("#define SC_ABI_VERSION")]
    for name in ("sc_kmeans_update", "sc_decode_qkv_f16", "sc_pick_token_f32", "sc_sample_token_f32", "sc_attention_variant"):
        assert name in log and name in declared_symbols()
    # ABI 7: the process-wide CU budget is gone from the header, the library and the binding; the INTEGRATION.md stub pins the SAME literal
    assert "sc_set_cu_budget" not in declared_symbols() and not hasattr(hip_lib, "sc_set_cu_budget")
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert int(re.search(r"^SC_ABI_VERSION\s*=\s*(\d+)", integ, re.M).group(1)) == ver
    assert "sc_abi_version() == SC_ABI_VERSION" in integ
    from streamchat_amd import _lib
    assert _lib.ABI_VERSION == ver


def test_headed_gemm_says_unsupported_before_any_device_work(hip_lib):
    """sc_gemm_headed_f16 serves the hand-scheduled kernel's shapes only and must say so (SC_ERR_UNSUPPORTED = -4) instead of computing
    something else: argument and shape checks run before any launch, so this is callable without a GPU (the pointers are never touched)."""
    from ctypes import c_float, c_void_p
    p = c_void_p(4096)
    rc = hip_lib.sc_gemm_headed_f16(p, 1024, p, p, p, 384, 512, 384, 1024, 4, p, 4096, 0, 128, c_float(1.0), None)       # N = 384: not a multiple of 256
    assert rc == -4 and b"N % 256" in hip_lib.sc_last_error()
    rc = hip_lib.sc_gemm_headed_f16(p, 1024, p, p, p, 512, 512, 512, 1024, 7, p, 4096, 0, 128, c_float(1.0), None)       # unknown mode
    assert rc == -1 and b"mode" in hip_lib.sc_last_error()
    rc = hip_lib.sc_gemm_headed_f16(p, 1024, p, p, p, 512, 512, 512, 1024, 4, p, 4096, 0, 100, c_float(1.0), None)       # lead_cols not a head multiple
    assert rc == -1
    rc = hip_lib.sc_gemm_headed_f16(p, 1024, p, p, p, 512, 512, 512, 1024, 4, p, 600, 100, 128, c_float(1.0), None)       # ABI 4: rows 100..611 of a 600-row rotary table
    assert rc == -1 and b"exceed the rotary table" in hip_lib.sc_last_error()
    rc = hip_lib.sc_rope_f32in_f16(p, 512, p, 600, None, 590, 16, 4, 128, 0, p, 512, None)                                # the same for the fp32-in rotary kernel
    assert rc == -1 and b"exceed the rotary table" in hip_lib.sc_last_error()
    rc = hip_lib.sc_attention_f16(p, 128, p, 128, p, 128, p, 128, 1, 16, 16, 1, 1, 128, c_float(1.0), 8, None, 1, None, 0, 0, 0, 0, 0, None)   # unknown flag bit
    assert rc == -1 and b"flags" in hip_lib.sc_last_error()


def test_no_cpu_fallback():
    import pytest
    import torch
    from streamchat_amd import ops
    from streamchat_amd._lib import StreamChatHipError
    with pytest.raises(StreamChatHipError):
        ops.kmeans_fit(torch.zeros(8, 16), 2, [0, 1])        # CPU tensor: refused, never computed on the host


def test_rope_qkv_rows_rejects_bad_arguments_before_any_device_work(hip_lib):
    """ABI 6 entry point: argument errors come back as SC_ERR_* with a message, without touching the device (runs without a GPU)."""
    import ctypes
    f = hip_lib.sc_rope_qkv_rows_f16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    hip_lib.sc_last_error.restype = ctypes.c_char_p
    assert f(None, 4608, None, None, 64, None, 2, 28, 4, 128, None, 3584, None, 0, 1024, 16, None) < 0
    assert b"null pointer" in hip_lib.sc_last_error()
    p = ctypes.c_void_p(4096)                                 # any aligned non-null value: the checks below fire before it is dereferenced
    assert f(p, 100, p, p, 64, p, 2, 28, 4, 128, p, 3584, p, 0, 1024, 16, None) < 0          # ldx smaller than (28 + 8) * 128
    assert b"leading dimensions" in hip_lib.sc_last_error()
    assert f(p, 4608, p, p, 64, p, 2, 28, 4, 100, p, 3584, p, 0, 1024, 16, None) < 0         # Dh % 8 != 0
    assert b"bad sizes" in hip_lib.sc_last_error()
