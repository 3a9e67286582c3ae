"""ctypes binding of libstreamchat_hip.so — the ONLY compute path of the package.

There is deliberately no fallback: if the shared library is missing or a symbol cannot be bound the
import fails loudly, and every wrapper refuses non-CUDA tensors.  Signatures mirror
include/streamchat_hip.h one to one."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_LIB") or os.path.join(_HERE, "libstreamchat_hip.so")     # SC_LIB: A/B builds of the kernels


class StreamChatHipError(RuntimeError):
    pass


# sc_kmeans_exchange_fn: int (*)(void* ctx, int what, sc_stream_t stream)
KMEANS_EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_void_p)


# name -> (restype, [argtypes])   — keep in sync with include/streamchat_hip.h
ABI_VERSION = 8            # include/streamchat_hip.h SC_ABI_VERSION this binding was written against
SIGNATURES = {
    "sc_abi_version": (c_int, []),
    "sc_last_error": (c_char_p, []),
    "sc_build_info": (c_char_p, []),
    "sc_stream_create_masked": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "sc_stream_destroy": (c_int, [c_void_p]),
    "sc_device_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "sc_kmeans_workspace_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "sc_kmeans_fit": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_kmeans_fit_cols": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, KMEANS_EXCHANGE_FN, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "sc_kmeans_assign": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_kmeans_update": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_preprocess_u8": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p, c_void_p]),
    "sc_preprocess_patchify_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p, c_int, c_void_p]),
    "sc_gemm_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_void_p]),
    "sc_vit_embed_ln_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sc_gemv_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_float, c_void_p]),
    "sc_decode_qkv_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                   c_int, c_int, c_float, c_void_p]),
    "sc_layernorm_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sc_rmsnorm_f16": (c_int, [c_void_p, c_int, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sc_attention_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_float, c_int, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int64, c_int64, c_void_p]),
    "sc_attention_variant": (c_int, [c_int, c_int, c_int]),
    "sc_bert_embed_ln_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sc_pool_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "sc_avgpool_tokens_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sc_gather_rows_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sc_gemm_headed_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "sc_rope_table_f32": (c_int, [c_void_p, c_int, c_int, c_float, c_float, c_void_p]),
    "sc_rope_qkv_rows_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "sc_rope_f32in_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sc_decode_qkv_tab_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                      c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "sc_decode_advance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sc_rope_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "sc_rope_qk_row_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "sc_rope_row_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "sc_patchify_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "sc_counter_uniform_f32": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "sc_pick_token_workspace_bytes": (c_size_t, [c_int]),
    "sc_pick_token_f32": (c_int, [c_void_p, c_int, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_sample_token_workspace_bytes": (c_size_t, [c_int]),
    "sc_sample_token_f32": (c_int, [c_void_p, c_int, c_int, c_int64, c_float, c_int, c_float, c_float, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "sc_sim_topk": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load and bind the library (raises StreamChatHipError if it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StreamChatHipError(
            f"{LIB_PATH} not found: build it with `python -m streamchat_amd.build` (hipcc, gfx950). "
            "streamchat_amd has no CPU / PyTorch fallback path.")
    # The library's HIP calls must bind to the SAME runtime torch uses: torch ships its own libamdhip64 / libhsa-runtime64 and the .so links
    # the system ones; whichever is in the process first provides the symbols.  Loaded before torch, the library would talk to a second
    # runtime that finds no device once torch's has opened it ("no ROCm-capable device", seen when build() and smoke() ran in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise StreamChatHipError(f"libstreamchat_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sc_abi_version() != ABI_VERSION:
        raise StreamChatHipError(f"ABI version mismatch: library {lib.sc_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sc_last_error()
        raise StreamChatHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """device pointer of a CUDA tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise StreamChatHipError("streamchat_amd kernels take CUDA (HIP) tensors only; there is no CPU fallback")
    if not t.is_contiguous():
        raise StreamChatHipError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


import threading as _threading

_tls = _threading.local()


def move_to_stream_when(condition, stream):
    """THIS host thread's launches move to `stream` (ordered behind everything it has enqueued so far) at its first kernel launch after
    `condition()` turns true - session.py / the entry point's --overlap: a reader / updater job that started on its CU partition takes the
    whole chip as soon as the answer decode beside it has finished.  `condition` None cancels.  Checked in stream_ptr(), i.e. at every
    launch of the library; never while the thread's stream is being captured."""
    _tls.pending = None if condition is None else (condition, stream)


def stream_ptr(device=None):
    import torch
    p = getattr(_tls, "pending", None)
    if p is not None and p[0]() and not torch.cuda.is_current_stream_capturing():
        _tls.pending = None
        cur = torch.cuda.current_stream(device)
        if cur.cuda_stream != p[1].cuda_stream:
            ev = torch.cuda.Event()
            ev.record(cur)
            p[1].wait_event(ev)
            torch.cuda.set_stream(p[1])
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


DTYPE_CODE = {"torch.float16": 0, "torch.bfloat16": 1, "torch.float32": 2}
