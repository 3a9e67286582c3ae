"""Thin tensor-level wrappers over the C ABI (include/streamchat_hip.h).

PyTorch is used for device memory, streams and nothing else: every function here launches
hand-written gfx950 kernels on `torch.cuda.current_stream()` with caller-visible tensors as
buffers.  No function has a PyTorch / CPU fallback."""
import ctypes
from ctypes import c_float, c_int, c_int64, c_size_t

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr, move_to_stream_when, StreamChatHipError   # noqa: F401

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # openai/clip-vit-large-patch14-336 preprocessor_config
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise StreamChatHipError("streamchat_amd kernels take CUDA (HIP) tensors only; there is no CPU fallback")


def _code(t: torch.Tensor) -> int:
    try:
        return _lib.DTYPE_CODE[str(t.dtype)]
    except KeyError:
        raise StreamChatHipError(f"unsupported dtype {t.dtype}")


_ws_cache = {}


class KernelTimer:
    """Optional live timing of kernel launches with HIP events on the launch stream (bench.py's `roofline`
    leg): `with ops.KernelTimer() as kt: ...; kt.summary()` -> {name: (launches, total_ms, total_work)}."""
    active = None

    def __init__(self):
        self.records = []
        self.aux = {}               # side records (e.g. the device-side k-means info words of every fit)
        self.tag = ""               # optional stage label set by the caller ("encode", "prefill", ...): records become "<tag>/<kernel>"

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *a):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, work, e0, e1 in self.records:
            n, ms, w = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), w + work)
        return out


class _timed:
    def __init__(self, name, work):
        self.kt = KernelTimer.active
        if self.kt is not None and torch.cuda.is_current_stream_capturing():
            self.kt = None                      # events recorded inside a hipGraph capture cannot be timed
        if self.kt is not None:
            self.name, self.work = (f"{self.kt.tag}/{name}" if self.kt.tag else name), work
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.kt is not None:
            self.e0.record()

    def __exit__(self, *a):
        if self.kt is not None:
            self.e1.record()
            self.kt.records.append((self.name, self.work, self.e0, self.e1))


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch per (device, current stream) (caller-owned memory handed to the library; 256-byte aligned).  Per STREAM since round 5:
    two host threads driving two streams (session.py: the QA thread beside the reader / updater) must not hand the same scratch to kernels
    that run at the same time; a buffer that is replaced stays alive until its stream has passed it (record_stream)."""
    d = torch.device(device)
    dev = d.index if d.index is not None else torch.cuda.current_device()
    st = torch.cuda.current_stream(dev)
    key = (dev, st.cuda_stream)
    buf = _ws_cache.get(key)
    if buf is not None and buf.numel() < nbytes:
        buf.record_stream(st)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def device_info():
    lib = _lib.load()
    cu, ok, mem = c_int(0), c_int(0), c_size_t(0)
    check(lib.sc_device_info(ctypes.byref(cu), ctypes.byref(ok), ctypes.byref(mem)), "sc_device_info")
    return dict(cu_count=cu.value, is_gfx950=bool(ok.value), hbm_bytes=mem.value)


# ------------------------------------------------------------------------------------------------
def kmeans_fit(X: torch.Tensor, K: int, init_idx, reseed_idx=None, weights=None, max_iter: int = 10, tol: float = 1e-4):
    """Lloyd k-means on the device (reference utiles.py:294-318 semantics, fp32-canonical).

    X [T, D] f16/bf16/f32 CUDA; init_idx [K] ints; reseed_idx optional ints consumed on empty clusters.
    Returns (centroids [K, D] fp32, labels [T] int64, wsum [K] fp32, info [4] int32) — all CUDA tensors,
    nothing is synchronised."""
    _require_cuda(X)
    lib = _lib.load()
    if X.dim() != 2:
        raise StreamChatHipError("kmeans_fit: X must be [T, D]")
    X = X.contiguous()
    T, D = X.shape
    dev = X.device
    init = torch.as_tensor(init_idx, dtype=torch.int32).to(dev).contiguous()
    if init.numel() != K:
        raise StreamChatHipError("kmeans_fit: init_idx must have K entries")
    rs = None if reseed_idx is None else torch.as_tensor(reseed_idx, dtype=torch.int32).to(dev).contiguous()
    w = None if weights is None else weights.to(device=dev, dtype=torch.float32).contiguous()
    C = torch.empty((K, D), dtype=torch.float32, device=dev)
    labels = torch.empty(T, dtype=torch.int64, device=dev)
    wsum = torch.empty(K, dtype=torch.float32, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    need = lib.sc_kmeans_workspace_bytes(T, D, K)
    ws = _workspace(need, dev)
    with torch.cuda.device(dev), _timed("kmeans_fit", float(T) * D * X.element_size() + 2.0 * K * D * 4):
        check(lib.sc_kmeans_fit(ptr(X), _code(X), T, c_int64(D), K, ptr(w), ptr(init), ptr(rs), 0 if rs is None else rs.numel(),
                                max_iter, c_float(tol), ptr(C), ptr(labels), ptr(wsum), ptr(info), ptr(ws), c_size_t(ws.numel()),
                                stream_ptr(dev)), "sc_kmeans_fit")
    if KernelTimer.active is not None:
        KernelTimer.active.aux.setdefault("kmeans_info", []).append(info)
    return C, labels, wsum, info


KM_GROUP, KM_SEGMENTS = 2048, 32          # SC-KM2 constants (csrc/kmeans.hip GW / NSEG): columns per fp64 group, segments per distance


def kmeans_column_slabs(D: int, world: int):
    """How sc_kmeans_fit_cols deals the D columns of a matrix to `world` ranks: whole SC-KM2 segments, the non-empty ones as even as they go
    (a matrix of fewer than 32 x seg_groups groups leaves its last segments empty: they go to the last rank, which owns the matrix's tail).
    Returns (seg_groups, [(seg_first, seg_count, col_lo, col_hi)] per rank), or None when there are fewer non-empty segments than ranks
    (few columns: the caller clusters on one rank instead)."""
    ng = (D + KM_GROUP - 1) // KM_GROUP
    seg_groups = (ng + KM_SEGMENTS - 1) // KM_SEGMENTS
    live = (ng + seg_groups - 1) // seg_groups                 # segments that hold at least one group
    if live < world:
        return None
    out = []
    for r in range(world):
        s0, s1 = live * r // world, (live * (r + 1) // world if r + 1 < world else KM_SEGMENTS)
        out.append((s0, s1 - s0, min(D, s0 * seg_groups * KM_GROUP), min(D, s1 * seg_groups * KM_GROUP)))
    return seg_groups, out


def kmeans_fit_cols(X: torch.Tensor, K: int, init_idx, reseed_idx, seg_groups: int, seg_first: int, seg_count: int, exchange,
                    weights=None, max_iter: int = 10, tol: float = 1e-4):
    """The data-parallel Lloyd (sc_kmeans_fit_cols): X [T, D_local] is this rank's column slab (`kmeans_column_slabs`), `exchange(what, table)`
    completes the [32, n] fp64 segment table `table` on every rank from the row windows the ranks own (what 0: distances, 1: shifts) - on the
    current stream, or after synchronising it.  Returns (C_local [K, D_local] fp32, labels, wsum, info): everything but C_local identical
    on every rank and bit-identical to kmeans_fit on the whole matrix."""
    _require_cuda(X)
    lib = _lib.load()
    if X.dim() != 2:
        raise StreamChatHipError("kmeans_fit_cols: X must be [T, D_local]")
    X = X.contiguous()
    T, D = X.shape
    dev = X.device
    init = torch.as_tensor(init_idx, dtype=torch.int32).to(dev).contiguous()
    if init.numel() != K:
        raise StreamChatHipError("kmeans_fit_cols: init_idx must have K entries")
    rs = None if reseed_idx is None else torch.as_tensor(reseed_idx, dtype=torch.int32).to(dev).contiguous()
    w = None if weights is None else weights.to(device=dev, dtype=torch.float32).contiguous()
    C = torch.empty((K, D), dtype=torch.float32, device=dev)
    labels = torch.empty(T, dtype=torch.int64, device=dev)
    wsum = torch.empty(K, dtype=torch.float32, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    tables = (torch.zeros((KM_SEGMENTS, T * K), dtype=torch.float64, device=dev), torch.zeros((KM_SEGMENTS, K), dtype=torch.float64, device=dev))
    failure = []

    def _cb(_ctx, what, _stream):
        try:
            exchange(int(what), tables[int(what)])
            return 0
        except BaseException as e:          # an exception must not unwind through the C frames: report it through the return code
            failure.append(e)
            return 1

    cb = _lib.KMEANS_EXCHANGE_FN(_cb)
    ws = _workspace(lib.sc_kmeans_workspace_bytes(T, D, K), dev)
    with torch.cuda.device(dev):
        rc = lib.sc_kmeans_fit_cols(ptr(X), _code(X), T, c_int64(D), K, ptr(w), ptr(init), ptr(rs), 0 if rs is None else rs.numel(),
                                    max_iter, c_float(tol), ptr(C), ptr(labels), ptr(wsum), ptr(info), c_int64(seg_groups), seg_first, seg_count,
                                    ptr(tables[0]), ptr(tables[1]), cb, None, ptr(ws), c_size_t(ws.numel()), stream_ptr(dev))
    if failure:
        raise failure[0]
    check(rc, "sc_kmeans_fit_cols")
    for t in tables:
        t.record_stream(torch.cuda.current_stream(dev))
    return C, labels, wsum, info


def kmeans_assign(X: torch.Tensor, C: torch.Tensor, return_dist2: bool = False):
    """labels [T] int64 (and optionally dist2 [T, K] fp64) of rows X against fp32 centroids C."""
    _require_cuda(X, C)
    lib = _lib.load()
    X = X.contiguous()
    C = C.to(torch.float32).contiguous()
    T, D = X.shape
    K = C.shape[0]
    dev = X.device
    labels = torch.empty(T, dtype=torch.int64, device=dev)
    d2 = torch.empty((T, K), dtype=torch.float64, device=dev) if return_dist2 else None
    ws = _workspace(lib.sc_kmeans_workspace_bytes(T, D, K), dev)
    with torch.cuda.device(dev):
        check(lib.sc_kmeans_assign(ptr(X), _code(X), T, c_int64(D), K, ptr(C), ptr(labels), ptr(d2), ptr(ws), c_size_t(ws.numel()),
                                   stream_ptr(dev)), "sc_kmeans_assign")
    return (labels, d2) if return_dist2 else labels


def kmeans_update(X: torch.Tensor, labels: torch.Tensor, C_old: torch.Tensor, weights=None, empty="zero", fill_idx=None):
    """One centroid update from given labels (C ABI sc_kmeans_update): returns (C_new [K, D] fp32, wsum [K] fp32, shift2 [K] fp64 =
    ||C_old[k] - C_new[k]||^2).  empty="zero": an empty cluster's centre is the zero vector (torch_kmeans); empty="fill": the j-th
    empty cluster takes row fill_idx[j] of X (kmeans_pytorch)."""
    _require_cuda(X, labels, C_old)
    lib = _lib.load()
    X = X.contiguous()
    T, D = X.shape
    C_old = C_old.to(torch.float32).contiguous()
    K = C_old.shape[0]
    dev = X.device
    labels = labels.to(device=dev, dtype=torch.int64).contiguous()
    w = None if weights is None else weights.to(device=dev, dtype=torch.float32).contiguous()
    fi = None if fill_idx is None else torch.as_tensor(fill_idx, dtype=torch.int32).to(dev).contiguous()
    C = torch.empty((K, D), dtype=torch.float32, device=dev)
    wsum = torch.empty(K, dtype=torch.float32, device=dev)
    shift2 = torch.empty(K, dtype=torch.float64, device=dev)
    ws = _workspace(lib.sc_kmeans_workspace_bytes(T, D, K), dev)
    with torch.cuda.device(dev):
        check(lib.sc_kmeans_update(ptr(X), _code(X), T, c_int64(D), K, ptr(w), ptr(labels), ptr(C_old), 1 if empty == "zero" else 0, ptr(fi),
                                   0 if fi is None else fi.numel(), ptr(C), ptr(wsum), ptr(shift2), ptr(ws), c_size_t(ws.numel()), stream_ptr(dev)),
              "sc_kmeans_update")
    return C, wsum, shift2


# ------------------------------------------------------------------------------------------------
def _f3(v):
    return (c_float * 3)(*[float(x) for x in v])


def preprocess_u8(frames: torch.Tensor, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """uint8 [N, H, W, 3] -> fp16 [N, 3, H, W], ((x/255)-mean)/std  (reference utiles.py:71-87 + .to(fp16))."""
    _require_cuda(frames)
    lib = _lib.load()
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise StreamChatHipError("preprocess_u8: frames must be uint8 [N, H, W, 3]")
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    out = torch.empty((n, 3, h, w), dtype=torch.float16, device=frames.device)
    with torch.cuda.device(frames.device):
        check(lib.sc_preprocess_u8(ptr(frames), n, h, w, _f3(mean), _f3(std), ptr(out), stream_ptr(frames.device)), "sc_preprocess_u8")
    return out


def preprocess_patchify_u8(frames: torch.Tensor, patch: int, ld: int, mean=CLIP_MEAN, std=CLIP_STD, out=None) -> torch.Tensor:
    """uint8 [N, H, W, 3] -> fp16 patch rows [N*(H/p)*(W/p), ld] (normalised, im2col order c,py,px; zero padded)."""
    _require_cuda(frames)
    lib = _lib.load()
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise StreamChatHipError("preprocess_patchify_u8: frames must be uint8 [N, H, W, 3]")
    frames = frames.contiguous()
    n, h, w, _ = frames.shape
    if h % patch or w % patch:
        raise StreamChatHipError(f"preprocess_patchify_u8: {h}x{w} frames are not a multiple of the patch size {patch} "
                                 "(resize + centre-crop to the tower's image_size first: mm_utils.resize_center_crop_u8)")
    rows = n * (h // patch) * (w // patch)
    if out is None:
        out = torch.empty((rows, ld), dtype=torch.float16, device=frames.device)
    elif out.shape[0] < rows or out.shape[1] != ld or out.dtype != torch.float16 or not out.is_contiguous():
        raise StreamChatHipError(f"preprocess_patchify_u8: out must be a contiguous fp16 [>= {rows}, {ld}] buffer, got {tuple(out.shape)}")
    with torch.cuda.device(frames.device):
        check(lib.sc_preprocess_patchify_u8(ptr(frames), n, h, w, patch, _f3(mean), _f3(std), ptr(out), ld, stream_ptr(frames.device)),
              "sc_preprocess_patchify_u8")
    return out


# ------------------------------------------------------------------------------------------------
def sim_topk(q: torch.Tensor, docs: torch.Tensor, k: int = 1, metric: str = "cos"):
    """(idx [k] int32, score [k] fp32) of the k best documents; cosine (desc) or squared L2 (asc); ties -> lowest index."""
    _require_cuda(q, docs)
    lib = _lib.load()
    q = q.to(torch.float32).reshape(-1).contiguous()
    docs = docs.to(torch.float32).contiguous()
    M, d = docs.shape
    if q.numel() != d:
        raise StreamChatHipError("sim_topk: q and docs disagree on d")
    idx = torch.empty(k, dtype=torch.int32, device=docs.device)
    score = torch.empty(k, dtype=torch.float32, device=docs.device)
    with torch.cuda.device(docs.device):
        check(lib.sc_sim_topk(ptr(q), ptr(docs), M, d, k, 0 if metric == "cos" else 1, ptr(idx), ptr(score), stream_ptr(docs.device)),
              "sc_sim_topk")
    return idx, score


def counter_uniform(seeds: torch.Tensor, counter=None, add: int = 0, out=None):
    """u [B] fp32 in [0, 1): a pure function of (seeds[b], counter[0] + add) - sc_counter_uniform_f32.  seeds [B] int64 (device); counter: device
    int64 scalar tensor or None.  No generator state: graph-capturable, the same numbers eagerly, in a graph and on any thread."""
    _require_cuda(seeds)
    if seeds.dtype != torch.int64 or (counter is not None and counter.dtype != torch.int64):
        raise StreamChatHipError("counter_uniform: int64 seeds / counter expected")
    seeds = seeds.contiguous().view(-1)
    B = seeds.numel()
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=seeds.device)
    with torch.cuda.device(seeds.device):
        check(_lib.load().sc_counter_uniform_f32(ptr(seeds), B, ptr(counter), c_int64(int(add)), ptr(out), stream_ptr(seeds.device)), "sc_counter_uniform_f32")
    return out


def counter_uniform_host(seed: int, n: int) -> float:
    """the same number on the host (tests; documentation of the formula)"""
    M = (1 << 64) - 1
    x = (seed + n * 0x9E3779B97F4A7C15) & M
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M
    x ^= x >> 31
    return (x >> 40) / 16777216.0


def pick_token(logits: torch.Tensor, temperature: float = 0.0, u=None, out=None, ws=None):
    """Next token ids [B] int64 from fp32 logits [B, V] (or [V]): arg-max (temperature <= 0, lowest index on ties) or a sample from
    softmax(logits / temperature) at the uniform draws `u` [B] (device fp32 in [0, 1)).  No host sync; graph-capturable when `ws`
    (a private uint8 buffer of sc_pick_token_workspace_bytes(B)) is passed."""
    _require_cuda(logits)
    lib = _lib.load()
    if logits.dtype != torch.float32:
        raise StreamChatHipError("pick_token: fp32 logits expected")
    lg = logits if logits.dim() == 2 else logits.view(1, -1)
    if lg.stride(1) != 1:
        raise StreamChatHipError("pick_token: logits rows must be contiguous")
    B, V = lg.shape
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=lg.device)
    need = lib.sc_pick_token_workspace_bytes(B)
    if ws is None:
        ws = _workspace(need, lg.device)
    if temperature > 0:
        if u is None:
            raise StreamChatHipError("pick_token: sampling needs the uniform draws u")
        u = u.to(device=lg.device, dtype=torch.float32).contiguous().view(-1)
        if u.numel() != B:
            raise StreamChatHipError("pick_token: u must have one entry per row")
    from ctypes import c_void_p
    with torch.cuda.device(lg.device):
        check(lib.sc_pick_token_f32(c_void_p(lg.data_ptr()), B, V, c_int64(lg.stride(0)), c_float(float(temperature)), ptr(u) if temperature > 0 else None,
                                    ptr(out), ptr(ws), c_size_t(ws.numel()), stream_ptr(lg.device)), "sc_pick_token_f32")
    return out


def sample_token_workspace_bytes(B: int) -> int:
    return int(_lib.load().sc_sample_token_workspace_bytes(B))


def sample_token(logits: torch.Tensor, temperature: float = 0.0, u=None, top_k: int = 0, top_p: float = 1.0, repetition_penalty: float = 1.0,
                 prev_ids=None, n_prev=None, out=None, ws=None):
    """Next token ids [B] int64 from fp32 logits [B, V] through HF's processor chain (repetition penalty over the DISTINCT ids of
    prev_ids[b, :n_prev[b]] -> temperature -> top-k (ties at the k-th value kept) -> top-p) and one inverse-CDF draw at u [B] (top_k <= 64: the
    candidate kernels; top_k > 64 or a nucleus without top-k: the full-vocabulary radix-select kernel, which also leaves per row
    {threshold logit bits, kept count} in ws[0 : 2B] as uint32 for diagnostics);
    temperature <= 0: arg-max of the penalised logits.  `logits` is modified in place when repetition_penalty != 1.  n_prev: int (all
    rows) or a device int32 tensor [B] (graph-safe).  No host sync; graph-capturable with a private `ws`."""
    _require_cuda(logits)
    lib = _lib.load()
    if logits.dtype != torch.float32:
        raise StreamChatHipError("sample_token: fp32 logits expected")
    lg = logits if logits.dim() == 2 else logits.view(1, -1)
    if lg.stride(1) != 1:
        raise StreamChatHipError("sample_token: logits rows must be contiguous")
    B, V = lg.shape
    if out is None:
        out = torch.empty(B, dtype=torch.int64, device=lg.device)
    need = lib.sc_sample_token_workspace_bytes(B)
    if ws is None:
        ws = _workspace(need, lg.device)
    if temperature > 0:
        if u is None:
            raise StreamChatHipError("sample_token: sampling needs the uniform draws u")
        u = u.to(device=lg.device, dtype=torch.float32).contiguous().view(-1)
        if u.numel() != B:
            raise StreamChatHipError("sample_token: u must have one entry per row")
    n_dev, n_host, prev_ld = None, 0, 0
    if prev_ids is not None and repetition_penalty != 1.0:
        if prev_ids.dtype != torch.int64 or not prev_ids.is_cuda or prev_ids.stride(-1) != 1:
            raise StreamChatHipError("sample_token: prev_ids must be a device int64 tensor [B, n] with unit stride")
        pv = prev_ids if prev_ids.dim() == 2 else prev_ids.view(1, -1)
        prev_ld = pv.stride(0) if pv.shape[0] > 1 else pv.shape[1]
        if isinstance(n_prev, torch.Tensor):
            if n_prev.dtype != torch.int32 or not n_prev.is_cuda or n_prev.numel() != B:
                raise StreamChatHipError("sample_token: n_prev tensor must be device int32 [B]")
            n_dev = n_prev
        else:
            n_host = int(pv.shape[1] if n_prev is None else n_prev)
            if n_host > pv.shape[1]:
                raise StreamChatHipError("sample_token: n_prev exceeds prev_ids")
    from ctypes import c_void_p
    with torch.cuda.device(lg.device):
        check(lib.sc_sample_token_f32(c_void_p(lg.data_ptr()), B, V, c_int64(lg.stride(0)), c_float(float(temperature)), int(top_k or 0),
                                      c_float(float(1.0 if top_p is None else top_p)), c_float(float(repetition_penalty or 1.0)),
                                      ptr(prev_ids) if prev_ld else None, c_int64(prev_ld), ptr(n_dev), n_host, ptr(u) if temperature > 0 else None,
                                      ptr(out), ptr(ws), c_size_t(ws.numel()), stream_ptr(lg.device)), "sc_sample_token_f32")
    return out


# ------------------------------------------------------------------------------------------------
# dense blocks (MFMA GEMM + norms)
# ------------------------------------------------------------------------------------------------
EPI = {"none": 0, "quick_gelu": 1, "gelu": 2, "swiglu": 3}


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, residual=None, epilogue: str = "none", out=None, out_f32: bool = False,
         a_rows=None, M=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias) (+ residual); fp16 operands, fp32 MFMA accumulate.
    `a` may be a row-strided view (stride(1) == 1).  a_rows=(grp, stride, off) with explicit M reads logical
    row m from storage row (m//grp)*stride + off + m%grp."""
    _require_cuda(a, w)
    lib = _lib.load()
    if a.dtype != torch.float16 or w.dtype != torch.float16:
        raise StreamChatHipError("gemm: fp16 operands only")
    K = a.shape[1]
    M = a.shape[0] if M is None else M
    N, K2 = w.shape
    if K2 != K or a.stride(1) != 1 or not w.is_contiguous():
        raise StreamChatHipError("gemm: shape/stride mismatch")
    if out is None:
        out = torch.empty((M, N // 2 if epilogue == "swiglu" else N), dtype=torch.float32 if out_f32 else torch.float16, device=a.device)
    if out.stride(1) != 1 or (residual is not None and residual.stride(1) != 1):
        raise StreamChatHipError("gemm: out/residual must be row-major")
    from ctypes import c_void_p
    P = lambda t: None if t is None else c_void_p(t.data_ptr())
    with torch.cuda.device(a.device), _timed("k_gemm", 2.0 * M * N * K):
        check(lib.sc_gemm_f16(P(a), a.stride(0), P(w), P(bias), P(residual), 0 if residual is None else residual.stride(0), P(out),
                              out.stride(0), M, N, K, EPI[epilogue], 1 if out.dtype == torch.float32 else 0,
                              *((0, 0, 0) if a_rows is None else a_rows), stream_ptr(a.device)),
              "sc_gemm_f16")
    return out


LOG2E = 1.4426950408889634


def gemm_headed_ok(N: int, K: int, a, w, bias, out) -> bool:
    """True where sc_gemm_headed_f16 (the hand-scheduled kernel) serves the call - the SAME preconditions the C entry point checks before it
    returns SC_ERR_UNSUPPORTED (gemm.hip): N % 256 == 0, K % 128 == 0, row strides of A and C multiples of 8 elements and >= K / N, A's row
    stride below 2^31 / 512 bytes, 16-byte aligned A / W / C / bias.  Anything else takes gemm(out_f32=True) + rope_f32in / a plain scale."""
    if N % 256 or K % 128:
        return False
    lda, ldc = (a.stride(0) if a.dim() == 2 else K), (out.stride(0) if out.dim() == 2 else N)
    if lda < K or lda % 8 or lda * 512 >= 2 ** 31 or ldc < N or ldc % 8:
        return False
    return all(t is None or t.data_ptr() % 16 == 0 for t in (a, w, bias, out))


def gemm_headed(a, w, bias, out, mode: str, lead_cols: int, rope_tab=None, pos0: int = 0, col_scale: float = 1.0):
    """out[M,N] = headed_epilogue(a @ w^T + bias), fp16 out.  mode "rope": columns [0, lead_cols) are rotary heads of width 128 rotated with
    rows pos0.. of the fp32 table `rope_tab` (one rounding); mode "colscale": columns < lead_cols scaled by col_scale.  Only where
    gemm_headed_ok() holds."""
    _require_cuda(a, w, out)
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if a.dtype != torch.float16 or w.dtype != torch.float16 or out.dtype != torch.float16 or a.stride(1) != 1 or out.stride(1) != 1 or not w.is_contiguous():
        raise StreamChatHipError("gemm_headed: fp16 row-major operands only")
    from ctypes import c_void_p
    P = lambda t: None if t is None else c_void_p(t.data_ptr())
    with torch.cuda.device(a.device), _timed("k_gemm", 2.0 * M * N * K):
        check(lib.sc_gemm_headed_f16(P(a), a.stride(0), P(w), P(bias), P(out), out.stride(0), M, N, K, {"rope": 4, "colscale": 5}[mode], P(rope_tab), 0 if rope_tab is None else int(rope_tab.shape[0]), int(pos0),
                                     int(lead_cols), c_float(col_scale), stream_ptr(a.device)), "sc_gemm_headed_f16")
    return out


_rope_tables = {}


def rope_table(max_pos: int, Dh: int, theta: float, scale: float, device) -> torch.Tensor:
    """fp32 [>= max_pos, 2, Dh/2] (cos | sin) * scale, built once per (device, Dh, theta, scale) and grown in powers of two."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), Dh, float(theta), float(scale))
    t = _rope_tables.get(key)
    if t is None or t.shape[0] < max_pos:
        n = 1 << max(12, (int(max_pos) - 1).bit_length())
        t = torch.empty((n, 2, Dh // 2), dtype=torch.float32, device=dev)
        from ctypes import c_void_p
        with torch.cuda.device(dev):
            check(_lib.load().sc_rope_table_f32(c_void_p(t.data_ptr()), n, Dh, c_float(theta), c_float(scale), stream_ptr(dev)), "sc_rope_table_f32")
        _rope_tables[key] = t
    return t


def rope_f32in(x32, tab, heads: int, Dh: int, out, plain_cols: int = 0, pos0: int = 0, positions=None):
    """x32 [rows, >= heads*Dh + plain_cols] fp32 (projection + bias) -> out fp16: `heads` rotary heads rotated with the table (one rounding), then
    `plain_cols` columns cast unchanged."""
    _require_cuda(x32, tab, out)
    if x32.dtype != torch.float32 or out.dtype != torch.float16 or x32.stride(-1) != 1 or out.stride(-1) != 1:
        raise StreamChatHipError("rope_f32in: fp32 in, fp16 out, unit last stride")
    x2, o2 = x32.reshape(-1, x32.shape[-1]) if x32.dim() == 1 else x32, out.reshape(-1, out.shape[-1]) if out.dim() == 1 else out
    pos = None if positions is None else positions.to(device=x32.device, dtype=torch.int32).contiguous()
    from ctypes import c_void_p
    with torch.cuda.device(x32.device):
        check(_lib.load().sc_rope_f32in_f16(c_void_p(x2.data_ptr()), x2.stride(0), c_void_p(tab.data_ptr()), int(tab.shape[0]), ptr(pos), int(pos0), x2.shape[0], heads, Dh, plain_cols,
                                            c_void_p(o2.data_ptr()), o2.stride(0), stream_ptr(x32.device)), "sc_rope_f32in_f16")
    return out


def rope_qkv_rows(x32, tab_q, tab_k, positions, q_heads: int, kv_heads: int, Dh: int, q_out, cache):
    """Batched decode: x32 [B, (q_heads + 2 kv_heads) * Dh] fp32 (fused q|k|v projection + bias of one new token per sequence) -> q_out [B, q_heads * Dh]
    fp16 (rotated, pre-scaled table) and row positions[b] of cache[b] ([B, cap, 2 * kv_heads * Dh]: rotated K | V).  One launch; the numbers of
    rope_f32in on q and on k|v + index_copy_."""
    _require_cuda(x32, tab_q, tab_k, positions, q_out, cache)
    if x32.dtype != torch.float32 or q_out.dtype != torch.float16 or cache.dtype != torch.float16 or positions.dtype != torch.int32 or cache.dim() != 3:
        raise StreamChatHipError("rope_qkv_rows: fp32 x, fp16 q_out / cache [B, cap, 2 * kv_heads * Dh], int32 positions")
    if x32.stride(-1) != 1 or q_out.stride(-1) != 1 or cache.stride(-1) != 1 or tab_q.shape[0] != tab_k.shape[0] or not positions.is_contiguous():
        raise StreamChatHipError("rope_qkv_rows: unit last strides, equal table lengths, contiguous positions")
    from ctypes import c_void_p
    with torch.cuda.device(x32.device):
        check(_lib.load().sc_rope_qkv_rows_f16(c_void_p(x32.data_ptr()), x32.stride(0), c_void_p(tab_q.data_ptr()), c_void_p(tab_k.data_ptr()), int(tab_q.shape[0]),
                                               ptr(positions), x32.shape[0], q_heads, kv_heads, Dh, c_void_p(q_out.data_ptr()), q_out.stride(0),
                                               c_void_p(cache.data_ptr()), cache.stride(0), cache.stride(1), cache.shape[1], stream_ptr(x32.device)), "sc_rope_qkv_rows_f16")
    return q_out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out=None):
    _require_cuda(x, gamma, beta)
    lib = _lib.load()
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float16, device=x.device)
    from ctypes import c_void_p
    with torch.cuda.device(x.device):
        check(lib.sc_layernorm_f16(c_void_p(x.data_ptr()), x.stride(0), ptr(gamma), ptr(beta), c_float(eps), c_void_p(out.data_ptr()),
                                   out.stride(0), rows, cols, stream_ptr(x.device)), "sc_layernorm_f16")
    return out


def rmsnorm(x: torch.Tensor, gamma: torch.Tensor, eps: float, out=None):
    _require_cuda(x, gamma)
    lib = _lib.load()
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float16, device=x.device)
    from ctypes import c_void_p
    with torch.cuda.device(x.device):
        check(lib.sc_rmsnorm_f16(c_void_p(x.data_ptr()), x.stride(0), ptr(gamma), c_float(eps), c_void_p(out.data_ptr()), out.stride(0),
                                 rows, cols, stream_ptr(x.device)), "sc_rmsnorm_f16")
    return out


def build_info() -> str:
    """how libstreamchat_hip.so was built (sc_build_info): machine scheduler of attention.hip (or its FALLBACK), kernarg preload of the decode kernels"""
    return _lib.load().sc_build_info().decode()


_masked_streams = {}


def masked_stream(cu_first: int, cu_count: int, device=None):
    """A torch stream restricted to the CUs [cu_first, cu_first + cu_count) (sc_stream_create_masked: hipExtStreamCreateWithCUMask; consecutive
    CU-mask bits go round-robin over the 8 XCDs x 4 shader engines: use multiples of 32).  Round 5: the HBM-bound answer decode on one partition
    beside the MFMA-bound encode / prefill of the next segment on the other.  ONE stream per (device, range) and process: a second request returns
    the same object (the library keeps the partition size of at most 32 masked streams; they are never destroyed - torch may hold events
    recorded on them until interpreter exit)."""
    from ctypes import byref, c_void_p
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(cu_first), int(cu_count))
    if key in _masked_streams:
        return _masked_streams[key]
    h = c_void_p()
    with torch.cuda.device(dev):
        check(_lib.load().sc_stream_create_masked(int(cu_first), int(cu_count), 0, byref(h)), "sc_stream_create_masked")
    s = torch.cuda.ExternalStream(h.value, device=dev)
    s._sc_handle = h
    _masked_streams[key] = s
    return s


def attention_workspace_bytes(B: int, Hq: int, Sq: int, nsplit: int, Dh: int) -> int:
    return B * Hq * Sq * nsplit * (Dh + 2) * 4 if nsplit > 1 else 0


def attention_variant(Dh: int, Sq: int, nsplit: int = 1) -> int:
    """Kernel sc_attention_f16 dispatches for this shape in this process (0 k_attn, 1 k_attn long-prefill variant, 2 k_attn_fat, 3 k_attn_decode for non-causal calls)."""
    return int(_lib.load().sc_attention_variant(Dh, Sq, nsplit))


def attention(q, k, v, Hq: int, Hkv: int, Dh: int, scale: float, causal: bool = False, kv_len=None, out=None, nsplit: int = 1,
              q_head_stride: int = 0, o_head_stride: int = 0, out_ld=None, ws=None, q_prescaled: bool = False):
    """Fused attention.  q [B, Sq, >=Hq*Dh], k/v [B, Skv, >=Hkv*Dh] fp16 (may be strided column slices of one fused
    QKV buffer: only stride(-1) == 1 and a common row stride per tensor are required).  Returns [B, Sq, Hq*Dh].
    q_prescaled: q already carries scale * log2(e) from its producer (gemm_headed, decode_qkv_tab, rope_f32in with a scaled table)."""
    _require_cuda(q, k, v)
    lib = _lib.load()
    B, Sq = q.shape[0], q.shape[1]
    Skv = k.shape[1]
    for t in ((k, v) if q_head_stride else (q, k, v)):
        if t.stride(-1) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
            raise StreamChatHipError("attention: tensors must be [B, S, *] with dense batch stride and unit last stride")
    if out is None:
        out = torch.empty((B, Sq, Hq * Dh), dtype=torch.float16, device=q.device)
    from ctypes import c_void_p
    P = lambda t: None if t is None else c_void_p(t.data_ptr())
    ldo = out.stride(1) if out_ld is None else out_ld
    kl = None if kv_len is None else kv_len.to(device=q.device, dtype=torch.int32).contiguous()
    if nsplit > 1:              # split-KV partials: the shared grow-only scratch, or a caller-owned buffer (anything a hipGraph captures)
        need = attention_workspace_bytes(B, Hq, Sq, nsplit, Dh)
        if ws is None:
            ws = _workspace(need, q.device)
        elif ws.numel() * ws.element_size() < need:
            raise StreamChatHipError(f"attention: workspace {ws.numel() * ws.element_size()} < required {need}")
    else:
        ws = None
    with torch.cuda.device(q.device), _timed("k_attn", 4.0 * B * Hq * Sq * Skv * Dh * (0.5 if causal else 1.0)):
        check(lib.sc_attention_f16(P(q), q.stride(1), P(k), k.stride(1), P(v), v.stride(1), P(out), ldo, B, Sq, Skv, Hq, Hkv, Dh,
                                   c_float(scale), (1 if causal else 0) | (2 if q_prescaled else 0), P(kl), nsplit, P(ws), c_size_t(0 if ws is None else ws.numel() * ws.element_size()),
                                   q_head_stride, o_head_stride, c_int64(q.stride(0) if B > 1 else 0), c_int64(out.stride(0) if B > 1 else 0),
                                   stream_ptr(q.device)), "sc_attention_f16")
    return out


def vit_embed_ln(patch, cls, pos, gamma, beta, eps: float, N: int, P: int, out=None):
    """[N*P, D] patch embeddings -> [N*(P+1), D] tokens: (cls | patches) + position embedding, then pre-LayerNorm."""
    _require_cuda(patch, cls, pos, gamma, beta)
    lib = _lib.load()
    D = patch.shape[1]
    if out is None:
        out = torch.empty((N * (P + 1), D), dtype=torch.float16, device=patch.device)
    with torch.cuda.device(patch.device):
        check(lib.sc_vit_embed_ln_f16(ptr(patch), ptr(cls), ptr(pos), ptr(gamma), ptr(beta), c_float(eps), ptr(out), N, P, D,
                                      stream_ptr(patch.device)), "sc_vit_embed_ln_f16")
    return out


def patchify_f16(pixel_values: torch.Tensor, patch: int, ld: int, out=None) -> torch.Tensor:
    """fp16 [N, 3, H, W] -> patch rows [N*(H/p)*(W/p), ld] (im2col order c,py,px; zero padded)."""
    _require_cuda(pixel_values)
    lib = _lib.load()
    x = pixel_values.contiguous()
    if x.dtype != torch.float16:
        raise StreamChatHipError("patchify_f16: fp16 pixel values expected")
    n, _, h, w = x.shape
    if h % patch or w % patch:
        raise StreamChatHipError(f"patchify_f16: {h}x{w} images are not a multiple of the patch size {patch}")
    rows = n * (h // patch) * (w // patch)
    if out is None:
        out = torch.empty((rows, ld), dtype=torch.float16, device=x.device)
    elif out.shape[0] < rows or out.shape[1] != ld or out.dtype != torch.float16 or not out.is_contiguous():
        raise StreamChatHipError(f"patchify_f16: out must be a contiguous fp16 [>= {rows}, {ld}] buffer, got {tuple(out.shape)}")
    with torch.cuda.device(x.device):
        check(lib.sc_patchify_f16(ptr(x), n, h, w, patch, ptr(out), ld, stream_ptr(x.device)), "sc_patchify_f16")
    return out


def bert_embed_ln(ids, word, pos, type0, gamma, beta, eps: float, out=None):
    """ids int32 [B, L] -> fp16 [B*L, H] = LN(word[ids] + pos[t] + type0)."""
    _require_cuda(ids, word)
    lib = _lib.load()
    B, L = ids.shape
    H = word.shape[1]
    ids = ids.to(torch.int32).contiguous()
    if out is None:
        out = torch.empty((B * L, H), dtype=torch.float16, device=word.device)
    with torch.cuda.device(word.device):
        check(lib.sc_bert_embed_ln_f16(ptr(ids), ptr(word), ptr(pos), ptr(type0), ptr(gamma), ptr(beta), c_float(eps), ptr(out), B, L, H,
                                       word.shape[0], stream_ptr(word.device)), "sc_bert_embed_ln_f16")
    return out


def pool(hidden, lengths=None, mode: str = "cls", normalize: bool = False):
    """hidden fp16 [B, L, H] -> fp32 [B, H]: CLS row or masked mean (+ optional L2 normalisation)."""
    _require_cuda(hidden)
    lib = _lib.load()
    B, L, H = hidden.shape
    hidden = hidden.contiguous()
    ln = None if lengths is None else lengths.to(device=hidden.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, H), dtype=torch.float32, device=hidden.device)
    with torch.cuda.device(hidden.device):
        check(lib.sc_pool_f16(ptr(hidden), ptr(ln), ptr(out), B, L, H, 0 if mode == "cls" else 1, 1 if normalize else 0,
                              stream_ptr(hidden.device)), "sc_pool_f16")
    return out


def gather_rows(ids, table, out=None):
    """out[r] = table[ids[r]] (fp16 rows; negative ids give zero rows)."""
    _require_cuda(ids, table)
    lib = _lib.load()
    ids = ids.to(torch.int32).contiguous().view(-1)
    rows, H = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((rows, H), dtype=torch.float16, device=table.device)
    from ctypes import c_void_p
    with torch.cuda.device(table.device):
        check(lib.sc_gather_rows_f16(ptr(ids), ptr(table), c_void_p(out.data_ptr()), rows, H, out.stride(0), table.shape[0],
                                     stream_ptr(table.device)), "sc_gather_rows_f16")
    return out


def avgpool_tokens(x, r: int):
    """[B, P*P, D] fp16 ViT token maps -> [B, (P//r)**2, D]: r x r spatial mean (reference compress_spatial_features)."""
    _require_cuda(x)
    lib = _lib.load()
    x = x.contiguous()
    B, S, D = x.shape
    P = int(round(S ** 0.5))
    if P * P != S:
        raise StreamChatHipError(f"avgpool_tokens: {S} tokens are not a square grid")
    g = P // r
    out = torch.empty((B, g * g, D), dtype=torch.float16, device=x.device)
    from ctypes import c_void_p
    with torch.cuda.device(x.device):
        check(lib.sc_avgpool_tokens_f16(c_void_p(x.data_ptr()), c_void_p(out.data_ptr()), B, P, D, r, stream_ptr(x.device)), "sc_avgpool_tokens_f16")
    return out


def rope_(x, heads: int, Dh: int, theta: float, pos0: int = 0, positions=None):
    """in-place rotate-half RoPE on x [rows, >= heads*Dh] (row-strided view allowed)."""
    _require_cuda(x)
    lib = _lib.load()
    from ctypes import c_void_p
    pos = None if positions is None else positions.to(device=x.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(x.device):
        check(lib.sc_rope_f16(c_void_p(x.data_ptr()), x.stride(0), ptr(pos), pos0, x.shape[0], heads, Dh, c_float(theta),
                              stream_ptr(x.device)), "sc_rope_f16")
    return x


def gemv(w, x, bias=None, residual=None, epilogue: str = "none", out=None, out_f32: bool = False, out_row=None, rms_gamma=None, rms_eps=1e-6):
    """y[N] = w[N,K] @ x[K] (+ bias) (+ residual): batch-1 decode projection (weights streamed once)."""
    _require_cuda(w, x)
    lib = _lib.load()
    N, K = w.shape
    x = x.reshape(-1)
    n_out = N // 2 if epilogue == "swiglu" else N
    if out is None:
        out = torch.empty(n_out, dtype=torch.float32 if out_f32 else torch.float16, device=w.device)
    from ctypes import c_void_p
    with torch.cuda.device(w.device), _timed("k_gemv", 2.0 * N * K):
        if out_row is not None:                 # `out` is a [rows, ld] buffer; the row index lives on the device
            check(lib.sc_gemv_f16(ptr(w), ptr(x), ptr(bias), None if residual is None else ptr(residual.reshape(-1)), c_void_p(out.data_ptr()), N, K,
                                  EPI[epilogue], 0, ptr(out_row), out.stride(0), ptr(rms_gamma), c_float(rms_eps), stream_ptr(w.device)), "sc_gemv_f16")
        else:
            check(lib.sc_gemv_f16(ptr(w), ptr(x), ptr(bias), None if residual is None else ptr(residual.reshape(-1)), ptr(out.reshape(-1)), N, K,
                                  EPI[epilogue], 1 if out.dtype == torch.float32 else 0, None, 0, ptr(rms_gamma), c_float(rms_eps), stream_ptr(w.device)),
                  "sc_gemv_f16")
    return out


def decode_qkv(wq, wkv, bq, bkv, x, rms_gamma, rms_eps: float, q_out, cache, row_index, q_heads: int, kv_heads: int, Dh: int, theta: float):
    """one launch: q / k / v projections of ONE token (fused RMSNorm), RoPE of q and k at position row_index[0] (device int32),
    q -> q_out [q_heads*Dh], k | v -> cache[row_index[0]].  Bit-identical to gemv + gemv + rope_qk_row_."""
    _require_cuda(wq, wkv, x, q_out, cache, row_index)
    lib = _lib.load()
    from ctypes import c_void_p
    with torch.cuda.device(x.device), _timed("k_gemv", 2.0 * (wq.shape[0] + wkv.shape[0]) * wq.shape[1]):
        check(lib.sc_decode_qkv_f16(ptr(wq), ptr(wkv), ptr(bq), ptr(bkv), ptr(x.reshape(-1)), ptr(rms_gamma), c_float(rms_eps), c_void_p(q_out.data_ptr()),
                                    c_void_p(cache.data_ptr()), cache.stride(0), ptr(row_index), q_heads, kv_heads, Dh, wq.shape[1], c_float(theta),
                                    stream_ptr(x.device)), "sc_decode_qkv_f16")
    return q_out


def decode_qkv_tab(wq, wkv, bq, bkv, x, rms_gamma, rms_eps: float, q_out, cache, row_index, q_heads: int, kv_heads: int, Dh: int, tab_q, tab_k):
    """decode_qkv with the one-rounding rotary arithmetic of the prefill GEMM epilogue: RoPE of the fp32 sums with the fp32 tables (tab_q carries
    the softmax scale * log2 e: q_out is pre-scaled), rounded once.  Bit-identical to gemv(out_f32) + rope_f32in."""
    _require_cuda(wq, wkv, x, q_out, cache, row_index, tab_q, tab_k)
    lib = _lib.load()
    from ctypes import c_void_p
    with torch.cuda.device(x.device), _timed("k_gemv", 2.0 * (wq.shape[0] + wkv.shape[0]) * wq.shape[1]):
        check(lib.sc_decode_qkv_tab_f16(ptr(wq), ptr(wkv), ptr(bq), ptr(bkv), ptr(x.reshape(-1)), ptr(rms_gamma), c_float(rms_eps), c_void_p(q_out.data_ptr()),
                                        c_void_p(cache.data_ptr()), cache.stride(0), ptr(row_index), q_heads, kv_heads, Dh, wq.shape[1],
                                        c_void_p(tab_q.data_ptr()), c_void_p(tab_k.data_ptr()), int(min(tab_q.shape[0], tab_k.shape[0])), stream_ptr(x.device)),
              "sc_decode_qkv_tab_f16")
    return q_out


def decode_advance(nxt, ring, cnt, tok, pos, kv_len, nprev):
    """one launch: ring[cnt] = nxt; tok = nxt; cnt, pos, kv_len, nprev += 1 (device scalars of a captured decode step)"""
    _require_cuda(nxt, ring, cnt, tok, pos, kv_len, nprev)
    if (nxt.dtype, ring.dtype, cnt.dtype, tok.dtype, pos.dtype, kv_len.dtype, nprev.dtype) != (torch.int64, torch.int64, torch.int64, torch.int32, torch.int32, torch.int32, torch.int32):
        raise StreamChatHipError("decode_advance: dtypes must be int64 (next, ring, index) / int32 (token, pos, kv_len, n_prev)")
    with torch.cuda.device(nxt.device):
        check(_lib.load().sc_decode_advance(ptr(nxt), ptr(ring), ptr(cnt), ptr(tok), ptr(pos), ptr(kv_len), ptr(nprev), stream_ptr(nxt.device)), "sc_decode_advance")


def rope_qk_row_(q, q_heads: int, cache, row_index, kv_heads: int, Dh: int, theta: float):
    """decode step: in-place RoPE of the query row `q` [q_heads*Dh] and of the K part of cache row `row_index` (device int32) - one launch."""
    _require_cuda(q, cache, row_index)
    lib = _lib.load()
    from ctypes import c_void_p
    with torch.cuda.device(q.device):
        check(lib.sc_rope_qk_row_f16(c_void_p(q.data_ptr()), q_heads, c_void_p(cache.data_ptr()), cache.stride(0), ptr(row_index), kv_heads, Dh,
                                     c_float(theta), stream_ptr(q.device)), "sc_rope_qk_row_f16")
    return q


def rope_row_(buf, row_index, heads: int, Dh: int, theta: float):
    """in-place RoPE on ONE row of `buf` [rows, ld]; the row number (= the token position) is read from the device int32 tensor
    `row_index` — hipGraph-replayable KV-cache append."""
    _require_cuda(buf, row_index)
    lib = _lib.load()
    from ctypes import c_void_p
    with torch.cuda.device(buf.device):
        check(lib.sc_rope_row_f16(c_void_p(buf.data_ptr()), buf.stride(0), ptr(row_index), heads, Dh, c_float(theta), stream_ptr(buf.device)),
              "sc_rope_row_f16")
    return buf
