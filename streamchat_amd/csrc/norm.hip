// Row normalisations (HBM-bound): LayerNorm (CLIP / BERT, fp32 statistics as torch.nn.LayerNorm does for
// fp16 inputs) and RMSNorm (Qwen2: x.float() -> x*rsqrt(mean(x^2)+eps) -> .to(fp16) -> * weight;
// transformers modeling_qwen2.py Qwen2RMSNorm).  One wave per row, 16-byte loads, the row stays in
// registers between the statistics pass and the write (one HBM read + one write per element).
#include "sc_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// NV = 16-byte vectors per lane (cols <= NV*512)
template <int NV, bool RMS>
__global__ __launch_bounds__(256) void k_norm(const _Float16* __restrict__ x, int ldx, const _Float16* __restrict__ gamma,
                                              const _Float16* __restrict__ beta, float eps, _Float16* __restrict__ y, int ldy, int rows,
                                              int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const _Float16* xr = x + (size_t)row * (size_t)ldx;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < cols) {
            sc_h8 h = *reinterpret_cast<const sc_h8*>(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)h[e]; s += RMS ? v[i][e] * v[i][e] : v[i][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < cols) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    _Float16* yr = y + (size_t)row * (size_t)ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < cols) {
            sc_h8 g = *reinterpret_cast<const sc_h8*>(gamma + c);
            sc_h8 o;
            if (RMS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const _Float16 t = (_Float16)(v[i][e] * rstd); o[e] = (_Float16)((float)g[e] * (float)t); }
            } else {
                sc_h8 b = *reinterpret_cast<const sc_h8*>(beta + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            }
            *reinterpret_cast<sc_h8*>(yr + c) = o;
        }
    }
}

// ViT token assembly + pre-LN: one wave per output token
template <int NV>
__global__ __launch_bounds__(256) void k_vit_embed_ln(const _Float16* __restrict__ patch, const _Float16* __restrict__ cls,
                                                      const _Float16* __restrict__ pos, const _Float16* __restrict__ gamma,
                                                      const _Float16* __restrict__ beta, float eps, _Float16* __restrict__ out, int rows,
                                                      int P, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n = row / (P + 1), tkn = row - n * (P + 1);
    const _Float16* src = tkn == 0 ? cls : patch + ((size_t)n * P + (tkn - 1)) * (size_t)D;
    const _Float16* pr = pos + (size_t)tkn * D;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < D) {
            const sc_h8 a = *reinterpret_cast<const sc_h8*>(src + c);
            const sc_h8 b = *reinterpret_cast<const sc_h8*>(pr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)(_Float16)(a[e] + b[e]); s += v[i][e]; }   // fp16 add like HF (fp16 embeddings)
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)D + eps);
    _Float16* yr = out + (size_t)row * (size_t)D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < D) {
            const sc_h8 g = *reinterpret_cast<const sc_h8*>(gamma + c);
            const sc_h8 b = *reinterpret_cast<const sc_h8*>(beta + c);
            sc_h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            *reinterpret_cast<sc_h8*>(yr + c) = o;
        }
    }
}

// BERT embeddings: word[ids] + position[pos] + token_type[0] -> LayerNorm (HF BertEmbeddings); one wave per token
template <int NV>
__global__ __launch_bounds__(256) void k_bert_embed_ln(const int* __restrict__ ids, const _Float16* __restrict__ word,
                                                       const _Float16* __restrict__ pos, const _Float16* __restrict__ type0,
                                                       const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta, float eps,
                                                       _Float16* __restrict__ out, int rows, int L, int H, int vocab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const _Float16* wr = word + (size_t)id * H;
    const _Float16* pr = pos + (size_t)(row % L) * H;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < H) {
            const sc_h8 a = *reinterpret_cast<const sc_h8*>(wr + c);
            const sc_h8 b = *reinterpret_cast<const sc_h8*>(pr + c);
            const sc_h8 t = *reinterpret_cast<const sc_h8*>(type0 + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)a[e] + (float)t[e] + (float)b[e]; s += v[i][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < H) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)H + eps);
    _Float16* yr = out + (size_t)row * (size_t)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < H) {
            const sc_h8 g = *reinterpret_cast<const sc_h8*>(gamma + c);
            const sc_h8 b = *reinterpret_cast<const sc_h8*>(beta + c);
            sc_h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((v[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            *reinterpret_cast<sc_h8*>(yr + c) = o;
        }
    }
}

// sentence pooling: mode 0 = CLS row, 1 = masked mean over the first len[b] tokens; optional L2 normalisation; fp32 out
__global__ __launch_bounds__(256) void k_pool(const _Float16* __restrict__ h, const int* __restrict__ len, float* __restrict__ out, int L, int H,
                                              int mode, int normalize) {
    const int b = blockIdx.x;
    __shared__ float red[4];
    const _Float16* hb = h + (size_t)b * L * H;
    const int n = mode == 0 ? 1 : (len ? (len[b] < L ? len[b] : L) : L);
    float ss = 0.f;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float a = 0.f;
        for (int t = 0; t < n; ++t) a += (float)hb[(size_t)t * H + c];
        a = mode == 0 ? a : a / fmaxf((float)n, 1e-9f);
        out[(size_t)b * H + c] = a;
        ss += a * a;
    }
    if (!normalize) return;
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
    for (int c = threadIdx.x; c < H; c += blockDim.x) out[(size_t)b * H + c] *= inv;
}

template <bool RMS>
int launch_norm(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* y, int ldy, int rows, int cols,
                hipStream_t s, const char* name) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nv = (cols + 511) / 512;
#define SC_NORM(NV)                                                                                                         \
    hipLaunchKernelGGL((k_norm<NV, RMS>), grid, block, 0, s, (const _Float16*)x, ldx, (const _Float16*)gamma, (const _Float16*)beta, eps, \
                       (_Float16*)y, ldy, rows, cols)
    if (nv <= 1) SC_NORM(1);
    else if (nv <= 2) SC_NORM(2);
    else if (nv <= 4) SC_NORM(4);
    else if (nv <= 8) SC_NORM(8);
    else return sc_fail(SC_ERR_UNSUPPORTED, "%s: cols %d > 4096 unsupported", name, cols);
#undef SC_NORM
    SC_CHECK_LAUNCH(name);
    return SC_OK;
}

int check_norm_args(const void* x, int ldx, const void* gamma, void* y, int ldy, int rows, int cols, const char* name) {
    SC_REQUIRE(x && gamma && y, "%s: null pointer argument", name);
    SC_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0, "%s: cols must be a positive multiple of 8", name);
    SC_REQUIRE(ldx >= cols && ldy >= cols && ldx % 8 == 0 && ldy % 8 == 0, "%s: leading dimensions must be >= cols and multiples of 8", name);
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma)) & 15) == 0,
               "%s: pointers must be 16-byte aligned", name);
    return SC_OK;
}

}  // namespace

extern "C" int sc_layernorm_f16(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* y, int ldy, int rows,
                                int cols, sc_stream_t stream) {
    if (int rc = check_norm_args(x, ldx, gamma, y, ldy, rows, cols, "sc_layernorm_f16")) return rc;
    SC_REQUIRE(beta && (reinterpret_cast<uintptr_t>(beta) & 15) == 0, "sc_layernorm_f16: beta must be non-null and 16-byte aligned");
    return launch_norm<false>(x, ldx, gamma, beta, eps, y, ldy, rows, cols, (hipStream_t)stream, "sc_layernorm_f16");
}

extern "C" int sc_rmsnorm_f16(const void* x, int ldx, const void* gamma, float eps, void* y, int ldy, int rows, int cols,
                              sc_stream_t stream) {
    if (int rc = check_norm_args(x, ldx, gamma, y, ldy, rows, cols, "sc_rmsnorm_f16")) return rc;
    return launch_norm<true>(x, ldx, gamma, nullptr, eps, y, ldy, rows, cols, (hipStream_t)stream, "sc_rmsnorm_f16");
}

extern "C" int sc_vit_embed_ln_f16(const void* patch, const void* cls, const void* pos, const void* gamma, const void* beta, float eps,
                                   void* out, int N, int P, int D, sc_stream_t stream) {
    SC_REQUIRE(patch && cls && pos && gamma && beta && out, "sc_vit_embed_ln_f16: null pointer argument");
    SC_REQUIRE(N > 0 && P > 0 && D > 0 && D % 8 == 0 && D <= 4096, "sc_vit_embed_ln_f16: need N, P > 0 and D a multiple of 8, <= 4096");
    const int rows = N * (P + 1);
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nv = (D + 511) / 512;
    hipStream_t s = (hipStream_t)stream;
#define SC_EMB(NV) hipLaunchKernelGGL((k_vit_embed_ln<NV>), grid, block, 0, s, (const _Float16*)patch, (const _Float16*)cls, (const _Float16*)pos, \
                                      (const _Float16*)gamma, (const _Float16*)beta, eps, (_Float16*)out, rows, P, D)
    if (nv <= 1) SC_EMB(1); else if (nv <= 2) SC_EMB(2); else if (nv <= 4) SC_EMB(4); else SC_EMB(8);
#undef SC_EMB
    SC_CHECK_LAUNCH("sc_vit_embed_ln_f16");
    return SC_OK;
}

extern "C" int sc_bert_embed_ln_f16(const int32_t* ids, const void* word, const void* pos, const void* type0, const void* gamma,
                                    const void* beta, float eps, void* out, int B, int L, int H, int vocab, sc_stream_t stream) {
    SC_REQUIRE(ids && word && pos && type0 && gamma && beta && out, "sc_bert_embed_ln_f16: null pointer argument");
    SC_REQUIRE(B > 0 && L > 0 && H > 0 && H % 8 == 0 && H <= 4096 && vocab > 0, "sc_bert_embed_ln_f16: bad sizes");
    const int rows = B * L;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    const int nv = (H + 511) / 512;
    hipStream_t s = (hipStream_t)stream;
#define SC_EMB(NV) hipLaunchKernelGGL((k_bert_embed_ln<NV>), grid, block, 0, s, ids, (const _Float16*)word, (const _Float16*)pos, (const _Float16*)type0, \
                                      (const _Float16*)gamma, (const _Float16*)beta, eps, (_Float16*)out, rows, L, H, vocab)
    if (nv <= 1) SC_EMB(1); else if (nv <= 2) SC_EMB(2); else if (nv <= 4) SC_EMB(4); else SC_EMB(8);
#undef SC_EMB
    SC_CHECK_LAUNCH("sc_bert_embed_ln_f16");
    return SC_OK;
}

extern "C" int sc_pool_f16(const void* hidden, const int32_t* len, float* out, int B, int L, int H, int mode, int normalize,
                           sc_stream_t stream) {
    SC_REQUIRE(hidden && out, "sc_pool_f16: null pointer argument");
    SC_REQUIRE(B > 0 && L > 0 && H > 0 && (mode == 0 || mode == 1), "sc_pool_f16: bad sizes / mode");
    hipLaunchKernelGGL(k_pool, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)hidden, len, out, L, H, mode, normalize);
    SC_CHECK_LAUNCH("sc_pool_f16");
    return SC_OK;
}
