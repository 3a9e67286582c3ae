// Small HBM-bound pieces of the Qwen2 path (reference llava_qwen.py:137-155 -> transformers Qwen2ForCausalLM, and the
// embedding splice of llava_arch.py:208-343): token-embedding row gather, rotary position embedding.
#include "sc_common.h"

namespace {

// out[r] = table[ids[r]]  (ids < 0 -> zeros: the -200 image sentinel rows are filled by the splice afterwards)
__global__ __launch_bounds__(256) void k_gather_rows(const int* __restrict__ ids, const _Float16* __restrict__ table, _Float16* __restrict__ out,
                                                     int rows, int H, int ldo, int vocab) {
    const int per_row = H / 8;
    const size_t total = (size_t)rows * per_row;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(g / per_row), c = (int)(g - (size_t)r * per_row) * 8;
        const int id = ids[r];
        sc_h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (id >= 0 && id < vocab) v = *reinterpret_cast<const sc_h8*>(table + (size_t)id * H + c);
        *reinterpret_cast<sc_h8*>(out + (size_t)r * ldo + c) = v;
    }
}

// Rotate-half RoPE in place on `heads` heads of width Dh starting at column 0 of each row (row stride ld):
//   x'[i] = x[i]*cos[i] - x[i+Dh/2]*sin[i];  x'[i+Dh/2] = x[i+Dh/2]*cos[i] + x[i]*sin[i],  i < Dh/2
// cos/sin = fp16(cos/sin(pos * theta^(-2i/Dh))) as HF builds them (fp32 trig, cast to the model dtype); products and the sum
// are rounded to fp16 like the fp16 tensor ops of apply_rotary_pos_emb.
__global__ __launch_bounds__(256) void k_rope(_Float16* __restrict__ x, int ld, const int* __restrict__ pos, int pos0, int rows, int heads, int Dh,
                                              float log2_theta) {
    const int half = Dh / 2;
    const int per_row = heads * (half / 4);            // 4 rotation pairs per thread (8-byte accesses)
    const size_t total = (size_t)rows * per_row;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(g / per_row);
        const int rem = (int)(g - (size_t)r * per_row);
        const int h = rem / (half / 4), i0 = (rem - h * (half / 4)) * 4;
        const float p = (float)(pos ? pos[r] : pos0 + r);
        _Float16* base = x + (size_t)r * ld + h * Dh;
        sc_h4 a = *reinterpret_cast<const sc_h4*>(base + i0);
        sc_h4 b = *reinterpret_cast<const sc_h4*>(base + half + i0);
        sc_h4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float inv_freq = exp2f(-log2_theta * (float)(2 * (i0 + e)) / (float)Dh);
            float sn, cs;
            sincosf(p * inv_freq, &sn, &cs);
            const _Float16 c16 = (_Float16)cs, s16 = (_Float16)sn;
            const _Float16 t1 = (_Float16)((float)a[e] * (float)c16), t2 = (_Float16)((float)b[e] * (float)s16);
            const _Float16 t3 = (_Float16)((float)b[e] * (float)c16), t4 = (_Float16)((float)a[e] * (float)s16);
            oa[e] = (_Float16)((float)t1 - (float)t2);
            ob[e] = (_Float16)((float)t3 + (float)t4);
        }
        *reinterpret_cast<sc_h4*>(base + i0) = oa;
        *reinterpret_cast<sc_h4*>(base + half + i0) = ob;
    }
}

// single-row variant for decode: row (= position) read from device memory
__global__ __launch_bounds__(256) void k_rope_row(_Float16* __restrict__ x, int ld, const int* __restrict__ row_index, int heads, int Dh, float log2_theta) {
    const int half = Dh / 2, per_row = heads * (half / 4);
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= per_row) return;
    const int row = row_index[0];
    const int h = rem / (half / 4), i0 = (rem - h * (half / 4)) * 4;
    const float p = (float)row;
    _Float16* base = x + (size_t)row * ld + h * Dh;
    sc_h4 a = *reinterpret_cast<const sc_h4*>(base + i0);
    sc_h4 b = *reinterpret_cast<const sc_h4*>(base + half + i0);
    sc_h4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float inv_freq = exp2f(-log2_theta * (float)(2 * (i0 + e)) / (float)Dh);
        float sn, cs;
        sincosf(p * inv_freq, &sn, &cs);
        const _Float16 c16 = (_Float16)cs, s16 = (_Float16)sn;
        const _Float16 t1 = (_Float16)((float)a[e] * (float)c16), t2 = (_Float16)((float)b[e] * (float)s16);
        const _Float16 t3 = (_Float16)((float)b[e] * (float)c16), t4 = (_Float16)((float)a[e] * (float)s16);
        oa[e] = (_Float16)((float)t1 - (float)t2);
        ob[e] = (_Float16)((float)t3 + (float)t4);
    }
    *reinterpret_cast<sc_h4*>(base + i0) = oa;
    *reinterpret_cast<sc_h4*>(base + half + i0) = ob;
}


// decode step: RoPE of the new query row (position row_index[0]) and of the K part of cache row row_index[0] in ONE launch
// (each of the two separate launches is ~4.5 us of pure latency inside the captured decode graph)
__global__ __launch_bounds__(256) void k_rope_qk_row(_Float16* __restrict__ q, int q_heads, _Float16* __restrict__ cache, int ld,
                                                     const int* __restrict__ row_index, int kv_heads, int Dh, float log2_theta) {
    const int half = Dh / 2, per_q = q_heads * (half / 4), per_k = kv_heads * (half / 4);
    int rem = blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= per_q + per_k) return;
    const int row = row_index[0];
    _Float16* x = q;
    if (rem >= per_q) { rem -= per_q; x = cache + (size_t)row * ld; }
    const int h = rem / (half / 4), i0 = (rem - h * (half / 4)) * 4;
    const float p = (float)row;
    _Float16* base = x + h * Dh;
    sc_h4 a = *reinterpret_cast<const sc_h4*>(base + i0);
    sc_h4 b = *reinterpret_cast<const sc_h4*>(base + half + i0);
    sc_h4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float inv_freq = exp2f(-log2_theta * (float)(2 * (i0 + e)) / (float)Dh);
        float sn, cs;
        sincosf(p * inv_freq, &sn, &cs);
        const _Float16 c16 = (_Float16)cs, s16 = (_Float16)sn;
        const _Float16 t1 = (_Float16)((float)a[e] * (float)c16), t2 = (_Float16)((float)b[e] * (float)s16);
        const _Float16 t3 = (_Float16)((float)b[e] * (float)c16), t4 = (_Float16)((float)a[e] * (float)s16);
        oa[e] = (_Float16)((float)t1 - (float)t2);
        ob[e] = (_Float16)((float)t3 + (float)t4);
    }
    *reinterpret_cast<sc_h4*>(base + i0) = oa;
    *reinterpret_cast<sc_h4*>(base + half + i0) = ob;
}

// ---- RoPE from fp32 projections, ONE rounding (round 3) -------------------------------------------------------------------------
// tab[pos][0][j] = cos(pos * theta^(-2j/Dh)) * scale, tab[pos][1][j] = sin(...) * scale, j < Dh/2 (fp32).  `scale` folds the softmax
// scale * log2 e into the QUERY table: q' = rope(q) * scale is what sc_attention_f16 takes with SC_ATTN_Q_PRESCALED and needs no
// per-score multiply; the key table has scale 1.  The rotation is applied to the fp32 accumulators + bias of the projection (GEMM
// epilogue, the decode GEMV, or k_rope_f32in below for the small-M paths) and the result is rounded to fp16 ONCE - HF's fp16 path rounds
// the projection, cos / sin, both products and the sum (what k_rope above reproduces); against an fp32 reference this is the closer one.
__global__ __launch_bounds__(256) void k_rope_table(float* __restrict__ tab, int max_pos, int half, float log2_theta, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_pos * half) return;
    const int pos = i / half, j = i - pos * half;
    const float inv_freq = exp2f(-log2_theta * (float)(2 * j) / (float)(2 * half));
    float sn, cs;
    sincosf((float)pos * inv_freq, &sn, &cs);
    tab[(size_t)pos * 2 * half + j] = cs * scale;
    tab[(size_t)pos * 2 * half + half + j] = sn * scale;
}

// x [rows, ldx] fp32 = projection + bias -> out [rows, ldo] fp16: the first `heads` heads of width Dh rotated with the table row of the
// row's position (positions[r] or pos0 + r), then `plain` further columns copied with a cast (the V part of a k|v projection)
__global__ __launch_bounds__(256) void k_rope_f32in(const float* __restrict__ x, int ldx, const float* __restrict__ tab, int tab_rows, const int* __restrict__ pos, int pos0,
                                                    int rows, int heads, int Dh, int plain, _Float16* __restrict__ out, int ldo) {
    const int half = Dh / 2;
    const int per_row = heads * (half / 4) + plain / 4;
    const size_t total = (size_t)rows * per_row;
    for (size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(gi / per_row);
        int rem = (int)(gi - (size_t)r * per_row);
        const float* xr = x + (size_t)r * ldx;
        _Float16* orow = out + (size_t)r * ldo;
        if (rem >= heads * (half / 4)) {
            const int c = heads * Dh + (rem - heads * (half / 4)) * 4;
            const sc_f4 v = *reinterpret_cast<const sc_f4*>(xr + c);
            *reinterpret_cast<sc_h4*>(orow + c) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            continue;
        }
        const int h = rem / (half / 4), i0 = (rem - h * (half / 4)) * 4;
        int p = pos ? pos[r] : pos0 + r;
        p = p < 0 ? 0 : (p >= tab_rows ? tab_rows - 1 : p);           // device-side positions are clamped to the table (host-side ones are checked before the launch)
        const float* t = tab + (size_t)p * Dh;
        const sc_f4 a = *reinterpret_cast<const sc_f4*>(xr + h * Dh + i0), b = *reinterpret_cast<const sc_f4*>(xr + h * Dh + half + i0);
        const sc_f4 cs = *reinterpret_cast<const sc_f4*>(t + i0), sn = *reinterpret_cast<const sc_f4*>(t + half + i0);
        sc_h4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ra = __builtin_fmaf(-b[e], sn[e], a[e] * cs[e]), rb = __builtin_fmaf(a[e], sn[e], b[e] * cs[e]);
            asm volatile("" : "+v"(ra), "+v"(rb));                  // fp32 fma, THEN the fp16 rounding: never one v_fma_mix (see k_decode_qkv, gemv.hip)
            oa[e] = (_Float16)ra;
            ob[e] = (_Float16)rb;
        }
        *reinterpret_cast<sc_h4*>(orow + h * Dh + i0) = oa;
        *reinterpret_cast<sc_h4*>(orow + h * Dh + half + i0) = ob;
    }
}

// The three launches behind a BATCHED decode step's q|k|v projection in one (round 5, ABI 6): x [B, (Hq + 2 Hkv) Dh] fp32 = fused projection +
// bias of one new token per sequence.  q heads rotated with tab_q (scaled table) -> q_out [B, Hq Dh]; k heads rotated with tab_k and the v
// columns cast -> row pos[b] of sequence b's KV cache (cache + b * cache_bs + pos[b] * cache_ld).  Element for element the arithmetic of
// k_rope_f32in (same fma sequence, one rounding): bit-identical to sc_rope_f32in_f16 x 2 + a row scatter.
__global__ __launch_bounds__(256) void k_rope_qkv_rows(const float* __restrict__ x, int ldx, const float* __restrict__ tab_q, const float* __restrict__ tab_k, int tab_rows,
                                                       const int* __restrict__ pos, int B, int Hq, int Hkv, int Dh, _Float16* __restrict__ q_out, int ldq,
                                                       _Float16* __restrict__ cache, long cache_bs, int cache_ld, int cache_rows) {
    const int half = Dh / 2, hq4 = half / 4;
    const int per_row = (Hq + Hkv) * hq4 + Hkv * Dh / 4;
    const int total = B * per_row;
    for (int gi = blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += gridDim.x * blockDim.x) {
        const int b = gi / per_row;
        int rem = gi - b * per_row;
        const float* xr = x + (size_t)b * ldx;
        int p = pos[b];
        const int prow = p < 0 ? 0 : (p >= cache_rows ? cache_rows - 1 : p);       // device-side positions are clamped (cache row and table row)
        p = p < 0 ? 0 : (p >= tab_rows ? tab_rows - 1 : p);
        _Float16* crow = cache + (size_t)b * (size_t)cache_bs + (size_t)prow * (size_t)cache_ld;
        if (rem >= (Hq + Hkv) * hq4) {                                             // v: cast only
            const int c = (rem - (Hq + Hkv) * hq4) * 4;
            const sc_f4 v = *reinterpret_cast<const sc_f4*>(xr + (Hq + Hkv) * Dh + c);
            *reinterpret_cast<sc_h4*>(crow + Hkv * Dh + c) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            continue;
        }
        const int h = rem / hq4, i0 = (rem - h * hq4) * 4;
        const bool isq = h < Hq;
        const float* t = (isq ? tab_q : tab_k) + (size_t)p * Dh;
        const sc_f4 a = *reinterpret_cast<const sc_f4*>(xr + h * Dh + i0), bb = *reinterpret_cast<const sc_f4*>(xr + h * Dh + half + i0);
        const sc_f4 cs = *reinterpret_cast<const sc_f4*>(t + i0), sn = *reinterpret_cast<const sc_f4*>(t + half + i0);
        sc_h4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ra = __builtin_fmaf(-bb[e], sn[e], a[e] * cs[e]), rb = __builtin_fmaf(a[e], sn[e], bb[e] * cs[e]);
            asm volatile("" : "+v"(ra), "+v"(rb));                  // fp32 fma, THEN the fp16 rounding (see k_rope_f32in)
            oa[e] = (_Float16)ra;
            ob[e] = (_Float16)rb;
        }
        _Float16* o = isq ? q_out + (size_t)b * ldq + h * Dh : crow + (h - Hq) * Dh;
        *reinterpret_cast<sc_h4*>(o + i0) = oa;
        *reinterpret_cast<sc_h4*>(o + half + i0) = ob;
    }
}

// out[b, y*g + x, d0..d0+7] = mean of in[b, (y*r + dy)*P + (x*r + dx), d0..d0+7] over the r x r window (fp32 accumulation)
__global__ void k_avgpool_tokens(const _Float16* __restrict__ in, _Float16* __restrict__ out, int P, int D, int r, int g, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;        // one thread = 8 channels of one output token
    if (i >= total) return;
    const int dv = D / 8;
    const int d0 = (int)(i % dv) * 8;
    const long tok = i / dv;
    const int x = (int)(tok % g), y = (int)((tok / g) % g);
    const long b = tok / ((long)g * g);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < r; ++dy)
        for (int dx = 0; dx < r; ++dx) {
            const sc_h8 v = *reinterpret_cast<const sc_h8*>(in + ((b * P + (y * r + dy)) * P + (x * r + dx)) * (long)D + d0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    const float inv = 1.0f / (float)(r * r);
    sc_h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[e] * inv);
    *reinterpret_cast<sc_h8*>(out + tok * (long)D + d0) = o;
}
}  // namespace

extern "C" int sc_rope_row_f16(void* buf, int ld, const int32_t* row_index, int heads, int Dh, float theta, sc_stream_t stream) {
    SC_REQUIRE(buf && row_index, "sc_rope_row_f16: null pointer argument");
    SC_REQUIRE(heads > 0 && Dh > 0 && Dh % 8 == 0 && ld >= heads * Dh && ld % 4 == 0 && theta > 1.f, "sc_rope_row_f16: bad sizes");
    const int per_row = heads * (Dh / 8);
    hipLaunchKernelGGL(k_rope_row, dim3((unsigned)((per_row + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (_Float16*)buf, ld, row_index, heads, Dh,
                       log2f(theta));
    SC_CHECK_LAUNCH("sc_rope_row_f16");
    return SC_OK;
}

extern "C" int sc_gather_rows_f16(const int32_t* ids, const void* table, void* out, int rows, int H, int ldo, int vocab, sc_stream_t stream) {
    SC_REQUIRE(ids && table && out, "sc_gather_rows_f16: null pointer argument");
    SC_REQUIRE(rows > 0 && H > 0 && H % 8 == 0 && ldo >= H && ldo % 8 == 0 && vocab > 0, "sc_gather_rows_f16: bad sizes");
    const size_t total = (size_t)rows * (H / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(k_gather_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, ids, (const _Float16*)table, (_Float16*)out, rows, H, ldo, vocab);
    SC_CHECK_LAUNCH("sc_gather_rows_f16");
    return SC_OK;
}

extern "C" int sc_rope_f16(void* x, int ld, const int32_t* positions, int pos0, int rows, int heads, int Dh, float theta, sc_stream_t stream) {
    SC_REQUIRE(x, "sc_rope_f16: null pointer argument");
    SC_REQUIRE(rows > 0 && heads > 0 && Dh > 0 && Dh % 8 == 0 && ld >= heads * Dh && ld % 4 == 0 && theta > 1.f, "sc_rope_f16: bad sizes");
    const size_t total = (size_t)rows * heads * (Dh / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(k_rope, dim3(grid), dim3(256), 0, (hipStream_t)stream, (_Float16*)x, ld, positions, pos0, rows, heads, Dh, log2f(theta));
    SC_CHECK_LAUNCH("sc_rope_f16");
    return SC_OK;
}

extern "C" int sc_avgpool_tokens_f16(const void* in, void* out, int B, int P, int D, int r, sc_stream_t stream) {
    SC_REQUIRE(in && out, "sc_avgpool_tokens_f16: null pointer argument");
    SC_REQUIRE(B > 0 && P > 0 && D > 0 && D % 8 == 0 && r >= 1 && r <= P, "sc_avgpool_tokens_f16: bad sizes (D %% 8 == 0, 1 <= r <= P)");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "sc_avgpool_tokens_f16: pointers must be 16-byte aligned");
    const int g = P / r;
    const long total = (long)B * g * g * (D / 8);
    hipLaunchKernelGGL(k_avgpool_tokens, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in, (_Float16*)out, P, D, r, g, total);
    SC_CHECK_LAUNCH("sc_avgpool_tokens_f16");
    return SC_OK;
}

extern "C" int sc_rope_qk_row_f16(void* q, int q_heads, void* cache, int ld, const int32_t* row_index, int kv_heads, int Dh, float theta,
                                  sc_stream_t stream) {
    SC_REQUIRE(q && cache && row_index, "sc_rope_qk_row_f16: null pointer argument");
    SC_REQUIRE(q_heads > 0 && kv_heads > 0 && Dh > 0 && Dh % 8 == 0 && ld >= kv_heads * Dh && ld % 4 == 0 && theta > 1.f, "sc_rope_qk_row_f16: bad sizes");
    const int total = (q_heads + kv_heads) * (Dh / 8);
    hipLaunchKernelGGL(k_rope_qk_row, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (_Float16*)q, q_heads, (_Float16*)cache, ld,
                       row_index, kv_heads, Dh, log2f(theta));
    SC_CHECK_LAUNCH("sc_rope_qk_row_f16");
    return SC_OK;
}

extern "C" int sc_rope_table_f32(float* tab, int max_pos, int Dh, float theta, float scale, sc_stream_t stream) {
    SC_REQUIRE(tab && max_pos > 0 && Dh > 0 && Dh % 8 == 0 && theta > 1.f, "sc_rope_table_f32: bad arguments");
    const int n = max_pos * (Dh / 2);
    hipLaunchKernelGGL(k_rope_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tab, max_pos, Dh / 2, log2f(theta), scale);
    SC_CHECK_LAUNCH("sc_rope_table_f32");
    return SC_OK;
}

extern "C" int sc_rope_qkv_rows_f16(const float* x, int ldx, const float* tab_q, const float* tab_k, int tab_rows, const int32_t* positions, int B, int q_heads,
                                    int kv_heads, int Dh, void* q_out, int ldq, void* cache, int64_t cache_batch_stride, int cache_ld, int cache_rows, sc_stream_t stream) {
    SC_REQUIRE(x && tab_q && tab_k && positions && q_out && cache, "sc_rope_qkv_rows_f16: null pointer argument");
    SC_REQUIRE(B > 0 && q_heads > 0 && kv_heads > 0 && Dh > 0 && Dh % 8 == 0 && tab_rows > 0 && cache_rows > 0, "sc_rope_qkv_rows_f16: bad sizes (Dh %% 8 == 0)");
    SC_REQUIRE(ldx >= (q_heads + 2 * kv_heads) * Dh && ldx % 4 == 0 && ldq >= q_heads * Dh && ldq % 4 == 0 && cache_ld >= 2 * kv_heads * Dh && cache_ld % 4 == 0 &&
               cache_batch_stride % 4 == 0, "sc_rope_qkv_rows_f16: leading dimensions too small or not multiples of 4");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(tab_q) | reinterpret_cast<uintptr_t>(tab_k)) & 15) == 0 &&
               ((reinterpret_cast<uintptr_t>(q_out) | reinterpret_cast<uintptr_t>(cache)) & 7) == 0, "sc_rope_qkv_rows_f16: x / tables must be 16-byte aligned, outputs 8-byte");
    const long total = (long)B * ((q_heads + kv_heads) * (Dh / 8) + kv_heads * Dh / 4);
    hipLaunchKernelGGL(k_rope_qkv_rows, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, tab_q, tab_k, tab_rows,
                       positions, B, q_heads, kv_heads, Dh, (_Float16*)q_out, ldq, (_Float16*)cache, cache_batch_stride, cache_ld, cache_rows);
    SC_CHECK_LAUNCH("sc_rope_qkv_rows_f16");
    return SC_OK;
}

extern "C" int sc_rope_f32in_f16(const float* x, int ldx, const float* tab, int tab_rows, const int32_t* positions, int pos0, int rows, int heads, int Dh, int plain_cols,
                                 void* out, int ldo, sc_stream_t stream) {
    SC_REQUIRE(x && tab && out, "sc_rope_f32in_f16: null pointer argument");
    SC_REQUIRE(tab_rows > 0 && (positions || (pos0 >= 0 && (long long)pos0 + rows <= (long long)tab_rows)),
               "sc_rope_f32in_f16: positions %d..%lld exceed the rotary table (%d rows)", pos0, (long long)pos0 + rows - 1, tab_rows);
    SC_REQUIRE(rows > 0 && heads >= 0 && Dh > 0 && Dh % 8 == 0 && plain_cols >= 0 && plain_cols % 4 == 0 && ldx >= heads * Dh + plain_cols && ldx % 4 == 0 &&
                   ldo >= heads * Dh + plain_cols && ldo % 4 == 0,
               "sc_rope_f32in_f16: bad sizes");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(tab)) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
               "sc_rope_f32in_f16: x / table must be 16-byte aligned, out 8-byte");
    const size_t total = (size_t)rows * (heads * (Dh / 8) + plain_cols / 4);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(k_rope_f32in, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, tab, tab_rows, positions, pos0, rows, heads, Dh, plain_cols, (_Float16*)out, ldo);
    SC_CHECK_LAUNCH("sc_rope_f32in_f16");
    return SC_OK;
}
