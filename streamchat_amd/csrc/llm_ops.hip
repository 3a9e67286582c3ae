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
