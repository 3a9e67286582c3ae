// Decode-time projections: y[N] = W[N,K] . x[K] (+ bias) (+ residual), fp16 weights streamed ONCE from HBM (batch-1 decode
// is a pure weight stream: 14.1 GB per token for Qwen2-7B — SURVEY.md §8(d)); replaces the M = 1 calls of
// transformers' nn.Linear inside Qwen2ForCausalLM.generate (reference llava_qwen.py:155).
// One wave owns RPW = 4 output rows: per K-slice of 512 it loads 16 bytes of x per lane once (L1/L2 resident) and 16 bytes of
// each of its 4 weight rows (1 KiB coalesced per row), accumulates with v_dot2_f32_f16, and finishes with a wave reduction.
// No LDS: the operand is streamed once per block and never shared (cdna guide §5 "GEMV / M <= 16" row).
#include "sc_common.h"
#include <stdlib.h>

#ifndef SC_WPB_QKV
#define SC_WPB_QKV 4
#define SC_WPB_FEW 4
#define SC_WPB_LONG 7          // down projection: 3584 rows, 2 per wave -> 256 workgroups of 7 waves = one per CU (+0.8 % tok/s; the others: +-0.2 %)
#endif

namespace {

// Qwen2 RMSNorm of the activation vector, ONCE per workgroup, into LDS (fp16, HF's rounding: gamma * fp16(x * rstd)): the 256 threads
// share the sum of squares (fixed reduction order: 8-element chunks t, t + 256, ... per thread, xor-shuffle tree per wave, the four
// wave sums added in order) and each writes its chunks of the normalised vector.  Before (round 2) every WAVE recomputed the norm and
// re-normalised x in every K-step of its weight stream - ~40 VALU instructions per 16-byte weight load in an in-order loop, and the
// weight loads could not start before the wave's own reduction was done (k_decode_qkv: 2.6 TB/s, gate/up: 51 us against 44 without
// the fused norm).  Shared by k_gemv<NORM> and k_decode_qkv so that the eager decode step and the captured graph stay bit-identical.
__device__ __forceinline__ void block_rmsnorm_to_lds(const _Float16* __restrict__ x, const _Float16* __restrict__ gamma, int K, float eps, _Float16* xn,
                                                     float* red) {
    // The reduction order is that of 256 VIRTUAL threads whatever the workgroup size (2, 3 or 4 waves: the launchers pick the wave count
    // that deals the rows evenly over the CUs): virtual thread v sums the 8-element chunks v, v + 256, ...; a virtual wave is reduced
    // with the xor-shuffle tree; the four virtual-wave sums are added in order.
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    for (int vw = wave; vw < 4; vw += nw) {
        float ss = 0.f;
        for (int k = (vw * 64 + lane) * 8; k < K; k += 2048) {
            const sc_h8 xv = *reinterpret_cast<const sc_h8*>(x + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)xv[e] * (float)xv[e];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
        if (lane == 0) red[vw] = ss;
    }
    __syncthreads();
    const float rstd = rsqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)K + eps);
    for (int k = t * 8; k < K; k += (int)blockDim.x * 8) {
        const sc_h8 xv = *reinterpret_cast<const sc_h8*>(x + k), gv = *reinterpret_cast<const sc_h8*>(gamma + k);
        sc_h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)gv[e] * (float)(_Float16)((float)xv[e] * rstd));
        *reinterpret_cast<sc_h8*>(xn + k) = o;
    }
    __syncthreads();
}

template <bool SWIGLU, bool OUT_F32, int RPW, int UNR = 4>
__global__ __launch_bounds__(512) void k_gemv(      // 2 or 4 waves per workgroup (launcher's choice: whichever deals the rows evenly over the CUs)
const _Float16* __restrict__ W, const _Float16* __restrict__ x, const _Float16* __restrict__ bias,
                                              const _Float16* __restrict__ res, void* __restrict__ y_base, int N, int K,
                                              const int* __restrict__ y_row, int y_ld, const _Float16* __restrict__ gamma, float eps) {
    // optional dynamic output row (KV-cache append at a device-resident position: keeps a decode step hipGraph-replayable)
    void* y = y_row ? (void*)(reinterpret_cast<_Float16*>(y_base) + (size_t)y_row[0] * (size_t)y_ld) : y_base;
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (int)(blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW;
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
    const _Float16* wp[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) wp[r] = W + (size_t)(row0 + r < N ? row0 + r : N - 1) * (size_t)K + lane * 8;
    // optional fused Qwen2 RMSNorm of x (gamma != NULL): the workgroup normalises x once into LDS (block_rmsnorm_to_lds) and the weight
    // stream below reads the finished vector; the first weight loads of every row are requested BEFORE that, so that the norm's
    // load -> reduce -> barrier chain runs under their latency instead of in front of it
    // what the epilogue adds (bias, residual) is requested NOW: read at the end, these dependent scalar-sized loads were a full memory
    // latency each on the tail of every wave of a 6 us kernel
    float eb[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = row0 + r < N ? row0 + r : N - 1;
        eb[r] = (SWIGLU || row0 >= N) ? 0.f : ((bias ? (float)bias[n] : 0.f) + (res ? (float)res[n] : 0.f));
    }
    constexpr int PF = RPW >= 4 ? 2 : 4;                           // K-slices requested ahead of the norm
    _Float16* xn = reinterpret_cast<_Float16*>(gemv_smem);
    sc_h8 w0[PF][RPW];
    if (gamma) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                w0[u][r] = lane * 8 + u * 512 < K ? __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp[r] + u * 512)) : sc_h8{0, 0, 0, 0, 0, 0, 0, 0};
        block_rmsnorm_to_lds(x, gamma, K, eps, xn, reinterpret_cast<float*>(gemv_smem + (size_t)K * 2));
    }
    if (row0 >= N) return;                                        // (after the barriers of the norm)
    auto ldx = [&](int k) { return gamma ? *reinterpret_cast<const sc_h8*>(xn + k) : *reinterpret_cast<const sc_h8*>(x + k); };
    int kb = lane * 8;
    if (gamma) {                                                  // the K-slices requested ahead
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (kb < K) {
                const sc_h8 xv = *reinterpret_cast<const sc_h8*>(xn + kb);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const sc_h2 a = {w0[u][r][e], w0[u][r][e + 1]}, b = {xv[e], xv[e + 1]};
                        acc[r] = __builtin_amdgcn_fdot2(a, b, acc[r], false);
                    }
                }
                kb += 512;
            }
        }
    }
#pragma unroll UNR
    for (int k = kb; k < K; k += 512) {                             // (unrolled: >= 4 weight loads in flight per lane at RPW = 1)
        const sc_h8 xv = ldx(k);
        sc_h8 wv[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp[r] + (k - lane * 8)));
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const sc_h2 a = {wv[r][e], wv[r][e + 1]}, b = {xv[e], xv[e + 1]};
                acc[r] = __builtin_amdgcn_fdot2(a, b, acc[r], false);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[r] += __shfl_xor(acc[r], m, 64);
    }
    if (lane == 0) {
        if (SWIGLU) {          // rows (g g u u): two outputs per 4 rows (RPW == 4)
            const float g0 = acc[0], g1 = acc[RPW > 1 ? 1 : 0], u0 = acc[RPW > 2 ? 2 : 0], u1 = acc[RPW > 3 ? 3 : 0];
            _Float16* o = reinterpret_cast<_Float16*>(y) + (row0 >> 1);
            o[0] = (_Float16)(g0 / (1.0f + __expf(-g0)) * u0);
            o[1] = (_Float16)(g1 / (1.0f + __expf(-g1)) * u1);
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = row0 + r;
                if (n < N) {
                    float v = acc[r] + eb[r];
                    if (OUT_F32) reinterpret_cast<float*>(y)[n] = v;
                    else reinterpret_cast<_Float16*>(y)[n] = (_Float16)v;
                }
            }
        }
    }
}

// Fused decode-step projection block (one launch instead of three: q GEMV, kv GEMV, RoPE — each of the small ones is a few us of
// pure launch + latency inside the captured decode graph):
//   [q | k | v] = W . rmsnorm(x) + b;  rotate-half RoPE of q and k at position p = pos[0];  q -> q_out,  k | v -> cache row p.
// One wave owns the two rows (h, j) and (h, j + Dh/2) of a head, i.e. one RoPE rotation pair, so the rotation happens in the
// epilogue of lane 0 with the arithmetic of k_rope_qk_row (llm_ops.hip: fp32 trig, cos / sin and every product rounded to fp16 like
// HF's apply_rotary_pos_emb) on the fp16-rounded projections: bit-identical to the three-launch path.  V rows ride along unrotated.
// TAB: the rotation uses the fp32 cos / sin tables of sc_rope_table_f32 (tab_q carries the softmax scale * log2 e: the query leaves
// pre-scaled for sc_attention_f16's SC_ATTN_Q_PRESCALED mode) on the fp32 accumulators + bias, rounded to fp16 ONCE - the numerics of
// the prefill GEMM's rotary epilogue (gemm.hip) and of k_rope_f32in.
template <bool TAB>
__global__ __launch_bounds__(512) void k_decode_qkv(const _Float16* __restrict__ Wq, const _Float16* __restrict__ Wkv, const _Float16* __restrict__ bq,
                                                    const _Float16* __restrict__ bkv, const _Float16* __restrict__ x, const _Float16* __restrict__ gamma,
                                                    float eps, _Float16* __restrict__ q_out, _Float16* __restrict__ cache, int ld,
                                                    const int* __restrict__ pos, int Hq, int Hkv, int Dh, int K, float log2_theta,
                                                    const float* __restrict__ tab_q, const float* __restrict__ tab_k, int tab_rows) {
    const int lane = threadIdx.x & 63;
    const int half = Dh >> 1;
    const int task = blockIdx.x * (int)(blockDim.x >> 6) + (threadIdx.x >> 6);   // (head, j) over q heads, then k heads, then v heads
    const int ntask = (Hq + 2 * Hkv) * half;
    const bool live = task < ntask;                                       // (dead waves of the last workgroup still take part in the norm's barriers)
    const int hh = (live ? task : 0) / half, j = (live ? task : 0) - hh * half;
    const _Float16 *w0, *b0;
    int kind, h;                                                          // 0 = q, 1 = k, 2 = v
    if (hh < Hq) { kind = 0; h = hh; w0 = Wq + (size_t)(h * Dh + j) * (size_t)K; b0 = bq ? bq + h * Dh + j : nullptr; }
    else if (hh < Hq + Hkv) { kind = 1; h = hh - Hq; w0 = Wkv + (size_t)(h * Dh + j) * (size_t)K; b0 = bkv ? bkv + h * Dh + j : nullptr; }
    else { kind = 2; h = hh - Hq - Hkv; w0 = Wkv + (size_t)((Hkv + h) * Dh + j) * (size_t)K; b0 = bkv ? bkv + (Hkv + h) * Dh + j : nullptr; }
    const _Float16* w1 = w0 + (size_t)half * (size_t)K;
    // everything the epilogue needs besides the two dot products is requested NOW (position, bias pair, table row): read at the end, these
    // were four dependent memory latencies (~2 us) on the tail of every wave of a 10 us kernel
    const int row = pos[0];
    const float bias0 = (live && b0) ? (float)b0[0] : 0.f, bias1 = (live && b0) ? (float)b0[half] : 0.f;
    float tcs = 1.f, tsn = 0.f;
    if (TAB && live && kind != 2) {
        const int trow = row < 0 ? 0 : (row >= tab_rows ? tab_rows - 1 : row);      // the position lives in device memory: clamp to the table, never read past it
        const float* t = (kind == 0 ? tab_q : tab_k) + (size_t)trow * (size_t)Dh;
        tcs = t[j]; tsn = t[half + j];
    }
    // RMSNorm of x once per workgroup into LDS; the first four K-slices of both weight rows are requested before it (see k_gemv)
    extern __shared__ __attribute__((aligned(16))) char qkv_smem[];
    _Float16* xn = reinterpret_cast<_Float16*>(qkv_smem);
#ifndef SC_QKV_PF
#define SC_QKV_PF 4
#endif
    constexpr int PF = SC_QKV_PF;                  // K-slices of both rows requested before the norm (8 = the whole row at K <= 4096)
    sc_h8 pa[PF], pb[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int k = lane * 8 + u * 512;
        pa[u] = (live && k < K) ? __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(w0 + k)) : sc_h8{0, 0, 0, 0, 0, 0, 0, 0};
        pb[u] = (live && k < K) ? __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(w1 + k)) : sc_h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (gamma) block_rmsnorm_to_lds(x, gamma, K, eps, xn, reinterpret_cast<float*>(qkv_smem + (size_t)K * 2));
    if (!live) return;
    auto ldx = [&](int k) { return gamma ? *reinterpret_cast<const sc_h8*>(xn + k) : *reinterpret_cast<const sc_h8*>(x + k); };
    float acc0 = 0.f, acc1 = 0.f;
    int kb = lane * 8;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (kb < K) {
            const sc_h8 xv = ldx(kb);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const sc_h2 b = {xv[e], xv[e + 1]};
                acc0 = __builtin_amdgcn_fdot2(sc_h2{pa[u][e], pa[u][e + 1]}, b, acc0, false);
                acc1 = __builtin_amdgcn_fdot2(sc_h2{pb[u][e], pb[u][e + 1]}, b, acc1, false);
            }
            kb += 512;
        }
    }
#pragma unroll 4
    for (int k = kb; k < K; k += 512) {
        const sc_h8 xv = ldx(k);
        const sc_h8 wa = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(w0 + k));
        const sc_h8 wb = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(w1 + k));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const sc_h2 b = {xv[e], xv[e + 1]};
            acc0 = __builtin_amdgcn_fdot2(sc_h2{wa[e], wa[e + 1]}, b, acc0, false);
            acc1 = __builtin_amdgcn_fdot2(sc_h2{wb[e], wb[e + 1]}, b, acc1, false);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { acc0 += __shfl_xor(acc0, m, 64); acc1 += __shfl_xor(acc1, m, 64); }
    if (lane == 0) {
        const _Float16 a = (_Float16)(acc0 + bias0), b = (_Float16)(acc1 + bias1);
        _Float16* dst = kind == 0 ? q_out + h * Dh + j : cache + (size_t)row * (size_t)ld + (kind == 1 ? 0 : Hkv * Dh) + h * Dh + j;
        if (kind == 2) { dst[0] = a; dst[half] = b; return; }
        if (TAB) {
            const float a32 = acc0 + bias0, b32 = acc1 + bias1, cs = tcs, sn = tsn;
            // fp32 fma FIRST, then the rounding to fp16 - as k_rope_f32in and the GEMM epilogue do it: left alone, hipcc fuses the two into
            // v_fma_mixlo_f16 (ONE rounding of the exact product-sum), which differs from fma -> cvt in the last fp16 bit on ties and made the
            // captured decode graph and the eager step disagree by one ulp in a K row every few tokens
            float ra = __builtin_fmaf(-b32, sn, a32 * cs), rb = __builtin_fmaf(a32, sn, b32 * cs);
            asm volatile("" : "+v"(ra), "+v"(rb));
            dst[0] = (_Float16)ra;
            dst[half] = (_Float16)rb;
            return;
        }
        const float inv_freq = exp2f(-log2_theta * (float)(2 * j) / (float)Dh);
        float sn, cs;
        sincosf((float)row * inv_freq, &sn, &cs);
        const _Float16 c16 = (_Float16)cs, s16 = (_Float16)sn;
        const _Float16 t1 = (_Float16)((float)a * (float)c16), t2 = (_Float16)((float)b * (float)s16);
        const _Float16 t3 = (_Float16)((float)b * (float)c16), t4 = (_Float16)((float)a * (float)s16);
        dst[0] = (_Float16)((float)t1 - (float)t2);
        dst[half] = (_Float16)((float)t3 + (float)t4);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: shape-specialised decode projections - K = NS * 512 and the workgroup size are template parameters, the kernel body is
// straight-line code (every loop below unrolls completely), all loads are ordinary loads in the program order the queue should have.
// What the generic kernels above leave to hipcc, and what it does with it (the .s of the round-4 library):
//  * `#pragma unroll N` over the weight loop with a run-time trip count: the trip % N remainder is peeled into a one-slice-per-trip loop
//    IN FRONT of the unrolled body - load, s_waitcnt vmcnt(0), dot, repeat: the down projection (37 slices, N = 8) opened every wave
//    with five fully serialised memory round trips, the o projection (7 slices, N = 4) ran three; inside the unrolled body the
//    activation slice is loaded BEHIND the weights it multiplies (vmcnt retires in order: the first dot drains the whole batch) and no
//    batch is requested before the previous one has been consumed;
//  * `bias ? bias[n] : 0` / `res ? res[n] : 0`: each load in its own branch with `s_waitcnt vmcnt(0)` behind it - one (o) or two (down)
//    round trips before the first weight byte is requested;
//  * the fused RMSNorm reads gamma only after its first barrier (a second dependent latency), reads x twice, and - in-order again -
//    cannot see x before the weight slices requested ahead of it have landed.
// Two attempts to fix this inside the generic kernels failed on the compiler: explicit two-buffer batching in plain C++ (with loads
// behind wave-uniform branches the wait-count pass falls back to vmcnt(0) at every first use) and inline-asm loads with hand-counted
// waits (hipcc merges the buffers of different control-flow arms with v_mov copies placed right behind the loads: it copies registers
// whose data is still in flight).  Without control flow neither problem exists: the queue of a wave is, in issue order,
//     [bias / residual]  [x (and gamma) chunks of this thread]  [weight batch 0]  |  per batch b: [batch b + 1]
// x is waited for with batch 0 outstanding (the norm's reduce -> barrier -> normalise -> barrier chain runs under the stream's first
// latency), batch b with batch b + 1 outstanding, and hipcc's own counted waits are exact.  The activation vector is always staged
// in LDS (normalised or copied), so the stream has no other vector-memory access in it.  Each lane adds its slices in ascending
// order, the norm sums its chunks in the order of the generic kernel: all results are bit-identical to the generic kernels'.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NS, int WPB, bool NORM>
struct XStageU {
    static constexpr int T = 64 * WPB, K = NS * 512, CH = K / 8, NX = (CH + T - 1) / T;
    static_assert(!NORM || (WPB == 4 && NX <= 2), "the norm's reduction order is that of 256 threads: thread t sums chunks t, t + 256");
    sc_h8 xr[NX], gr[NORM ? NX : 1];
    __device__ __forceinline__ void issue(const _Float16* __restrict__ x, const _Float16* __restrict__ gamma) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < NX; ++i) {                  // (a chunk past the end is clamped to the LAST chunk: loaded, processed and stored a second
            const int k = (t + i * T) * 8, kc = k < K ? k : K - 8;      //  time with the same values - no load sits behind a condition hipcc could sink it into)
            xr[i] = *reinterpret_cast<const sc_h8*>(x + kc);
            if (NORM) gr[i] = *reinterpret_cast<const sc_h8*>(gamma + kc);
        }
    }
    __device__ __forceinline__ void finish(float eps, _Float16* xn, float* red) {
        const int t = threadIdx.x;
        if (NORM) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NX; ++i)
                if ((t + i * T) * 8 < K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss += (float)xr[i][e] * (float)xr[i][e];
                }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
            if ((t & 63) == 0) red[t >> 6] = ss;
            __syncthreads();
            const float rstd = rsqrtf((((red[0] + red[1]) + red[2]) + red[3]) / (float)K + eps);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int k = (t + i * T) * 8, kc = k < K ? k : K - 8;
                sc_h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)gr[i][e] * (float)(_Float16)((float)xr[i][e] * rstd));
                *reinterpret_cast<sc_h8*>(xn + kc) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) { const int k = (t + i * T) * 8, kc = k < K ? k : K - 8; *reinterpret_cast<sc_h8*>(xn + kc) = xr[i]; }
        }
        __syncthreads();
    }
};

// the weight stream of a wave: RPW rows (wp[r] already offset by lane * 8), NS slices in batches of UNR, two register buffers
template <int RPW, int NS, int UNR>
struct StreamU {
    static constexpr int NBATCH = (NS + UNR - 1) / UNR;
    sc_h8 buf[2][UNR][RPW];
    template <int B>
    __device__ __forceinline__ void load(const _Float16* const (&wp)[RPW]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (B * UNR + u < NS) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) buf[B & 1][u][r] = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp[r] + (B * UNR + u) * 512));
            }
    }
    template <int B>
    __device__ __forceinline__ void run(const _Float16* const (&wp)[RPW], const _Float16* xn, int lane, float (&acc)[RPW]) {
        if constexpr (B < NBATCH) {
            if constexpr (B + 1 < NBATCH) load<B + 1>(wp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (B * UNR + u < NS) {
                    const sc_h8 xv = *reinterpret_cast<const sc_h8*>(xn + lane * 8 + (B * UNR + u) * 512);
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2)
                            acc[r] = __builtin_amdgcn_fdot2(sc_h2{buf[B & 1][u][r][e], buf[B & 1][u][r][e + 1]}, sc_h2{xv[e], xv[e + 1]}, acc[r], false);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            run<B + 1>(wp, xn, lane, acc);
        }
    }
};

template <bool SWIGLU, bool OUT_F32, bool NORM, int RPW, int NS, int WPB, int UNR>
__global__ __launch_bounds__(64 * WPB) void k_gemv_u(const _Float16* __restrict__ W, const _Float16* __restrict__ x, const _Float16* __restrict__ bias,
                                                     const _Float16* __restrict__ res, void* __restrict__ y, int N, const _Float16* __restrict__ gamma, float eps) {
    constexpr int K = NS * 512;
    extern __shared__ __attribute__((aligned(16))) char gemv_smem[];
    _Float16* xn = reinterpret_cast<_Float16*>(gemv_smem);
    float* red = reinterpret_cast<float*>(gemv_smem + (size_t)K * 2);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * WPB + (int)(threadIdx.x >> 6)) * RPW;
    const _Float16* wp[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) wp[r] = W + (size_t)(row0 + r < N ? row0 + r : N - 1) * (size_t)K + lane * 8;
    // what the epilogue adds: requested first with branch-free loads (a null pointer reads W[0], the value is dropped), converted last
    _Float16 eb_b[RPW], eb_r[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int n = row0 + r < N ? row0 + r : N - 1;
        eb_b[r] = *(bias ? bias + n : W); eb_r[r] = *(res ? res + n : W);
    }
    XStageU<NS, WPB, NORM> xs;
    xs.issue(x, gamma);
    StreamU<RPW, NS, UNR> ws;
    ws.template load<0>(wp);
    __builtin_amdgcn_sched_barrier(0);
    xs.finish(eps, xn, red);
    // (no early exit for a wave past the last row: it streams row N - 1 again and stores nothing.  With `if (row0 >= N) return;` here
    //  hipcc sinks the first weight batch below the staging's barriers, into the block that uses it)
    float acc[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) acc[r] = 0.f;
    ws.template run<0>(wp, xn, lane, acc);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        // a use of the epilogue operands in THIS block: without it hipcc sinks their loads into the lane-0 block below, where each
        // is a full memory round trip on the tail of the wave
        asm volatile("" ::"v"((unsigned)__builtin_bit_cast(unsigned short, eb_b[r])), "v"((unsigned)__builtin_bit_cast(unsigned short, eb_r[r])));
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[r] += __shfl_xor(acc[r], m, 64);
    }
    if (lane == 0 && row0 < N) {
        if (SWIGLU) {          // rows (g g u u): two outputs per 4 rows (RPW == 4)
            const float g0 = acc[0], g1 = acc[RPW > 1 ? 1 : 0], u0 = acc[RPW > 2 ? 2 : 0], u1 = acc[RPW > 3 ? 3 : 0];
            _Float16* o = reinterpret_cast<_Float16*>(y) + (row0 >> 1);
            o[0] = (_Float16)(g0 / (1.0f + __expf(-g0)) * u0);
            o[1] = (_Float16)(g1 / (1.0f + __expf(-g1)) * u1);
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = row0 + r;
                if (n < N) {
                    const float v = acc[r] + ((bias ? (float)eb_b[r] : 0.f) + (res ? (float)eb_r[r] : 0.f));
                    if (OUT_F32) reinterpret_cast<float*>(y)[n] = v;
                    else reinterpret_cast<_Float16*>(y)[n] = (_Float16)v;
                }
            }
        }
    }
}

// k_decode_qkv<true> for K = NS * 512 and four waves per workgroup (see k_gemv_u); queue: [bias pair] [x, gamma] [both weight rows,
// all NS slices] [rotary table pair - its address needs pos[0], a scalar load: behind the weights, the stream does not wait for it]
template <int NS>
__global__ __launch_bounds__(256) void k_decode_qkv_u(const _Float16* __restrict__ Wq, const _Float16* __restrict__ Wkv, const _Float16* __restrict__ bq,
                                                      const _Float16* __restrict__ bkv, const _Float16* __restrict__ x, const _Float16* __restrict__ gamma,
                                                      float eps, _Float16* __restrict__ q_out, _Float16* __restrict__ cache, int ld,
                                                      const int* __restrict__ pos, int Hq, int Hkv, int Dh,
                                                      const float* __restrict__ tab_q, const float* __restrict__ tab_k, int tab_rows) {
    constexpr int K = NS * 512;
    const int lane = threadIdx.x & 63;
    const int half = Dh >> 1;
    const int task = blockIdx.x * 4 + (int)(threadIdx.x >> 6);            // (head, j) over q heads, then k heads, then v heads
    const int ntask = (Hq + 2 * Hkv) * half;
    const bool live = task < ntask;                                       // (dead waves of the last workgroup still take part in the staging's barriers)
    const int hh = (live ? task : 0) / half, j = (live ? task : 0) - hh * half;
    const _Float16 *w0, *b0;
    int kind, h;                                                          // 0 = q, 1 = k, 2 = v
    if (hh < Hq) { kind = 0; h = hh; w0 = Wq + (size_t)(h * Dh + j) * (size_t)K; b0 = bq ? bq + h * Dh + j : nullptr; }
    else if (hh < Hq + Hkv) { kind = 1; h = hh - Hq; w0 = Wkv + (size_t)(h * Dh + j) * (size_t)K; b0 = bkv ? bkv + h * Dh + j : nullptr; }
    else { kind = 2; h = hh - Hq - Hkv; w0 = Wkv + (size_t)((Hkv + h) * Dh + j) * (size_t)K; b0 = bkv ? bkv + (Hkv + h) * Dh + j : nullptr; }
    const _Float16* wp[2] = {w0 + lane * 8, w0 + (size_t)half * (size_t)K + lane * 8};
    extern __shared__ __attribute__((aligned(16))) char qkv_smem[];
    _Float16* xn = reinterpret_cast<_Float16*>(qkv_smem);
    float* red = reinterpret_cast<float*>(qkv_smem + (size_t)K * 2);
    const _Float16 braw0 = *(b0 ? b0 : Wq), braw1 = *(b0 ? b0 + half : Wq);
    XStageU<NS, 4, true> xs;
    xs.issue(x, gamma);
    StreamU<2, NS, NS> ws;
    ws.template load<0>(wp);
    __builtin_amdgcn_sched_barrier(0);
    const int row = pos[0];
    const int trow = row < 0 ? 0 : (row >= tab_rows ? tab_rows - 1 : row);      // the position lives in device memory: clamp to the table, never read past it
    const float* t = (kind == 0 ? tab_q : tab_k) + (size_t)trow * (size_t)Dh;   // (a V row reads a table row it ignores)
    const float tcs = t[j], tsn = t[half + j];
    __builtin_amdgcn_sched_barrier(0);
    xs.finish(eps, xn, red);
    float acc[2] = {0.f, 0.f};                                            // (dead waves stream task 0 again and store nothing: see k_gemv_u)
    ws.template run<0>(wp, xn, lane, acc);
    float acc0 = acc[0], acc1 = acc[1];
    asm volatile("" ::"v"((unsigned)__builtin_bit_cast(unsigned short, braw0)), "v"((unsigned)__builtin_bit_cast(unsigned short, braw1)), "v"(tcs), "v"(tsn));   // (see k_gemv_u)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { acc0 += __shfl_xor(acc0, m, 64); acc1 += __shfl_xor(acc1, m, 64); }
    if (lane == 0 && live) {
        const float bias0 = b0 ? (float)braw0 : 0.f, bias1 = b0 ? (float)braw1 : 0.f;
        _Float16* dst = kind == 0 ? q_out + h * Dh + j : cache + (size_t)row * (size_t)ld + (kind == 1 ? 0 : Hkv * Dh) + h * Dh + j;
        if (kind == 2) { dst[0] = (_Float16)(acc0 + bias0); dst[half] = (_Float16)(acc1 + bias1); return; }
        const float a32 = acc0 + bias0, b32 = acc1 + bias1;
        // fp32 fma FIRST, then the rounding to fp16 (see k_decode_qkv)
        float ra = __builtin_fmaf(-b32, tsn, a32 * tcs), rb = __builtin_fmaf(a32, tsn, b32 * tcs);
        asm volatile("" : "+v"(ra), "+v"(rb));
        dst[0] = (_Float16)ra;
        dst[half] = (_Float16)rb;
    }
}

// Waves per workgroup of the decode kernels, per kind of launch (0 q/k/v block, 1 short projection, 2 long rows / two rows per wave,
// 3 everything else).  Every workgroup of these launches is resident at once, so a launch lasts as long as the CU with the most
// workgroups: the counts below deal the 7B shapes evenly over 256 CUs where that measured faster (profiles/r03_run21_gemv_waves_per_block.md);
// SC_GEMV_WPB_<kind>=n pins another count for experiments (block_rmsnorm_to_lds gives the same bits for any of them).
int pick_wpb(int kind) {
    static int forced[4] = {-1, -1, -1, -1};
    static const int dflt[4] = {SC_WPB_QKV, SC_WPB_FEW, SC_WPB_LONG, 4};
    if (forced[kind] < 0) {
        char name[24];
        snprintf(name, sizeof(name), "SC_GEMV_WPB_%d", kind);
        const char* e = getenv(name);
        const int v = e ? atoi(e) : 0;
        forced[kind] = (v >= 1 && v <= 8) ? v : dflt[kind];
    }
    return forced[kind];
}

// SC_GEMV_GENERIC=1 pins the generic kernels (A/B runs); any SC_GEMV_WPB_* / SC_GEMV_RPW_FEW knob does the same (they tune the generic kernels)
bool use_specialised() {
    static const bool on = [] {
        const char* g = getenv("SC_GEMV_GENERIC");
        if (g && g[0] == '1') return false;
        for (const char* n : {"SC_GEMV_WPB_0", "SC_GEMV_WPB_1", "SC_GEMV_WPB_2", "SC_GEMV_WPB_3", "SC_GEMV_RPW_FEW"}) if (getenv(n)) return false;
        return true;
    }();
    return on;
}

}  // namespace

extern "C" int sc_decode_qkv_f16(const void* Wq, const void* Wkv, const void* bq, const void* bkv, const void* x, const void* rms_gamma, float rms_eps,
                                 void* q_out, void* cache, int cache_ld, const int32_t* pos, int q_heads, int kv_heads, int Dh, int K, float theta,
                                 sc_stream_t stream) {
    SC_REQUIRE(Wq && Wkv && x && q_out && cache && pos, "sc_decode_qkv_f16: null pointer argument");
    SC_REQUIRE(q_heads > 0 && kv_heads > 0 && Dh > 0 && Dh % 2 == 0 && K > 0 && K % 8 == 0, "sc_decode_qkv_f16: bad sizes");
    SC_REQUIRE(cache_ld >= 2 * kv_heads * Dh, "sc_decode_qkv_f16: cache row stride too small for K | V");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(Wq) | reinterpret_cast<uintptr_t>(Wkv) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(rms_gamma)) & 15) == 0,
               "sc_decode_qkv_f16: weights, x and gamma must be 16-byte aligned");
    const int ntask = (q_heads + 2 * kv_heads) * (Dh / 2);
    const int wpb = pick_wpb(0);
    hipLaunchKernelGGL(k_decode_qkv<false>, dim3((unsigned)((ntask + wpb - 1) / wpb)), dim3(64 * wpb), rms_gamma ? (size_t)K * 2 + 16 : 0, (hipStream_t)stream, (const _Float16*)Wq, (const _Float16*)Wkv,
                       (const _Float16*)bq, (const _Float16*)bkv, (const _Float16*)x, (const _Float16*)rms_gamma, rms_eps, (_Float16*)q_out, (_Float16*)cache,
                       cache_ld, pos, q_heads, kv_heads, Dh, K, log2f(theta), (const float*)nullptr, (const float*)nullptr, 0);
    SC_CHECK_LAUNCH("sc_decode_qkv_f16");
    return SC_OK;
}

extern "C" int sc_decode_qkv_tab_f16(const void* Wq, const void* Wkv, const void* bq, const void* bkv, const void* x, const void* rms_gamma, float rms_eps,
                                     void* q_out, void* cache, int cache_ld, const int32_t* pos, int q_heads, int kv_heads, int Dh, int K,
                                     const float* tab_q, const float* tab_k, int tab_rows, sc_stream_t stream) {
    SC_REQUIRE(Wq && Wkv && x && q_out && cache && pos && tab_q && tab_k && tab_rows > 0, "sc_decode_qkv_tab_f16: null pointer argument / empty rotary table");
    SC_REQUIRE(q_heads > 0 && kv_heads > 0 && Dh > 0 && Dh % 2 == 0 && K > 0 && K % 8 == 0, "sc_decode_qkv_tab_f16: bad sizes");
    SC_REQUIRE(cache_ld >= 2 * kv_heads * Dh, "sc_decode_qkv_tab_f16: cache row stride too small for K | V");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(Wq) | reinterpret_cast<uintptr_t>(Wkv) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(rms_gamma)) & 15) == 0,
               "sc_decode_qkv_tab_f16: weights, x and gamma must be 16-byte aligned");
    const int ntask = (q_heads + 2 * kv_heads) * (Dh / 2);
    if (K == 3584 && rms_gamma && use_specialised()) {              // the 7B shape: straight-line kernel (k_decode_qkv_u)
        hipLaunchKernelGGL(k_decode_qkv_u<7>, dim3((unsigned)((ntask + 3) / 4)), dim3(256), (size_t)K * 2 + 16, (hipStream_t)stream, (const _Float16*)Wq, (const _Float16*)Wkv,
                           (const _Float16*)bq, (const _Float16*)bkv, (const _Float16*)x, (const _Float16*)rms_gamma, rms_eps, (_Float16*)q_out, (_Float16*)cache,
                           cache_ld, pos, q_heads, kv_heads, Dh, tab_q, tab_k, tab_rows);
        SC_CHECK_LAUNCH("sc_decode_qkv_tab_f16");
        return SC_OK;
    }
    const int wpb = pick_wpb(0);
    hipLaunchKernelGGL(k_decode_qkv<true>, dim3((unsigned)((ntask + wpb - 1) / wpb)), dim3(64 * wpb), rms_gamma ? (size_t)K * 2 + 16 : 0, (hipStream_t)stream, (const _Float16*)Wq, (const _Float16*)Wkv,
                       (const _Float16*)bq, (const _Float16*)bkv, (const _Float16*)x, (const _Float16*)rms_gamma, rms_eps, (_Float16*)q_out, (_Float16*)cache,
                       cache_ld, pos, q_heads, kv_heads, Dh, K, 0.f, tab_q, tab_k, tab_rows);
    SC_CHECK_LAUNCH("sc_decode_qkv_tab_f16");
    return SC_OK;
}

extern "C" int sc_gemv_f16(const void* W, const void* x, const void* bias, const void* residual, void* y, int N, int K, int epilogue,
                           int out_f32, const int32_t* y_row, int y_ld, const void* rms_gamma, float rms_eps, sc_stream_t stream) {
    SC_REQUIRE(W && x && y, "sc_gemv_f16: null pointer argument");
    SC_REQUIRE(N > 0 && K > 0 && K % 8 == 0, "sc_gemv_f16: K must be a positive multiple of 8");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(x)) & 15) == 0, "sc_gemv_f16: W and x must be 16-byte aligned");
    SC_REQUIRE(epilogue == SC_EPI_NONE || epilogue == SC_EPI_SWIGLU, "sc_gemv_f16: epilogue must be NONE or SWIGLU");
    SC_REQUIRE(!y_row || (!out_f32 && y_ld > 0), "sc_gemv_f16: a dynamic output row needs fp16 output and y_ld > 0");
    SC_REQUIRE(epilogue != SC_EPI_SWIGLU || (N % 4 == 0 && !bias && !residual && !out_f32), "sc_gemv_f16: SwiGLU needs N % 4 == 0, no bias/residual, fp16 out");
    hipStream_t s = (hipStream_t)stream;
    const _Float16 *w = (const _Float16*)W, *xx = (const _Float16*)x, *b = (const _Float16*)bias, *r = (const _Float16*)residual;
    // rows per wave: 4 amortises the x loads when there are plenty of rows; 1 keeps >= ~900 workgroups in flight for the
    // 3584-row projections (224 workgroups at 4 rows/wave left most CUs with a single latency-bound workgroup)
    const size_t gsm = rms_gamma ? (size_t)K * 2 + 16 : 0;          // the normalised activation vector + 4 wave sums (block_rmsnorm_to_lds)
    SC_REQUIRE(gsm <= 65536, "sc_gemv_f16: fused RMSNorm needs K <= 32760");
    const bool few = N < 16384 && epilogue != SC_EPI_SWIGLU;
    // the 7B shapes (K = 3584 = 7 slices, K = 18944 = 37 slices): straight-line kernels (k_gemv_u); everything else: the generic ones
    if (!y_row && use_specialised() && (K == 3584 || (K == 18944 && few && !out_f32 && !rms_gamma))) {
        const _Float16* gm = (const _Float16*)rms_gamma;
        const size_t lds = (size_t)K * 2 + 16;
#define SC_GU(SW, F32, RPW, NS, WPB, UNR)                                                                                                             \
    do {                                                                                                                                              \
        const dim3 g_((unsigned)((N + (WPB) * (RPW) - 1) / ((WPB) * (RPW)))), b_(64 * (WPB));                                                         \
        if (gm) hipLaunchKernelGGL((k_gemv_u<SW, F32, true, RPW, NS, WPB, UNR>), g_, b_, lds, s, w, xx, b, r, y, N, gm, rms_eps);                      \
        else hipLaunchKernelGGL((k_gemv_u<SW, F32, false, RPW, NS, WPB, UNR>), g_, b_, lds, s, w, xx, b, r, y, N, gm, rms_eps);                        \
    } while (0)
        if (K == 18944) hipLaunchKernelGGL((k_gemv_u<false, false, false, 2, 37, 7, 6>), dim3((unsigned)((N + 13) / 14)), dim3(448), lds, s, w, xx, b, r, y, N, gm, rms_eps);
        else if (epilogue == SC_EPI_SWIGLU) SC_GU(true, false, 4, 7, 4, 2);
        else if (out_f32) { if (few) SC_GU(false, true, 1, 7, 4, 7); else SC_GU(false, true, 4, 7, 4, 2); }
        else { if (few) SC_GU(false, false, 1, 7, 4, 7); else SC_GU(false, false, 4, 7, 4, 2); }
#undef SC_GU
        SC_CHECK_LAUNCH("sc_gemv_f16");
        return SC_OK;
    }
    static int rpw_few = -1;                        // SC_GEMV_RPW_FEW=1|2|4: rows per wave of the small projections (A/B runs)
    if (rpw_few < 0) { const char* e = getenv("SC_GEMV_RPW_FEW"); const int v = e ? atoi(e) : 1; rpw_few = (v == 2 || v == 4) ? v : 1; }   // anything else -> 1
    // the tuning knob only applies where a kernel for that row count is instantiated (fp16 output); the fp32-output path of a short
    // projection always runs one row per wave, and the grid is derived from the rows per wave of the kernel actually launched
    const int rpw = few ? (out_f32 ? 1 : rpw_few) : 4;
    const int wpb = pick_wpb(few ? 1 : 3), wpb2 = pick_wpb(2);
    const dim3 grid((unsigned)((N + wpb * rpw - 1) / (wpb * rpw))), block(64 * wpb);
    if (few && !out_f32 && (rpw == 2 || rpw == 4 || (rpw_few == 1 && K >= 8192))) {
        // long rows (the down projection, K = 18 944): two rows per wave and 8 x 16 B per lane in flight per row stream the 136 MB at
        // 5.56 TB/s against 5.11 (profiles/r02_run18: A/B of rows-per-wave x unroll on a >1 GB weight cycle); short rows: 1 row, unroll 4
        if (rpw == 4) hipLaunchKernelGGL((k_gemv<false, false, 4>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps);
        else hipLaunchKernelGGL((k_gemv<false, false, 2, 8>), dim3((unsigned)((N + 2 * wpb2 - 1) / (2 * wpb2))), dim3(64 * wpb2), gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps);
        SC_CHECK_LAUNCH("sc_gemv_f16");
        return SC_OK;
    }
    if (epilogue == SC_EPI_SWIGLU) hipLaunchKernelGGL((k_gemv<true, false, 4>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps);
    else if (out_f32) { if (few) hipLaunchKernelGGL((k_gemv<false, true, 1>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps);
                        else hipLaunchKernelGGL((k_gemv<false, true, 4>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps); }
    else { if (few) hipLaunchKernelGGL((k_gemv<false, false, 1>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps);
           else hipLaunchKernelGGL((k_gemv<false, false, 4>), grid, block, gsm, s, w, xx, b, r, y, N, K, y_row, y_ld, (const _Float16*)rms_gamma, rms_eps); }
    SC_CHECK_LAUNCH("sc_gemv_f16");
    return SC_OK;
}
