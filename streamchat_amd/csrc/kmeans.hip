// Selective-frame k-means for gfx950 (CDNA4) — the HBM-streaming replacement of the reference's
// `weighted_kmeans_feature` (utiles.py:291-330).  Each "point" is a whole frame of
// D = 576*3584 = 2,064,384 features and K is tiny (5..8), so this is a pure streaming reduce:
// the reference's [T,K,D] broadcast temporary (utiles.py:299, 8.3 GB at T=400) is never formed.
//
// Kernels (one Lloyd iteration = assign -> reduce -> labels/order -> update -> decide):
//   km_assign   one wave per 512-column chunk; centroid slice lives in VGPRs, rows stream through
//               as 16-byte/lane coalesced loads (1 KiB per wave-instruction); the TT*KB fp32 lane
//               partials of a row group are reduced across the wave with a halving butterfly
//               (v_permlane32_swap / v_permlane16_swap + 4 shuffle levels: ~2.5 ops per value
//               instead of 12) and written as one coalesced 256-byte line per group.
//   km_reduce   fp64 two-level (32 segments) sum of the per-chunk partials.
//   km_argmin   fp64 totals -> argmin (first minimum), one thread per row.
//   km_order    single block: stable counting sort of rows by label (ballot ranks), W[k], empty-cluster ranks.
//   km_update   one wave per chunk: per cluster, rows in ascending order, fp32 sequential
//               weighted sum / W; shift partials for the convergence test.
//   km_decide   sum_k ||C_i - C'||_2 < tol ? -> device-side `done` flag (no host round trip).
// The arithmetic order is the "SC-KM1" spec shared bit-for-bit with oracle/kmeans_oracle.c.
// Compiled with -ffp-contract=off: every fma below is explicit.
#include "sc_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int CH = 512;      // columns per chunk = 64 lanes x 8 elements
constexpr int NSEG = 32;     // fp64 segments
constexpr int WPB = 4;       // waves per block in the streaming kernels (2 / 8: +-0.3 %, profiles/r04_run9_kmeans_knobs.md)
typedef float sc_f2 __attribute__((ext_vector_type(2)));

struct KmState {
    int done, exit_iter, cur, reseed_pos, status, n_empty, pad0, pad1;
};

template <typename Tag> struct Raw8;
template <> struct Raw8<ScF16> {
    uint4 a;
    __device__ __forceinline__ void load(const void* b, size_t off) {        // X is streamed once per pass: non-temporal
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(reinterpret_cast<const _Float16*>(b) + off));
        a = make_uint4(v[0], v[1], v[2], v[3]);
    }
    __device__ __forceinline__ void zero() { a = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void unpack(float (&o)[8]) const {
        sc_h8 v = __builtin_bit_cast(sc_h8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    }
};
template <> struct Raw8<ScBF16> {
    uint4 a;
    __device__ __forceinline__ void load(const void* b, size_t off) { a = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(b) + off); }
    __device__ __forceinline__ void zero() { a = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void unpack(float (&o)[8]) const {
        unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    }
};
template <> struct Raw8<ScF32> {
    uint4 a, b;
    __device__ __forceinline__ void load(const void* p, size_t off) {
        const uint4* q = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(p) + off);
        a = q[0]; b = q[1];
    }
    __device__ __forceinline__ void zero() { a = make_uint4(0, 0, 0, 0); b = a; }
    __device__ __forceinline__ void unpack(float (&o)[8]) const {
        o[0] = __uint_as_float(a.x); o[1] = __uint_as_float(a.y); o[2] = __uint_as_float(a.z); o[3] = __uint_as_float(a.w);
        o[4] = __uint_as_float(b.x); o[5] = __uint_as_float(b.y); o[6] = __uint_as_float(b.z); o[7] = __uint_as_float(b.w);
    }
};

// generic (unaligned D) 8-element fetch with per-element bounds
template <typename Tag>
__device__ __forceinline__ void load8_guard(const void* base, size_t rowoff, int64_t col, int64_t D, float (&o)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (col + e < D) ? sc_load1<Tag>(base, rowoff + (size_t)(col + e)) : 0.f;
}

template <int M>
__device__ __forceinline__ void bfly_step(float (&v)[64], int lane) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float send = up ? v[i] : v[i + M];
        const float keep = up ? v[i + M] : v[i];
        v[i] = keep + __shfl_xor(send, M, 64);
    }
}

// Halving butterfly: on entry lane L holds 64 values v[i] (item i, this lane's partial);
// on exit v[0] of lane L holds the SC-KM1 tree sum over all 64 lanes of item L.
__device__ __forceinline__ void butterfly64(float (&v)[64], int lane) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    bfly_step<8>(v, lane);
    bfly_step<4>(v, lane);
    bfly_step<2>(v, lane);
    bfly_step<1>(v, lane);
}

__host__ __device__ constexpr int tt_for(int kb) { return (64 / kb) < 16 ? (64 / kb) : 16; }

// ---------------------------------------------------------------------------------------------
template <typename Tag, int KB, bool VEC>
__global__ __launch_bounds__(WPB * 64) void km_assign(const void* __restrict__ X, const float* __restrict__ Ca,
                                                      const float* __restrict__ Cb, const KmState* __restrict__ st,
                                                      float* __restrict__ partial, int T, int64_t D, int K, int k0,
                                                      int64_t nchunks) {
    constexpr int TT = tt_for(KB);
    if (st->done) return;
    const float* __restrict__ C = st->cur ? Cb : Ca;
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t col = c * CH + lane * 8;
    const bool active = col < D;
    const size_t I = (size_t)T * (size_t)K;

    float cr[KB][8];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        if (VEC) {
            if (active) sc_load8<ScF32>(C, (size_t)(k0 + k) * (size_t)D + (size_t)col, cr[k]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) cr[k][e] = 0.f;
            }
        } else {
            load8_guard<ScF32>(C, (size_t)(k0 + k) * (size_t)D, col, D, cr[k]);
        }
    }

    const int ngroups = (T + TT - 1) / TT;
    Raw8<Tag> cur[TT], nxt[TT];
    float xg[VEC ? 1 : TT][8];   // unaligned path keeps converted rows instead of raw vectors
    (void)xg;

    auto load_group = [&](int g, Raw8<Tag>(&buf)[TT]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = g * TT + tt;
            if (t < T && active) buf[tt].load(X, (size_t)t * (size_t)D + (size_t)col);
            else buf[tt].zero();
        }
    };
    if (VEC) load_group(0, cur);

    for (int g = 0; g < ngroups; ++g) {
        if (VEC && g + 1 < ngroups) load_group(g + 1, nxt);
        float p[64];
#pragma unroll
        for (int i = TT * KB; i < 64; ++i) p[i] = 0.f;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float x[8];
            if (VEC) cur[tt].unpack(x);
            else {
                const int t = g * TT + tt;
                if (t < T) load8_guard<Tag>(X, (size_t)t * (size_t)D, col, D, x);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                // packed fp32 math (v_pk_add_f32 / v_pk_fma_f32): component .x carries the even elements' accumulator, .y the odd
                // ones' — exactly the SC-KM1 lane partial (each packed op is IEEE per component)
                sc_f2 acc = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const sc_f2 xv = {x[e], x[e + 1]}, cv = {cr[k][e], cr[k][e + 1]};
                    const sc_f2 d = xv - cv;
                    acc = __builtin_elementwise_fma(d, d, acc);
                }
                p[tt * KB + k] = acc.x + acc.y;
            }
        }
        butterfly64(p, lane);
        if (lane < TT * KB) {
            const int tt = lane / KB, k = lane - tt * KB;
            const int t = g * TT + tt;
            if (t < T) partial[(size_t)c * I + (size_t)t * K + (k0 + k)] = p[0];
        }
        if (VEC) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) cur[tt] = nxt[tt];
        }
    }
}

// level-1 fp64 reduce: seg[s][item] = sum_{c in segment s, ascending} partial[c][item]
__global__ void km_reduce(const float* __restrict__ partial, const KmState* __restrict__ st, double* __restrict__ seg,
                          size_t I, int64_t nchunks, int check_done) {
    if (check_done && st->done) return;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= I) return;
    const int s = blockIdx.y;
    const int64_t seglen = (nchunks + NSEG - 1) / NSEG;
    int64_t lo = (int64_t)s * seglen, hi = lo + seglen;
    if (hi > nchunks) hi = nchunks;
    double a = 0.0;
    int64_t c = lo;
    for (; c + 8 <= hi; c += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(c + u) * I + item];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (double)v[u];
    }
    for (; c < hi; ++c) a += (double)partial[(size_t)c * I + item];
    seg[(size_t)s * I + item] = a;
}

// dist2 totals -> labels (first minimum); one thread per row, rows spread over the grid
__global__ __launch_bounds__(64) void km_argmin(const double* __restrict__ seg, const KmState* __restrict__ st, int* __restrict__ labels32,
                                                double* __restrict__ dist2_out, int T, int K, int check_done) {
    if (check_done && st->done) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const size_t I = (size_t)T * K;
    int best = 0;
    double bv = 0.0;
    for (int k = 0; k < K; ++k) {
        double part[NSEG];
#pragma unroll
        for (int s = 0; s < NSEG; ++s) part[s] = seg[(size_t)s * I + (size_t)t * K + k];       // independent loads, summed in order
        double tot = 0.0;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) tot += part[s];
        if (dist2_out) dist2_out[(size_t)t * K + k] = tot;
        if (k == 0 || tot < bv) { bv = tot; best = k; }
    }
    labels32[t] = best;
}

// single block: stable counting sort of the rows by label (ballot ranks), W[k], empty-cluster ranks
__global__ __launch_bounds__(1024) void km_order(KmState* __restrict__ st, const float* __restrict__ w, const int* __restrict__ labels32,
                                                 int* __restrict__ order, int* __restrict__ start, float* __restrict__ W,
                                                 int* __restrict__ empty_rank, int T, int K, int check_done) {
    if (check_done && st->done) return;
    extern __shared__ int sm[];               // counts[K] | wave_cnt[16]
    int* counts = sm;
    int* wcnt = sm + K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = threadIdx.x; k < K; k += blockDim.x) counts[k] = 0;
    __syncthreads();
    // pass 1: cluster sizes
    for (int t = threadIdx.x; t < T; t += blockDim.x) atomicAdd(&counts[labels32[t]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0, ne = 0;
        for (int k = 0; k < K; ++k) { start[k] = acc; acc += counts[k]; }
        start[K] = acc;
        for (int k = 0; k < K; ++k) {
            const bool empty = w ? false : (counts[k] == 0);        // unweighted: W = count (exact in any order); weighted: decided below
            empty_rank[k] = empty ? ne++ : -1;
            if (!w) W[k] = (float)counts[k];
        }
        if (!w) st->n_empty = ne;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) counts[k] = 0;            // becomes the running fill position per cluster
    __syncthreads();
    // pass 2: stable scatter, tiles of blockDim rows in ascending t; rank inside the tile by per-cluster ballots
    for (int t0 = 0; t0 < T; t0 += blockDim.x) {
        const int t = t0 + threadIdx.x;
        const int lab = t < T ? labels32[t] : -1;
        for (int k = 0; k < K; ++k) {
            const unsigned long long m = __ballot(lab == k);
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            if (lab == k) {
                int before = 0;
                for (int i = 0; i < wave; ++i) before += wcnt[i];
                order[start[k] + counts[k] + before + __popcll(m & ((1ull << lane) - 1ull))] = t;
            }
            __syncthreads();
            if (threadIdx.x == 0) { int tot = 0; for (int i = 0; i < nw; ++i) tot += wcnt[i]; counts[k] += tot; }
            __syncthreads();
        }
    }
    // weighted: W[k] = sequential fp32 sum over the cluster's rows in ascending order (SC-KM1)
    if (w) {
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float ws = 0.f;
            for (int i = start[k]; i < start[k + 1]; ++i) ws = ws + w[order[i]];
            W[k] = ws;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int ne = 0;
            for (int k = 0; k < K; ++k) empty_rank[k] = (W[k] > 0.f) ? -1 : ne++;
            st->n_empty = ne;
        }
    }
}

// one wave per chunk: new centroids + shift partials
template <typename Tag, bool VEC>
__global__ __launch_bounds__(WPB * 64) void km_update(const void* __restrict__ X, float* __restrict__ Ca, float* __restrict__ Cb,
                                                      KmState* __restrict__ st, const float* __restrict__ w,
                                                      const int* __restrict__ order, const int* __restrict__ start,
                                                      const float* __restrict__ W, const int* __restrict__ empty_rank,
                                                      const int* __restrict__ reseed_idx, int n_reseed,
                                                      float* __restrict__ dpart, int T, int64_t D, int K, int64_t nchunks, int empty_zero) {
    if (st->done) return;
    const float* __restrict__ Cold = st->cur ? Cb : Ca;
    float* __restrict__ Cnew = st->cur ? Ca : Cb;
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t col = c * CH + lane * 8;
    const bool active = col < D;
    const int rbase = st->reseed_pos;

    for (int k = 0; k < K; ++k) {
        float cn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cn[e] = 0.f;
        const float Wk = W[k];
        if (Wk > 0.f) {
            const int lo = start[k], hi = start[k + 1];
            constexpr int U = 8;
            for (int i0 = lo; i0 < hi; i0 += U) {
                float x[U][8];
                float wt[U];
                Raw8<Tag> raw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u;
                    if (i < hi) {
                        const int t = order[i];
                        wt[u] = w ? w[t] : 1.0f;
                        if (VEC) {
                            if (active) raw[u].load(X, (size_t)t * (size_t)D + (size_t)col);
                            else raw[u].zero();
                        } else load8_guard<Tag>(X, (size_t)t * (size_t)D, col, D, x[u]);
                    } else {
                        wt[u] = 0.f;
                        raw[u].zero();
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[u][e] = 0.f;
                    }
                }
                if (VEC) {
#pragma unroll
                    for (int u = 0; u < U; ++u) raw[u].unpack(x[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (i0 + u < hi) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) cn[e] = w ? cn[e] + wt[u] * x[u][e] : cn[e] + x[u][e];   // mul, then add (1.0f * x == x exactly)
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) cn[e] = cn[e] / Wk;
        } else if (!empty_zero) {                 // empty_zero: the centre of an empty cluster is the zero vector (torch_kmeans, utils.py:66)
            const int pos = rbase + empty_rank[k];
            int r = 0;
            if (reseed_idx && pos < n_reseed) r = reseed_idx[pos];
            if (r < 0 || r >= T) r = 0;
            if (VEC) { if (active) sc_load8<Tag>(X, (size_t)r * (size_t)D + (size_t)col, cn); }
            else load8_guard<Tag>(X, (size_t)r * (size_t)D, col, D, cn);
        }
        float co[8];
        if (VEC) {
            if (active) sc_load8<ScF32>(Cold, (size_t)k * (size_t)D + (size_t)col, co);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) co[e] = 0.f;
            }
        } else load8_guard<ScF32>(Cold, (size_t)k * (size_t)D, col, D, co);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float d0 = co[e] - cn[e], d1 = co[e + 1] - cn[e + 1];
            a0 = __builtin_fmaf(d0, d0, a0);
            a1 = __builtin_fmaf(d1, d1, a1);
        }
        const float wp = sc_wave_tree_sum(a0 + a1);
        if (lane == 0) dpart[(size_t)c * K + k] = wp;
        if (VEC) {
            if (active) {
                sc_f4* dst = reinterpret_cast<sc_f4*>(Cnew + (size_t)k * (size_t)D + (size_t)col);
                dst[0] = sc_f4{cn[0], cn[1], cn[2], cn[3]};
                dst[1] = sc_f4{cn[4], cn[5], cn[6], cn[7]};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col + e < D) Cnew[(size_t)k * (size_t)D + (size_t)(col + e)] = cn[e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: update(i) + assign(i + 1) in ONE pass over X (fp16, D % 512 == 0, K <= 8, T <= 8 * RW).
// km_assign and km_update each stream the whole [T, D] matrix: 2 n reads of X for n Lloyd iterations, 0.77 ms per iteration at T = 400,
// K = 5, D = 2 064 384 against 0.27 ms for one read at the HBM rate.  Both need the chunk's T x 1 KiB slab; what stops a single pass is
// where to keep 400 KiB between the two uses.  Here a persistent workgroup of EIGHT waves (two per SIMD, 256 registers each) keeps the
// slab of one 512-column chunk IN REGISTERS - wave w owns rows [w RW, (w + 1) RW) as the raw 16 bytes per lane and row, 4 VGPRs each,
// 200 of its 256 registers at RW = 50 - and walks chunks c = b, b + G, ...:
//   update   SC-KM1 adds a cluster's rows in ascending order, one fp32 accumulator per column: the sum of cluster k travels through the
//            waves in row order as a systolic chain - in phase p wave w adds its rows of cluster p - w to the running sums in LDS
//            (K + 7 phases, one barrier each); which rows those are is a 64-bit ballot mask per cluster, built once per launch;
//   finish   wave k divides cluster k by W_k (or takes the reseed row), writes C', the shift partial of the chunk and the new centroid
//            slice into LDS;
//   assign   every wave takes its rows once more - still in registers - against the K new slices (LDS): the lane partials of a row's K
//            distances are reduced with an 8-value halving butterfly (levels 32, 16, 8 halve the values, 4, 2, 1 are plain xor-adds: the
//            SC-KM1 tree, 8 live registers instead of butterfly64's 64);
//   reload   the registers of row j are free the moment its distances are done: the row of the NEXT chunk is requested right there, so the
//            stream runs under the whole assign pass and the next update finds its slab (mostly) landed.
// One read of X per Lloyd iteration (n + 1 for n iterations with the plain first assign).  Arithmetic order is SC-KM1 throughout: labels,
// centroids and exit iteration stay bit-identical to the two-kernel path and to oracle/kmeans_oracle.c.
// ---------------------------------------------------------------------------------------------
constexpr int FNW = 4;                                                        // waves per workgroup of the fused pass: one per SIMD, 512 registers each

// (the empty asm makes the conversion depend on THIS point of the program: without it hipcc hoists the unpacking of every row out of the
//  phase loop - 8 fp32 registers per row live across it, the slab's footprint tripled and spilled)
__device__ __forceinline__ void unpack_h8(uint4& a, float (&o)[8]) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
    const sc_h8 v = __builtin_bit_cast(sc_h8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}

// First half of the SC-KM1 wave tree for 8 values per lane (this lane's partials of items 0..7): the levels 32, 16, 8 halve the values;
// on return the lane holds ONE value: the partial of item (lane >> 3) over the lanes that differ from it in bits 5, 4, 3.
__device__ __forceinline__ float butterfly8_hi(float (&v)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                             // lanes ^ 32: items i | i + 4
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                             // lanes ^ 16: items i | i + 2
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 2]), false, false);
        v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const bool up = (lane & 8) != 0;                                          // lanes ^ 8: items 0 | 1
    const float send = up ? v[0] : v[1], keep = up ? v[1] : v[0];
    return keep + __shfl_xor(send, 8, 64);
}

// compile-time row loops (a `#pragma unroll` over 100 rows is refused by the unroller's budget, and raising the budget unrolls every other loop
// of the kernel too - 300 k instructions; a recursive template is unrolled by construction and nothing else is)
template <int J, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (J < N) { f(std::integral_constant<int, J>{}); static_for<J + 1, N>(f); }
}

template <int RW, int K, bool ASSIGN>                                         // rows per wave, clusters, whether the next iteration's partials are wanted
__global__ __launch_bounds__(FNW * 64) void km_fused(const _Float16* __restrict__ X, float* __restrict__ Ca, float* __restrict__ Cb,
                                                     const KmState* __restrict__ st, const float* __restrict__ w,
                                                     const int* __restrict__ labels32, const float* __restrict__ W,
                                                     const int* __restrict__ empty_rank, const int* __restrict__ reseed_idx, int n_reseed,
                                                     float* __restrict__ dpart, float* __restrict__ partial, int T, int64_t D,
                                                     int64_t nchunks) {
    if (st->done) return;
    constexpr int NM = (RW + 63) / 64;                                        // 64-row mask words per cluster
    constexpr int GR = 4, NG = (RW + GR - 1) / GR;                            // rows whose butterflies interleave in the assign pass (8: 130 - 290 spilled registers)
    __shared__ __attribute__((aligned(16))) float sums[K][CH];               // running column sums per cluster (the chain), then the new slices
    const float* __restrict__ Cold = st->cur ? Cb : Ca;
    float* __restrict__ Cnew = st->cur ? Ca : Cb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r0 = wave * RW;
    const int rbase = st->reseed_pos;
    const size_t I = (size_t)T * (size_t)K;
    // which of this wave's rows belong to cluster k: lane l of mask word m stands for row r0 + 64 m + l
    unsigned long long mask[K][NM], live[NM];
    float myw[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int jr = 64 * m + lane, myrow = r0 + jr;
        const bool ok = jr < RW && myrow < T;
        const int mylab = ok ? labels32[myrow] : -1;
        myw[m] = (w && ok) ? w[myrow] : 1.0f;
        live[m] = __ballot(ok);
#pragma unroll
        for (int k = 0; k < K; ++k) mask[k][m] = __ballot(mylab == k);
    }

    uint4 row[RW];
    // one running pointer per load sweep (rows are consecutive; past T it stops advancing: the last row is re-read and never used) - a
    // hundred precomputed row addresses are two hundred live scalars
    const size_t row_first = (size_t)(r0 < T ? r0 : T - 1) * (size_t)D + (size_t)lane * 8;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    int64_t c = blockIdx.x;
    if (c >= nchunks) return;
    {
        const _Float16* xp = X + (size_t)c * CH + row_first;
        static_for<0, RW>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(xp));
            row[j] = make_uint4(v[0], v[1], v[2], v[3]);
            if (r0 + j + 1 < T) xp += D;
        });
    }
#pragma unroll 1
    for (; c < nchunks; c += gridDim.x) {
        const int64_t cn_next = c + gridDim.x;
        const int64_t col = c * CH + lane * 8;
        // ---- update: the ordered row sums as a chain through the waves ----
#pragma unroll 1
        for (int p = 0; p < K + FNW - 1; ++p) {
            const int k = p - wave;
            if (k >= 0 && k < K) {
                float sacc[8];
                if (wave == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sacc[e] = 0.f;
                } else {
                    const sc_f4 a = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8]), b = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8 + 4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sacc[e] = a[e]; sacc[4 + e] = b[e]; }
                }
                unsigned long long mk[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) {                                // mask[k] with a run-time k: a select chain over K scalar pairs
                    mk[m] = 0;
#pragma unroll
                    for (int kk = 0; kk < K; ++kk) mk[m] = (k == kk) ? mask[kk][m] : mk[m];
                }
                static_for<0, RW>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    __builtin_amdgcn_sched_barrier(0);                        // (one row at a time: interleaved, the unrolled rows' temporaries spill the slab)
                    if ((mk[j >> 6] >> (j & 63)) & 1ull) {                    // (wave-uniform)
                        float x[8];
                        unpack_h8(row[j], x);
                        if (w) {
                            const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myw[j >> 6]), j & 63));
#pragma unroll
                            for (int e = 0; e < 8; ++e) sacc[e] = sacc[e] + wt * x[e];        // mul, then add (no contraction: -ffp-contract=off)
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) sacc[e] = sacc[e] + x[e];
                        }
                    }
                });
                *reinterpret_cast<sc_f4*>(&sums[k][lane * 8]) = sc_f4{sacc[0], sacc[1], sacc[2], sacc[3]};
                *reinterpret_cast<sc_f4*>(&sums[k][lane * 8 + 4]) = sc_f4{sacc[4], sacc[5], sacc[6], sacc[7]};
            }
            __syncthreads();
        }
        // ---- finish: wave k % 4 owns cluster k: C' = sums / W (or the reseed row), shift partial, global store; the new slice replaces the sums ----
#pragma unroll 1
        for (int k = wave; k < K; k += FNW) {
            float cn[8];
            const float Wk = W[k];
            if (Wk > 0.f) {
                const sc_f4 a = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8]), b = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cn[e] = a[e] / Wk; cn[4 + e] = b[e] / Wk; }
            } else {
                const int pos = rbase + empty_rank[k];
                int r = 0;
                if (reseed_idx && pos < n_reseed) r = reseed_idx[pos];
                if (r < 0 || r >= T) r = 0;
                sc_load8<ScF16>(X, (size_t)r * (size_t)D + (size_t)col, cn);
            }
            float co[8];
            sc_load8<ScF32>(Cold, (size_t)k * (size_t)D + (size_t)col, co);
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float d0 = co[e] - cn[e], d1 = co[e + 1] - cn[e + 1];
                a0 = __builtin_fmaf(d0, d0, a0);
                a1 = __builtin_fmaf(d1, d1, a1);
            }
            const float wp = sc_wave_tree_sum(a0 + a1);
            if (lane == 0) dpart[(size_t)c * K + k] = wp;
            sc_f4* dst = reinterpret_cast<sc_f4*>(Cnew + (size_t)k * (size_t)D + (size_t)col);
            dst[0] = sc_f4{cn[0], cn[1], cn[2], cn[3]};
            dst[1] = sc_f4{cn[4], cn[5], cn[6], cn[7]};
            *reinterpret_cast<sc_f4*>(&sums[k][lane * 8]) = sc_f4{cn[0], cn[1], cn[2], cn[3]};
            *reinterpret_cast<sc_f4*>(&sums[k][lane * 8 + 4]) = sc_f4{cn[4], cn[5], cn[6], cn[7]};
        }
        __syncthreads();
        // ---- assign for the next iteration (against C', held in registers: 8 x K) + rolling reload of the slab for the next chunk ----
        float cr[K][8];
        if (ASSIGN) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const sc_f4 a = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8]), b = *reinterpret_cast<const sc_f4*>(&sums[k][lane * 8 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cr[k][e] = a[e]; cr[k][4 + e] = b[e]; }
            }
        }
        const bool more = cn_next < nchunks;                                  // (wave-uniform)
        const _Float16* xn = X + (size_t)(more ? cn_next : c) * CH + row_first;
        static_for<0, NG>([&](auto gc) {
            constexpr int j0 = decltype(gc)::value * GR;
            __builtin_amdgcn_sched_barrier(0);                                // (GR rows at a time: their butterflies interleave, the rest of the slab stays put)
            float half[GR];                                                   // per row: the value left after the levels 32, 16, 8
            static_for<0, GR>([&](auto jjc) {
                constexpr int jj = decltype(jjc)::value, j = j0 + jj;
                half[jj] = 0.f;
                if constexpr (j < RW) {
                    float x[8];
                    if (ASSIGN) unpack_h8(row[j], x);
                    if (more) {                                               // the registers of row j are free: next chunk's row j
                        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(xn));
                        row[j] = make_uint4(v[0], v[1], v[2], v[3]);
                        if (r0 + j + 1 < T) xn += D;
                    }
                    if (ASSIGN) {
                        float pv[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) pv[k] = 0.f;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            sc_f2 acc = {0.f, 0.f};
#pragma unroll
                            for (int e = 0; e < 8; e += 2) {
                                const sc_f2 xv = {x[e], x[e + 1]}, cv = {cr[k][e], cr[k][e + 1]};
                                const sc_f2 d = xv - cv;
                                acc = __builtin_elementwise_fma(d, d, acc);
                            }
                            pv[k] = acc.x + acc.y;
                        }
                        half[jj] = butterfly8_hi(pv, lane);
                    }
                }
            });
            if (ASSIGN) {
                // second half of the tree (levels 4, 2, 1: plain xor-adds), the eight rows' chains interleaved
#pragma unroll
                for (int jj = 0; jj < GR; ++jj) half[jj] = half[jj] + __shfl_xor(half[jj], 4, 64);
#pragma unroll
                for (int jj = 0; jj < GR; ++jj) half[jj] = half[jj] + __shfl_xor(half[jj], 2, 64);
#pragma unroll
                for (int jj = 0; jj < GR; ++jj) half[jj] = half[jj] + __shfl_xor(half[jj], 1, 64);
                const int k = lane >> 3;
#pragma unroll
                for (int jj = 0; jj < GR; ++jj) {
                    const int j = j0 + jj;
                    if (j < RW && ((live[j >> 6] >> (j & 63)) & 1ull) && (lane & 7) == 0 && k < K) partial[(size_t)c * I + (size_t)(r0 + j) * K + k] = half[jj];
                }
            }
        });
        __syncthreads();                                                      // `sums` is rewritten by the next chunk
    }
}

// per-cluster squared shift totals ||C_k - C'_k||^2 of clusters [kb, kb + kn) from the per-chunk partials (SC-KM1: 32 contiguous chunk
// segments in fp64, then the segment sums in fp64); result in tot[0..kn) (shared), valid after the trailing barrier
__device__ __forceinline__ void km_shift_totals(const float* __restrict__ dpart, int K, int kb, int kn, int64_t nchunks, double* segs, double* tot) {
    const int64_t seglen = (nchunks + NSEG - 1) / NSEG;
    for (int p = threadIdx.x; p < NSEG * kn; p += blockDim.x) {
        const int s = p / kn, k = kb + p % kn;
        int64_t lo = (int64_t)s * seglen, hi = lo + seglen;
        if (hi > nchunks) hi = nchunks;
        double a = 0.0;
        int64_t c = lo;
        for (; c + 8 <= hi; c += 8) {                 // 8 independent loads in flight, added in ascending chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dpart[(size_t)(c + u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        for (; c < hi; ++c) a += (double)dpart[(size_t)c * K + k];
        segs[s * 64 + (k - kb)] = a;
    }
    __syncthreads();
    if ((int)threadIdx.x < kn) {
        double t = 0.0;
        for (int s = 0; s < NSEG; ++s) t += segs[s * 64 + threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
}

// single block: shift = sum_k sqrt(total_k); decide convergence; advance state
__global__ __launch_bounds__(256) void km_decide(const float* __restrict__ dpart, KmState* __restrict__ st, int K, int64_t nchunks,
                                                 int iter, int max_iter, float tol, int n_reseed) {
    if (st->done) return;
    __shared__ double segs[NSEG * 64];   // K <= 64 per pass
    __shared__ double tot[64];
    double diff = 0.0;
    for (int kb = 0; kb < K; kb += 64) {
        const int kn = (K - kb) < 64 ? (K - kb) : 64;
        km_shift_totals(dpart, K, kb, kn, nchunks, segs, tot);
        if (threadIdx.x == 0)
            for (int k = 0; k < kn; ++k) diff += sqrt(tot[k]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (st->n_empty > 0) {
            if (st->reseed_pos + st->n_empty > n_reseed) st->status = 1;
            st->reseed_pos += st->n_empty;
        }
        if (diff < (double)tol) {
            st->done = 1;
            st->exit_iter = iter;
        } else {
            st->cur ^= 1;
            if (iter == max_iter - 1) { st->done = 1; st->exit_iter = iter; }
        }
    }
}

// single block: shift2[k] = ||C_k - C'_k||^2 (fp64) for the caller of sc_kmeans_update; W -> wsum
__global__ __launch_bounds__(256) void km_shift_out(const float* __restrict__ dpart, const float* __restrict__ W, double* __restrict__ shift2,
                                                    float* __restrict__ wsum, int K, int64_t nchunks) {
    __shared__ double segs[NSEG * 64];
    __shared__ double tot[64];
    for (int kb = 0; kb < K; kb += 64) {
        const int kn = (K - kb) < 64 ? (K - kb) : 64;
        km_shift_totals(dpart, K, kb, kn, nchunks, segs, tot);
        if ((int)threadIdx.x < kn && shift2) shift2[kb + threadIdx.x] = tot[threadIdx.x];
        __syncthreads();
    }
    if (wsum)
        for (int k = threadIdx.x; k < K; k += blockDim.x) wsum[k] = W[k];
}

__global__ void km_labels_in(const int64_t* __restrict__ labels, int* __restrict__ labels32, int T, int K) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) { const int64_t l = labels[t]; labels32[t] = (l < 0 || l >= K) ? 0 : (int)l; }
}

// C[k] = X[init_idx[k]] as fp32 (+ state reset).  Grid (column blocks, K): 8 consecutive columns per thread as one 16-byte load and two 16-byte
// stores where D allows it (round 4: the element-wise version spent 49 us on a 64-bit division per element; 41 MB at K = 5)
template <typename Tag, bool VEC>
__global__ __launch_bounds__(256) void km_init(const void* __restrict__ X, const int* __restrict__ init_idx, float* __restrict__ C, KmState* st,
                                                 int T, int64_t D, int K) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { st->done = 0; st->exit_iter = 0; st->cur = 0; st->reseed_pos = 0; st->status = 0; st->n_empty = 0; }
    const int k = blockIdx.y;
    int r = init_idx[k];
    if (r < 0 || r >= T) r = 0;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (j >= D) return;
    float v[8];
    if (VEC) {
        sc_load8<Tag>(X, (size_t)r * (size_t)D + (size_t)j, v);
        sc_f4* dst = reinterpret_cast<sc_f4*>(C + (size_t)k * (size_t)D + (size_t)j);
        dst[0] = sc_f4{v[0], v[1], v[2], v[3]};
        dst[1] = sc_f4{v[4], v[5], v[6], v[7]};
    } else {
        for (int e = 0; e < 8; ++e)
            if (j + e < D) C[(size_t)k * (size_t)D + (size_t)(j + e)] = sc_load1<Tag>(X, (size_t)r * (size_t)D + (size_t)(j + e));
    }
}

__global__ void km_finalize(const float* __restrict__ Ca, const float* __restrict__ Cb, const KmState* __restrict__ st,
                            const int* __restrict__ labels32, const float* __restrict__ W, float* __restrict__ Cout,
                            int64_t* __restrict__ labels, float* __restrict__ wsum, int* __restrict__ info, int T, int64_t D, int K) {
    const float* C = st->cur ? Cb : Ca;
    const size_t n = (size_t)K * (size_t)D;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(Cout) | reinterpret_cast<uintptr_t>(C)) & 15) == 0) {
        const sc_f4* src = reinterpret_cast<const sc_f4*>(C);
        sc_f4* dst = reinterpret_cast<sc_f4*>(Cout);
        for (size_t i = gid; i < n / 4; i += gsz) dst[i] = src[i];
    } else {
        for (size_t i = gid; i < n; i += gsz) Cout[i] = C[i];
    }
    for (size_t t = gid; t < (size_t)T; t += gsz) labels[t] = labels32[t];
    for (size_t k = gid; k < (size_t)K; k += gsz) wsum[k] = W[k];
    if (gid == 0) { info[0] = st->exit_iter; info[1] = st->status; info[2] = st->reseed_pos; info[3] = 0; }
}

__global__ void km_set_state(KmState* st, int cur) {
    st->done = 0; st->exit_iter = 0; st->cur = cur; st->reseed_pos = 0; st->status = 0; st->n_empty = 0;
}
__global__ void km_labels_out(const int* __restrict__ labels32, int64_t* __restrict__ labels, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) labels[t] = labels32[t];
}

// ---- workspace carve --------------------------------------------------------------------------
struct KmWs {
    KmState* st; float* Ca; float* Cb; float* partial; double* seg; float* dpart; int* labels32; int* order; int* start;
    float* W; int* empty_rank; size_t bytes;
};
KmWs carve(void* base, int T, int64_t D, int K) {
    const int64_t nch = (D + CH - 1) / CH;
    const size_t I = (size_t)T * K;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = sc_align_up(off + bytes, 256); return p ? (void*)(p + o) : (void*)nullptr; };
    KmWs w;
    w.st = (KmState*)take(sizeof(KmState));
    w.Ca = (float*)take(sizeof(float) * (size_t)K * D);
    w.Cb = (float*)take(sizeof(float) * (size_t)K * D);
    w.partial = (float*)take(sizeof(float) * (size_t)nch * I);
    w.seg = (double*)take(sizeof(double) * NSEG * I);
    w.dpart = (float*)take(sizeof(float) * (size_t)nch * K);
    w.labels32 = (int*)take(sizeof(int) * T);
    w.order = (int*)take(sizeof(int) * T);
    w.start = (int*)take(sizeof(int) * (K + 1));
    w.W = (float*)take(sizeof(float) * K);
    w.empty_rank = (int*)take(sizeof(int) * K);
    w.bytes = off;
    return w;
}

template <typename Tag, int KB>
void launch_assign_kb(bool vec, const void* X, const KmWs& w, int T, int64_t D, int K, int k0, int64_t nch, hipStream_t s) {
    const dim3 grid((unsigned)((nch + WPB - 1) / WPB)), block(WPB * 64);
    if (vec) hipLaunchKernelGGL((km_assign<Tag, KB, true>), grid, block, 0, s, X, w.Ca, w.Cb, w.st, w.partial, T, D, K, k0, nch);
    else hipLaunchKernelGGL((km_assign<Tag, KB, false>), grid, block, 0, s, X, w.Ca, w.Cb, w.st, w.partial, T, D, K, k0, nch);
}
template <typename Tag>
void launch_assign(bool vec, const void* X, const KmWs& w, int T, int64_t D, int K, int64_t nch, hipStream_t s) {
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int kb = (K - k0) < 16 ? (K - k0) : 16;
        switch (kb) {
#define SC_CASE(n) case n: launch_assign_kb<Tag, n>(vec, X, w, T, D, K, k0, nch, s); break;
            SC_CASE(1) SC_CASE(2) SC_CASE(3) SC_CASE(4) SC_CASE(5) SC_CASE(6) SC_CASE(7) SC_CASE(8)
            SC_CASE(9) SC_CASE(10) SC_CASE(11) SC_CASE(12) SC_CASE(13) SC_CASE(14) SC_CASE(15) SC_CASE(16)
#undef SC_CASE
        }
    }
}

template <typename Tag>
int fit_impl(const void* X, int T, int64_t D, int K, const float* wts, const int32_t* init_idx, const int32_t* reseed_idx,
             int n_reseed, int max_iter, float tol, float* C, int64_t* labels, float* wsum, int32_t* info, void* ws,
             hipStream_t s) {
    const KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH;
    const size_t I = (size_t)T * K;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    {
        const dim3 igrid((unsigned)((D + 2047) / 2048), (unsigned)K);
        if (vec) hipLaunchKernelGGL((km_init<Tag, true>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
        else hipLaunchKernelGGL((km_init<Tag, false>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
    }
    const dim3 sgrid((unsigned)((nch + WPB - 1) / WPB)), sblock(WPB * 64);
    const dim3 rgrid((unsigned)((I + 255) / 256), NSEG);
    // one-pass iterations (km_fused: update(i) + assign(i + 1) from a register-resident slab): fp16 rows, whole 512-column chunks, K = 5 or 8,
    // T <= 400.  OPT-IN (SC_KM_FUSED=1): bit-identical to the two-kernel path (tests/test_gpu_kmeans_fused.py) and it does read X once per
    // iteration, but it is SLOWER on MI355X - 1.47 ms per Lloyd iteration against 0.66 at T = 400, K = 5 (0.26 against 0.20 at T = 64, K = 8;
    // profiles/r05_run_j_kmeans_fused.md): holding the 400 KiB slab takes one wave per SIMD with all 512 registers, and with nothing to
    // switch to every dependent step of a row's distance tree (packed fma chain, three lane swaps, four LDS-crossbar shuffles) is paid at
    // full latency - ~2000 cycles per row and wave, 6x the HBM time of the row.  km_assign hides exactly that with 8+ waves per SIMD.
    static int fused_on = -1, n_cu = 0;
    if (fused_on < 0) { const char* e = getenv("SC_KM_FUSED"); fused_on = (e && e[0] == '1') ? 1 : 0; }
    const bool fused = fused_on && std::is_same<Tag, ScF16>::value && vec && D % CH == 0 && (K == 5 || K == 8) && T <= FNW * 100;
    if (fused && n_cu == 0) { int dev = 0, n = 0; (void)hipGetDevice(&dev); n_cu = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
    auto launch_fused = [&](int do_assign) {
        const unsigned g = (unsigned)(nch < sc_launch_cu_count(n_cu, s) ? nch : sc_launch_cu_count(n_cu, s));
        const int rw = (T + FNW - 1) / FNW;
#define SC_KF3(RWV, KV, AV) hipLaunchKernelGGL((km_fused<RWV, KV, AV>), dim3(g), dim3(FNW * 64), 0, s, (const _Float16*)X, w.Ca, w.Cb, w.st, wts, w.labels32, w.W, \
                                               w.empty_rank, reseed_idx, n_reseed, w.dpart, w.partial, T, D, nch)
#define SC_KF2(RWV, KV) do { if (do_assign) SC_KF3(RWV, KV, true); else SC_KF3(RWV, KV, false); } while (0)
#define SC_KF(RWV) do { if (K == 5) SC_KF2(RWV, 5); else SC_KF2(RWV, 8); } while (0)
        if (rw <= 16) SC_KF(16); else if (rw <= 32) SC_KF(32); else if (rw <= 64) SC_KF(64); else SC_KF(100);
#undef SC_KF
#undef SC_KF2
#undef SC_KF3
    };
    for (int it = 0; it < max_iter; ++it) {
        if (!fused || it == 0) launch_assign<Tag>(vec, X, w, T, D, K, nch, s);          // (fused: the partials of iteration it > 0 were written by km_fused(it - 1))
        hipLaunchKernelGGL(km_reduce, rgrid, dim3(256), 0, s, w.partial, w.st, w.seg, I, nch, 1);
        hipLaunchKernelGGL(km_argmin, dim3((T + 63) / 64), dim3(64), 0, s, w.seg, w.st, w.labels32, (double*)nullptr, T, K, 1);
        hipLaunchKernelGGL(km_order, dim3(1), dim3(1024), sizeof(int) * (K + 16), s, w.st, wts, w.labels32, w.order, w.start, w.W, w.empty_rank, T, K, 1);
        if (fused) launch_fused(it + 1 < max_iter ? 1 : 0);
        else if (vec) hipLaunchKernelGGL((km_update<Tag, true>), sgrid, sblock, 0, s, X, w.Ca, w.Cb, w.st, wts, w.order, w.start, w.W,
                                         w.empty_rank, reseed_idx, n_reseed, w.dpart, T, D, K, nch, 0);
        else hipLaunchKernelGGL((km_update<Tag, false>), sgrid, sblock, 0, s, X, w.Ca, w.Cb, w.st, wts, w.order, w.start, w.W,
                                w.empty_rank, reseed_idx, n_reseed, w.dpart, T, D, K, nch, 0);
        hipLaunchKernelGGL(km_decide, dim3(1), dim3(256), 0, s, w.dpart, w.st, K, nch, it, max_iter, tol, n_reseed);
    }
    hipLaunchKernelGGL(km_finalize, dim3(1024), dim3(256), 0, s, w.Ca, w.Cb, w.st, w.labels32, w.W, C, labels, wsum, info, T, D, K);
    SC_CHECK_LAUNCH("sc_kmeans_fit");
    return SC_OK;
}

template <typename Tag>
int assign_impl(const void* X, int T, int64_t D, int K, const float* C, int64_t* labels, double* dist2, void* ws, hipStream_t s) {
    KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH;
    const size_t I = (size_t)T * K;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    hipLaunchKernelGGL(km_set_state, dim3(1), dim3(1), 0, s, w.st, 0);
    w.Ca = const_cast<float*>(C);   // read-only use: state.cur == 0 selects Ca
    launch_assign<Tag>(vec, X, w, T, D, K, nch, s);
    hipLaunchKernelGGL(km_reduce, dim3((unsigned)((I + 255) / 256), NSEG), dim3(256), 0, s, w.partial, w.st, w.seg, I, nch, 0);
    hipLaunchKernelGGL(km_argmin, dim3((T + 63) / 64), dim3(64), 0, s, w.seg, w.st, w.labels32, dist2, T, K, 0);
    hipLaunchKernelGGL(km_labels_out, dim3((T + 255) / 256), dim3(256), 0, s, w.labels32, labels, T);
    SC_CHECK_LAUNCH("sc_kmeans_assign");
    return SC_OK;
}

// one centroid update from given labels: C_new[k] = sum_{t: label t = k} w_t x_t / W_k (rows in ascending order, SC-KM1), an empty
// cluster takes row fill_idx[its rank among the empty clusters] (empty_mode 0) or the zero vector (empty_mode 1)
template <typename Tag>
int update_impl(const void* X, int T, int64_t D, int K, const float* wts, const int64_t* labels, const float* C_old, int empty_mode,
                const int32_t* fill_idx, int n_fill, float* C_new, float* wsum, double* shift2, void* ws, hipStream_t s) {
    KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C_old) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(C_new) & 15) == 0);
    hipLaunchKernelGGL(km_set_state, dim3(1), dim3(1), 0, s, w.st, 0);
    hipLaunchKernelGGL(km_labels_in, dim3((T + 255) / 256), dim3(256), 0, s, labels, w.labels32, T, K);
    hipLaunchKernelGGL(km_order, dim3(1), dim3(1024), sizeof(int) * (K + 16), s, w.st, wts, w.labels32, w.order, w.start, w.W, w.empty_rank, T, K, 0);
    const dim3 sgrid((unsigned)((nch + WPB - 1) / WPB)), sblock(WPB * 64);
    float* Ca = const_cast<float*>(C_old);        // state.cur == 0: Ca is read (old centroids), Cb written
    if (vec) hipLaunchKernelGGL((km_update<Tag, true>), sgrid, sblock, 0, s, X, Ca, C_new, w.st, wts, w.order, w.start, w.W, w.empty_rank,
                                fill_idx, n_fill, w.dpart, T, D, K, nch, empty_mode);
    else hipLaunchKernelGGL((km_update<Tag, false>), sgrid, sblock, 0, s, X, Ca, C_new, w.st, wts, w.order, w.start, w.W, w.empty_rank,
                            fill_idx, n_fill, w.dpart, T, D, K, nch, empty_mode);
    hipLaunchKernelGGL(km_shift_out, dim3(1), dim3(256), 0, s, w.dpart, w.W, shift2, wsum, K, nch);
    SC_CHECK_LAUNCH("sc_kmeans_update");
    return SC_OK;
}

}  // namespace

extern "C" size_t sc_kmeans_workspace_bytes(int T, int64_t D, int K) {
    if (T <= 0 || D <= 0 || K <= 0) return 0;
    return carve(nullptr, T, D, K).bytes;
}

extern "C" int sc_kmeans_fit(const void* X, int dtype, int T, int64_t D, int K, const float* w, const int32_t* init_idx,
                             const int32_t* reseed_idx, int n_reseed, int max_iter, float tol, float* C, int64_t* labels,
                             float* wsum, int32_t* info, void* ws, size_t ws_bytes, sc_stream_t stream) {
    SC_REQUIRE(X && init_idx && C && labels && wsum && info && ws, "sc_kmeans_fit: null pointer argument");
    SC_REQUIRE(T > 0 && D > 0 && K > 0 && max_iter > 0, "sc_kmeans_fit: T, D, K, max_iter must be positive");
    SC_REQUIRE(n_reseed >= 0 && (n_reseed == 0 || reseed_idx), "sc_kmeans_fit: n_reseed > 0 needs reseed_idx");
    if (ws_bytes < sc_kmeans_workspace_bytes(T, D, K))
        return sc_fail(SC_ERR_WORKSPACE, "sc_kmeans_fit: workspace %zu < required %zu", ws_bytes, sc_kmeans_workspace_bytes(T, D, K));
    SC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "sc_kmeans_fit: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case SC_F16: return fit_impl<ScF16>(X, T, D, K, w, init_idx, reseed_idx, n_reseed, max_iter, tol, C, labels, wsum, info, ws, s);
        case SC_BF16: return fit_impl<ScBF16>(X, T, D, K, w, init_idx, reseed_idx, n_reseed, max_iter, tol, C, labels, wsum, info, ws, s);
        case SC_F32: return fit_impl<ScF32>(X, T, D, K, w, init_idx, reseed_idx, n_reseed, max_iter, tol, C, labels, wsum, info, ws, s);
    }
    return sc_fail(SC_ERR_ARG, "sc_kmeans_fit: unknown dtype %d", dtype);
}

extern "C" int sc_kmeans_update(const void* X, int dtype, int T, int64_t D, int K, const float* w, const int64_t* labels, const float* C_old,
                                int empty_mode, const int32_t* fill_idx, int n_fill, float* C_new, float* wsum, double* shift2, void* ws,
                                size_t ws_bytes, sc_stream_t stream) {
    SC_REQUIRE(X && labels && C_old && C_new && ws, "sc_kmeans_update: null pointer argument");
    SC_REQUIRE(T > 0 && D > 0 && K > 0, "sc_kmeans_update: T, D, K must be positive");
    SC_REQUIRE(C_old != C_new, "sc_kmeans_update: C_new must not alias C_old");
    SC_REQUIRE(empty_mode == 0 || empty_mode == 1, "sc_kmeans_update: empty_mode must be 0 (fill rows) or 1 (zero vector)");
    SC_REQUIRE(n_fill >= 0 && (n_fill == 0 || fill_idx), "sc_kmeans_update: n_fill > 0 needs fill_idx");
    if (ws_bytes < sc_kmeans_workspace_bytes(T, D, K))
        return sc_fail(SC_ERR_WORKSPACE, "sc_kmeans_update: workspace %zu < required %zu", ws_bytes, sc_kmeans_workspace_bytes(T, D, K));
    SC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "sc_kmeans_update: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case SC_F16: return update_impl<ScF16>(X, T, D, K, w, labels, C_old, empty_mode, fill_idx, n_fill, C_new, wsum, shift2, ws, s);
        case SC_BF16: return update_impl<ScBF16>(X, T, D, K, w, labels, C_old, empty_mode, fill_idx, n_fill, C_new, wsum, shift2, ws, s);
        case SC_F32: return update_impl<ScF32>(X, T, D, K, w, labels, C_old, empty_mode, fill_idx, n_fill, C_new, wsum, shift2, ws, s);
    }
    return sc_fail(SC_ERR_ARG, "sc_kmeans_update: unknown dtype %d", dtype);
}

extern "C" int sc_kmeans_assign(const void* X, int dtype, int T, int64_t D, int K, const float* C, int64_t* labels, double* dist2,
                                void* ws, size_t ws_bytes, sc_stream_t stream) {
    SC_REQUIRE(X && C && labels && ws, "sc_kmeans_assign: null pointer argument");
    SC_REQUIRE(T > 0 && D > 0 && K > 0, "sc_kmeans_assign: T, D, K must be positive");
    if (ws_bytes < sc_kmeans_workspace_bytes(T, D, K))
        return sc_fail(SC_ERR_WORKSPACE, "sc_kmeans_assign: workspace %zu < required %zu", ws_bytes, sc_kmeans_workspace_bytes(T, D, K));
    SC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "sc_kmeans_assign: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case SC_F16: return assign_impl<ScF16>(X, T, D, K, C, labels, dist2, ws, s);
        case SC_BF16: return assign_impl<ScBF16>(X, T, D, K, C, labels, dist2, ws, s);
        case SC_F32: return assign_impl<ScF32>(X, T, D, K, C, labels, dist2, ws, s);
    }
    return sc_fail(SC_ERR_ARG, "sc_kmeans_assign: unknown dtype %d", dtype);
}
