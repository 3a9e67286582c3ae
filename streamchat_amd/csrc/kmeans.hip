// Selective-frame k-means for gfx950 (CDNA4) — the HBM-streaming replacement of the reference's
// `weighted_kmeans_feature` (utiles.py:291-330).  Each "point" is a whole frame of
// D = 576*3584 = 2,064,384 features and K is tiny (5..8), so this is a pure streaming reduce:
// the reference's [T,K,D] broadcast temporary (utiles.py:299, 8.3 GB at T=400) is never formed.
//
// Reduction spec "SC-KM2" (round 6), shared bit for bit with oracle/kmeans_oracle.c:
//   cell  = 8 columns: d = x - c, even / odd fma chains, p = acc0 + acc1          (what a LANE that owns 8 columns computes)
//   slice = KM_SW columns: adjacent-pair tree over its cells                      (8- / 16-lane butterfly, or one THREAD that owns the row)
//   group = 2048 columns: slice partials added in ascending order in fp64
//   total = 32 contiguous segments of groups in fp64, then the segment sums; argmin = first minimum of the fp64 totals
//   update sums: per column 16 chains per cluster - rank j of the cluster's rows (ascending) feeds chain ((gs_k + (j >> 3)) & 7, j & 1), gs_k =
//           the cluster's first 8-row group when the clusters are laid out one after the other in whole 8-row groups; chains sequential fp32,
//           no contraction; u_a = chain(a, 0) + chain(a, 1), S = u_0 + u_1 + .. + u_7 in that order, C' = S / W; shift: the distance structure on (C - C')^2
// SC-KM1 (rounds 1-5) had a 64-lane tree per (row, cluster, 512-column chunk); that tree is why the one-read pass of round 5 (km_fused) could
// only hold its slab in the register file of ONE wave per SIMD and ran 1.6 - 2.2x slower than two passes.  Under SC-KM2 nothing above the
// cell needs a cross-lane step when a thread owns a row, so the slab can live in LDS and the pass runs at normal occupancy.
//
// Two kernel families, same bits:
//   km2_pass<K, RGW, MODE>   (fp16 rows, D % 64 == 0, 2 <= K <= 8, T >= 120 (K <= 5) / 144 (K >= 6), T + 7 K <= 448)   ONE read of X per Lloyd iteration.
//               A workgroup owns a 2048-column group and walks its 32 slices; the [T, 64] fp16 slab of a slice is gathered into LDS by
//               LDS-DMA (buffer_load ... lds) SORTED BY CLUSTER and used twice - the update of iteration i and, against the C' that
//               comes out of it, the distances of iteration i + 1.  Every wave owns the same 8-row groups of the slab for the DMA, the
//               update and the assign, so the slab needs no barrier and the next slice's rows are requested while this slice's are
//               measured (details at the kernel).  MODE 2 = assign only (iteration 0 and sc_kmeans_assign), 3 = both; the last iteration's
//               update-only pass and every shape outside the range above run on
//   km_assign / km_update        (any dtype, any D, K, T)   two passes over X per iteration; lane = 8 columns, rows streamed through registers.
// Both are followed per iteration by km_reduce (segments) -> km_argmin -> km_order (stable counting sort, W[k], empty ranks) and km_decide
// (sum_k ||C_i - C'||_2 < tol ? -> device-side `done` flag, no host round trip).
// Compiled with -ffp-contract=off: every fma below is explicit.
#include "sc_common.h"
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

#ifndef KM_SW
#define KM_SW 64             // columns per slice (spec constant: oracle/kmeans_oracle.c SC_SLICE)
#endif
constexpr int CH = 512;      // columns per chunk of the lane-mapped kernels = 64 lanes x 8 elements
constexpr int GW = 2048;     // columns per fp64 group (spec constant) = the 4 chunks of one workgroup
constexpr int SW = KM_SW;
constexpr int NCELL = SW / 8;            // cells per slice
constexpr int NSL = GW / SW;             // slices per group
constexpr int SPC = CH / SW;             // slices per chunk
constexpr int NSEG = 32;     // fp64 segments
constexpr int WPB = 4;       // waves per block in the lane-mapped kernels: one group
static_assert(SW == 64 || SW == 128, "KM_SW");
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

// One level of the halving butterfly: the lanes that differ in bit log2(L) split the N live values between them - the lane with the bit
// clear keeps v[0 .. N/2), the other v[N/2 .. N) (stored back at 0 .. N/2) - and each adds the partner's partial of the values it keeps.
template <int L, int N>
__device__ __forceinline__ void bfly_halve(float (&v)[64], int lane) {
    const bool up = (lane & L) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float send = up ? v[i] : v[i + N / 2];
        const float keep = up ? v[i + N / 2] : v[i];
        v[i] = keep + __shfl_xor(send, L, 64);
    }
}

// SC-KM2 slice tree for 64 items at once: on entry lane L holds its cell partial v[i] of 64 items; on exit v[0 .. NV) (NV = 64 / NCELL)
// hold, for the slice made of this lane's NCELL-lane row, the slice partials of the items
//   item(i) = i + sum_b bit_b(lane) * (32 >> b),  b < log2(NCELL)
// (levels = lane bits 0, 1, 2(, 3): adjacent-pair order).  ~1.9 ops per value.
constexpr int NV = 64 / NCELL;
__device__ __forceinline__ void butterfly_slice(float (&v)[64], int lane) {
    bfly_halve<1, 64>(v, lane);
    bfly_halve<2, 32>(v, lane);
    bfly_halve<4, 16>(v, lane);
    if (NCELL == 16) bfly_halve<8, 8>(v, lane);
}
__device__ __forceinline__ int butterfly_item(int i, int lane) {
    int it = i + (lane & 1) * 32 + ((lane >> 1) & 1) * 16 + ((lane >> 2) & 1) * 8;
    if (NCELL == 16) it += ((lane >> 3) & 1) * 4;
    return it;
}
// slice partial of one value per lane (a cell partial): every lane of the NCELL-lane row returns the row's tree sum
__device__ __forceinline__ float slice_tree_sum(float v) {
#pragma unroll
    for (int m = 1; m < NCELL; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

__host__ __device__ constexpr int tt_for(int kb) { return (64 / kb) < 16 ? (64 / kb) : 16; }

// ---------------------------------------------------------------------------------------------
template <typename Tag, int KB, bool VEC>
__global__ __launch_bounds__(WPB * 64) void km_assign(const void* __restrict__ X, const float* __restrict__ Ca,
                                                      const float* __restrict__ Cb, const KmState* __restrict__ st,
                                                      double* __restrict__ gpart, int T, int64_t D, int K, int k0,
                                                      int64_t nchunks) {
    constexpr int TT = tt_for(KB);
    if (st->done) return;
    __shared__ float sl[64][NSL + 1];                  // slice partials of the row group's 64 items over the NSL slices of this workgroup's group
    const float* __restrict__ C = st->cur ? Cb : Ca;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * WPB + wave;           // (a chunk past the end contributes zeros: its waves still join the barriers)
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
        butterfly_slice(p, lane);
        {
            const int sidx = wave * SPC + lane / NCELL;           // slice of the group
#pragma unroll
            for (int i = 0; i < NV; ++i) sl[butterfly_item(i, lane)][sidx] = p[i];
        }
        __syncthreads();
        if (threadIdx.x < TT * KB) {                              // group total of one item: its NSL slice partials in ascending order, fp64
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < NSL; ++q) a += (double)sl[threadIdx.x][q];
            const int tt = threadIdx.x / KB, k = threadIdx.x - tt * KB;
            const int t = g * TT + tt;
            if (t < T) gpart[(size_t)blockIdx.x * I + (size_t)t * K + (k0 + k)] = a;
        }
        __syncthreads();
        if (VEC) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) cur[tt] = nxt[tt];
        }
    }
}

// level-1 fp64 reduce: seg[s][item] = sum_{g in segment s, ascending} gpart[g][item].  `seglen` groups per segment; the matrix at hand holds
// segments [seg_first, seg_first + seg_count) of the 32 (the whole matrix: 0, 32; a column slab of sc_kmeans_fit_cols: the segments it owns -
// its local group 0 is the first group of segment seg_first, the other rows of `seg` are left to the exchange)
__global__ void km_reduce(const double* __restrict__ partial, const KmState* __restrict__ st, double* __restrict__ seg,
                          size_t I, int64_t nchunks, int check_done, int64_t seglen, int seg_first, int seg_count) {
    if (check_done && st->done) return;
    const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= I) return;
    const int s = blockIdx.y;
    if (s < seg_first || s >= seg_first + seg_count) return;
    int64_t lo = (int64_t)(s - seg_first) * seglen, hi = lo + seglen;
    if (lo > nchunks) lo = nchunks;
    if (hi > nchunks) hi = nchunks;
    double a = 0.0;
    int64_t c = lo;
    for (; c + 8 <= hi; c += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(c + u) * I + item];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; c < hi; ++c) a += partial[(size_t)c * I + item];
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
// dynamic LDS of km_order: counts[K] | wave_cnt[16] | (weighted fits of up to KM_ORDER_STAGE rows) order[T] | w[T]
constexpr int KM_ORDER_STAGE = 4096;
inline size_t km_order_lds(int T, int K) { return sizeof(int) * (size_t)(K + 16) + (T <= KM_ORDER_STAGE ? 8u * (size_t)T : 0u); }
__device__ __forceinline__ void km_order_body(KmState* __restrict__ st, const float* __restrict__ w, const int* __restrict__ labels32,
                                              int* __restrict__ order, int* __restrict__ start, float* __restrict__ W,
                                              int* __restrict__ empty_rank, int T, int K, int* sm) {
    int* counts = sm;                         // counts[K] | wave_cnt[16]
    int* wcnt = sm + K;
    const bool stage = w && T <= KM_ORDER_STAGE;                 // weighted: the sequential W sums read their rows out of LDS
    int* ord_s = sm + K + 16;
    float* w_s = reinterpret_cast<float*>(ord_s + T);
    if (stage)
        for (int t = threadIdx.x; t < T; t += blockDim.x) w_s[t] = w[t];
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
                const int pos = start[k] + counts[k] + before + __popcll(m & ((1ull << lane) - 1ull));
                order[pos] = t;
                if (stage) ord_s[pos] = t;
            }
            __syncthreads();
            if (threadIdx.x == 0) { int tot = 0; for (int i = 0; i < nw; ++i) tot += wcnt[i]; counts[k] += tot; }
            __syncthreads();
        }
    }
    // weighted: W[k] = sequential fp32 sum over the cluster's rows in ascending order (SC-KM1).  The ADDS are sequential; the loads go out eight
    // at a time and, for fits of up to KM_ORDER_STAGE rows, read LDS copies of `order` and `w` (two dependent GLOBAL round trips per row made
    // this loop 20 of the 25 us of the launch at T = 400 - and every call of weighted_kmeans_feature is a weighted one: it passes ones)
    if (w) {
        __syncthreads();                      // `order` / ord_s / w_s were written by other threads of this block
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float ws = 0.f;
            int i = start[k];
            const int e = start[k + 1];
            for (; i + 8 <= e; i += 8) {
                int o[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = stage ? ord_s[i + u] : order[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = stage ? w_s[o[u]] : w[o[u]];
#pragma unroll
                for (int u = 0; u < 8; ++u) ws = ws + v[u];
            }
            for (; i < e; ++i) ws = ws + (stage ? w_s[ord_s[i]] : w[order[i]]);
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

__global__ __launch_bounds__(1024) void km_order(KmState* __restrict__ st, const float* __restrict__ w, const int* __restrict__ labels32,
                                                 int* __restrict__ order, int* __restrict__ start, float* __restrict__ W,
                                                 int* __restrict__ empty_rank, int T, int K, int check_done) {
    if (check_done && st->done) return;
    extern __shared__ int sm[];
    km_order_body(st, w, labels32, order, start, W, empty_rank, T, K, sm);
}

// one wave per chunk: new centroids + shift partials
template <typename Tag, bool VEC>
__global__ __launch_bounds__(WPB * 64) void km_update(const void* __restrict__ X, float* __restrict__ Ca, float* __restrict__ Cb,
                                                      KmState* __restrict__ st, const float* __restrict__ w,
                                                      const int* __restrict__ order, const int* __restrict__ start,
                                                      const float* __restrict__ W, const int* __restrict__ empty_rank,
                                                      const int* __restrict__ reseed_idx, int n_reseed,
                                                      double* __restrict__ dgpart, int T, int64_t D, int K, int64_t nchunks, int empty_zero) {
    if (st->done) return;
    extern __shared__ float sh[];                      // [K][NSL] slice partials of the shift over this workgroup's group
    const float* __restrict__ Cold = st->cur ? Cb : Ca;
    float* __restrict__ Cnew = st->cur ? Ca : Cb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * WPB + wave;           // (a chunk past the end contributes zeros and still joins the barrier)
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
            // SC-KM2 update: rank j of the cluster's rows feeds chain ((gs_k + (j >> 3)) & 7, j & 1), gs_k = the cluster's first 8-row group in
            // the cluster-sorted layout: chain group a takes the 8-rank batches b with (gs_k + b) & 7 == a, even ranks of a batch into s0, odd
            // ones into s1; u_a = s0 + s1; the u_a are added in ascending a
            int gsk = 0;
            for (int kk = 0; kk < k; ++kk) gsk += (start[kk + 1] - start[kk] + 7) >> 3;
            for (int a = 0; a < 8; ++a) {
                float s0[8], s1[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
                for (int i0 = lo + U * ((a - gsk) & 7); i0 < hi; i0 += 8 * U) {
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
                            for (int e = 0; e < 8; ++e) {                     // mul, then add (1.0f * x == x exactly)
                                if (u & 1) s1[e] = w ? s1[e] + wt[u] * x[u][e] : s1[e] + x[u][e];
                                else s0[e] = w ? s0[e] + wt[u] * x[u][e] : s0[e] + x[u][e];
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float u = s0[e] + s1[e]; cn[e] = (a == 0) ? u : cn[e] + u; }
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
        const float sp = slice_tree_sum(a0 + a1);
        if (lane % NCELL == 0) sh[k * NSL + wave * SPC + lane / NCELL] = sp;
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
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {            // group total: slice partials in ascending order, fp64
        double a = 0.0;
        for (int q = 0; q < NSL; ++q) a += (double)sh[k * NSL + q];
        dgpart[(size_t)blockIdx.x * K + k] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6: update(i) + assign(i + 1) in ONE pass over X from an LDS-resident slab (see the header).  fp16 rows, D % KM_SW == 0.
// ---------------------------------------------------------------------------------------------
// compile-time loops (a recursive template is unrolled by construction, whatever the unroller's budget says)
template <int J, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (J < N) { f(std::integral_constant<int, J>{}); static_for<J + 1, N>(f); }
}

// 16-byte LDS-DMA through a raw buffer resource: lane l's 16 bytes land at lds + 16 l
__device__ __forceinline__ void km2_dma16(const void* base, unsigned extent, char* lds, unsigned voff, unsigned soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)extent, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
// v + (v of the lane a DPP control selects): one v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float km2_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

#ifdef KM2_TRACE
// diagnostic build (tools/trace_km2.py): shader cycles of wave 0 per phase, summed over all workgroups and slices
__device__ unsigned long long km2_trace[8];
#define KM2_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); tr[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define KM2_STAMP(i) do { } while (0)
#endif
constexpr int KM2_NW = 8;                                                     // waves per workgroup
constexpr int KM2_ROWB = SW * 2;                                              // bytes per slab row
// LDS: slab | new centroid slice [K][64] | per-wave chain sums [8][K][64] | weight of every slab position
__host__ __device__ constexpr size_t km2_lds_bytes(int K, int RGW) {
    return (size_t)RGW * 64 * KM2_ROWB + (size_t)K * SW * 4 + (size_t)KM2_NW * K * SW * 4 + (size_t)RGW * 64 * 4;
}
// waves per SIMD the register allocation must leave room for: as many workgroups as the LDS admits on a CU (at most 32 waves), over 4 SIMDs
__host__ __device__ constexpr int km2_min_waves_per_simd(int K, int RGW) {
    int wgs = (int)((160 * 1024) / km2_lds_bytes(K, RGW));
    if (wgs * KM2_NW > 32) wgs = 32 / KM2_NW;
    if (wgs < 1) wgs = 1;
    const int wps = wgs * KM2_NW / 4;
    return wps > 4 ? 4 : wps;                                                 // (128 registers: the K = 8 centroid cells alone take 64)
}

// K clusters; RGW = 8-row groups per wave (the slab holds RGW * 64 rows); MODE 1 = update, 2 = assign, 3 = both.
//   slab      [RGW * 64][64] fp16, one 128-byte line per row; a DMA instruction brings one 8-row group (1 KiB, lane-linear).  In the update
//             modes the rows are gathered SORTED BY CLUSTER, every cluster padded to whole 8-row groups with rows that read as zeros
//             (out-of-range DMA): an 8-row group belongs to one cluster.
//   ownership wave wv owns the groups i = wv, wv + 8, .. of the slab for EVERYTHING - it brings them in, adds them into the cluster sums and
//             measures their rows - so no wave ever waits for another wave's rows: the slab needs no barrier, and the DMA of the next slice's
//             group i is issued the moment the wave has read group i for the last time (a per-wave software pipeline; the only workgroup
//             barriers are the two around the exchange of the K x 64 new centroid values).
//   update    lane = (row parity = lane / 32, column pair = lane % 32): one 4-byte LDS read per row pair; the two half-waves are the two
//             chains of the wave (SC-KM2: chain = (group index & 7, rank & 1)).  The half-waves are added (lanes ^ 32), the 8 waves' sums go
//             through LDS and 64 K threads add them in wave order, divide by W and write C' (LDS + global).
//   assign    lane = (row r8 = lane / 8 of an 8-row group, cell = lane % 8): the group's 1 KiB is read lane-linear (16 bytes per lane), the K
//             centroid cells of the lane's cell sit in 8 K registers for the whole slice; cell partial in the lane, slice tree = three DPP adds
//             across the 8 cell lanes (quad_perm 1-0-3-2, 2-3-0-1, row_half_mirror: pairs (a, a^1), (a, a^2), then the two halves - the
//             adjacent-pair tree); lane (r8, cell = k) keeps the fp64 group total of (row, k).
template <int N> __device__ __forceinline__ void km2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int K, int RGW, int MODE>
__global__ __launch_bounds__(KM2_NW * 64, km2_min_waves_per_simd(K, RGW)) void km2_pass(
        const _Float16* __restrict__ X, float* Ca, float* Cb, const KmState* __restrict__ st, const float* __restrict__ w,
        const int* __restrict__ order, const int* __restrict__ start, const float* __restrict__ W, const int* __restrict__ empty_rank,
        const int* __restrict__ reseed_idx, int n_reseed, double* __restrict__ dgpart, double* __restrict__ gpart, int T, int64_t D) {
    static_assert(SW == 64 && NCELL == 8 && K <= 8, "km2_pass is written for 64-column slices");
    constexpr bool UPD = (MODE & 1) != 0, ASG = (MODE & 2) != 0;
    constexpr int NW = KM2_NW, ROWS = RGW * 64;
    if (st->done) return;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const slab = smem;
    float* const cnew = reinterpret_cast<float*>(smem + (size_t)ROWS * KM2_ROWB);
    float* const wsum = cnew + K * SW;                                        // [NW][K][64]
    float* const wpos = wsum + NW * K * SW;                                   // [ROWS]
    const float* Cold = st->cur ? Cb : Ca;
    float* Cnew = st->cur ? Ca : Cb;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cell = lane & 7, r8 = lane >> 3;
    const int64_t g = blockIdx.x;
    const int64_t gcol = g * GW;
    const int nsl = (int)((D - gcol) / SW < NSL ? (D - gcol) / SW : NSL);     // slices of this group (the last group may be short)
    const unsigned extent = (unsigned)((size_t)T * (size_t)D * 2);
    const unsigned rowbytes = (unsigned)(D * 2);
    const int rbase = UPD ? st->reseed_pos : 0;

    // layout of the slab: cluster k's run starts at group gs[k] and has ng[k] groups (update modes); plain row order otherwise
    int gs[K], ng[K], ngtot;
    if constexpr (UPD) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) { gs[k] = acc; ng[k] = (start[k + 1] - start[k] + 7) >> 3; acc += ng[k]; }
        ngtot = acc;
    } else {
        ngtot = (T + 7) >> 3;
    }
    // this wave's groups i = wv + 8 ii: the cluster of the group (wave-uniform) and, per lane, the row at position r8 of the group
    // (-1: padding, reads zeros) - the row the lane's DMA brings and whose distances the lane owns
    int drow[RGW], kg[RGW];
#pragma unroll
    for (int ii = 0; ii < RGW; ++ii) {
        const int i = wv + ii * NW;
        int t = -1, kk = -1;
        if constexpr (UPD) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (i >= gs[k] && i < gs[k] + ng[k]) {
                    kk = k;
                    const int j = (i - gs[k]) * 8 + r8;
                    if (j < start[k + 1] - start[k]) t = order[start[k] + j];
                }
            }
            if (cell == 0) wpos[i * 8 + r8] = (w && t >= 0) ? w[t] : 1.0f;
        } else {
            if (i * 8 + r8 < T) t = i * 8 + r8;
        }
        drow[ii] = t;
        kg[ii] = kk;
    }
    double acc64[RGW];
#pragma unroll
    for (int n = 0; n < RGW; ++n) acc64[n] = 0.0;
    double dsh = 0.0;                                                         // thread (k, cell 0): shift total of cluster k over the group
    const int sk = threadIdx.x / NCELL, scell = threadIdx.x % NCELL;          // (cluster, cell) of the shift threads
    float Wmine = 0.f;                                                        // thread (k, column) of the final sum: W_k (fetched once: a load inside
    if constexpr (UPD) { if (threadIdx.x < K * SW) Wmine = W[threadIdx.x / SW]; }   //  the slice loop would wait for the DMA that is in flight around it)
    auto dma_group = [&](int ii, int64_t col0) {                              // this wave's group ii of the slice at column col0 -> LDS
        const int i = wv + ii * NW;
        const unsigned voff = drow[ii] >= 0 ? (unsigned)drow[ii] * rowbytes + (unsigned)(cell << 4) : extent;
        km2_dma16(X, extent, slab + i * 1024, voff, (unsigned)(col0 * 2));
    };
#ifdef KM2_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    if constexpr (UPD && K * SW > 0) __syncthreads();                         // wpos is read by other lanes of the same wave only, but keep the launch simple
#pragma unroll
    for (int ii = 0; ii < RGW; ++ii) dma_group(ii, gcol);                      // prologue: the first slice
#pragma unroll 1
    for (int sidx = 0; sidx < nsl; ++sidx) {
        const int64_t col0 = gcol + (int64_t)sidx * SW;
        const bool more = sidx + 1 < nsl;                                     // (uniform)
        float co[8];                                                          // the old centroid cell of the shift threads
        float cr[K][8];                                                       // assign: the K centroid cells of this lane's cell
        if (UPD && threadIdx.x < K * NCELL) sc_load8<ScF32>(Cold, (size_t)sk * (size_t)D + (size_t)col0 + (size_t)scell * 8, co);
        if constexpr (ASG && !UPD) {
#pragma unroll
            for (int k = 0; k < K; ++k) sc_load8<ScF32>(Cold, (size_t)k * (size_t)D + (size_t)col0 + (size_t)cell * 8, cr[k]);
        }
        KM2_STAMP(0);
        km2_wait_vm<0>();                                                     // this wave's groups of the slice have landed (its own DMA: no barrier)
        // (the centroid loads above are complete as well.  Say so to the compiler HERE: it counts vmcnt for its own loads only, and a wait it
        //  places after the next slice's DMA has been issued would wait for that DMA too)
        if constexpr (UPD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(co[e]));
        }
        if constexpr (ASG && !UPD) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(cr[k][e]));
        }
        KM2_STAMP(1);
        // ---- B: update ----
        if constexpr (UPD) {
            const int par = lane >> 5, cp = lane & 31;
            sc_f2 cs[K];
#pragma unroll
            for (int k = 0; k < K; ++k) cs[k] = sc_f2{0.f, 0.f};
            static_for<0, RGW>([&](auto ic) {
                constexpr int ii = decltype(ic)::value;
                const int i = wv + ii * NW;
                if (kg[ii] >= 0) {                                            // (wave-uniform)
                    const char* base = slab + i * 1024 + par * KM2_ROWB + cp * 4;
                    sc_f2 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const sc_h2 v = *reinterpret_cast<const sc_h2*>(base + q * 2 * KM2_ROWB);
                        x[q] = sc_f2{(float)v[0], (float)v[1]};
                    }
                    if (w) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { const float wt = wpos[i * 8 + 2 * q + par]; x[q] = sc_f2{wt * x[q].x, wt * x[q].y}; }   // mul, then add
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if (kg[ii] == k) {                                    // (wave-uniform: a scalar branch)
#pragma unroll
                            for (int q = 0; q < 4; ++q) cs[k] = cs[k] + x[q];
                        }
                    }
                }
            });
#pragma unroll
            for (int k = 0; k < K; ++k) {                                     // u = chain(par 0) + chain(par 1)
                cs[k].x = cs[k].x + __shfl_xor(cs[k].x, 32, 64);
                cs[k].y = cs[k].y + __shfl_xor(cs[k].y, 32, 64);
                if (par == 0) *reinterpret_cast<sc_f2*>(wsum + (wv * K + k) * SW + cp * 2) = cs[k];
            }
            if constexpr (!ASG) {                                             // update only: the slab has been read for the last time
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (more) {
#pragma unroll
                    for (int ii = 0; ii < RGW; ++ii) dma_group(ii, col0 + SW);
                }
            }
            KM2_STAMP(2);
            __syncthreads();
            KM2_STAMP(3);
            if (threadIdx.x < K * SW) {                                       // thread (k, column): the 8 waves' sums in wave order, / W
                const int k = threadIdx.x / SW, col = threadIdx.x % SW;
                const float Wk = Wmine;
                float cn;
                if (Wk > 0.f) {
                    float S = wsum[k * SW + col];
#pragma unroll
                    for (int a = 1; a < NW; ++a) S = S + wsum[(a * K + k) * SW + col];
                    cn = S / Wk;
                } else {                                                      // empty cluster: the reseed row, straight from memory
                    const int pos = rbase + empty_rank[k];
                    int r = 0;
                    if (reseed_idx && pos < n_reseed) r = reseed_idx[pos];
                    if (r < 0 || r >= T) r = 0;
                    cn = (float)X[(size_t)r * (size_t)D + (size_t)col0 + (size_t)col];
                }
                cnew[k * SW + col] = cn;
                Cnew[(size_t)k * (size_t)D + (size_t)col0 + (size_t)col] = cn;
            }
            __syncthreads();
            KM2_STAMP(4);
            // ---- C: shift partial of the slice, one thread per (cluster, cell) ----
            if (wv < (K * NCELL + 63) / 64) {
                float a0 = 0.f, a1 = 0.f;
                if (threadIdx.x < K * NCELL) {
                    const sc_f4 n0 = *reinterpret_cast<const sc_f4*>(cnew + sk * SW + scell * 8), n1 = *reinterpret_cast<const sc_f4*>(cnew + sk * SW + scell * 8 + 4);
                    const float cnv[8] = {n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3]};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float d0 = co[e] - cnv[e], d1 = co[e + 1] - cnv[e + 1];
                        a0 = __builtin_fmaf(d0, d0, a0);
                        a1 = __builtin_fmaf(d1, d1, a1);
                    }
                }
                const float sp = slice_tree_sum(a0 + a1);
                if (threadIdx.x < K * NCELL && scell == 0) dsh += (double)sp;
            }
            if constexpr (ASG) {                                              // the new centroid cells of this lane's cell
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const sc_f4 n0 = *reinterpret_cast<const sc_f4*>(cnew + k * SW + cell * 8), n1 = *reinterpret_cast<const sc_f4*>(cnew + k * SW + cell * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cr[k][e] = n0[e]; cr[k][4 + e] = n1[e]; }
                }
            }
        }
        KM2_STAMP(5);
        // ---- D: assign - lane (r8, cell) of this wave's groups; group ii of the NEXT slice is requested as soon as group ii has been read ----
        if constexpr (ASG) {
            static_for<0, RGW>([&](auto ic) {
                constexpr int ii = decltype(ic)::value;
                const int i = wv + ii * NW;
                sc_h8 v;
                const bool live = i < ngtot;                                  // (wave-uniform)
                if (live) {
                    v = *reinterpret_cast<const sc_h8*>(slab + i * 1024 + lane * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");          // the 16 bytes are in registers: their LDS lines are free
                }
                __builtin_amdgcn_sched_barrier(0);                            // (the request stays HERE: sunk below the arithmetic it would start too late)
                if (more) dma_group(ii, col0 + SW);
                __builtin_amdgcn_sched_barrier(0);
                if (live) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
                    float mine = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        sc_f2 acc = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const sc_f2 xv = {x[e], x[e + 1]}, c2 = {cr[k][e], cr[k][e + 1]};
                            const sc_f2 d = xv - c2;
                            acc = __builtin_elementwise_fma(d, d, acc);
                        }
                        float p = acc.x + acc.y;                              // cell partial
                        p = km2_dpp_add<0xB1>(p);                             // + cell ^ 1   (quad_perm [1,0,3,2])
                        p = km2_dpp_add<0x4E>(p);                             // + pair ^ 2   (quad_perm [2,3,0,1])
                        p = km2_dpp_add<0x141>(p);                            // + the other half of the 8 (row_half_mirror)
                        mine = (cell == k) ? p : mine;
                    }
                    acc64[ii] += (double)mine;
                }
            });
        }
        KM2_STAMP(6);
    }
#ifdef KM2_TRACE
    if (threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&km2_trace[i], tr[i]);
#endif
    if constexpr (ASG) {
        const size_t I = (size_t)T * K;
#pragma unroll
        for (int ii = 0; ii < RGW; ++ii)
            if (cell < K && drow[ii] >= 0) gpart[(size_t)g * I + (size_t)drow[ii] * K + cell] = acc64[ii];
    }
    if constexpr (UPD) {
        if (threadIdx.x < K * NCELL && scell == 0) dgpart[(size_t)g * K + sk] = dsh;
    }
}

// per-cluster squared shift totals ||C_k - C'_k||^2 of clusters [kb, kb + kn) from the fp64 group totals (SC-KM2: 32 contiguous group
// segments, then the segment sums); result in tot[0..kn) (shared), valid after the trailing barrier.  (`nchunks` = number of groups)
__device__ __forceinline__ void km_shift_totals(const double* __restrict__ dpart, int K, int kb, int kn, int64_t nchunks, double* segs, double* tot) {
    const int64_t seglen = (nchunks + NSEG - 1) / NSEG;
    for (int p = threadIdx.x; p < NSEG * kn; p += blockDim.x) {
        const int s = p / kn, k = kb + p % kn;
        int64_t lo = (int64_t)s * seglen, hi = lo + seglen;
        if (hi > nchunks) hi = nchunks;
        double a = 0.0;
        int64_t c = lo;
        for (; c + 8 <= hi; c += 8) {                 // 8 independent loads in flight, added in ascending group order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dpart[(size_t)(c + u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; c < hi; ++c) a += dpart[(size_t)c * K + k];
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
__global__ __launch_bounds__(256) void km_decide(const double* __restrict__ dpart, KmState* __restrict__ st, int K, int64_t nchunks,
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

// ---- column-sharded fit (sc_kmeans_fit_cols): the two halves of km_decide with the exchange of the segment sums between them ----
// single block: sseg[s][k] = sum_{g in segment s of this slab, ascending} dgpart[g][k] for the slab's segments [seg_first, seg_first + seg_count)
__global__ __launch_bounds__(256) void km_shift_segs(const double* __restrict__ dpart, const KmState* __restrict__ st, double* __restrict__ sseg, int K,
                                                     int64_t nchunks, int64_t seglen, int seg_first, int seg_count) {
    if (st->done) return;
    for (int p = threadIdx.x; p < seg_count * K; p += blockDim.x) {
        const int sl = p / K, k = p % K;
        int64_t lo = (int64_t)sl * seglen, hi = lo + seglen;
        if (lo > nchunks) lo = nchunks;
        if (hi > nchunks) hi = nchunks;
        double a = 0.0;
        int64_t c = lo;
        for (; c + 8 <= hi; c += 8) {                 // (the order of km_shift_totals)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dpart[(size_t)(c + u) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; c < hi; ++c) a += dpart[(size_t)c * K + k];
        sseg[(size_t)(seg_first + sl) * K + k] = a;
    }
}
// single block: km_decide from all 32 segment rows (identical on every rank after the exchange)
__global__ __launch_bounds__(64) void km_decide_segs(const double* __restrict__ sseg, KmState* __restrict__ st, int K, int iter, int max_iter, float tol,
                                                     int n_reseed) {
    if (st->done) return;
    if (threadIdx.x == 0) {
        double diff = 0.0;
        for (int k = 0; k < K; ++k) {
            double t = 0.0;
            for (int s = 0; s < NSEG; ++s) t += sseg[(size_t)s * K + k];
            diff += sqrt(t);
        }
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
__global__ __launch_bounds__(256) void km_shift_out(const double* __restrict__ dpart, const float* __restrict__ W, double* __restrict__ shift2,
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
    KmState* st; float* Ca; float* Cb; double* gpart; double* seg; double* dgpart; int* labels32; int* order; int* start;
    float* W; int* empty_rank; size_t bytes;
};
KmWs carve(void* base, int T, int64_t D, int K) {
    const int64_t ng = (D + GW - 1) / GW;
    const size_t I = (size_t)T * K;
    char* p = reinterpret_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = sc_align_up(off + bytes, 256); return p ? (void*)(p + o) : (void*)nullptr; };
    KmWs w;
    w.st = (KmState*)take(sizeof(KmState));
    w.Ca = (float*)take(sizeof(float) * (size_t)K * D);
    w.Cb = (float*)take(sizeof(float) * (size_t)K * D);
    w.gpart = (double*)take(sizeof(double) * (size_t)ng * I);
    w.seg = (double*)take(sizeof(double) * NSEG * I);
    w.dgpart = (double*)take(sizeof(double) * (size_t)ng * K);
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
    if (vec) hipLaunchKernelGGL((km_assign<Tag, KB, true>), grid, block, 0, s, X, w.Ca, w.Cb, w.st, w.gpart, T, D, K, k0, nch);
    else hipLaunchKernelGGL((km_assign<Tag, KB, false>), grid, block, 0, s, X, w.Ca, w.Cb, w.st, w.gpart, T, D, K, k0, nch);
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
void launch_update(bool vec, const void* X, float* Ca, float* Cb, const KmWs& w, const float* wts, const int32_t* fill_idx, int n_fill,
                   int T, int64_t D, int K, int64_t nch, int empty_mode, hipStream_t s) {
    const dim3 sgrid((unsigned)((nch + WPB - 1) / WPB)), sblock(WPB * 64);
    const size_t sh = sizeof(float) * (size_t)K * NSL;
    if (vec) hipLaunchKernelGGL((km_update<Tag, true>), sgrid, sblock, sh, s, X, Ca, Cb, w.st, wts, w.order, w.start, w.W, w.empty_rank,
                                fill_idx, n_fill, w.dgpart, T, D, K, nch, empty_mode);
    else hipLaunchKernelGGL((km_update<Tag, false>), sgrid, sblock, sh, s, X, Ca, Cb, w.st, wts, w.order, w.start, w.W, w.empty_rank,
                            fill_idx, n_fill, w.dgpart, T, D, K, nch, empty_mode);
}

// ---- the one-read pass: which shapes take it, and its launch ----------------------------------
// SC_KM_FUSED=0 forces the two-pass kernels everywhere (same bits: tests/test_gpu_kmeans_fused.py compares the two paths).
bool km2_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SC_KM_FUSED"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}
// rows-per-wave bucket: the slab holds RGW * 64 rows and must take the T rows sorted by cluster with every cluster padded to whole 8-row groups.
// With few rows a slice is too little work per workgroup step (two barriers, one DMA round trip), and the K = 8 instantiation carries 64
// centroid registers per lane: measured on one box (profiles/r06_run_i_kmeans_T_sweep.jsonl), one-read pass against two-pass kernels, ms per
// iteration: K = 5: T = 100 0.245 / 0.232, 130 0.269 / 0.288, 200 0.336 / 0.402, 400 0.519 / 0.734; K = 8: T = 64 0.26 / 0.21, 80 0.298 / 0.242,
// 150 0.360 / 0.387, 392 0.774 / 0.851 - the pass takes over where it wins.
int km2_rgw(int T, int K) {
    if (T < (K >= 6 ? 144 : 120)) return 0;                                    // (measured at K = 5 and 8; the K between take the nearer one's threshold)
    const int need = T + 7 * K;
    return need <= 256 ? 4 : need <= 448 ? 7 : 0;
}
template <typename Tag>
bool km2_eligible(const void* X, int T, int64_t D, int K) {
    if (!std::is_same<Tag, ScF16>::value || !km2_enabled()) return false;
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0 || D % SW != 0 || K < 2 || K > 8) return false;     // (the entry point's --num_clusters: 5 as shipped, any K <= 8 takes the pass)
    const int rgw = km2_rgw(T, K);
    if (rgw == 0) return false;
    return (uint64_t)rgw * 64ull * (uint64_t)D * 2ull < (1ull << 32);           // the DMA's 32-bit row offsets (padded rows included)
}
template <int K, int RGW, int MODE>
void km2_launch_inst(const void* X, float* Ca, float* Cb, const KmWs& w, const float* wts, const int32_t* reseed_idx, int n_reseed, int T, int64_t D,
                     hipStream_t s) {
    constexpr size_t lds = km2_lds_bytes(K, RGW);
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)km2_pass<K, RGW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    const int64_t ng = (D + GW - 1) / GW;
    hipLaunchKernelGGL((km2_pass<K, RGW, MODE>), dim3((unsigned)ng), dim3(KM2_NW * 64), lds, s, (const _Float16*)X, Ca, Cb, w.st, wts,
                       w.order, w.start, w.W, w.empty_rank, reseed_idx, n_reseed, w.dgpart, w.gpart, T, D);
}
template <int K, int MODE>
void km2_launch_k(const void* X, float* Ca, float* Cb, const KmWs& w, const float* wts, const int32_t* reseed_idx, int n_reseed, int T, int64_t D, hipStream_t s) {
    switch (km2_rgw(T, K)) {
        case 4: km2_launch_inst<K, 4, MODE>(X, Ca, Cb, w, wts, reseed_idx, n_reseed, T, D, s); break;
        default: km2_launch_inst<K, 7, MODE>(X, Ca, Cb, w, wts, reseed_idx, n_reseed, T, D, s); break;
    }
}
template <int MODE>
void km2_launch(const void* X, float* Ca, float* Cb, const KmWs& w, const float* wts, const int32_t* reseed_idx, int n_reseed, int T, int64_t D, int K,
                hipStream_t s) {
    switch (K) {
#define SC_CASE(n) case n: km2_launch_k<n, MODE>(X, Ca, Cb, w, wts, reseed_idx, n_reseed, T, D, s); break;
        SC_CASE(2) SC_CASE(3) SC_CASE(4) SC_CASE(5) SC_CASE(6) SC_CASE(7) SC_CASE(8)
#undef SC_CASE
    }
}

template <typename Tag>
int fit_impl(const void* X, int T, int64_t D, int K, const float* wts, const int32_t* init_idx, const int32_t* reseed_idx,
             int n_reseed, int max_iter, float tol, float* C, int64_t* labels, float* wsum, int32_t* info, void* ws,
             hipStream_t s) {
    const KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH, ng = (D + GW - 1) / GW;
    const size_t I = (size_t)T * K;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    {
        const dim3 igrid((unsigned)((D + 2047) / 2048), (unsigned)K);
        if (vec) hipLaunchKernelGGL((km_init<Tag, true>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
        else hipLaunchKernelGGL((km_init<Tag, false>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
    }
    const dim3 rgrid((unsigned)((I + 255) / 256), NSEG);
    // one-read iterations (km2_pass): the pass of iteration i computes C'(i) AND the distances of iteration i + 1 against it from one
    // LDS-resident slab per slice: n + 1 reads of X for n Lloyd iterations instead of 2 n
    const bool fused = km2_eligible<Tag>(X, T, D, K);
    for (int it = 0; it < max_iter; ++it) {
        if (!fused) launch_assign<Tag>(vec, X, w, T, D, K, nch, s);
        else if (it == 0) km2_launch<2>(X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, s);
        hipLaunchKernelGGL(km_reduce, rgrid, dim3(256), 0, s, w.gpart, w.st, w.seg, I, ng, 1, (ng + NSEG - 1) / NSEG, 0, NSEG);
        hipLaunchKernelGGL(km_argmin, dim3((T + 63) / 64), dim3(64), 0, s, w.seg, w.st, w.labels32, (double*)nullptr, T, K, 1);
        hipLaunchKernelGGL(km_order, dim3(1), dim3(1024), km_order_lds(T, K), s, w.st, wts, w.labels32, w.order, w.start, w.W, w.empty_rank, T, K, 1);
        // (the last iteration has no next assign: its update alone is a load-latency-bound pass for km2_pass - 0.49 ms against 0.42 for km_update)
        if (!fused || it + 1 == max_iter) launch_update<Tag>(vec, X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, nch, 0, s);
        else km2_launch<3>(X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, s);
        hipLaunchKernelGGL(km_decide, dim3(1), dim3(256), 0, s, w.dgpart, w.st, K, ng, it, max_iter, tol, n_reseed);
    }
    hipLaunchKernelGGL(km_finalize, dim3(1024), dim3(256), 0, s, w.Ca, w.Cb, w.st, w.labels32, w.W, C, labels, wsum, info, T, D, K);
    SC_CHECK_LAUNCH("sc_kmeans_fit");
    return SC_OK;
}

// The Lloyd loop of fit_impl on a COLUMN SLAB of X that holds whole segments of the SC-KM2 distance structure: what leaves the slab per
// iteration are its rows of the two segment tables (distances [32][T K], shifts [32][K], fp64); `exchange` completes both tables on every
// rank, after which arg-min, ordering and the convergence decision are the replicated kernels of the 1-GPU fit on identical inputs.
template <typename Tag>
int fit_cols_impl(const void* X, int T, int64_t D, int K, const float* wts, const int32_t* init_idx, const int32_t* reseed_idx,
                  int n_reseed, int max_iter, float tol, float* C, int64_t* labels, float* wsum, int32_t* info, int64_t seglen, int seg_first,
                  int seg_count, double* seg_dist, double* seg_shift, sc_kmeans_exchange_fn exchange, void* xctx, void* ws, hipStream_t s) {
    const KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH, ng = (D + GW - 1) / GW;
    const size_t I = (size_t)T * K;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    {
        const dim3 igrid((unsigned)((D + 2047) / 2048), (unsigned)K);
        if (vec) hipLaunchKernelGGL((km_init<Tag, true>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
        else hipLaunchKernelGGL((km_init<Tag, false>), igrid, dim3(256), 0, s, X, init_idx, w.Ca, w.st, T, D, K);
    }
    const dim3 rgrid((unsigned)((I + 255) / 256), NSEG);
    const bool fused = km2_eligible<Tag>(X, T, D, K);
    for (int it = 0; it < max_iter; ++it) {
        if (!fused) launch_assign<Tag>(vec, X, w, T, D, K, nch, s);
        else if (it == 0) km2_launch<2>(X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, s);
        hipLaunchKernelGGL(km_reduce, rgrid, dim3(256), 0, s, w.gpart, w.st, seg_dist, I, ng, 1, seglen, seg_first, seg_count);
        SC_CHECK_LAUNCH("sc_kmeans_fit_cols");
        if (exchange(xctx, 0, (sc_stream_t)s) != 0) return sc_fail(SC_ERR_LAUNCH, "sc_kmeans_fit_cols: the exchange of the distance segments failed (iteration %d)", it);
        hipLaunchKernelGGL(km_argmin, dim3((T + 63) / 64), dim3(64), 0, s, seg_dist, w.st, w.labels32, (double*)nullptr, T, K, 1);
        hipLaunchKernelGGL(km_order, dim3(1), dim3(1024), km_order_lds(T, K), s, w.st, wts, w.labels32, w.order, w.start, w.W, w.empty_rank, T, K, 1);
        if (!fused || it + 1 == max_iter) launch_update<Tag>(vec, X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, nch, 0, s);
        else km2_launch<3>(X, w.Ca, w.Cb, w, wts, reseed_idx, n_reseed, T, D, K, s);
        hipLaunchKernelGGL(km_shift_segs, dim3(1), dim3(256), 0, s, w.dgpart, w.st, seg_shift, K, ng, seglen, seg_first, seg_count);
        SC_CHECK_LAUNCH("sc_kmeans_fit_cols");
        if (exchange(xctx, 1, (sc_stream_t)s) != 0) return sc_fail(SC_ERR_LAUNCH, "sc_kmeans_fit_cols: the exchange of the shift segments failed (iteration %d)", it);
        hipLaunchKernelGGL(km_decide_segs, dim3(1), dim3(64), 0, s, seg_shift, w.st, K, it, max_iter, tol, n_reseed);
    }
    hipLaunchKernelGGL(km_finalize, dim3(1024), dim3(256), 0, s, w.Ca, w.Cb, w.st, w.labels32, w.W, C, labels, wsum, info, T, D, K);
    SC_CHECK_LAUNCH("sc_kmeans_fit_cols");
    return SC_OK;
}

template <typename Tag>
int assign_impl(const void* X, int T, int64_t D, int K, const float* C, int64_t* labels, double* dist2, void* ws, hipStream_t s) {
    KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH, ng = (D + GW - 1) / GW;
    const size_t I = (size_t)T * K;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    hipLaunchKernelGGL(km_set_state, dim3(1), dim3(1), 0, s, w.st, 0);
    w.Ca = const_cast<float*>(C);   // read-only use: state.cur == 0 selects Ca
    if (km2_eligible<Tag>(X, T, D, K) && (reinterpret_cast<uintptr_t>(C) & 31) == 0) km2_launch<2>(X, w.Ca, w.Cb, w, nullptr, nullptr, 0, T, D, K, s);
    else launch_assign<Tag>(vec, X, w, T, D, K, nch, s);
    hipLaunchKernelGGL(km_reduce, dim3((unsigned)((I + 255) / 256), NSEG), dim3(256), 0, s, w.gpart, w.st, w.seg, I, ng, 0, (ng + NSEG - 1) / NSEG, 0, NSEG);
    hipLaunchKernelGGL(km_argmin, dim3((T + 63) / 64), dim3(64), 0, s, w.seg, w.st, w.labels32, dist2, T, K, 0);
    hipLaunchKernelGGL(km_labels_out, dim3((T + 255) / 256), dim3(256), 0, s, w.labels32, labels, T);
    SC_CHECK_LAUNCH("sc_kmeans_assign");
    return SC_OK;
}

// one centroid update from given labels: C_new[k] = sum_{t: label t = k} w_t x_t / W_k (rows in ascending order), an empty
// cluster takes row fill_idx[its rank among the empty clusters] (empty_mode 0) or the zero vector (empty_mode 1)
template <typename Tag>
int update_impl(const void* X, int T, int64_t D, int K, const float* wts, const int64_t* labels, const float* C_old, int empty_mode,
                const int32_t* fill_idx, int n_fill, float* C_new, float* wsum, double* shift2, void* ws, hipStream_t s) {
    KmWs w = carve(ws, T, D, K);
    const int64_t nch = (D + CH - 1) / CH, ng = (D + GW - 1) / GW;
    const bool vec = (D % 8 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C_old) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(C_new) & 15) == 0);
    hipLaunchKernelGGL(km_set_state, dim3(1), dim3(1), 0, s, w.st, 0);
    hipLaunchKernelGGL(km_labels_in, dim3((T + 255) / 256), dim3(256), 0, s, labels, w.labels32, T, K);
    hipLaunchKernelGGL(km_order, dim3(1), dim3(1024), km_order_lds(T, K), s, w.st, wts, w.labels32, w.order, w.start, w.W, w.empty_rank, T, K, 0);
    float* Ca = const_cast<float*>(C_old);        // state.cur == 0: Ca is read (old centroids), Cb written
    launch_update<Tag>(vec, X, Ca, C_new, w, wts, fill_idx, n_fill, T, D, K, nch, empty_mode, s);
    hipLaunchKernelGGL(km_shift_out, dim3(1), dim3(256), 0, s, w.dgpart, w.W, shift2, wsum, K, ng);
    SC_CHECK_LAUNCH("sc_kmeans_update");
    return SC_OK;
}

}  // namespace

#ifdef KM2_TRACE
extern "C" int sc_km2_trace_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(km2_trace), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(km2_trace), z, sizeof(z)); }
    return 0;
}
#endif

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

extern "C" int sc_kmeans_fit_cols(const void* X, int dtype, int T, int64_t D_local, int K, const float* w, const int32_t* init_idx,
                                  const int32_t* reseed_idx, int n_reseed, int max_iter, float tol, float* C_local, int64_t* labels,
                                  float* wsum, int32_t* info, int64_t seg_groups, int seg_first, int seg_count, double* seg_dist,
                                  double* seg_shift, sc_kmeans_exchange_fn exchange, void* exchange_ctx, void* ws, size_t ws_bytes,
                                  sc_stream_t stream) {
    SC_REQUIRE(X && init_idx && C_local && labels && wsum && info && ws && seg_dist && seg_shift && exchange, "sc_kmeans_fit_cols: null pointer argument");
    SC_REQUIRE(T > 0 && D_local > 0 && K > 0 && max_iter > 0, "sc_kmeans_fit_cols: T, D_local, K, max_iter must be positive");
    SC_REQUIRE(n_reseed >= 0 && (n_reseed == 0 || reseed_idx), "sc_kmeans_fit_cols: n_reseed > 0 needs reseed_idx");
    SC_REQUIRE(seg_groups > 0 && seg_first >= 0 && seg_count > 0 && seg_first + seg_count <= NSEG, "sc_kmeans_fit_cols: bad segment window");
    {
        // the slab holds whole segments: all of its groups but the last are whole (only the matrix's last group may be short), and they
        // fill segments seg_first .. seg_first + seg_count - 1 in order (the last one possibly short or empty: the matrix's tail)
        const int64_t ngl = (D_local + GW - 1) / GW;
        SC_REQUIRE(ngl <= (int64_t)seg_count * seg_groups, "sc_kmeans_fit_cols: %lld groups do not fit %d segments of %lld", (long long)ngl, seg_count, (long long)seg_groups);
        SC_REQUIRE(seg_first + seg_count == NSEG || ngl == (int64_t)seg_count * seg_groups,
                   "sc_kmeans_fit_cols: a slab that is not the matrix's last one holds seg_count * seg_groups whole groups");
        SC_REQUIRE(seg_first + seg_count == NSEG || D_local % GW == 0, "sc_kmeans_fit_cols: only the matrix's last slab may end in a partial group");
    }
    if (ws_bytes < sc_kmeans_workspace_bytes(T, D_local, K))
        return sc_fail(SC_ERR_WORKSPACE, "sc_kmeans_fit_cols: workspace %zu < required %zu", ws_bytes, sc_kmeans_workspace_bytes(T, D_local, K));
    SC_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0 && (reinterpret_cast<uintptr_t>(seg_dist) & 7) == 0 && (reinterpret_cast<uintptr_t>(seg_shift) & 7) == 0,
               "sc_kmeans_fit_cols: workspace must be 256-byte aligned, the segment tables 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
#define SC_FC(Tag) return fit_cols_impl<Tag>(X, T, D_local, K, w, init_idx, reseed_idx, n_reseed, max_iter, tol, C_local, labels, wsum, info, seg_groups, seg_first, \
                                            seg_count, seg_dist, seg_shift, exchange, exchange_ctx, ws, s)
        case SC_F16: SC_FC(ScF16);
        case SC_BF16: SC_FC(ScBF16);
        case SC_F32: SC_FC(ScF32);
#undef SC_FC
    }
    return sc_fail(SC_ERR_ARG, "sc_kmeans_fit_cols: unknown dtype %d", dtype);
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
