// Next-token selection over the fp32 logits of the LM head (V = 152 064 for Qwen2-7B) — replaces the torch.argmax /
// torch.softmax + torch.multinomial the decode loops used (HF `generate` behind reference llava_qwen.py:155: greedy when
// do_sample=False, else temperature softmax sampling; inference_streaming_longva_v2.py:252-253, utiles.py:551-552).
//
// Two launches, no host round trip, hipGraph-capturable:
//   k_pick_partial  grid (NBLK, B): every block scans a contiguous slice of a row with 16-byte loads and reduces it with wavefront
//                   shuffles to (max, first arg-max, sum of exp((x - max) / T)).
//   k_pick_final    one block per row: merges the NBLK partials (arg-max: lowest index among equal maxima = torch.argmax /
//                   numpy semantics), and for sampling inverts the CDF at u * total: the slice that contains the target is
//                   found by a scan over the block partials, then ONE block-wide prefix scan over that slice (<= 4 KiB elements)
//                   locates the element.  Deterministic for a given u; u comes from the caller's (torch) generator.
// HBM/L2 traffic = one read of the logits (608 KB per row) + a re-read of one slice: latency bound (~10 us), 0.3 % of a decode step.
#include "sc_common.h"

namespace {

constexpr int NBLK = 64;      // slices per row
constexpr int THREADS = 256;

struct Part { float mx; int arg; float sum; int pad; };

__device__ __forceinline__ void better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

__global__ __launch_bounds__(THREADS) void k_pick_partial(const float* __restrict__ logits, int V, int64_t ld, float inv_t, Part* __restrict__ part) {
    const int row = blockIdx.y, blk = blockIdx.x;
    const float* x = logits + (size_t)row * (size_t)ld;
    const int per = ((V + NBLK - 1) / NBLK + 3) & ~3;                     // slice length, multiple of 4 (16-byte loads)
    const int lo = blk * per, hi = min(V, lo + per);
    float bv = -INFINITY; int bi = 0x7fffffff;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += THREADS * 4) {
            if (i + 4 <= hi) {
                const sc_f4 v = *reinterpret_cast<const sc_f4*>(x + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) better(bv, bi, v[e], i + e);
            } else {
                for (int e = 0; i + e < hi; ++e) better(bv, bi, x[i + e], i + e);
            }
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += THREADS) better(bv, bi, x[i], i);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) better(bv, bi, __shfl_xor(bv, s, 64), __shfl_xor(bi, s, 64));
    __shared__ float sv[THREADS / 64];
    __shared__ int si[THREADS / 64];
    __shared__ float ssum[THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) better(bv, bi, sv[w], si[w]);
    // sum of exp((x - slice max) / T): the slice (<= 2.4k elements) is L1/L2-resident from the first pass
    float s = 0.f;
    if (inv_t > 0.f && bv > -INFINITY)
        for (int i = lo + threadIdx.x; i < hi; i += THREADS) s += __expf((x[i] - bv) * inv_t);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) ssum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < THREADS / 64; ++w) tot += ssum[w];
        part[(size_t)row * NBLK + blk] = Part{bv, bi, tot, 0};
    }
}

__global__ __launch_bounds__(THREADS) void k_pick_final(const float* __restrict__ logits, int V, int64_t ld, float inv_t, const float* __restrict__ u,
                                                        const Part* __restrict__ part, int64_t* __restrict__ out) {
    const int row = blockIdx.x;
    const Part* p = part + (size_t)row * NBLK;
    __shared__ float wgt[NBLK];
    __shared__ float scan[THREADS];
    __shared__ int sel_blk;
    __shared__ float sel_target;
    // every thread computes the global max / arg-max from the 64 partials (cheap, avoids a broadcast)
    float M = -INFINITY; int A = 0x7fffffff;
    for (int b = 0; b < NBLK; ++b) better(M, A, p[b].mx, p[b].arg);
    if (inv_t <= 0.f || u == nullptr) {                                   // greedy
        if (threadIdx.x == 0) out[row] = A == 0x7fffffff ? 0 : A;
        return;
    }
    if (threadIdx.x < NBLK) wgt[threadIdx.x] = p[threadIdx.x].mx > -INFINITY ? p[threadIdx.x].sum * __expf((p[threadIdx.x].mx - M) * inv_t) : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int b = 0; b < NBLK; ++b) tot += wgt[b];
        float target = fminf(fmaxf(u[row], 0.f), 0.99999994f) * tot;      // u in [0, 1)
        int b = 0;
        for (; b < NBLK - 1; ++b) {
            if (target < wgt[b]) break;
            target -= wgt[b];
        }
        while (b > 0 && wgt[b] <= 0.f) --b;                               // rounding pushed the target past the last non-empty slice
        sel_blk = b;
        sel_target = fminf(target, wgt[b]);
    }
    __syncthreads();
    const float* x = logits + (size_t)row * (size_t)ld;
    const int per = ((V + NBLK - 1) / NBLK + 3) & ~3;
    const int lo = sel_blk * per, hi = min(V, lo + per);
    const int n = hi - lo, chunk = (n + THREADS - 1) / THREADS;
    const int c0 = lo + threadIdx.x * chunk, c1 = min(hi, c0 + chunk);
    float loc = 0.f;
    for (int i = c0; i < c1; ++i) loc += __expf((x[i] - M) * inv_t);
    scan[threadIdx.x] = loc;
    __syncthreads();
    if (threadIdx.x == 0) {                                               // 256-entry serial scan: ~1 us, keeps the inversion exact and simple
        float acc = 0.f;
        int t = 0;
        for (; t < THREADS - 1; ++t) {
            if (sel_target < acc + scan[t]) break;
            acc += scan[t];
        }
        while (t > 0 && (lo + t * chunk >= hi || scan[t] <= 0.f)) --t;
        const int a0 = lo + t * chunk, a1 = min(hi, a0 + chunk);
        int pick = a0;
        float run = acc;
        for (int i = a0; i < a1; ++i) {
            const float e = __expf((x[i] - M) * inv_t);
            pick = i;
            if (sel_target < run + e) break;
            run += e;
        }
        out[row] = pick;
    }
}

}  // namespace

extern "C" size_t sc_pick_token_workspace_bytes(int B) { return B > 0 ? (size_t)B * NBLK * sizeof(Part) : 0; }

extern "C" int sc_pick_token_f32(const float* logits, int B, int V, int64_t ld, float temperature, const float* u, int64_t* out, void* ws,
                                 size_t ws_bytes, sc_stream_t stream) {
    SC_REQUIRE(logits && out && ws, "sc_pick_token_f32: null pointer argument");
    SC_REQUIRE(B > 0 && V > 0 && ld >= V, "sc_pick_token_f32: bad sizes");
    SC_REQUIRE(!(temperature > 0.f) || u, "sc_pick_token_f32: sampling (temperature > 0) needs the uniform draws u[B]");
    if (ws_bytes < sc_pick_token_workspace_bytes(B))
        return sc_fail(SC_ERR_WORKSPACE, "sc_pick_token_f32: workspace %zu < required %zu", ws_bytes, sc_pick_token_workspace_bytes(B));
    hipStream_t s = (hipStream_t)stream;
    const float inv_t = temperature > 0.f ? 1.0f / temperature : 0.f;
    hipLaunchKernelGGL(k_pick_partial, dim3(NBLK, B), dim3(THREADS), 0, s, logits, V, ld, inv_t, (Part*)ws);
    hipLaunchKernelGGL(k_pick_final, dim3(B), dim3(THREADS), 0, s, logits, V, ld, inv_t, temperature > 0.f ? u : nullptr, (const Part*)ws, out);
    SC_CHECK_LAUNCH("sc_pick_token_f32");
    return SC_OK;
}
