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
//
// sc_sample_token_f32 adds what HF `generate` applies when the checkpoint's generation_config.json asks for it (Qwen2-7B-Instruct
// ships repetition_penalty 1.05, top_k 20, top_p 0.8; the reference passes temperature and top_p explicitly, inference_streaming_
// longva_v2.py:252-256, and inherits the rest): RepetitionPenaltyLogitsProcessor -> TemperatureLogitsWarper -> TopKLogitsWarper ->
// TopPLogitsWarper, then one draw.  Three more launches, all on the device:
//   k_rep_penalty   one block per row: every DISTINCT previously generated id is penalised once (x < 0 ? x * r : x / r), as HF's
//                   gather / scatter does; an LDS bitmap over the vocabulary deduplicates.
//   k_topk_partial  grid (NBLK, B): the slice goes to LDS, top_k rounds of block arg-max (lowest index on ties) -> k candidates.
//   k_topk_final    one block per row: the NBLK * k candidates -> the k largest in descending order (+ further candidates equal to
//                   the k-th value: HF keeps ties), softmax at temperature T, nucleus cut (remove the ascending-cumulative tail
//                   <= 1 - top_p, keep >= 1), then the inverse CDF at u over the kept tokens IN INDEX ORDER (the convention of
//                   sc_pick_token_f32, so that top_k >= V / top_p = 1 reproduce it).
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


// ---- HF logits processors / warpers ----
constexpr int TK_MAX = 64;        // top_k limit
constexpr int TK_KEEP = 128;      // kept tokens incl. ties at the k-th value
struct Cand { float v; int i; };

__global__ __launch_bounds__(THREADS) void k_rep_penalty(float* __restrict__ logits, int V, int64_t ld, const int64_t* __restrict__ prev, int64_t prev_ld,
                                                         const int32_t* __restrict__ n_prev_dev, int n_prev_host, float pen) {
    extern __shared__ unsigned bitmap[];
    const int row = blockIdx.x;
    const int n = n_prev_dev ? n_prev_dev[row] : n_prev_host;
    const int words = (V + 31) >> 5;
    for (int i = threadIdx.x; i < words; i += THREADS) bitmap[i] = 0u;
    __syncthreads();
    float* x = logits + (size_t)row * (size_t)ld;
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const int64_t id = prev[(size_t)row * (size_t)prev_ld + i];
        if (id < 0 || id >= V) continue;
        const unsigned bit = 1u << (id & 31);
        if (atomicOr(&bitmap[id >> 5], bit) & bit) continue;             // a repeated id is penalised once (HF: gather, then scatter)
        const float v = x[id];
        x[id] = v < 0.f ? v * pen : v / pen;
    }
}

// block-wide arg-best of (v, i) over all threads; every thread returns the winner
__device__ __forceinline__ void block_best(float& bv, int& bi, float* sv, int* si) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) better(bv, bi, __shfl_xor(bv, s, 64), __shfl_xor(bi, s, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                                      // the previous round's readers are done with sv / si
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) better(bv, bi, sv[w], si[w]);
}

__global__ __launch_bounds__(THREADS) void k_topk_partial(const float* __restrict__ logits, int V, int64_t ld, int k, Cand* __restrict__ cand) {
    extern __shared__ float sl[];
    __shared__ float sv[THREADS / 64];
    __shared__ int si[THREADS / 64];
    const int row = blockIdx.y, blk = blockIdx.x;
    const float* x = logits + (size_t)row * (size_t)ld;
    const int per = ((V + NBLK - 1) / NBLK + 3) & ~3;
    const int lo = blk * per, hi = min(V, lo + per), n = max(hi - lo, 0);
    for (int i = threadIdx.x; i < n; i += THREADS) sl[i] = x[lo + i];
    __syncthreads();
    Cand* out = cand + ((size_t)row * NBLK + blk) * (size_t)k;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < n; i += THREADS) better(bv, bi, sl[i], i);
        block_best(bv, bi, sv, si);
        if (threadIdx.x == 0) {
            out[r] = Cand{bv, bi == 0x7fffffff ? 0x7fffffff : lo + bi};
            if (bi != 0x7fffffff) sl[bi] = -INFINITY;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(THREADS) void k_topk_final(const Cand* __restrict__ cand, int k, float inv_t, float top_p, const float* __restrict__ u,
                                                        int64_t* __restrict__ out) {
    extern __shared__ Cand pool[];                                        // NBLK * k candidates
    __shared__ float sv[THREADS / 64];
    __shared__ int si[THREADS / 64];
    __shared__ Cand top[TK_KEEP];
    __shared__ float prob[TK_KEEP];
    __shared__ int n_top;
    const int row = blockIdx.x, total = NBLK * k;
    const Cand* c = cand + (size_t)row * (size_t)total;
    for (int i = threadIdx.x; i < total; i += THREADS) pool[i] = c[i];
    __syncthreads();
    // the k largest in descending order (equal values: lowest vocabulary index first), then everything equal to the k-th value
    for (int r = 0; r < TK_KEEP; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff, bp = 0;
        for (int i = threadIdx.x; i < total; i += THREADS) {
            const float v = pool[i].v; const int idx = pool[i].i;
            if (v > bv || (v == bv && idx < bi)) { bv = v; bi = idx; bp = i; }
        }
        // reduce on (v, index); the pool slot of the winner is found again by its (unique) vocabulary index
        float wv = bv; int wi = bi;
        block_best(wv, wi, sv, si);
        const bool stop = (wi == 0x7fffffff) || (r >= k && wv < top[k - 1].v);
        if (stop) { if (threadIdx.x == 0) n_top = r; break; }
        if (bi == wi && bv == wv) pool[bp].v = -INFINITY, pool[bp].i = 0x7fffffff;     // exactly one thread owns the winner
        if (threadIdx.x == 0) { top[r] = Cand{wv, wi}; n_top = r + 1; }
        __syncthreads();
    }
    __syncthreads();
    const int n = n_top;
    if (n == 0) { if (threadIdx.x == 0) out[row] = 0; return; }
    if (inv_t <= 0.f || u == nullptr) { if (threadIdx.x == 0) out[row] = top[0].i; return; }          // arg-max of the processed logits
    if (threadIdx.x < n) prob[threadIdx.x] = __expf((top[threadIdx.x].v - top[0].v) * inv_t);
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int j = 0; j < n; ++j) tot += prob[j];
        int keep = n;
        if (top_p < 1.f) {                                                // HF TopPLogitsWarper: drop the ascending tail whose cumulative probability <= 1 - top_p
            float tail = 0.f;
            const float cut = (1.f - top_p) * tot;
            for (int j = n - 1; j >= 1; --j) {
                tail += prob[j];
                if (tail <= cut) keep = j; else break;
            }
        }
        // insertion sort of the kept tokens by vocabulary index (keep <= 128, typically <= 20), then the CDF walk
        for (int a = 1; a < keep; ++a) {
            const Cand ca = top[a]; const float pa = prob[a];
            int b = a - 1;
            while (b >= 0 && top[b].i > ca.i) { top[b + 1] = top[b]; prob[b + 1] = prob[b]; --b; }
            top[b + 1] = ca; prob[b + 1] = pa;
        }
        float kept = 0.f;
        for (int j = 0; j < keep; ++j) kept += prob[j];
        const float target = fminf(fmaxf(u[row], 0.f), 0.99999994f) * kept;
        float run = 0.f;
        int pick = top[0].i;
        for (int j = 0; j < keep; ++j) {
            pick = top[j].i;
            if (target < run + prob[j]) break;
            run += prob[j];
        }
        out[row] = pick;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Full-vocabulary top-k / top-p (round 3): the corners HF `generate` accepts that the candidate path above does not serve - a nucleus
// without top-k (the reference forwards --top_p, inference_streaming_longva_v2.py:245-256; HF's default top_k = 50 only applies when the
// checkpoint or the caller leaves it) and top_k > 64.  One block of 1024 threads per row, ~10 coalesced scans of the row (L2-resident,
// 608 KB), everything in integers so that the result does not depend on the order of atomics:
//   * keys: the logits as order-preserving uint32; thresholds by 4-pass radix select over 8 bits with LDS histograms -
//     COUNT histograms for the k-th largest key (TopKLogitsWarper keeps everything >= the k-th value), then MASS histograms of
//     w_i = exp((x_i - max) / T) in 2^40 fixed point (uint64 LDS atomics: exact sums) for the nucleus: the largest key t with
//     mass(key >= t) >= top_p * mass(kept by top-k) - i.e. a token stays iff the mass of strictly larger tokens is < top_p, which is
//     TopPLogitsWarper's "remove the ascending tail whose cumulative probability <= 1 - top_p" (a group of EQUAL logits is kept or
//     dropped as a whole);
//   * the draw: inverse CDF at u over the kept tokens in INDEX order (the convention of sc_pick_token_f32), per-thread contiguous
//     chunks, fp64 prefix over the 1024 chunk sums.
// diag (optional, device): per row {threshold logit as float bits, kept count} - what the tests compare with HF's processors.
// ------------------------------------------------------------------------------------------------------------------
constexpr int FT = 1024;

__device__ __forceinline__ unsigned f2key(float x) { const unsigned u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(FT) void k_full_filter(const float* __restrict__ logits, int V, int64_t ld, float inv_t, int top_k, float top_p,
                                                    const float* __restrict__ u, int64_t* __restrict__ out, unsigned* __restrict__ diag) {
    __shared__ unsigned long long hist[256];
    __shared__ float red[FT / 64];
    __shared__ unsigned sh_sel;
    __shared__ unsigned long long sh_left, sh_total;
    __shared__ double chunk[FT];
    __shared__ int sh_owner;
    __shared__ double sh_before;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)row * (size_t)ld;
    // ---- max ----
    float mx = -INFINITY;
    for (int i = tid; i < V; i += FT) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int s2 = 32; s2 >= 1; s2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s2, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < FT / 64; ++w) mx = fmaxf(mx, red[w]);
    auto wfix = [&](float v) -> unsigned long long { return (unsigned long long)(__expf((v - mx) * inv_t) * 1099511627776.0f); };   // 2^40
    // 4-pass radix select, descending: the largest key t such that measure(key >= t) >= need.  MASS: measure = fixed-point weight, else count.
    // Candidates of a pass are the keys that match the prefix chosen so far AND are >= floor_key (the top-k threshold during the mass search).
    auto radix_select = [&](unsigned long long need, bool mass, unsigned floor_key) -> unsigned {
        unsigned prefix = 0;
        unsigned long long left = need;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const unsigned hi_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int b = tid; b < 256; b += FT) hist[b] = 0ull;
            __syncthreads();
            for (int i = tid; i < V; i += FT) {
                const float v = x[i];
                const unsigned k = f2key(v);
                if ((k & hi_mask) == prefix && k >= floor_key) atomicAdd(&hist[(k >> shift) & 255u], mass ? wfix(v) : 1ull);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned long long acc = 0ull;
                int b = 255;
                for (; b > 0; --b) { if (acc + hist[b] >= left) break; acc += hist[b]; }
                sh_sel = (unsigned)b; sh_left = left - acc;          // (b == 0: everything that is left lies in the lowest bin)
            }
            __syncthreads();
            prefix |= sh_sel << shift;
            left = sh_left;
            __syncthreads();
        }
        return prefix;
    };
    unsigned thr = 0u;                                                   // keep key >= thr
    if (top_k > 0 && top_k < V) thr = radix_select((unsigned long long)top_k, false, 0u);
    if (top_p < 1.f) {
        // mass kept by top-k, then the nucleus threshold inside it
        unsigned long long part = 0ull;
        for (int i = tid; i < V; i += FT) { const float v = x[i]; if (f2key(v) >= thr) part += wfix(v); }
        if (tid == 0) sh_total = 0ull;
        __syncthreads();
        atomicAdd(&sh_total, part);
        __syncthreads();
        const double need_d = (double)top_p * (double)sh_total;
        unsigned long long need = (unsigned long long)need_d;
        if ((double)need < need_d) ++need;                                // ceil: mass(key >= t) >= top_p * total
        if (need < 1ull) need = 1ull;
        const unsigned tp = radix_select(need, true, thr);
        thr = tp > thr ? tp : thr;
    }
    // ---- inverse CDF over the kept tokens in index order ----
    const int per = (V + FT - 1) / FT, lo = tid * per, hi = min(V, lo + per);
    double mine = 0.0;
    int cnt = 0;
    for (int i = lo; i < hi; ++i) { const float v = x[i]; if (f2key(v) >= thr) { mine += (double)__expf((v - mx) * inv_t); ++cnt; } }
    chunk[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int t = 0; t < FT; ++t) tot += chunk[t];
        const double target = (double)fminf(fmaxf(u[row], 0.f), 0.99999994f) * tot;
        double run = 0.0;
        int owner = 0;
        for (int t = 0; t < FT; ++t) { if (chunk[t] > 0.0) { owner = t; if (target < run + chunk[t]) break; run += chunk[t]; } }
        sh_owner = owner; sh_before = run;
    }
    if (diag) {                                                           // kept count (integer atomics) + the threshold logit
        if (tid == 0) sh_total = 0ull;
        __syncthreads();
        atomicAdd(&sh_total, (unsigned long long)cnt);
        __syncthreads();
        if (tid == 0) { diag[2 * row] = __float_as_uint(key2f(thr)); diag[2 * row + 1] = (unsigned)sh_total; }
    }
    __syncthreads();
    if (tid == sh_owner) {
        const double target = (double)fminf(fmaxf(u[row], 0.f), 0.99999994f);
        double tot = 0.0;
        for (int t = 0; t < FT; ++t) tot += chunk[t];
        const double tg = target * tot;
        double run = sh_before;
        int pick = -1;
        for (int i = lo; i < hi; ++i) {
            const float v = x[i];
            if (f2key(v) >= thr) { pick = i; const double w = (double)__expf((v - mx) * inv_t); if (tg < run + w) break; run += w; }
        }
        out[row] = pick < 0 ? 0 : pick;
    }
}

}  // namespace

namespace {
// bookkeeping of one batch-1 decode step, ONE launch instead of six elementwise ones inside the captured graph: the picked token becomes the
// next input id and is appended to the output ring; position, valid key count, ring index and history length advance by one
__global__ void k_decode_advance(const int64_t* __restrict__ nxt, int64_t* __restrict__ ring, int64_t* __restrict__ cnt, int32_t* __restrict__ tok,
                                 int32_t* __restrict__ pos, int32_t* __restrict__ len, int32_t* __restrict__ nprev) {
    if (threadIdx.x == 0) {
        const int64_t t = nxt[0], c = cnt[0];
        ring[c] = t;
        tok[0] = (int32_t)t;
        cnt[0] = c + 1;
        pos[0] += 1; len[0] += 1; nprev[0] += 1;
    }
}
}  // namespace

extern "C" int sc_decode_advance(const int64_t* next_token, int64_t* ring, int64_t* ring_index, int32_t* token, int32_t* pos, int32_t* kv_len,
                                 int32_t* n_prev, sc_stream_t stream) {
    SC_REQUIRE(next_token && ring && ring_index && token && pos && kv_len && n_prev, "sc_decode_advance: null pointer argument");
    hipLaunchKernelGGL(k_decode_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, next_token, ring, ring_index, token, pos, kv_len, n_prev);
    SC_CHECK_LAUNCH("sc_decode_advance");
    return SC_OK;
}

namespace {
// u[b] = a uniform in [0, 1) that is a pure function of (seeds[b], n), n = counter[0] + counter_add: splitmix64 of seed + n * golden, top 24 bits.
// The n-th sampled token of a sequence draws the same number whether the sequence is decoded alone or in a batch, eagerly or from a replayed
// graph, on whichever host thread: nothing is read from (or advanced in) a generator on the device.
__global__ void k_counter_uniform(const int64_t* __restrict__ seeds, const int64_t* __restrict__ counter, int64_t add, int B, float* __restrict__ u) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long n = (unsigned long long)((counter ? counter[0] : 0) + add);
    unsigned long long x = (unsigned long long)seeds[b] + n * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    u[b] = (float)(x >> 40) * (1.0f / 16777216.0f);
}
}  // namespace

extern "C" int sc_counter_uniform_f32(const int64_t* seeds, int B, const int64_t* counter, int64_t counter_add, float* u, sc_stream_t stream) {
    SC_REQUIRE(seeds && u && B > 0, "sc_counter_uniform_f32: null pointer argument or B <= 0");
    hipLaunchKernelGGL(k_counter_uniform, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, seeds, counter, counter_add, B, u);
    SC_CHECK_LAUNCH("sc_counter_uniform_f32");
    return SC_OK;
}

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

extern "C" size_t sc_sample_token_workspace_bytes(int B) {
    return B > 0 ? sc_align_up(sc_pick_token_workspace_bytes(B), 256) + (size_t)B * NBLK * TK_MAX * sizeof(Cand) : 0;
}

extern "C" int sc_sample_token_f32(float* logits, int B, int V, int64_t ld, float temperature, int top_k, float top_p, float repetition_penalty,
                                   const int64_t* prev_ids, int64_t prev_ld, const int32_t* n_prev_dev, int n_prev_host, const float* u, int64_t* out,
                                   void* ws, size_t ws_bytes, sc_stream_t stream) {
    SC_REQUIRE(logits && out && ws, "sc_sample_token_f32: null pointer argument");
    SC_REQUIRE(B > 0 && V > 0 && ld >= V, "sc_sample_token_f32: bad sizes");
    SC_REQUIRE(top_k >= 0, "sc_sample_token_f32: top_k must be >= 0 (0 = off)");
    SC_REQUIRE(top_p > 0.f && repetition_penalty > 0.f, "sc_sample_token_f32: top_p and repetition_penalty must be positive");
    SC_REQUIRE(!(temperature > 0.f) || u, "sc_sample_token_f32: sampling (temperature > 0) needs the uniform draws u[B]");
    if (ws_bytes < sc_sample_token_workspace_bytes(B))
        return sc_fail(SC_ERR_WORKSPACE, "sc_sample_token_f32: workspace %zu < required %zu", ws_bytes, sc_sample_token_workspace_bytes(B));
    hipStream_t s = (hipStream_t)stream;
    if (repetition_penalty != 1.f && prev_ids && (n_prev_dev || n_prev_host > 0)) {
        const size_t bm = (size_t)((V + 31) / 32) * 4;
        SC_REQUIRE(bm <= 64 * 1024, "sc_sample_token_f32: vocabulary too large for the repetition bitmap");
        hipLaunchKernelGGL(k_rep_penalty, dim3(B), dim3(THREADS), bm, s, logits, V, ld, prev_ids, prev_ld, n_prev_dev, n_prev_host, repetition_penalty);
    }
    if (top_k > V) top_k = V;                                          // HF: top_k = min(top_k, vocabulary)
    if ((top_k == 0 || top_k == V) && top_p >= 1.f)                       // nothing to cut: the plain arg-max / temperature sample
        return sc_pick_token_f32(logits, B, V, ld, temperature, u, out, ws, ws_bytes, stream);
    if (!(temperature > 0.f))                                             // arg-max: no warper can remove the largest logit
        return sc_pick_token_f32(logits, B, V, ld, temperature, u, out, ws, ws_bytes, stream);
    if (top_k == 0 || top_k == V || top_k > TK_MAX) {                      // a nucleus without top-k, or more candidates than the candidate path keeps
        hipLaunchKernelGGL(k_full_filter, dim3(B), dim3(FT), 0, s, (const float*)logits, V, ld, 1.0f / temperature, top_k == V ? 0 : top_k, top_p, u, out,
                           (unsigned*)ws);                                // ws[0 .. 2B) <- per row {threshold logit bits, kept count} (diagnostics)
        SC_CHECK_LAUNCH("sc_sample_token_f32");
        return SC_OK;
    }
    Cand* cand = (Cand*)((char*)ws + sc_align_up(sc_pick_token_workspace_bytes(B), 256));
    const int per = ((V + NBLK - 1) / NBLK + 3) & ~3;
    const float inv_t = temperature > 0.f ? 1.0f / temperature : 0.f;
    hipLaunchKernelGGL(k_topk_partial, dim3(NBLK, B), dim3(THREADS), (size_t)per * sizeof(float), s, logits, V, ld, top_k, cand);
    hipLaunchKernelGGL(k_topk_final, dim3(B), dim3(THREADS), (size_t)NBLK * top_k * sizeof(Cand), s, cand, top_k, inv_t, top_p, temperature > 0.f ? u : nullptr, out);
    SC_CHECK_LAUNCH("sc_sample_token_f32");
    return SC_OK;
}
