// Similarity top-k (launch-latency bound: M is tens to hundreds of captions / dialogue docs).
// Replaces cos_sim + running argmax of the tree search (reference utiles.py:732-740,768-771) and the
// FAISS flat-L2 search of the dialogue memory (memory_bank/memory_retrieval/local_doc_qa.py:270).
// One block: each wave scores documents (fp32 lane partials -> fp64 wave sum), then k rounds of a
// block-wide arg-best with lowest-index tie-break select the results on the device.
#include "sc_common.h"

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// dynamic LDS: scores[M] doubles
__global__ __launch_bounds__(256) void k_sim_topk(const float* __restrict__ q, const float* __restrict__ docs, int M, int d, int k,
                                                  int metric, int* __restrict__ idx, float* __restrict__ score) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sc = reinterpret_cast<double*>(smem);
    __shared__ double red_v[4];
    __shared__ int red_i[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    double qn = 0.0;
    for (int j = lane; j < d; j += 64) qn += (double)q[j] * (double)q[j];
    qn = wave_sum_f64(qn);
    for (int m = wave; m < M; m += nw) {
        const float* x = docs + (size_t)m * d;
        double dot = 0.0, xn = 0.0, l2 = 0.0;
        for (int j = lane; j < d; j += 64) {
            const double a = q[j], b = x[j];
            dot += a * b; xn += b * b; l2 += (a - b) * (a - b);
        }
        dot = wave_sum_f64(dot); xn = wave_sum_f64(xn); l2 = wave_sum_f64(l2);
        if (lane == 0) sc[m] = metric == 0 ? dot / (fmax(sqrt(qn), 1e-12) * fmax(sqrt(xn), 1e-12)) : -l2;   // larger = better
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        double bv = -INFINITY; int bi = 0x7fffffff;
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            const double v = sc[m];
            if (!(v != v) && (v > bv || (v == bv && m < bi)) ) { bv = v; bi = m; }
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const double ov = __shfl_xor(bv, s, 64); const int oi = __shfl_xor(bi, s, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w2 = 1; w2 < nw; ++w2)
                if (red_v[w2] > bv || (red_v[w2] == bv && red_i[w2] < bi)) { bv = red_v[w2]; bi = red_i[w2]; }
            if (bi == 0x7fffffff) { bi = -1; }
            idx[r] = bi;
            score[r] = (float)(metric == 0 ? bv : -bv);
            if (bi >= 0) sc[bi] = NAN;   // consumed (NaN never compares as best)
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int sc_sim_topk(const float* q, const float* docs, int M, int d, int k, int metric, int32_t* idx, float* score,
                           sc_stream_t stream) {
    SC_REQUIRE(q && docs && idx && score, "sc_sim_topk: null pointer argument");
    SC_REQUIRE(M > 0 && d > 0 && k > 0 && k <= 64, "sc_sim_topk: need M > 0, d > 0, 0 < k <= 64");
    SC_REQUIRE(k <= M, "sc_sim_topk: k (%d) > M (%d)", k, M);
    SC_REQUIRE(metric == 0 || metric == 1, "sc_sim_topk: metric must be 0 (cosine) or 1 (L2)");
    SC_REQUIRE((size_t)M * 8 <= 64 * 1024, "sc_sim_topk: M too large for the single-block kernel (max 8192)");
    hipLaunchKernelGGL(k_sim_topk, dim3(1), dim3(256), (size_t)M * sizeof(double), (hipStream_t)stream, q, docs, M, d, k, metric, idx, score);
    SC_CHECK_LAUNCH("sc_sim_topk");
    return SC_OK;
}
