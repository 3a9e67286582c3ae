/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU oracle for the selective-frame k-means of StreamChat
 * (reference: utiles.py:291-330 `weighted_kmeans_feature` / inner `weighted_kmeans_torch`
 * :294-318).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this.  Pinned against the golden vectors tests/golden/kmeans_*.npz, which were produced by
 * running the reference's own function (tools/make_golden.py).
 *
 * Semantics restated (fp32-canonical, SURVEY.md §0 surprise 4: the reference overflows in fp16):
 *   C_0 = X[init_idx]                                               utiles.py:295-296
 *   loop i < max_iter:
 *     labels_i[t] = argmin_k ||X[t]-C_i[k]||   (first minimum)      :299-302
 *     S[k] = sum_{t:label=k} w_t X[t] ; W[k] = sum w_t              :303-308
 *     C'[k] = S[k]/W[k] if W[k]>0 else X[reseed.pop()]              :309-313
 *     if sum_k ||C_i[k]-C'[k]||_2 < tol: break  (C_i, labels_i kept) :314-316
 *     C_{i+1} = C'                                                  :317
 *   return C, labels, W, i        (on exhaustion C is one update ahead of labels — Q3)
 *
 * Reduction spec "SC-KM2" (round 6; shared, bit for bit, with streamchat_amd/csrc/kmeans.hip so that
 * labels are identical by construction, ties included).  SC-KM1 (rounds 1-5) reduced every 512-column
 * chunk with a 64-lane tree per (row, cluster): that tree is what kept a one-read Lloyd pass from
 * fitting on chip with more than one wave per SIMD.  SC-KM2 keeps SC-KM1's cell (8
 * columns, two fma chains) and replaces everything above it by a structure a thread that owns a ROW
 * can compute alone (no cross-lane step), while a lane that owns COLUMNS still can (an 8- or 16-lane
 * butterfly):
 *   - cell    = 8 consecutive columns c*8 + e, e < 8; columns >= D contribute 0.
 *               d = x - c (fp32); even e feed acc0 = fmaf(d,d,acc0), odd e feed acc1; p = acc0 + acc1.
 *   - slice   = SC_SLICE (64) consecutive columns = SC_SLICE/8 cells; slice partial = adjacent-pair
 *               tree over the cells: level j = 0,1,.. adds the partial of cell a + 2^j to cell a
 *               (a a multiple of 2^(j+1)), fp32.
 *   - group   = 2048 consecutive columns; group total = the group's slice partials added in ascending
 *               order to 0.0 in fp64.
 *   - total   = the groups are cut into 32 contiguous segments of ceil(ngroups/32); each segment is
 *               summed in ascending group order in fp64, then the 32 segment sums in ascending order.
 *   - argmin on the fp64 totals (sqrt is monotone; first minimum wins).
 *   - update: the clusters are laid out one after the other in 8-row groups: cluster k has
 *     ceil(n_k / 8) groups starting at group gs_k = sum_{k' < k} ceil(n_k' / 8).  The rows of a cluster
 *     in ascending order have ranks j = 0, 1, ..; rank j belongs to chain
 *     (wv, par) = ((gs_k + (j >> 3)) & 7, j & 1) - 16 chains: 8 waves x 2 half-waves each run one, and a
 *     wave owns the same 8-row groups of the layout for every cluster (balanced whatever the sizes).
 *     Per column: chain sum s[wv][par] = 0, then s = s + (w_t * x) over the chain's ranks ascending,
 *     fp32 without fma contraction; u[wv] = s[wv][0] + s[wv][1]; S = u[0], S = S + u[wv] for wv = 1..7;
 *     C' = S / W (fp32 division).  W = sequential fp32 sum of the cluster's weights, ascending.
 *   - shift: the same cell / slice / group / segment structure on (C_i - C')^2 per cluster, then
 *     sum_k sqrt(total_k) in fp64, compared with (double)tol.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC kmeans_oracle.c -o libsc_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef SC_SLICE
#define SC_SLICE 64             /* columns per slice (spec constant; kmeans.hip: KM_SW) */
#endif
#define SC_GROUP 2048           /* columns per fp64 group (spec constant; kmeans.hip: KM_GW) */
#define NCELL (SC_SLICE / 8)
#define NSL (SC_GROUP / SC_SLICE)
#define NSEG 32

static inline float h2f(uint16_t h) { /* IEEE binary16 -> binary32, exact */
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, o;
    if (e == 0) {
        if (m == 0) o = s;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3ffu; o = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) o = s | 0x7f800000u | (m << 13);
    else o = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &o, 4); return f;
}
static inline float b2f(uint16_t b) { uint32_t o = (uint32_t)b << 16; float f; memcpy(&f, &o, 4); return f; }

/* dtype: 0 = f16, 1 = bf16, 2 = f32 */
static inline float ldx(const void *X, int dtype, size_t i) {
    if (dtype == 2) return ((const float *)X)[i];
    uint16_t v = ((const uint16_t *)X)[i];
    return dtype == 0 ? h2f(v) : b2f(v);
}

/* adjacent-pair tree over the NCELL cell partials of a slice */
static float slice_tree(float *p) {
    for (int h = 1; h < NCELL; h <<= 1)
        for (int a = 0; a < NCELL; a += 2 * h) p[a] = p[a] + p[a + h];
    return p[0];
}

/* fp64 group total of sum_j (x_j - c_j)^2 over group g of row `rowoff` */
static double group_total_xc(const void *X, int dtype, size_t rowoff, const float *Crow, int64_t D, int64_t g) {
    double tot = 0.0;
    for (int s = 0; s < NSL; ++s) {
        float p[NCELL];
        for (int cl = 0; cl < NCELL; ++cl) {
            float a0 = 0.f, a1 = 0.f;
            for (int e = 0; e < 8; ++e) {
                int64_t col = g * SC_GROUP + (int64_t)s * SC_SLICE + cl * 8 + e;
                float x = 0.f, cc = 0.f;
                if (col < D) { x = ldx(X, dtype, rowoff + (size_t)col); cc = Crow[col]; }
                float d = x - cc;
                if (e & 1) a1 = fmaf(d, d, a1); else a0 = fmaf(d, d, a0);
            }
            p[cl] = a0 + a1;
        }
        tot += (double)slice_tree(p);
    }
    return tot;
}
static double group_total_cc(const float *A, const float *B, int64_t D, int64_t g) {
    double tot = 0.0;
    for (int s = 0; s < NSL; ++s) {
        float p[NCELL];
        for (int cl = 0; cl < NCELL; ++cl) {
            float a0 = 0.f, a1 = 0.f;
            for (int e = 0; e < 8; ++e) {
                int64_t col = g * SC_GROUP + (int64_t)s * SC_SLICE + cl * 8 + e;
                float d = 0.f;
                if (col < D) d = A[col] - B[col];
                if (e & 1) a1 = fmaf(d, d, a1); else a0 = fmaf(d, d, a0);
            }
            p[cl] = a0 + a1;
        }
        tot += (double)slice_tree(p);
    }
    return tot;
}

static double seg_total(const double *gp, int64_t ng) { /* gp[ng] fp64 group totals -> two-level sum */
    int64_t seglen = (ng + NSEG - 1) / NSEG;
    double tot = 0.0;
    for (int s = 0; s < NSEG; ++s) {
        double a = 0.0;
        int64_t lo = (int64_t)s * seglen, hi = lo + seglen; if (hi > ng) hi = ng;
        for (int64_t c = lo; c < hi; ++c) a += gp[c];
        tot += a;
    }
    return tot;
}

/* squared distances dist2[T*K] (fp64) of every row to every centroid, SC-KM2 order */
void sc_oracle_kmeans_dist2(const void *X, int dtype, int T, int64_t D, int K, const float *C, double *dist2) {
    int64_t nch = (D + SC_GROUP - 1) / SC_GROUP;
#pragma omp parallel
    {
        double *wp = (double *)malloc(sizeof(double) * (size_t)nch);
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < K; ++k) {
                for (int64_t c = 0; c < nch; ++c)
                    wp[c] = group_total_xc(X, dtype, (size_t)t * (size_t)D, C + (size_t)k * D, D, c);
                dist2[(size_t)t * K + k] = seg_total(wp, nch);
            }
        free(wp);
    }
}

/* Full fit.  trace_labels (may be NULL): [max_iter*T] int32, row i = labels of iteration i.
 * returns 0, or -1 on bad arguments.  reseed_idx may be NULL only if no cluster ever empties
 * (returns -2 otherwise). */
int sc_oracle_kmeans_fit(const void *X, int dtype, int T, int64_t D, int K, const float *w,
                         const int32_t *init_idx, const int32_t *reseed_idx, int n_reseed, int max_iter, float tol,
                         float *C /*[K*D] out*/, int64_t *labels /*[T] out*/, float *wsum /*[K] out*/, int *iters,
                         int32_t *trace_labels) {
    if (T <= 0 || D <= 0 || K <= 0 || max_iter <= 0) return -1;
    int64_t nch = (D + SC_GROUP - 1) / SC_GROUP;
    float *Ccur = (float *)malloc(sizeof(float) * (size_t)K * D), *Cnew = (float *)malloc(sizeof(float) * (size_t)K * D);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)T * K);
    double *wp = (double *)malloc(sizeof(double) * (size_t)nch);
    int rpos = 0, rc = 0, i;
    for (int k = 0; k < K; ++k) {
        if (init_idx[k] < 0 || init_idx[k] >= T) { rc = -1; goto out; }
        for (int64_t j = 0; j < D; ++j) Ccur[(size_t)k * D + j] = ldx(X, dtype, (size_t)init_idx[k] * D + j);
    }
    for (i = 0; i < max_iter; ++i) {
        sc_oracle_kmeans_dist2(X, dtype, T, D, K, Ccur, d2);
        for (int t = 0; t < T; ++t) {
            int best = 0; double bv = d2[(size_t)t * K];
            for (int k = 1; k < K; ++k) if (d2[(size_t)t * K + k] < bv) { bv = d2[(size_t)t * K + k]; best = k; }
            labels[t] = best;
            if (trace_labels) trace_labels[(size_t)i * T + t] = best;
        }
        int gs = 0;                                   /* first 8-row group of the cluster in the sorted layout */
        for (int k = 0; k < K; ++k) {
            float W = 0.f;
            int nk = 0;
            for (int t = 0; t < T; ++t) if (labels[t] == k) { W = W + (w ? w[t] : 1.0f); ++nk; }
            const int gsk = gs;
            gs += (nk + 7) / 8;
            wsum[k] = W;
            float *cn = Cnew + (size_t)k * D;
            if (W > 0.f) {
#pragma omp parallel for schedule(static)
                for (int64_t j = 0; j < D; ++j) {
                    float ch[8][2];
                    for (int a = 0; a < 8; ++a) ch[a][0] = ch[a][1] = 0.f;
                    int rank = 0;
                    for (int t = 0; t < T; ++t) if (labels[t] == k) {
                        float prod = (w ? w[t] : 1.0f) * ldx(X, dtype, (size_t)t * D + j);
                        float *c = &ch[(gsk + (rank >> 3)) & 7][rank & 1];
                        *c = *c + prod;
                        ++rank;
                    }
                    float s = ch[0][0] + ch[0][1];
                    for (int a = 1; a < 8; ++a) { float u = ch[a][0] + ch[a][1]; s = s + u; }
                    cn[j] = s / W;
                }
            } else {
                if (!reseed_idx || rpos >= n_reseed) { rc = -2; goto out; }
                int r = reseed_idx[rpos++];
                if (r < 0 || r >= T) { rc = -1; goto out; }
                for (int64_t j = 0; j < D; ++j) cn[j] = ldx(X, dtype, (size_t)r * D + j);
            }
        }
        double diff = 0.0;
        for (int k = 0; k < K; ++k) {
            for (int64_t c = 0; c < nch; ++c) wp[c] = group_total_cc(Ccur + (size_t)k * D, Cnew + (size_t)k * D, D, c);
            diff += sqrt(seg_total(wp, nch));
        }
        if (diff < (double)tol) break;
        float *tmp = Ccur; Ccur = Cnew; Cnew = tmp;
    }
    if (i == max_iter) i = max_iter - 1;   /* python: loop variable after exhaustion */
    *iters = i;
    memcpy(C, Ccur, sizeof(float) * (size_t)K * D);
out:
    free(Ccur); free(Cnew); free(d2); free(wp);
    return rc;
}

/* ---- retrieval oracle: cosine / flat-L2 top-k (utiles.py:732-740; local_doc_qa.py:270 FAISS flat L2) ----
 * metric 0: cosine, scores descending; metric 1: squared L2, ascending.  fp64 accumulate in index
 * order; ties -> lowest index.  idx/score sized k. */
void sc_oracle_topk(const float *q, const float *docs, int M, int d, int k, int metric, int32_t *idx, float *score) {
    double *s = (double *)malloc(sizeof(double) * (size_t)M);
    double qn = 0; for (int j = 0; j < d; ++j) qn += (double)q[j] * q[j];
    for (int m = 0; m < M; ++m) {
        const float *x = docs + (size_t)m * d; double dot = 0, xn = 0, l2 = 0;
        for (int j = 0; j < d; ++j) { dot += (double)q[j] * x[j]; xn += (double)x[j] * x[j]; double e = (double)q[j] - x[j]; l2 += e * e; }
        s[m] = metric == 0 ? dot / (fmax(sqrt(qn), 1e-12) * fmax(sqrt(xn), 1e-12)) : l2;
    }
    char *used = (char *)calloc((size_t)M, 1);
    for (int r = 0; r < k && r < M; ++r) {
        int best = -1;
        for (int m = 0; m < M; ++m) { if (used[m]) continue; if (best < 0 || (metric == 0 ? s[m] > s[best] : s[m] < s[best])) best = m; }
        used[best] = 1; idx[r] = best; score[r] = (float)s[best];
    }
    free(used); free(s);
}
