/* streamchat_hip.h — C ABI of libstreamchat_hip.so (hand-written gfx950 / CDNA4 kernels).
 *
 * The reference (hmxiong/StreamChat) is pure Python with NO plugin / FFI boundary
 * (SURVEY.md §2.2, §8(b)); the seams it exposes are Python call signatures.  This header is the
 * drop-in boundary the build defines underneath those seams: each entry point names the reference
 * call site(s) whose arithmetic it replaces.  The Python host mirror (the streamchat_amd Python package) binds
 * these symbols with ctypes on `tensor.data_ptr()`; INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / HIP C++ types in signatures
 *     (`sc_stream_t` is a `hipStream_t` passed as void*; NULL = the default stream).
 *   - return 0 on success, negative on error; `sc_last_error()` gives the text (thread-local).
 *     Never throws or aborts across the ABI.
 *   - every data pointer is a DEVICE pointer unless the parameter comment says "host".
 *   - the library allocates no device memory: scratch is an explicit caller-owned workspace whose
 *     size the matching *_workspace_bytes() reports.  Kernels are asynchronous on the given stream;
 *     there is no hidden synchronisation.  Re-entrant for distinct streams + workspaces.
 *   - row-major contiguous tensors only; leading dimensions are explicit where they may differ.
 *   - dtype codes: SC_F16 = 0, SC_BF16 = 1, SC_F32 = 2.
 */
#ifndef STREAMCHAT_HIP_H
#define STREAMCHAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sc_stream_t;

enum { SC_F16 = 0, SC_BF16 = 1, SC_F32 = 2 };
enum { SC_OK = 0, SC_ERR_ARG = -1, SC_ERR_WORKSPACE = -2, SC_ERR_LAUNCH = -3, SC_ERR_UNSUPPORTED = -4 };

/* ABI changelog (a binding asserts sc_abi_version() >= the version that introduced the newest symbol it uses):
 *   1  first release
 *   2  sc_attention_f16 gained batch strides; + sc_avgpool_tokens_f16
 *   3  + sc_kmeans_update          (one centroid update from given labels: torch_kmeans / kmeans_pytorch host loops)
 *      + sc_decode_qkv_f16         (q/k/v GEMV + RMSNorm + RoPE + KV-cache append in one launch)
 *      + sc_pick_token_f32, sc_pick_token_workspace_bytes     (arg-max / temperature sampling on device)
 *      + sc_sample_token_f32, sc_sample_token_workspace_bytes (repetition penalty / top-k / top-p chain)
 *      + sc_attention_variant      (which attention kernel a shape is dispatched to)
 *      round 3: + sc_gemm_headed_f16 (rotary / column-scale GEMM epilogues), sc_rope_table_f32, sc_rope_f32in_f16, sc_decode_qkv_tab_f16,
 *               sc_decode_advance;
 *               sc_attention_f16's `causal` argument became a flag word (bit 1 = SC_ATTN_Q_PRESCALED; 0 / 1 mean what they meant)
 *      (these seven shipped in round 2 under version 2 by mistake; 3 is the first version that guarantees them)
 *   4  round 4: the three consumers of a rotary table take the table's ROW COUNT (sc_gemm_headed_f16 `rope_tab_rows`, sc_rope_f32in_f16 and
 *      sc_decode_qkv_tab_f16 `tab_rows`): positions known on the host (pos0 + rows) beyond the table are SC_ERR_ARG, positions read from
 *      device memory (`positions[]`, `pos[0]`) are clamped to the last table row instead of reading past the allocation
 *   5  round 5: + sc_build_info; + sc_stream_create_masked / sc_stream_destroy / sc_set_cu_budget (CU-partitioned streams: the HBM-bound
 *      answer decode beside the MFMA-bound encode / prefill of the next segment); sc_attention_f16 decode path: grid and chunk shares
 *      changed (results of a split-KV call differ in the last bit from version 4's, every caller sees one consistent kernel)
 *   6  round 5: + sc_rope_qkv_rows_f16 (the batched decode step's rotary + KV-append in one launch)
 *   7  round 6: - sc_set_cu_budget (the one piece of process-wide mutable state in the library: a CU-masked stream has carried its own CU
 *      count since version 5, and that is the only way a persistent launch is sized now - "no global state besides the thread-local error
 *      text" holds again); k-means: reduction spec SC-KM2 (labels of non-tied inputs unchanged; centroids / fp64 distances differ from
 *      version 6's in the last bits), workspace layout changed (sc_kmeans_workspace_bytes says how much, as always);
 *      + sc_counter_uniform_f32 (the uniform draw of the n-th sampled token as a pure function of (seed, n): batched == one-by-one and
 *      graph == eager under sampling, no generator state on the device)
 *   8  round 6: + sc_kmeans_fit_cols (sc_kmeans_fit on a column slab of X that holds whole SC-KM2 segments, with a caller-supplied exchange of the
 *      two fp64 segment tables per iteration: the data-parallel Lloyd of a merge group, bit-identical to the 1-GPU fit at any rank count)
 */
#define SC_ABI_VERSION 8

int sc_abi_version(void);
const char* sc_last_error(void);
/* v5: how the library was built ("abi=5 attention=iterative-ilp decode=kernarg-preload"; "...(FALLBACK)" when the Makefile had to fall
 * back to the default machine scheduler for attention.hip): performance numbers of different builds are not comparable - bench.py prints it */
const char* sc_build_info(void);
/* v5: CU-partitioned streams (hipExtStreamCreateWithCUMask).  A stream restricted to CUs [cu_first, cu_first + cu_count) of the device's CU
 * mask.  Consecutive mask bits go round-robin over the 8 XCDs x 4 shader engines: use MULTIPLES OF 32 for both numbers (a partition that leaves
 * the engines of an XCD unequal is paced by its smallest one: 112 / 144 CUs measured 25 - 35 % slower than 96 / 128).  The stream carries its CU
 * count: the library's persistent launches (one workgroup per CU) size their grids for the stream they are launched on - per stream, no
 * process-wide setting (v7 removed sc_set_cu_budget).  The reference has no such notion; it runs reader / updater / QA as Python threads on
 * one default stream (reference previous_version/streaming_demo_llava_next_3.py:967-991). */
int sc_stream_create_masked(int cu_first, int cu_count, int high_priority, sc_stream_t* out);
int sc_stream_destroy(sc_stream_t stream);
/* host out-params: number of CUs, 1 if the device is gfx950, total HBM bytes */
int sc_device_info(int* cu_count, int* is_gfx950, size_t* hbm_bytes);

/* ----------------------------------------------------------------------------------------------
 * Selective-frame k-means.  Replaces `weighted_kmeans_feature` / inner `weighted_kmeans_torch`
 * (reference utiles.py:291-330, :294-318; callers utiles.py:587, inference_streaming_longva_v2.py:347).
 *   X          [T, D] row-major, dtype code `dtype` (D = P*Dm = 576*3584 for LongVA frames)
 *   w          [T] fp32 point weights or NULL (= all ones, reference default :292-293)
 *   init_idx   [K] int32 rows used as initial centroids (the reference draws torch.randperm :295;
 *              the draw is an explicit input here so results do not depend on a device RNG)
 *   reseed_idx [n_reseed] int32 rows consumed in order whenever a cluster is empty
 *              (reference: random.randint :313); may be NULL if n_reseed == 0
 *   C          [K, D] fp32 out — centroids as the reference returns them (Q3 semantics)
 *   labels     [T] int64 out — assignment of the last executed iteration
 *   wsum       [K] fp32 out — weights_sum of the last update
 *   info       [4] int32 out: {exit_iter, status (0 ok, 1 = reseed stream exhausted),
 *              reseeds consumed, reserved}
 * Lloyd iterations run entirely on the device (no host round trip); iterations after
 * convergence are skipped by a device-side flag.  Reduction order: "SC-KM2" (DESIGN.md section 2),
 * restated bit-for-bit by oracle/kmeans_oracle.c.
 */
size_t sc_kmeans_workspace_bytes(int T, int64_t D, int K);
int sc_kmeans_fit(const void* X, int dtype, int T, int64_t D, int K, const float* w,
                  const int32_t* init_idx, const int32_t* reseed_idx, int n_reseed,
                  int max_iter, float tol, float* C, int64_t* labels, float* wsum, int32_t* info,
                  void* ws, size_t ws_bytes, sc_stream_t stream);
/* v8: the same fit, DATA-PARALLEL OVER COLUMNS (`north_star`: "k-means data-parallel"; reference utiles.py:294-318 on one device).  Under SC-KM2
 * a distance is 32 contiguous SEGMENTS of 2048-column groups summed in fp64 and then the 32 segment sums in order; updates, reseeds and the
 * initial rows are per column; the convergence shift has the distance structure.  A rank that holds the columns of whole segments therefore
 * computes exactly the segment sums the 1-GPU fit computes, and once the two small segment tables are complete on every rank, the arg-min,
 * ordering and convergence kernels run replicated on identical inputs: labels, wsum, info and the rank's columns of C are bit-identical to
 * sc_kmeans_fit on the whole matrix, for any number of ranks.
 *   X            [T, D_local] row-major: columns [seg_first * seg_groups * 2048, ...) of the whole matrix - segments seg_first ..
 *                seg_first + seg_count - 1; seg_groups = ceil(ceil(D / 2048) / 32) of the WHOLE matrix (groups per segment).  Every slab but the
 *                matrix's last holds seg_count * seg_groups whole groups; the last one (seg_first + seg_count == 32) holds what is left
 *   C_local      [K, D_local] fp32 out: this slab's columns of the centroids;  labels / wsum / info: as sc_kmeans_fit, identical on every rank
 *   seg_dist     [32, T * K] fp64, seg_shift [32, K] fp64: caller-owned device tables.  The library writes this slab's rows
 *                [seg_first, seg_first + seg_count) and then calls `exchange(ctx, what, stream)` (what 0: seg_dist, 1: seg_shift), which must
 *                make ALL 32 rows of that table valid on `stream` order (an all-gather of the ranks' row windows; RCCL enqueues it, a host-staged
 *                transport synchronises the stream) and return 0; max_iter calls of each kind, on every rank, whatever the convergence flag says
 *   ws           sc_kmeans_workspace_bytes(T, D_local, K)
 * The call enqueues the whole loop like sc_kmeans_fit; it returns SC_ERR_LAUNCH if an exchange reports failure. */
typedef int (*sc_kmeans_exchange_fn)(void* ctx, int what, sc_stream_t stream);
int sc_kmeans_fit_cols(const void* X, int dtype, int T, int64_t D_local, int K, const float* w,
                       const int32_t* init_idx, const int32_t* reseed_idx, int n_reseed,
                       int max_iter, float tol, float* C_local, int64_t* labels, float* wsum, int32_t* info,
                       int64_t seg_groups, int seg_first, int seg_count, double* seg_dist, double* seg_shift,
                       sc_kmeans_exchange_fn exchange, void* exchange_ctx, void* ws, size_t ws_bytes, sc_stream_t stream);
/* One assignment step against given centroids (the `kmeans_predict` surface of
 * kmeans_pytorch/__init__.py:130 and torch_kmeans KMeans.predict): labels [T] int64,
 * dist2 [T, K] fp64 squared Euclidean distances (may be NULL). */
int sc_kmeans_assign(const void* X, int dtype, int T, int64_t D, int K, const float* C,
                     int64_t* labels, double* dist2, void* ws, size_t ws_bytes, sc_stream_t stream);

/* One centroid-update step from given labels — with sc_kmeans_assign the two halves of a Lloyd iteration, for the host-driven
 * loops of the vendored clustering APIs whose stopping rule / empty-cluster policy differ from weighted_kmeans_feature:
 * torch_kmeans KMeans._cluster (reference torch_kmeans/clustering/kmeans.py:517-570: mean of the assigned rows, an empty cluster's
 * centre becomes the zero vector, torch_kmeans/utils/utils.py:66) and kmeans_pytorch.kmeans (kmeans_pytorch/__init__.py:92-107:
 * an empty cluster takes a random row).
 *   labels  [T] int64 (device), values in [0, K)         w  [T] fp32 weights or NULL (all 1)
 *   C_old   [K, D] fp32 centroids the shift is measured against; C_new [K, D] fp32 out (must not alias C_old)
 *   empty_mode 0: the j-th empty cluster (ascending k) becomes row fill_idx[j] of X (row 0 if j >= n_fill); 1: zero vector
 *   wsum    [K] fp32 out or NULL: summed weights per cluster;  shift2 [K] fp64 out or NULL: ||C_old[k] - C_new[k]||^2 (SC-KM2 order)
 * Same summation order as sc_kmeans_fit (rows of a cluster in ascending order, fp32, no contraction). */
int sc_kmeans_update(const void* X, int dtype, int T, int64_t D, int K, const float* w, const int64_t* labels,
                     const float* C_old, int empty_mode, const int32_t* fill_idx, int n_fill, float* C_new,
                     float* wsum, double* shift2, void* ws, size_t ws_bytes, sc_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Frame preprocessing.  Replaces `process_images` → CLIPImageProcessor.preprocess rescale+normalise
 * (reference utiles.py:71-87) followed by `.to(torch.float16)` (inference_streaming_longva_v2.py:520).
 *   hwc   [n, h, w, 3] uint8 RGB frames (already resized/cropped: identity for 336x336 streams)
 *   mean/std  host float[3]
 *   out   [n, 3, h, w] fp16:  ((x/255) - mean[c]) / std[c]   (fp32 math, rounded once to fp16)
 */
int sc_preprocess_u8(const uint8_t* hwc, int n, int h, int w, const float* mean, const float* std,
                     void* out_f16, sc_stream_t stream);
/* Same arithmetic fused with the ViT patch gather (im2col of the Conv2d k=s=patch at
 * HF CLIPVisionEmbeddings; call site reference clip_encoder.py:76): writes
 *   out [n * (h/patch) * (w/patch), ld] fp16, column = c*patch*patch + py*patch + px, columns
 *   [3*patch*patch, ld) zero-filled (ld >= 3*patch*patch, multiple of 8). */
int sc_preprocess_patchify_u8(const uint8_t* hwc, int n, int h, int w, int patch, const float* mean,
                              const float* std, void* out_f16, int ld, sc_stream_t stream);

/* Patch gather for already-normalised fp16 pixel values [n, 3, h, w] (the tensor the reference hands to
 * `encode_images`, llava_arch.py:179): same output layout as sc_preprocess_patchify_u8. */
int sc_patchify_f16(const void* chw_f16, int n, int h, int w, int patch, void* out_f16, int ld,
                    sc_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Similarity top-k.  Replaces `cos_sim` + strict-> running argmax of the tree search
 * (reference utiles.py:732,738-740,768-771) and the FAISS flat-L2 `similarity_search_with_score`
 * of the dialogue memory (memory_bank/memory_retrieval/local_doc_qa.py:270).
 *   q [d] fp32, docs [M, d] fp32; metric 0 = cosine (descending), 1 = squared L2 (ascending)
 *   idx [k] int32, score [k] fp32 out, best first; ties -> lowest index.  k <= 64.
 */
int sc_sim_topk(const float* q, const float* docs, int M, int d, int k, int metric, int32_t* idx,
                float* score, sc_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Dense building blocks of the encoders (ViT-L/14-336 + mlp2x_gelu projector, BERT text encoders,
 * Qwen2 LLM).  They replace the third-party transformers==4.37.2 modules the reference calls at
 * longva/model/multimodal_encoder/clip_encoder.py:76, multimodal_projector/builder.py:41-48,
 * utiles.py:707,728 and longva/model/language_model/llava_qwen.py:155.
 */
enum { SC_EPI_NONE = 0, SC_EPI_QUICK_GELU = 1, SC_EPI_GELU_ERF = 2, SC_EPI_SWIGLU = 3, SC_EPI_ROPE = 4, SC_EPI_COLSCALE = 5 };
/* SC_EPI_SWIGLU: W rows are interleaved per 4 output columns as (gate_j, gate_j+1, up_j, up_j+1); the kernel writes
 * silu(gate) * up to C[M, N/2] (Qwen2MLP act_fn(gate_proj(x)) * up_proj(x) without the [M, 2I] intermediate). */
/* C[M,N] = epi(A[M,K] @ W[N,K]^T + bias[N]) (+ residual[M,N]).  fp16 in, fp32 accumulate (MFMA),
 * fp16 out (out_f32 = 0) or fp32 out (out_f32 = 1).  W has the torch.nn.Linear layout [N, K].
 * lda / ldr / ldc in elements.  Requirements: K % 64 == 0, N % 128 == 0, A and W 16-byte aligned,
 * lda % 8 == 0.  M is arbitrary.  bias and residual may be NULL.
 * Optional A row-group map (a_grp > 0): logical row m reads storage row
 * (m / a_grp) * a_grp_stride + a_grp_off + m % a_grp  — e.g. (576, 577, 1) drops the CLS token of every
 * frame (`feature_select`, reference clip_encoder.py:46-66) without a copy. */
int sc_gemm_f16(const void* A, int lda, const void* W, const void* bias, const void* residual, int ldr,
                void* C, int ldc, int M, int N, int K, int epilogue, int out_f32, int a_grp,
                int a_grp_stride, int a_grp_off, sc_stream_t stream);

/* The same GEMM with an epilogue that knows the output columns are HEADS of width 128 (round 3, ABI 3; served by the hand-scheduled
 * 256 x 256 kernel only: N % 256 == 0, K % 128 == 0, lda % 8 == 0, ldc % 8 == 0, 16-byte aligned A / W / C / bias - otherwise
 * SC_ERR_UNSUPPORTED and the caller takes sc_gemm_f16(out_f32 = 1) + sc_rope_f32in_f16, which produce the same numbers):
 *   mode SC_EPI_ROPE      Qwen2 q / k|v projections (HF Qwen2Attention behind reference llava_qwen.py:155): columns [0, lead_cols) are
 *                         rotary heads - rotate-half RoPE of the fp32 sum (acc + bias) with row pos0 + m of `rope_tab`
 *                         (sc_rope_table_f32), rounded to fp16 ONCE; columns >= lead_cols (the V part) get the bias only.
 *   mode SC_EPI_COLSCALE  (acc + bias) * col_scale for columns < lead_cols, plain beyond: the q third of CLIP's fused q|k|v projection
 *                         leaves pre-scaled for SC_ATTN_Q_PRESCALED (HF CLIPAttention scales q the same way: clip_encoder.py:76). */
int sc_gemm_headed_f16(const void* A, int lda, const void* W, const void* bias, void* C, int ldc, int M, int N, int K,
                       int mode, const float* rope_tab, int rope_tab_rows, int pos0, int lead_cols, float col_scale, sc_stream_t stream);
/* ViT token assembly + pre-LayerNorm (HF CLIPVisionEmbeddings + pre_layrnorm):
 *   out[n, 0]     = LN(cls + pos[0]);  out[n, 1 + p] = LN(patch[n*P + p] + pos[1 + p])
 * patch [N*P, D], cls [D], pos [P+1, D], out [N*(P+1), D], all fp16; D % 8 == 0, D <= 4096. */
int sc_vit_embed_ln_f16(const void* patch, const void* cls, const void* pos, const void* gamma,
                        const void* beta, float eps, void* out, int N, int P, int D, sc_stream_t stream);
/* Decode-time matrix-vector product y[N] = W[N,K] . x[K] (+ bias) (+ residual): weights streamed once (HBM-bound).
 * epilogue SC_EPI_NONE or SC_EPI_SWIGLU (interleaved gate/up rows, y has N/2 entries).  K % 8 == 0.
 * y_row (optional, device int32[1]): write to row y_row[0] of a [rows, y_ld] fp16 buffer starting at y (KV-cache append at a
 * device-resident position, so a whole decode step can be captured once in a hipGraph and replayed).
 * rms_gamma (optional, fp16 [K]): x is first RMS-normalised as Qwen2RMSNorm does (gamma * fp16(x * rsqrt(mean(x^2) + rms_eps))). */
int sc_gemv_f16(const void* W, const void* x, const void* bias, const void* residual, void* y, int N, int K,
                int epilogue, int out_f32, const int32_t* y_row, int y_ld, const void* rms_gamma, float rms_eps,
                sc_stream_t stream);
/* y = LayerNorm(x) * gamma + beta over the last dim, fp32 statistics; [rows, cols] fp16, cols % 8 == 0,
 * cols <= 4096. */
int sc_layernorm_f16(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* y,
                     int ldy, int rows, int cols, sc_stream_t stream);
/* Qwen2 RMSNorm: y = gamma * fp16(x * rsqrt(mean(x^2) + eps)), fp32 statistics. */
int sc_rmsnorm_f16(const void* x, int ldx, const void* gamma, float eps, void* y, int ldy, int rows,
                   int cols, sc_stream_t stream);

/* Qwen2 helpers.  sc_gather_rows_f16: out[r] = table[ids[r]] (embed_tokens; ids < 0 -> zero row, filled by the image
 * splice of llava_arch.py:208-343).  sc_rope_f16: rotate-half rotary embedding in place on `heads` heads of width Dh at
 * column 0 of every row (positions NULL -> pos0 + row), HF apply_rotary_pos_emb numerics. */
int sc_gather_rows_f16(const int32_t* ids, const void* table, void* out, int rows, int H, int ldo, int vocab,
                       sc_stream_t stream);
int sc_rope_f16(void* x, int ld, const int32_t* positions, int pos0, int rows, int heads, int Dh, float theta,
                sc_stream_t stream);
/* RoPE on ONE row of a [rows, ld] buffer; the row number (= token position) is read from device memory (graph-replayable). */
int sc_rope_row_f16(void* buf, int ld, const int32_t* row_index, int heads, int Dh, float theta, sc_stream_t stream);
/* Decode step: RoPE of the new query row q [q_heads*Dh] (position row_index[0]) and of the K part of cache row row_index[0]
 * ([rows, ld] buffer, K = the first kv_heads*Dh columns) in one launch. */
int sc_rope_qk_row_f16(void* q, int q_heads, void* cache, int ld, const int32_t* row_index, int kv_heads, int Dh,
                       float theta, sc_stream_t stream);
/* Rotary embedding from fp32 projections with ONE rounding (round 3, ABI 3).  sc_rope_table_f32: tab[pos][0][j] = cos(pos * theta^(-2j/Dh))
 * * scale, tab[pos][1][j] = sin(..) * scale, j < Dh/2, pos < max_pos (fp32; scale = softmax scale * log2 e for the QUERY table, 1 for keys).
 * sc_rope_f32in_f16: x [rows, ldx] fp32 (projection + bias) -> out [rows, ldo] fp16: the first `heads` heads of width Dh rotated with the
 * table row positions[r] (or pos0 + r), `plain_cols` further columns cast unchanged (the V part of a k|v projection).
 * sc_decode_qkv_tab_f16: sc_decode_qkv_f16 with this arithmetic (tab_q scaled, tab_k not): q leaves pre-scaled for SC_ATTN_Q_PRESCALED.
 * `tab_rows` / `rope_tab_rows` (ABI 4) = positions the table holds: a position beyond it is an error where the host knows it, clamped otherwise. */
int sc_rope_table_f32(float* tab, int max_pos, int Dh, float theta, float scale, sc_stream_t stream);
int sc_rope_f32in_f16(const float* x, int ldx, const float* tab, int tab_rows, const int32_t* positions, int pos0, int rows, int heads, int Dh,
                      int plain_cols, void* out, int ldo, sc_stream_t stream);
/* v6: one new token for each of B sequences (batched decode, llm.BatchDecoder; reference utiles.py:539-559 runs these generates one by one).
 * x [B, ldx] fp32 = fused q|k|v projection + bias ((q_heads + 2 kv_heads) * Dh columns).  q heads rotated with tab_q at positions[b] -> q_out
 * [B, ldq]; k heads rotated with tab_k and the v columns cast -> row positions[b] of sequence b's cache (cache + b * cache_batch_stride +
 * positions[b] * cache_ld elements; rows [K of kv_heads | V of kv_heads]).  Bit-identical to sc_rope_f32in_f16 on the q and on the k|v columns
 * followed by a row scatter.  positions are device data (graph-replayable) and clamped to the table / to cache_rows. */
int sc_rope_qkv_rows_f16(const float* x, int ldx, const float* tab_q, const float* tab_k, int tab_rows, const int32_t* positions, int B, int q_heads,
                         int kv_heads, int Dh, void* q_out, int ldq, void* cache, int64_t cache_batch_stride, int cache_ld, int cache_rows, sc_stream_t stream);
int sc_decode_qkv_tab_f16(const void* Wq, const void* Wkv, const void* bq, const void* bkv, const void* x, const void* rms_gamma,
                          float rms_eps, void* q_out, void* cache, int cache_ld, const int32_t* pos, int q_heads, int kv_heads, int Dh,
                          int K, const float* tab_q, const float* tab_k, int tab_rows, sc_stream_t stream);
/* Text encoders (BERT-large "mxbai-colbert" CLS embedding, reference utiles.py:704-708,725-729; MiniLM-L6 sentence
 * embedding behind HuggingFaceEmbeddings, memory_bank/memory_retrieval/local_doc_qa.py:193):
 *   sc_bert_embed_ln_f16: out[b*L + t] = LN(word[ids[b*L+t]] + pos[t] + type0)      (HF BertEmbeddings)
 *   sc_pool_f16: mode 0 = CLS row, mode 1 = mean over the first len[b] tokens (len NULL = L); optional L2
 *                normalisation; out [B, H] fp32. */
int sc_bert_embed_ln_f16(const int32_t* ids, const void* word, const void* pos, const void* type0,
                         const void* gamma, const void* beta, float eps, void* out, int B, int L, int H,
                         int vocab, sc_stream_t stream);
int sc_pool_f16(const void* hidden, const int32_t* len, float* out, int B, int L, int H, int mode,
                int normalize, sc_stream_t stream);
/* Spatial average pooling of ViT token maps (reference utiles.py:264-289 compress_spatial_features -> F.avg_pool2d, --compress_rate):
 * in [B, P*P, D] fp16 (row-major P x P token grid), out [B, g*g, D] fp16 with g = P / r (floor: trailing rows / columns are dropped),
 * out[b, y*g + x] = mean over the r x r window, accumulated in fp32.  D % 8 == 0. */
int sc_avgpool_tokens_f16(const void* in, void* out, int B, int P, int D, int r, sc_stream_t stream);
/* Fused softmax(Q K^T * scale [+ mask]) V, fp16 in/out, fp32 online softmax (flash-style, S x S never
 * materialised).  Token-major layouts with explicit row strides (so q/k/v may alias one fused QKV buffer):
 *   q   [B, Sq,  Hq,  Dh]  row stride ldq elements, head h at column h*Dh
 *   k,v [B, Skv, Hkv, Dh]  row strides ldk / ldv (GQA: Hq % Hkv == 0)
 *   out [B, Sq,  Hq,  Dh]  row stride ldo
 *   causal: bit 0: 0 = full, 1 = causal with the Sq queries aligned to the END of the Skv keys;
 *           bit 1 (SC_ATTN_Q_PRESCALED, ABI 3): q already carries scale * log2(e) - applied by its producer to the fp32 projection before
 *           the one rounding to fp16 (sc_gemm_headed_f16, sc_decode_qkv_tab_f16, sc_rope_f32in_f16 with a scaled table) - and `scale` is
 *           ignored: the kernel takes p = 2^(q.k - m) without a per-score multiply
 *   kv_len: optional [B] int32 valid key count per batch row (padding mask) or NULL.  Dh in {32, 64, 128}.
 *   nsplit > 1: split-KV ("flash-decoding") for few queries over a long cache — the key range is cut into nsplit slices
 *   processed by separate workgroups and merged; needs ws of B*Hq*Sq*nsplit*(Dh+2)*4 bytes.  nsplit = 1: ws may be NULL.
 *   q_head_stride / o_head_stride (elements; 0 = Dh): distance between consecutive heads inside a q / out row.  With
 *   ldq = Dh and q_head_stride = G*Dh the G query heads of a GQA group become G query ROWS of one KV head without any copy
 *   (decode: every K/V byte is then read once for the whole group).
 *   q_batch_stride / o_batch_stride (elements; 0 = Sq*ldq / Sq*ldo): distance between consecutive batch rows of q / out, so the
 *   head-packed addressing also works for B > 1 (batched decode: q is [B, Hq*Dh] and its batch stride is Hq*Dh, not G*Dh).
 *   k / v batch rows are Skv*ldk / Skv*ldv apart. */
enum { SC_ATTN_CAUSAL = 1, SC_ATTN_Q_PRESCALED = 2 };
int sc_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out,
                     int ldo, int B, int Sq, int Skv, int Hq, int Hkv, int Dh, float scale, int causal,
                     const int32_t* kv_len, int nsplit, void* ws, size_t ws_bytes, int q_head_stride,
                     int o_head_stride, int64_t q_batch_stride, int64_t o_batch_stride, sc_stream_t stream);

/* Which kernel sc_attention_f16 dispatches for a shape in this process: 0 = k_attn (32 queries per wave, whole 64-row tiles), 1 = k_attn
 * long-prefill variant (Dh = 128, Sq >= 2048, no split-KV: 48 queries per wave, half tiles), (2 = round 2's hand-scheduled k_attn_fat:
 * removed in round 3), 3 = k_attn_decode (Dh = 128, at most
 * 16 query rows, split-KV, non-causal = a decode step: every wave streams its own run of 32-row chunks).  Host-only query for tests and
 * profiles; the arithmetic contract of sc_attention_f16 does not depend on it. */
int sc_attention_variant(int Dh, int Sq, int nsplit);

/* Fused decode-step projection block: [q | k | v] = W . rmsnorm(x) + b, rotate-half RoPE (HF fp16 numerics) of q and k at position
 * pos[0] (device int32), q -> q_out [q_heads*Dh], k | v -> cache row pos[0] (row stride cache_ld, K at column 0, V at kv_heads*Dh).
 * One launch for the three small launches of a decode layer (q GEMV, kv GEMV, RoPE: HF Qwen2Attention q_proj / k_proj / v_proj +
 * apply_rotary_pos_emb behind reference llava_qwen.py:155); bit-identical to sc_gemv_f16 x2 + sc_rope_qk_row_f16.
 *   Wq [q_heads*Dh, K], Wkv [2*kv_heads*Dh, K] (k rows then v rows), bq / bkv biases or NULL, rms_gamma [K] or NULL (no norm). */
int sc_decode_qkv_f16(const void* Wq, const void* Wkv, const void* bq, const void* bkv, const void* x, const void* rms_gamma,
                      float rms_eps, void* q_out, void* cache, int cache_ld, const int32_t* pos, int q_heads, int kv_heads,
                      int Dh, int K, float theta, sc_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Next-token selection over fp32 LM-head logits.  Replaces what HF `generate` does after the forward pass
 * (reference llava_qwen.py:155 -> transformers GenerationMixin: argmax when do_sample=False; temperature softmax +
 * multinomial when do_sample=True — inference_streaming_longva_v2.py:252-253 temperature 0.2, utiles.py:551-552 0.1).
 *   logits [B, ld] fp32 rows of V valid entries;  out [B] int64 (device)
 *   temperature <= 0: arg-max, lowest index among equal maxima.
 *   temperature  > 0: sample from softmax(logits / temperature) by inverting the CDF at u[b] in [0, 1) (device floats the
 *                     caller draws, e.g. from a torch generator: the draw, not the kernel, carries the randomness).
 *   ws: sc_pick_token_workspace_bytes(B) bytes.  Two launches, no host synchronisation (hipGraph-capturable). */
/* Bookkeeping of one batch-1 decode step in ONE launch (round 3, ABI 3; all pointers device scalars except `ring`): ring[ring_index[0]] =
 * next_token[0]; token[0] = next_token[0]; ring_index, pos, kv_len, n_prev += 1.  Replaces six elementwise launches per replay of the
 * captured decode graph (llm.DecodeGraph). */
int sc_decode_advance(const int64_t* next_token, int64_t* ring, int64_t* ring_index, int32_t* token, int32_t* pos, int32_t* kv_len,
                      int32_t* n_prev, sc_stream_t stream);
/* v7: u[b] = uniform in [0, 1) = splitmix64(seeds[b] + (counter[0] + counter_add) * 0x9E3779B97F4A7C15) >> 40, scaled by 2^-24.  seeds [B] device
 * int64; counter: device int64 scalar or NULL (= 0).  The draw of the n-th sampled token of a sequence depends on its seed and n alone (HF draws
 * from the default generator in batch order, reference call sites inference_streaming_longva_v2.py:252-256, utiles.py:551-556: a batched
 * generate there is not the same random experiment as one-by-one generates; here it is). */
int sc_counter_uniform_f32(const int64_t* seeds, int B, const int64_t* counter, int64_t counter_add, float* u, sc_stream_t stream);
size_t sc_pick_token_workspace_bytes(int B);
int sc_pick_token_f32(const float* logits, int B, int V, int64_t ld, float temperature, const float* u, int64_t* out,
                      void* ws, size_t ws_bytes, sc_stream_t stream);

/* Next token with the logits processors / warpers HF `generate` applies when the checkpoint's generation_config.json (or the caller)
 * asks for them - RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper, in that order
 * (transformers generation/logits_process.py; reference call site inference_streaming_longva_v2.py:252-256, utiles.py:551-556) - and
 * one draw, all on the device (no host sync, hipGraph-capturable):
 *   logits [B, V] fp32, row stride ld; MODIFIED in place when repetition_penalty != 1 (x < 0 ? x * r : x / r at every DISTINCT id of
 *   prev_ids[row, 0 .. n_prev): the ids generated so far; n_prev_dev [B] device int32 or NULL -> n_prev_host for every row).
 *   temperature <= 0: arg-max of the processed logits (lowest index on ties); else a sample from softmax(logits / T) restricted to
 *   the top_k largest (0 = off; tokens equal to the k-th value stay, as in HF) and to the nucleus top_p (1 = off), by inverse CDF at u[B]
 *   over the kept tokens in index order.  top_k <= 64: candidate kernels; top_k > 64 or top_p < 1 without top_k (round 3): one block per row
 *   selects both thresholds over the full vocabulary by integer radix select (a group of equal logits is kept or dropped as a whole) and
 *   leaves, per row, {threshold logit as float bits, kept count} in ws[0 .. 2B) (uint32) for diagnostics.
 *   top_k = 0 and top_p = 1 is sc_pick_token_f32 after the penalty.  ws: sc_sample_token_workspace_bytes(B). */
size_t sc_sample_token_workspace_bytes(int B);
int sc_sample_token_f32(float* logits, int B, int V, int64_t ld, float temperature, int top_k, float top_p, float repetition_penalty,
                        const int64_t* prev_ids, int64_t prev_ld, const int32_t* n_prev_dev, int n_prev_host, const float* u,
                        int64_t* out, void* ws, size_t ws_bytes, sc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STREAMCHAT_HIP_H */
