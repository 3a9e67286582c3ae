// Decode-side attention kernels of sc_attention_f16 (attention.hip holds the tile kernel, the dispatch and the C entry points):
// the split-KV merge k_attn_combine and the per-wave streaming kernel k_attn_decode.  A translation unit of their own because
// attention.hip is built with hipcc's iterative-ILP machine scheduler (good for the MFMA-bound tile kernel: +3 % on the 49 k prefill),
// which costs these HBM-bound kernels 0.6-2 % (profiles/r03_run49_51_attn_sched_strategy.md): this file takes the default scheduler.
#include "sc_common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short sc_s4 __attribute__((ext_vector_type(4)));

// 16-byte LDS-DMA through a raw buffer resource (see attention.hip)
__device__ __forceinline__ void lds_load16(const void* base, int extent, char* lds, unsigned voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, extent, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// merge of the split-KV partials: out = sum_i O_i 2^(m_i - M) / sum_i l_i 2^(m_i - M).  One workgroup (DH threads) per
// (batch, head, query): the split weights are computed once by the first wave (one lane per split), then every thread owns one
// output dimension and sums the weighted partials with independent loads.
template <int DH>
__global__ void k_attn_combine(const float* __restrict__ part, _Float16* __restrict__ O, int ldo, int Sq, int Hq, int nsplit, int o_hs, long o_bs) {
    __shared__ float wgt[1024];
    __shared__ float inv_den;
    const int row = blockIdx.x;                       // (b*Hq + h)*Sq + q
    const int q = row % Sq, bh = row / Sq, h = bh % Hq, b = bh / Hq;
    const float* pp = part + (size_t)row * nsplit * (DH + 2);
    // NW groups of DH threads walk the partials in an interleaved order, 8 independent loads in flight each: with a single group the
    // kernel was one dependent load chain per output element (7.9 us for 128 partials: latency, not bytes).  The first batch of every
    // thread is requested BEFORE the split weights are computed (they only multiply it): the two load latencies overlap instead of adding.
    constexpr int NW = 1024 / DH < 8 ? 1024 / DH : 8;
    __shared__ float red[NW][DH];
    const int d = threadIdx.x % DH, w = threadIdx.x / DH;
    float v0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v0[u] = (w + u * NW < nsplit) ? pp[(w + u * NW) * (DH + 2) + d] : 0.f;
    if (threadIdx.x < 64) {
        float M = -INFINITY;
        for (int i = threadIdx.x; i < nsplit; i += 64) M = fmaxf(M, pp[i * (DH + 2) + DH]);
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) M = fmaxf(M, __shfl_xor(M, s, 64));
        float den = 0.f;
        for (int i = threadIdx.x; i < nsplit; i += 64) {
            const float m = pp[i * (DH + 2) + DH];
            const float w = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
            wgt[i] = w;
            den += pp[i * (DH + 2) + DH + 1] * w;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) den += __shfl_xor(den, s, 64);
        if (threadIdx.x == 0) inv_den = den > 0.f ? 1.0f / den : 0.f;
    }
    __syncthreads();
    float num = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (w + u * NW < nsplit) num += v0[u] * wgt[w + u * NW];
    int i = w + 8 * NW;
    for (; i + 7 * NW < nsplit; i += 8 * NW) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(i + u * NW) * (DH + 2) + d];
#pragma unroll
        for (int u = 0; u < 8; ++u) num += v[u] * wgt[i + u * NW];
    }
    for (; i < nsplit; i += NW) num += pp[i * (DH + 2) + d] * wgt[i];
    red[w][d] = num;
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int u = 1; u < NW; ++u) num += red[u][d];
        O[(size_t)b * (size_t)o_bs + (size_t)q * (size_t)ldo + h * o_hs + d] = (_Float16)(num * inv_den);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decode attention (few query rows against a long KV cache; batch-1 decode packs the G query heads of a KV group as G <= 16 query
// rows of one "head", llm.py _decode_one / DecodeGraph): HBM-bound, 2.8 GB of K/V per token at a 49 k context.  Every WAVE is an
// independent stream over its own contiguous run of 32-row chunks - no block barrier in the loop.
//
// Round 5: BOTH operands reach the wave by LDS-DMA (buffer_load ... lds) into per-wave rings of DEC_PD stages, every wait is a
// hand-counted s_waitcnt vmcnt and every LDS read is inline asm.  The round-3 kernel loaded K straight into MFMA operand registers
// with the buffer-load builtin next to the V DMA: hipcc cannot count a register load's position in a queue that also holds LDS-DMA
// ops across a loop back-edge, so it put `s_waitcnt vmcnt(0)` in front of the first S MFMA of every trip - the whole prefetch queue
// drained once per two chunks (22.9 us per layer at 49 k = 4.4 TB/s, the same time as the one-tile-in-flight tile kernel, which is
// what gave it away; cdna guide 5.x "mixing load KINDS in one k-loop mis-waits").
//   K: rows of 256 B land in LDS row-major with the 16-byte piece index XORed with (row & 15) on the SOURCE side (an LDS-DMA image is
//      lane-linear), read back as MFMA A fragments with ds_read_b128 - conflict-free (a 16-lane service group holds 16 distinct rows);
//      global side: every DMA instruction fetches 4 whole rows = 8 full 128-B lines (the register path fetched 16 rows x 64 B);
//   V: as before, swizzled rows read with ds_read_b64_tr_b16 (P.V needs V transposed);
//   queue order per chunk c: K(c+PD) is requested as soon as the S MFMAs have read K(c)'s fragments, V(c+PD) as soon as the transpose
//   reads of V(c) have returned, so each ring needs only DEC_PD stages: 2 x 16 KiB per wave, 128 KiB per workgroup;
//   rows past the wave's range are outside the buffer resource's extent: zeros, no memory traffic (the row part of every address
//   stays in the VGPR offset, which is what the range check covers).
// The MFMA operands are element for element those of the round-3 kernel: results are bit-identical to it.
// Exact online softmax per chunk (the loop is bandwidth-bound, the VALU work is free).  The four waves of a workgroup cover four
// consecutive quarters of one split; they merge their (O, m, l) through LDS at the end, so a workgroup leaves ONE partial per split
// in the layout k_attn_combine reads.
// ------------------------------------------------------------------------------------------------------------------
constexpr int DCH = 32;                            // kv rows per chunk
// Diagnostic build only (-DSC_DEC_TRACE, tools/trace_decode_attn.py): five 100-MHz wall-clock stamps per wave of the LAST launch
#ifdef SC_DEC_TRACE
__device__ unsigned long long sc_dec_trace_buf[8192 * 8];
#define DEC_STAMP(i) do { if (lane == 0) sc_dec_trace_buf[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 2048 * (DEC_NW * 8) + wave * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DEC_STAMP(i) do { } while (0)
#endif
#ifndef SC_DEC_ABL
#define SC_DEC_ABL 0
#endif
#ifndef SC_DEC_NW
#define SC_DEC_NW 4                                // streaming waves per workgroup (a power of two)
#endif
constexpr int DEC_NW = SC_DEC_NW, DEC_NW_LOG = DEC_NW == 8 ? 3 : (DEC_NW == 4 ? 2 : (DEC_NW == 2 ? 1 : 0));
static_assert((1 << DEC_NW_LOG) == DEC_NW, "1, 2, 4 or 8 waves");

// DEC_PD = chunks in flight per wave = stages of the K and of the V ring: 2 (128 KiB of LDS, one workgroup per CU) or 1 (64 KiB, two
// workgroups per CU: twice the waves, half the run-ahead each)
template <int DH, int DEC_PD>
__global__ __launch_bounds__(64 * DEC_NW) void k_attn_decode(const _Float16* __restrict__ Q, int ldq, const _Float16* __restrict__ Kp, int ldk,
                                                        const _Float16* __restrict__ Vp, int ldv, int Sq, int Skv, int Hq, int Hkv, float scale_log2,
                                                        const int* __restrict__ kv_len, float* __restrict__ part, int nsplit, int q_hs, long q_bs,
                                                        int group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(DH == 128, "the K ring is laid out for 256-byte rows");
    static_assert(DEC_PD >= 1 && 16 * DEC_PD - 8 < 64, "vmcnt is a 6-bit counter");
    constexpr int DS = DH / 32, DB = DH / 16, VROW = DH * 2, CHB = DCH * VROW;          // bytes of one K or V chunk (8 KiB at Dh = 128)
    constexpr int WAVE_LDS = 2 * DEC_PD * CHB;                                          // [K stages | V stages] of one wave
    constexpr int CNT = 16 * DEC_PD - 8;                                                // vm ops younger than the chunk part being waited for (see the loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 15, g = lane >> 4;
    // grid = (split, head, batch): no integer division in the prologue (round 4 decoded a flat block index with three of them, ~60
    // scalar instructions in front of the first memory request of a 22 us kernel)
#ifdef SC_DEC_HROT          // diagnostic: which head a grid row works on (does a slow row follow the grid position or the head's addresses?)
    const int split = blockIdx.x, h = (blockIdx.y + SC_DEC_HROT) % gridDim.y, b = blockIdx.z;
#else
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
#endif
    DEC_STAMP(0);
#ifdef SC_DEC_TRACE
    {   // where this wave runs: HW_ID (cu_id [11:8], sh_id [12], se_id [15:13]) and the XCC id
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) sc_dec_trace_buf[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 2048 * (DEC_NW * 8) + wave * 8 + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int hk = group == 1 ? h : h / group;                                          // group = Hq / Hkv (1 when the caller packs the G heads as query rows)
    const int kv_valid = kv_len ? min(kv_len[b], Skv) : Skv;
    // chunks of this split, then of this wave: EVEN shares (sizes differ by at most one chunk).  Round 4 cut ceil(nch / nsplit) chunks per
    // split and ceil(that / 4) per wave: at 49 153 keys (1537 chunks, 64 splits) that is 25 chunks for 61 workgroups, 12 for one and none
    // for two, and 7 | 7 | 7 | 4 inside a workgroup - the launch lasted 7 chunk times where 6.004 would do
#ifdef SC_DEC_OLD_PARTITION
    const int cps = ((kv_valid + DCH - 1) / DCH + nsplit - 1) / nsplit, nch = (kv_valid + DCH - 1) / DCH;
    const int s_lo = min(split * cps, nch), s_hi = min(s_lo + cps, nch);
    const int cpw = (s_hi - s_lo + DEC_NW - 1) >> DEC_NW_LOG;
    const int c_lo = min(s_lo + wave * cpw, s_hi), c_hi = min(c_lo + cpw, s_hi);
#else
    const int nch = (kv_valid + DCH - 1) / DCH;
    const int cq = nch / nsplit, cr = nch - cq * nsplit;                                // the first cr splits take cq + 1 chunks
    const int s_lo = split * cq + min(split, cr), ns = cq + (split < cr ? 1 : 0);
    const int c_lo = s_lo + ((ns * wave) >> DEC_NW_LOG), c_hi = s_lo + ((ns * (wave + 1)) >> DEC_NW_LOG);
#endif
    const int row_end = min(c_hi * DCH, kv_valid);                                      // rows of this wave: [c_lo * 32, row_end)

    // Q fragments (B operand: column = query row rl, k-slots = 8 head-dim elements): inline asm, so that no compiler-tracked vector
    // load is pending when the DMA queue starts (see the header); waited for together with the first K chunk
    sc_u4 qf[DS];
    {
        const int qr = rl < Sq ? rl : Sq - 1;
        const _Float16* qp = Q + (size_t)b * (size_t)q_bs + (size_t)qr * (size_t)ldq + h * q_hs + g * 8;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(qf[ds]) : "v"(qp), "n"(ds * 64) : "memory");
    }
    const _Float16* kbase = Kp + (size_t)b * Skv * (size_t)ldk + hk * DH;
    const _Float16* vbase = Vp + (size_t)b * Skv * (size_t)ldv + hk * DH;
    const bool any = c_lo < c_hi;
    const int k_ext = (any && row_end > 0) ? (int)(((unsigned)(row_end - 1) * (unsigned)ldk + DH) * 2u) : 0;
    const int v_ext = (any && row_end > 0) ? (int)(((unsigned)(row_end - 1) * (unsigned)ldv + DH) * 2u) : 0;
    // per-lane source offsets inside a chunk.  DMA granule j * 64 + lane lands at LDS byte (j * 64 + lane) * 16 = row j * 4 + (lane >> 4),
    // 16-byte slot lane & 15; it fetches the piece (lane & 15) ^ (row & 15) of that K row / (lane & 15) ^ ((row & 7) << 1) of that V row
    unsigned k_vo[8], v_vo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = j * 4 + (lane >> 4);
        k_vo[j] = ((unsigned)r * (unsigned)ldk + (unsigned)(((lane & 15) ^ (r & 15)) * 8)) * 2u;
        v_vo[j] = ((unsigned)r * (unsigned)ldv + (unsigned)(((lane & 15) ^ ((r & 7) << 1)) * 8)) * 2u;
    }
    char* kring = smem + wave * WAVE_LDS;
    char* vring = kring + DEC_PD * CHB;
    const unsigned kbase_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)(wave * WAVE_LDS);
    const unsigned vbase_lds = kbase_lds + (unsigned)(DEC_PD * CHB);
    // read offsets: K fragment (kvb, ds) of lane (rl, g) = row kvb * 16 + rl, piece ds * 4 + g -> slot piece ^ rl (kvb: +4096 bytes)
    unsigned k_off[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) k_off[ds] = (unsigned)(rl * VROW + (((ds * 4 + g) ^ rl) << 4));
    const int vrow = 4 * g + (rl >> 2);
    unsigned v_off[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) v_off[db] = (unsigned)(vrow * VROW + ((db ^ (vrow & 7)) << 5) + (rl & 3) * 8);

    auto issue_k = [&](int c) {                                  // 8 LDS-DMA ops
        const unsigned ro = (unsigned)c * (unsigned)(DCH * ldk * 2);
        char* dst = kring + (c % DEC_PD) * CHB;
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_load16(kbase, c < c_hi ? k_ext : 0, dst + j * 1024, ro + k_vo[j], 0);
    };
    auto issue_v = [&](int c) {                                  // 8 LDS-DMA ops
        const unsigned ro = (unsigned)c * (unsigned)(DCH * ldv * 2);
        char* dst = vring + (c % DEC_PD) * CHB;
#pragma unroll
        for (int j = 0; j < 8; ++j) lds_load16(vbase, c < c_hi ? v_ext : 0, dst + j * 1024, ro + v_vo[j], 0);
    };

    sc_f4 o[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) o[i] = sc_f4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // The vm queue of a wave, in issue order:  [q x DS]  K(c0) V(c0) ... K(c0+PD-1) V(c0+PD-1)   then per chunk c:  K(c+PD)  V(c+PD).
    // When S(c) starts, the ops younger than K(c) are V(c), K/V(c+1 .. c+PD-1) = 16 PD - 8; when P.V(c) starts, the ops younger than
    // V(c) are K/V(c+1 .. c+PD-1) and K(c+PD) = 16 PD - 8 as well: one constant serves both waits.
    if (any) {
#pragma unroll
        for (int u = 0; u < DEC_PD; ++u) { issue_k(c_lo + u); issue_v(c_lo + u); }          // (past c_hi: zero extent -> no traffic)
        DEC_STAMP(1);
        for (int c = c_lo; c < c_hi; ++c) {
            const unsigned stage = (unsigned)((c % DEC_PD) * CHB);
#if SC_DEC_ABL == 1                 // ablation: the stream alone (waits + re-issue, no LDS reads, no arithmetic)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
            issue_k(c + DEC_PD);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
            issue_v(c + DEC_PD);
            continue;
#endif
            // ---- S^T = K . Q^T for the 32 rows ----
            static_assert(DS == 4, "the fences below name 8 K fragments and 4 q fragments");
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]) : "n"(CNT) : "memory");
            sc_u4 kf[2][DS];
            const unsigned ka = kbase_lds + stage;
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0][ds]) : "v"(ka + k_off[ds]));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[1][ds]) : "v"(ka + k_off[ds]), "n"(16 * VROW));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0][0]), "+v"(kf[0][1]), "+v"(kf[0][2]), "+v"(kf[0][3]),
                                                  "+v"(kf[1][0]), "+v"(kf[1][1]), "+v"(kf[1][2]), "+v"(kf[1][3]) :: "memory");
            if (c == c_lo) DEC_STAMP(2);
            issue_k(c + DEC_PD);                                   // K(c)'s stage is free: its fragments are in registers
            sc_f4 s[2];
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb) {
                s[kvb] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ds = 0; ds < DS; ++ds)
                    s[kvb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_h8, kf[kvb][ds]), __builtin_bit_cast(sc_h8, qf[ds]), s[kvb], 0, 0, 0);
            }
            // ---- exact online-softmax step ----
            const int kv0 = c * DCH + g * 4;
            float tmax = -INFINITY;
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kvb][r] = (kv0 + kvb * 16 + r < row_end) ? s[kvb][r] : -INFINITY;
                    tmax = fmaxf(tmax, s[kvb][r]);
                }
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto bsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(bsw[0]), __uint_as_float(bsw[1])) * scale_log2;
            const float m_new = fmaxf(m_run, tmax);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db] *= alpha;
            sc_h8 pf;
            float ps = 0.f;
#pragma unroll
            for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kvb][r], scale_log2, -m_use));
                    ps += p;
                    pf[kvb * 4 + r] = (_Float16)p;
                }
            l_run += ps;
            // ---- O^T += V^T . P^T : V(c) must have landed ----
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
            const unsigned sa = vbase_lds + stage;
            sc_s4 lo[DB], hi[DB];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[db]) : "v"(sa + v_off[db]));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[db]) : "v"(sa + v_off[db]), "n"(16 * VROW));
            }
            static_assert(DB == 8, "the lgkmcnt fence below names 16 registers");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]),
                                                  "+v"(lo[4]), "+v"(hi[4]), "+v"(lo[5]), "+v"(hi[5]), "+v"(lo[6]), "+v"(hi[6]), "+v"(lo[7]), "+v"(hi[7]) :: "memory");
            issue_v(c + DEC_PD);                                   // V(c)'s stage is free
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                typedef short sc_s8 __attribute__((ext_vector_type(8)));
                const sc_s8 v8 = {lo[db][0], lo[db][1], lo[db][2], lo[db][3], hi[db][0], hi[db][1], hi[db][2], hi[db][3]};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_h8, v8), pf, o[db], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no LDS-DMA may still be in flight when the rings are reused / the block ends
    DEC_STAMP(3);
    // ---- merge the four waves of the split through LDS (aliases the rings: every wave is done with its own) ----
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    __syncthreads();
    float* mo = reinterpret_cast<float*>(smem);             // [NW][DH][16] O^T, then [NW][16] m, [NW][16] l
    float* mm = mo + DEC_NW * DH * 16;
    float* ml = mm + DEC_NW * 16;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) mo[(wave * DH + db * 16 + g * 4 + r) * 16 + rl] = o[db][r];
    if (g == 0) { mm[wave * 16 + rl] = m_run; ml[wave * 16 + rl] = l_run; }
    __syncthreads();
    for (int e = tid; e < Sq * DH; e += 64 * DEC_NW) {
        const int q = e / DH, d = e - q * DH;
        float M = mm[q];
#pragma unroll
        for (int w = 1; w < DEC_NW; ++w) M = fmaxf(M, mm[w * 16 + q]);
        float acc = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < DEC_NW; ++w) {
            const float mw = mm[w * 16 + q];
            const float wt = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - M);
            acc += mo[(w * DH + d) * 16 + q] * wt;
            l += ml[w * 16 + q] * wt;
        }
        float* pp = part + ((((size_t)b * Hq + h) * Sq + q) * nsplit + split) * (DH + 2);
        pp[d] = acc;
        if (d == 0) { pp[DH] = M; pp[DH + 1] = l; }
    }
    DEC_STAMP(4);
}

}  // namespace

const char* sc_decode_build_tag() {
#ifdef SC_KERNARG_PRELOAD
    return "decode=kernarg-preload";
#else
    return "decode=plain";
#endif
}

#ifdef SC_DEC_TRACE
extern "C" int sc_dec_trace_read(void* dst, int bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sc_dec_trace_buf), bytes); }
#endif

// launchers used by attention.hip
void sc_attn_combine_launch(int Dh, const float* part, void* out, int ldo, int B, int Sq, int Hq, int nsplit, int o_hs, long o_bs, hipStream_t s) {
    const dim3 grid((unsigned)(B * Hq * Sq));
    if (Dh == 128) hipLaunchKernelGGL((k_attn_combine<128>), grid, dim3(1024), 0, s, part, (_Float16*)out, ldo, Sq, Hq, nsplit, o_hs, o_bs);
    else if (Dh == 64) hipLaunchKernelGGL((k_attn_combine<64>), grid, dim3(512), 0, s, part, (_Float16*)out, ldo, Sq, Hq, nsplit, o_hs, o_bs);
    else hipLaunchKernelGGL((k_attn_combine<32>), grid, dim3(256), 0, s, part, (_Float16*)out, ldo, Sq, Hq, nsplit, o_hs, o_bs);
}

void sc_attn_decode_launch(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Sq, int Skv, int Hq, int Hkv, float scale_log2,
                           const int32_t* kv_len, float* part, int nsplit, int q_hs, long q_bs, hipStream_t s) {
    // SC_DEC_PD=1|2 pins the ring depth (A/B runs).  Default ONE stage per ring (64 KiB per workgroup): measured on the same box at a 49 k
    // context 326.2 tokens/s against 323.7 with two stages (64 splits; 325.6 / 315.5 at 128), and 11.53 against 11.47 ms on the 26-sequence
    // caption step - the run-ahead of a second stage buys nothing once the stream is all-DMA (profiles/r05_run_i_*)
    static std::atomic<int> pd_env{-1};
    if (pd_env < 0) { const char* e = getenv("SC_DEC_PD"); pd_env = e ? atoi(e) : 0; }
    const int pd = (pd_env == 1 || pd_env == 2) ? pd_env.load() : 1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto go = [&](auto pdc) {
        constexpr int PD = decltype(pdc)::value;
        constexpr int LDS_RING = DEC_NW * 2 * PD * DCH * 128 * 2, LDS_MERGE = DEC_NW * (128 * 16 + 32) * 4;   // NW waves x (K ring + V ring) x PD stages of 8 KiB
        constexpr int LDS_DEC = LDS_RING > LDS_MERGE ? LDS_RING : LDS_MERGE;
        static std::atomic<bool> attr_done[16];
        if (!attr_done[dev & 15]) { (void)hipFuncSetAttribute((const void*)k_attn_decode<128, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DEC); attr_done[dev & 15] = true; }
        hipLaunchKernelGGL((k_attn_decode<128, PD>), dim3((unsigned)nsplit, (unsigned)Hq, (unsigned)B), dim3(64 * DEC_NW), LDS_DEC, s, (const _Float16*)q, ldq, (const _Float16*)k, ldk,
                           (const _Float16*)v, ldv, Sq, Skv, Hq, Hkv, scale_log2, kv_len, part, nsplit, q_hs, q_bs, Hq / Hkv);
    };
    if (pd == 1) go(std::integral_constant<int, 1>{}); else go(std::integral_constant<int, 2>{});
}
