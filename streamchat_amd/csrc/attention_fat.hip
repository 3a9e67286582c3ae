// Long-prefill attention for gfx950 with a HAND-SCHEDULED steady-state loop: Dh = 128, causal / GQA, fp16 in/out (the Qwen2-7B prefill
// over 26 k - 49 k stacked frame tokens, reference llava_qwen.py:155 -> HF Qwen2 attention).  Same mathematics as k_attn<128> in
// attention.hip (transposed orientation: S^T = K.Q^T, O^T += V^T.P^T, one lane owns one query column, softmax statistics lane-local,
// fixed softmax reference in the steady state); what differs is the MFMA shape and how the loop is issued.
//
// Why (PMC + ablations, profiles/r02_run27..29): v_mfma_f32_16x16x32_f16 holds the SIMD's VALU issue port for 8 of its 16 pipe
// cycles, and the softmax needs 0.75 plain VALU (4.5 cycles) + 0.5 v_exp (9 cycles) per such MFMA: 8 + 3.4 + 4.5 = 15.9 port cycles
// per 16 pipe cycles - the VALU port alone is saturated, and everything else a wave issues (LDS reads, waits, DMA) comes on top.
// A first hand-scheduled version on 16x16x32 (tools/wip/attention_fat16x16_wip.hip.txt) confirmed it: MFMA-only 2.36 PF-equivalent,
// everything-but-MFMA 2.95, both together 1.24 = no overlap at all.  v_mfma_f32_32x32x16_f16 does twice the work per issue:
//   * ONE wave per SIMD with the whole 512-register file: 64 queries per wave (2 q-blocks of 32), 256 per workgroup; O^T (128
//     registers) and the Q fragments (64) live in AGPRs; every K / V fragment read from LDS (1 KiB) feeds TWO 32x32x16 MFMAs = 64
//     pipe cycles (the compiler-scheduled 32x32 experiment of round 2 had 32 queries per wave at three waves per SIMD: one MFMA per
//     fragment, LDS-bound);
//   * the KV loop is software-pipelined over 32-row chunks by q-block: step (c, q) issues, in a FIXED order of volatile asm
//     statements, the 8 P.V MFMAs of the previous q-block, the 8 S MFMAs of chunk c+1 for q-block q, and in their shadow the 40
//     VALU instructions of the softmax of (c, q): 8 v_pk_fma_f32, 16 v_exp_f32, 8 v_cvt_pk_f16_f32, 8 v_dot2_f32_f16 (row sums):
//     per MFMA slot (32 pipe cycles, 8 of them issue) at most two v_exp + one plain VALU = 22.5 port cycles;
//   * fragments are single-buffered and refilled right after their last use, 16 MFMA slots (~512 cycles) before the next one;
//   * K / V tiles (64 rows) go through a 4-slot LDS ring (128 KiB) filled by inline-asm `buffer_load ... lds` (two per step, tile
//     t+3 during tile t), retired with ONE counted `s_waitcnt vmcnt(8)` + ONE `s_barrier` per tile.
// Softmax reference: fixed at the row max of the first 32 keys (as the steady-state loop of k_attn: no running max, no O
// rescale); masks (causal diagonal, ragged end) are applied to P in the tiles that need them by a second instantiation of the
// loop body.  A score more than 2^16 above its reference makes O / l non-finite: the workgroup then redoes its rows with an
// exact online-softmax pass (compiler-scheduled 16x16x32 code, slow, never taken on real activations; tested).
//
// Fragment maps (32x32x16: A row / B column = lane & 31, k-slots 8 * (lane >> 5) + j; C: column = lane & 31, rows (r & 3) + 8 * (r >> 2)
// + 4 * (lane >> 5)): the S^T registers 0..7 / 8..15 of a lane are exactly the eight k-slots of P.V k-step 0 / 1 when V's rows are
// presented as k-slot (g, j) = row 16 * s + (j < 4 ? 4 * g + j : 8 + 4 * g + j - 4), which two ds_read_b64_tr_b16 deliver.
//
// Hazards the hardware does not interlock and the compiler cannot see inside asm: (1) VALU write -> MFMA read needs two wait
// states: the last v_cvt_pk of a step is followed by >= 2 instructions before the first P.V MFMA of the next step; (2) v_exp ->
// dependent VALU needs one instruction in between: guaranteed by the slot order below; (3) MFMA write -> VALU read: S is read 16
// MFMAs (512 cycles) after it was written, O after an explicit drain.
#include "sc_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short sc_s4 __attribute__((ext_vector_type(4)));
typedef short sc_s8 __attribute__((ext_vector_type(8)));
typedef float sc_f2 __attribute__((ext_vector_type(2)));
typedef float sc_f16x __attribute__((ext_vector_type(16)));
typedef unsigned sc_rsrc4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) sc_s4* lds_s4_ptr;

#ifndef FAT_ABL
#define FAT_ABL 0          // timing ablations (tools/build_fat_variants.sh): 1 no DMA, 2 no wait + barrier, 4 no LDS reads, 8 no softmax VALU, 16 no P.V MFMAs, 32 no S MFMAs
#endif
constexpr int F_KVT = 64, F_DH = 128;
constexpr int F_TILE = F_KVT * F_DH * 2;          // 16 KiB per K (or V) tile
constexpr int F_STAGE = 2 * F_TILE;               // K | V
constexpr int F_RING = 4;
constexpr int F_LDS = F_RING * F_STAGE;           // 128 KiB

__device__ __forceinline__ sc_rsrc4 fat_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    return sc_rsrc4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
                    (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// 16-byte LDS-DMA from inline asm (the compiler must not know that LDS is written: with the builtin it drains the whole DMA queue
// in front of every ds_read that may alias).  M0 = wave-uniform LDS byte address; s_nop: SALU -> M0 / SGPR -> buffer wait states.
__device__ __forceinline__ void fat_dma16(sc_rsrc4 rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 1) void k_attn_fat(const _Float16* __restrict__ Q, int ldq, const _Float16* __restrict__ Kp, int ldk,
                                                     const _Float16* __restrict__ Vp, int ldv, _Float16* __restrict__ O, int ldo, int Sq, int Skv,
                                                     int Hq, int Hkv, float scale_log2, const int* __restrict__ kv_len, int B, int q_hs, int o_hs,
                                                     long q_bs, long o_bs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hh = lane >> 5;
    // block -> (batch, kv head, query head, 256-query block): the per-XCD queues of k_attn (the G query heads of a KV group run
    // back to back on one XCD and share its L2 copy of the K / V stream), q-blocks longest-first
    const int nqb = (Sq + 255) / 256, G = Hq / Hkv;
    const int bid = blockIdx.x, xcd = bid & 7, jq = bid >> 3;
    const int pair = (jq / G) * 8 + xcd;
    if (pair >= nqb * Hkv * B) return;
    const int hk = pair % Hkv, h = hk * G + jq % G;
    const int qi = (pair / Hkv) % nqb, b = pair / (Hkv * nqb);
    const int qblk0 = (nqb - 1 - qi) * 256;
    const int qw0 = qblk0 + wave * 64;
    const int kv_valid = kv_len ? min(kv_len[b], Skv) : Skv;
    const int coff = Skv - Sq;
    int nt = (kv_valid + F_KVT - 1) / F_KVT;
    int n_full = kv_valid / F_KVT;                                    // tiles whose 64 keys are valid and visible to all 256 queries
    if (CAUSAL) {
        const int last_q = min(qblk0 + 256, Sq) - 1;
        nt = min(nt, (last_q + coff) / F_KVT + 1);
        n_full = min(n_full, (qblk0 + coff + 1) / F_KVT);
    }
    n_full = min(n_full, nt);
    const int nt_all = (kv_valid + F_KVT - 1) / F_KVT;

    _Float16* const obase = O + (size_t)b * (size_t)o_bs + h * o_hs;
    if (nt == 0) {                                                    // no visible key at all: zeros (k_attn's convention)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int qr = qw0 + q * 32 + ql;
            if (qr < Sq)
#pragma unroll
                for (int i = 0; i < 16; ++i) *reinterpret_cast<sc_h4*>(obase + (size_t)qr * (size_t)ldo + hh * 64 + i * 4) = sc_h4{0, 0, 0, 0};
        }
        return;
    }

    // ---- Q^T fragments: lane (q, hh) holds Q[q][16 * ks + 8 * hh .. +7] (B operand of the S MFMAs; only ever "a" operands) ----
    sc_h8 qf[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int qr = qw0 + q * 32 + ql;
        qr = qr < Sq ? qr : Sq - 1;
        const _Float16* qp = Q + (size_t)b * (size_t)q_bs + (size_t)qr * (size_t)ldq + h * q_hs + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[q][ks] = *reinterpret_cast<const sc_h8*>(qp + ks * 16);
    }

    // ---- staging: K granule (row r, 16-B column c) at r * 256 + ((c ^ (r & 15)) << 4), V at r * 256 + ((c ^ ((r & 3) << 2)) << 4)
    //      (the 8 (row, 32-B block) pairs of a transpose read's 32-lane service group fall on 8 distinct bank blocks);
    //      lane-linear DMA destination, swizzle on the source offset ----
    const _Float16* kbase = Kp + (size_t)b * Skv * (size_t)ldk + hk * F_DH;
    const _Float16* vbase = Vp + (size_t)b * Skv * (size_t)ldv + hk * F_DH;
    unsigned k_lo[4], v_lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qg = j * 256 + tid, r = qg >> 4;
        k_lo[j] = ((unsigned)r * (unsigned)ldk + (unsigned)(((qg & 15) ^ (r & 15)) * 8)) * 2u;
        v_lo[j] = ((unsigned)r * (unsigned)ldv + (unsigned)(((qg & 15) ^ ((r & 3) << 2)) * 8)) * 2u;
    }
    // extent = the valid rows only (rows past kv_valid of a cache hold allocator garbage; past the extent the DMA writes zeros)
    const unsigned k_bytes = ((unsigned)(kv_valid - 1) * (unsigned)ldk + F_DH) * 2u, v_bytes = ((unsigned)(kv_valid - 1) * (unsigned)ldv + F_DH) * 2u;
    const sc_rsrc4 k_rs = fat_rsrc(kbase, k_bytes), v_rs = fat_rsrc(vbase, v_bytes);
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    // DMA instruction e (0..3: K granule groups, 4..7: V) of tile u; tiles past the end fetch from offset = extent (zeros)
    auto dma = [&](int u, int e) {
        const int jj = e & 3;
        const unsigned dst = lds0 + (unsigned)((u & 3) * F_STAGE + (e >> 2) * F_TILE + (jj * 256 + wave * 64) * 16);
        const bool in = u < nt_all;
        if (e < 4) fat_dma16(k_rs, dst, k_lo[jj], in ? (unsigned)(u * F_KVT * ldk * 2) : k_bytes);
        else fat_dma16(v_rs, dst, v_lo[jj], in ? (unsigned)(u * F_KVT * ldv * 2) : v_bytes);
    };

    // ---- fragment read offsets inside a stage ----
    // K: row ql of a 32-row chunk, 16-B column 2 * ks + hh at (column ^ (row & 15)): offset(ks) = kA ^ (ks << 5)
    const int kA = ql * 256 + ((((ql & 15) & 14) | (hh ^ (ql & 1))) << 4);
    // V transpose read: lane group Gq = lane >> 4: d-half = Gq & 1, k-group g = Gq >> 1; the lane supplies row 4 * g + (t16 >> 2) (+8 for
    // the second read, +16 per k-step) and the 8-byte piece (t16 & 3) of the 16 columns [32 * db + 16 * d-half, +16)
    const int t16 = lane & 15, Gq = lane >> 4;
    const int vr0 = 4 * (Gq >> 1) + (t16 >> 2);
    int v_off[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) v_off[db] = F_TILE + vr0 * 256 + (((2 * db + (Gq & 1)) ^ ((vr0 & 3) << 1)) << 5) + (t16 & 3) * 8;   // rows +8 / +16: same (r & 3)

    sc_f16x o[4][2];                              // O^T accumulators [d-block of 32][q-block]: AGPRs, only touched by asm MFMAs until the epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][j][r] = 0.f;
    sc_f16x s[2][2];                              // S^T [parity of the chunk][q-block]: chunk c is read from s[c & 1] while chunk c+1 accumulates in the other
    sc_h8 kf[8], vf[8];                           // K fragments [d k-step], V^T fragments [kv k-step * 4 + d-block]
    sc_u4 pf[2][2][2];                            // P^T as MFMA B operand [parity of the chunk][q-block][kv k-step]
    float l[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // row sums: four partial accumulators per q-block (no dependent chain of adds)
    float nm[2];                                  // -(reference max) in the scaled log2 domain
    int qlim[2];                                  // last visible kv of the lane's query (minus its row offset 4 * hh)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int qpos = CAUSAL ? min(qw0 + q * 32 + ql + coff, kv_valid - 1) : kv_valid - 1;
        qlim[q] = qpos - hh * 4;
    }
    const float fzero = 0.f;

#define FAT_PV(I, QQ, PAR) do { if (!(FAT_ABL & 16)) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o[(I) & 3][QQ]) : "v"(vf[I]), "v"(pf[PAR][QQ][(I) >> 2])); } while (0)
// The last P.V MFMA of a loop pass / of the kernel, followed - when DRAIN != 0 - by ~130 cycles of s_nop INSIDE the same asm statement:
// behind the loop hipcc moves accumulators between register tuples (v_accvgpr_mov / v_mov of O and S), and it cannot know that the MFMAs
// of the last three slots are still in flight (found with 64 keys: one q-block lost the last chunk's P.V; at 2 k+ keys the error hid
// under the tolerance).  A separate drain statement would not do: the moves may be scheduled in front of it.
#define FAT_PV_DRAIN(I, QQ, PAR, DRAIN) do { if (!(FAT_ABL & 16)) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lfat_nd%=\n\t" \
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n.Lfat_nd%=:" \
        : "+a"(o[(I) & 3][QQ]) : "v"(vf[I]), "v"(pf[PAR][QQ][(I) >> 2]), "s"(DRAIN) : "scc"); } while (0)
#define FAT_S(PAR, Q_, I) do { if (FAT_ABL & 32) break; if ((I) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(s[PAR][Q_]) : "v"(kf[I]), "a"(qf[Q_][I])); \
                               else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s[PAR][Q_]) : "v"(kf[I]), "a"(qf[Q_][I])); } while (0)
#define FAT_FENCE asm volatile("" ::: "memory")
    auto rd_k = [&](int ks, const char* src) {    // src = first row of the 32-row chunk in the K half of a stage
        if (FAT_ABL & (4 | 1024)) return;
        FAT_FENCE;
        kf[ks] = *reinterpret_cast<const sc_h8*>(src + (kA ^ (ks << 5)));
        FAT_FENCE;
    };
    auto rd_v = [&](int i, const char* src) {     // i = kv k-step * 4 + d-block
        if (FAT_ABL & (4 | 2048)) return;
        FAT_FENCE;
        const char* vp = src + (i >> 2) * 16 * 256 + v_off[i & 3];
        const sc_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vp));
        const sc_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vp + 8 * 256));
        const sc_s8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        vf[i] = __builtin_bit_cast(sc_h8, v8);
        FAT_FENCE;
    };

    // ---- prologue: tiles 0..2 in flight, tile 0 landed ----
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) dma(u, e);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // fill: S of chunk 0 for both q-blocks (interleaved: a dependent MFMA must not follow its producer closely, see the loop),
    // reference = its row max over the visible keys; then the K fragments of chunk 1
#pragma unroll
    for (int i = 0; i < 8; ++i) rd_k(i, smem);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        FAT_S(0, 0, i);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        FAT_S(0, 1, i);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s[0][0]), "+v"(s[0][1]) : : "memory");     // asm MFMA results are read by VALU below (the statement owns them: no read can be scheduled in front of it)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, ((r & 3) + 8 * (r >> 2) > qlim[q]) ? -INFINITY : s[0][q][r]);
        const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
        nm[q] = tmax > -INFINITY ? -tmax * scale_log2 : 0.f;               // key 0 is visible to every query (coff >= 0, kv_valid >= 1)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) rd_k(i, smem + 32 * 256);
#pragma unroll
    for (int i = 0; i < 8; ++i) vf[i] = sc_h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pf[1][q][ks] = sc_u4{0u, 0u, 0u, 0u};

    // ---- one iteration = chunk c = 2t + H_: three mutually independent streams in one fixed issue order:
    //        S(c+1) -> s[H_ ^ 1]   (16 MFMAs, K fragments of chunk c+1; each fragment is then refilled with chunk c+2)
    //        P.V(c-1) -> O         (16 MFMAs, V fragments of chunk c-1 and pf[H_ ^ 1]; each fragment is then refilled with chunk c)
    //        softmax(c): s[H_] -> pf[H_], l   (80 VALU: q-block 0 in the first 16 MFMA slots, q-block 1 in the last 16)
    //      MFMA order S(q0,j) PV(q0,j) S(q1,j) PV(q1,j): a dependent MFMA (the S accumulation over j) follows its producer three
    //      MFMAs later - back to back the result is not forwarded and nothing interlocks inside asm. ----
    auto iter = [&](auto Hc, auto Mc, int t, const char* vsrc, const char* ksrc, int last) {
        constexpr int H_ = decltype(Hc)::value, HN = H_ ^ 1;
        constexpr bool MASK = decltype(Mc)::value;
        float x[2][16], e[2][16];
        int lim[2] = {0, 0};
        if (MASK) { lim[0] = qlim[0] - (t * F_KVT + H_ * 32); lim[1] = qlim[1] - (t * F_KVT + H_ * 32); }   // S register r (row (r & 3) + 8 * (r >> 2) + 4 * hh) is dead iff its row > lim
        // MFMA slot K: j = K >> 2; K & 3 = 0: S(q0, j) | 1: S(q1, j) | 2: P.V(q0, j) | 3: P.V(q1, j).  A fragment is refilled ONE MFMA after
        // its last use (K fragment j behind slot 4j+2, V fragment j behind slot 4j+4): a ds_read whose destination is a source of the
        // MFMA issued just before it waits until that MFMA has read its operands - ~30-70 cycles with the matrix pipe idle behind it
        // (ablation: the 16 + 32 fragment reads of a tile cost 38 % of the loop when placed right behind their last use)
#define FAT_M(K) do { if (((K) & 3) == 0) { FAT_S(HN, 0, (K) >> 2); if ((K) >= 4) rd_v(((K) >> 2) - 1, vsrc); } else if (((K) & 3) == 1) FAT_S(HN, 1, (K) >> 2); \
                      else if (((K) & 3) == 2) { FAT_PV((K) >> 2, 0, HN); rd_k((K) >> 2, ksrc); } \
                      else if ((K) == 31 && H_ == 1) FAT_PV_DRAIN(7, 1, HN, last); else FAT_PV((K) >> 2, 1, HN); } while (0)
        // plain fp32 VALU only: v_pk_*_f32 and v_dot2_f32_f16 do not execute under an MFMA (tools/probes/probe_fat.hip)
#define FAT_F(Q_, R) do { if (FAT_ABL & (8 | 64)) x[Q_][R] = s[H_][Q_][R]; else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[Q_][R]) : "v"(s[H_][Q_][R]), "s"(scale_log2), "v"(nm[Q_])); } while (0)
#define FAT_E(Q_, R) do { if (FAT_ABL & (8 | 128)) e[Q_][R] = x[Q_][R]; else asm volatile("v_exp_f32 %0, %1" : "=v"(e[Q_][R]) : "v"(x[Q_][R])); \
                          if (MASK) asm volatile("v_cmp_gt_i32 vcc, %2, %1\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(e[Q_][R]) : "v"(lim[Q_]), "n"(((R) & 3) + 8 * ((R) >> 2)), "v"(fzero) : "vcc"); } while (0)
#define FAT_C(Q_, J) do { if (FAT_ABL & (8 | 256)) break; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pf[H_][Q_][(J) >> 2][(J) & 3]) : "v"(e[Q_][2 * (J)]), "v"(e[Q_][2 * (J) + 1])); } while (0)
#define FAT_A(Q_, R) do { if (FAT_ABL & (8 | 512)) break; asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[Q_][(R) & 3]) : "v"(e[Q_][R])); } while (0)
#define FAT_DMA(E) do { if (!(FAT_ABL & 1)) dma(t + 3, H_ * 4 + (E)); } while (0)
#include "attention_fat_sched.inc"
        rd_v(7, vsrc);                                                   // behind the DMA of slot 31: P.V(q1, 7) has read the fragment by then
    };
    // one tile = two chunks; on entry: all waves are done with tile t-1 (its ring slot takes tile t+3) and tile t+1 (DMA issued
    // during tile t-2) has landed - at most the 8 DMA instructions of tile t+2 stay in flight
    auto tile = [&](auto Mc, int t, int last_) {
        const int last = __builtin_amdgcn_readfirstlane(last_);              // an SGPR for the s_cmp inside FAT_PV_DRAIN
        if (!(FAT_ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const char* vs = smem + (t & 3) * F_STAGE;
        const char* ks = smem + ((t + 1) & 3) * F_STAGE;
        iter(std::integral_constant<int, 0>{}, Mc, t, vs, ks, 0);
        iter(std::integral_constant<int, 1>{}, Mc, t, vs + 32 * 256, ks + 32 * 256, last);
    };
    // two loops, not one loop with two bodies: around a branch inside the loop hipcc moves every accumulator between copies
    for (int t = 0; t < n_full; ++t) tile(std::false_type{}, t, t + 1 == n_full);          // last pass of a loop: drain before the moves behind it
    for (int t = n_full; t < nt; ++t) tile(std::true_type{}, t, t + 1 == nt);
    // drain: P.V of the last chunk (parity 1), then let the MFMA pipe and the DMA queue run empty (zero-fill DMA must not land in
    // the LDS of the workgroup that follows on this CU)
#pragma unroll
    for (int i = 0; i < 7; ++i) { FAT_PV(i, 0, 1); FAT_PV(i, 1, 1); }
    FAT_PV(7, 0, 1);
    FAT_PV_DRAIN(7, 1, 1, 1);
    // ... and ONE statement that owns all eight accumulators: the epilogue's v_accvgpr_read of an O tile may otherwise be scheduled
    // right behind that tile's last asm MFMA (in flight for another 64 cycles) - which is what happened to the four q-block-0 tiles
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)"
                 : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[2][0]), "+a"(o[2][1]), "+a"(o[3][0]), "+a"(o[3][1]) : : "memory");

    // ---- overflow check (fp16 P beyond 2^16 of the reference -> inf / NaN in O or l) ----
    float chk = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        chk += (l[q][0] + l[q][1] + l[q][2] + l[q][3]) * 0.f;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) chk += o[db][q][r] * 0.f;
    }
    if (!__syncthreads_or(chk != chk) || FAT_ABL) {
        // lane (q, hh) holds O[q][32 * db + 8 * (r >> 2) + 4 * hh + (r & 3)]
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float ls = (l[q][0] + l[q][1]) + (l[q][2] + l[q][3]);
            ls += __shfl_xor(ls, 32, 64);
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            const int qr = qw0 + q * 32 + ql;
            if (qr < Sq) {
                _Float16* op = obase + (size_t)qr * (size_t)ldo + 4 * hh;
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        *reinterpret_cast<sc_h4*>(op + db * 32 + r4 * 8) = sc_h4{(_Float16)(o[db][q][4 * r4] * inv), (_Float16)(o[db][q][4 * r4 + 1] * inv),
                                                                                 (_Float16)(o[db][q][4 * r4 + 2] * inv), (_Float16)(o[db][q][4 * r4 + 3] * inv)};
            }
        }
        return;
    }

    // ---- exact redo (never on real activations): 16 queries at a time, single-buffered tiles, online softmax with O rescale;
    //      compiler-scheduled 16x16x32 code with the fragment maps of k_attn<128> (V swizzle (r & 7) << 1) ----
    const int rl = lane & 15, g = lane >> 4;
    unsigned v_lo16[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qg = j * 256 + tid, r = qg >> 4;
        v_lo16[j] = ((unsigned)r * (unsigned)ldv + (unsigned)(((qg & 15) ^ ((r & 7) << 1)) * 8)) * 2u;
    }
    int k_off[4], v_off16[8];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off[ds] = rl * 256 + (((ds * 4 + g) ^ rl) << 4);
    const int vrow = 4 * g + (rl >> 2);
#pragma unroll
    for (int db = 0; db < 8; ++db) v_off16[db] = F_TILE + vrow * 256 + ((db ^ (vrow & 7)) << 5) + (rl & 3) * 8;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        int qr = qw0 + q * 16 + rl;
        const int qr_c = qr < Sq ? qr : Sq - 1;
        const _Float16* qp = Q + (size_t)b * (size_t)q_bs + (size_t)qr_c * (size_t)ldq + h * q_hs + g * 8;
        sc_h8 qx[4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) qx[ds] = *reinterpret_cast<const sc_h8*>(qp + ds * 32);
        const int qlast = (CAUSAL ? min(qw0 + q * 16 + rl + coff, kv_valid - 1) : kv_valid - 1) - g * 4;
        sc_f4 o2[8];
#pragma unroll
        for (int db = 0; db < 8; ++db) o2[db] = sc_f4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int jj = e & 3;
                const unsigned dst = lds0 + (unsigned)((e >> 2) * F_TILE + (jj * 256 + wave * 64) * 16);
                if (e < 4) fat_dma16(k_rs, dst, k_lo[jj], (unsigned)(t * F_KVT * ldk * 2));
                else fat_dma16(v_rs, dst, v_lo16[jj], (unsigned)(t * F_KVT * ldv * 2));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            sc_f4 sx[4];
#pragma unroll
            for (int kvb = 0; kvb < 4; ++kvb) {
                sx[kvb] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ds = 0; ds < 4; ++ds) {
                    const sc_h8 kx = *reinterpret_cast<const sc_h8*>(smem + kvb * 4096 + k_off[ds]);
                    sx[kvb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kx, qx[ds], sx[kvb], 0, 0, 0);
                }
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int kvb = 0; kvb < 4; ++kvb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sx[kvb][r] = (t * F_KVT + kvb * 16 + r > qlast) ? -INFINITY : sx[kvb][r];
                    tmax = fmaxf(tmax, sx[kvb][r]);
                }
            const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
            const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1])) * scale_log2;
            const float m_new = fmaxf(m_run, tmax);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 8; ++db) o2[db] *= alpha;
            sc_h8 px[2];
#pragma unroll
            for (int kvb = 0; kvb < 4; ++kvb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sx[kvb][r] * scale_log2 - m_use);
                    const _Float16 ph = (_Float16)p;
                    l_run += (float)ph;
                    px[kvb >> 1][(kvb & 1) * 4 + r] = ph;
                }
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                for (int db = 0; db < 8; ++db) {
                    const char* vp = smem + pc * 32 * 256 + v_off16[db];
                    const sc_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vp));
                    const sc_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(vp + 16 * 256));
                    const sc_s8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o2[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_h8, v8), px[pc], o2[db], 0, 0, 0);
                }
        }
        float ls = l_run;
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        const float inv = ls > 0.f ? 1.0f / ls : 0.f;
        if (qr < Sq) {
            _Float16* op = obase + (size_t)qr * (size_t)ldo + g * 4;
#pragma unroll
            for (int db = 0; db < 8; ++db)
                *reinterpret_cast<sc_h4*>(op + db * 16) = sc_h4{(_Float16)(o2[db][0] * inv), (_Float16)(o2[db][1] * inv), (_Float16)(o2[db][2] * inv), (_Float16)(o2[db][3] * inv)};
        }
    }
}

}  // namespace

// Dispatch helper for sc_attention_f16 (attention.hip): Dh = 128, no split-KV.  Returns SC_OK or an error code.
int sc_attn_fat_launch(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Sq, int Skv, int Hq, int Hkv,
                       float scale, int causal, const int32_t* kv_len, int q_hs, int o_hs, long q_bs, long o_bs, hipStream_t s) {
    static bool attr_done[16][2];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int ci = causal ? 1 : 0;
    if (!attr_done[dev & 15][ci]) {
        const hipError_t e = hipFuncSetAttribute(causal ? (const void*)k_attn_fat<true> : (const void*)k_attn_fat<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_attention_f16: cannot reserve %d bytes of LDS: %s", F_LDS, hipGetErrorString(e));
        attr_done[dev & 15][ci] = true;
    }
    const int nqb = (Sq + 255) / 256, G = Hq / Hkv;
    const int npairs = nqb * Hkv * B;
    const dim3 grid((unsigned)(((npairs + 7) / 8) * 8 * G)), block(256);
    const float sl2 = scale * 1.4426950408889634f;
    if (causal)
        hipLaunchKernelGGL((k_attn_fat<true>), grid, block, F_LDS, s, (const _Float16*)q, ldq, (const _Float16*)k, ldk, (const _Float16*)v, ldv, (_Float16*)out, ldo, Sq,
                           Skv, Hq, Hkv, sl2, kv_len, B, q_hs, o_hs, q_bs, o_bs);
    else
        hipLaunchKernelGGL((k_attn_fat<false>), grid, block, F_LDS, s, (const _Float16*)q, ldq, (const _Float16*)k, ldk, (const _Float16*)v, ldv, (_Float16*)out, ldo, Sq,
                           Skv, Hq, Hkv, sl2, kv_len, B, q_hs, o_hs, q_bs, o_bs);
    SC_CHECK_LAUNCH("sc_attention_f16");
    return SC_OK;
}
