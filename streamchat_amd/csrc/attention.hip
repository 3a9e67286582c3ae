// Fused multi-head attention for gfx950 (CDNA4), fp16 in/out, fp32 online softmax.
//   ViT-L: S = 577, 16 heads x 64, non-causal  (HF CLIPAttention behind reference clip_encoder.py:76)
//   BERT : padding-masked (kv_len), 12..16 heads x 32/64 (reference utiles.py:707,728)
//   Qwen2: causal GQA 28/4 heads x 128 over 26k-49k stacked frame tokens (reference llava_qwen.py:155)
//
// Orientation: everything is computed TRANSPOSED so that one lane owns one query column end to end.
//   S^T[kv][q] = K[kv][:] . Q[q][:]      A = K fragment (LDS, ds_read_b128),  B = Q fragment (registers)
//   O^T[d][q] += V^T[d][kv] . P^T[kv][q]  A = V^T fragment (row-major V tile in LDS read with
//                                          ds_read_b64_tr_b16, the gfx950 transpose read), B = P^T (registers)
// With v_mfma_f32_16x16x32_f16 the C/D layout is col = lane&15, row = (lane>>4)*4 + r, so the P values a lane
// produced for its query column are exactly the B-operand k-slots it must feed to the second MFMA (the
// k-slot <-> kv-row bijection is applied to the V rows the transpose read fetches).  Row max / row sum /
// rescale factors are therefore per-lane scalars: no LDS round trip and no cross-lane traffic for P; the row max across the
// 4 lane groups is two VALU row swaps (v_permlane16_swap / v_permlane32_swap).
//
// Block = 4 waves x QB q-blocks of 16 queries (QB = 2; 3 for long Dh = 128 prefill); KV tiles of 64 rows double-buffered in
// LDS by 16-byte buffer_load ... lds (the resource extent is the valid cache length: rows past kv_len arrive as zeros);
// bank-conflict-free XOR swizzles on the per-lane source address and on the reads.  The KV loop is a general tile body plus a
// steady-state loop with no mask, no row max and no O rescale (see the notes at the loop); what bounds it - LDS fragment
// traffic and the SIMD issue port - is measured in tools/probes/ and summarised in DESIGN.md section 4.
#include "sc_common.h"
#include <stdlib.h>
#include <type_traits>

// attention_decode.hip (its own translation unit: default machine scheduler)
void sc_attn_combine_launch(int Dh, const float* part, void* out, int ldo, int B, int Sq, int Hq, int nsplit, int o_hs, long o_bs, hipStream_t s);
void sc_attn_decode_launch(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int B, int Sq, int Skv, int Hq, int Hkv, float scale_log2,
                           const int32_t* kv_len, float* part, int nsplit, int q_hs, long q_bs, hipStream_t s);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef short sc_s4 __attribute__((ext_vector_type(4)));
typedef float sc_f2 __attribute__((ext_vector_type(2)));

constexpr int KVT = 64;   // kv rows per tile

// 16-byte LDS-DMA through a raw buffer resource (base, extent in bytes): lane address = base + voff + soff, destination = the
// wave-uniform LDS pointer + lane * 16; out-of-range lanes write zeros.  A free function because an opaque
// __amdgpu_buffer_rsrc_t inside a lambda of the kernel silently drops the kernel's host stub.
__device__ __forceinline__ void lds_load16(const void* base, int extent, char* lds, unsigned voff, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, extent, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// PRE (SC_ATTN_Q_PRESCALED): the caller's Q already carries the softmax scale in the log2 domain (q * scale * log2 e, applied by the
// PRODUCER of q to its fp32 accumulators before the one rounding to fp16: the rotary / column-scale epilogues of gemm.hip, k_decode_qkv,
// k_rope_f32in).  In the steady-state loop the softmax reference then enters through the C operand of the first S MFMA of a k-step
// chain (accumulators start at -m instead of 0): the scores leave the matrix pipe as s - m and p = 2^that needs no FMA - one VALU
// instruction less per score in an issue-bound loop (+4 % at 49 k tokens, +3 % on the ViT shape, profiles/r03_run2_attn_prescale_ab.md).
// Scaling an fp16 q inside this kernel instead (a second rounding) was measured and rejected: error x2-7 on peaky rows.
template <int DH, int QB, bool CAUSAL, int CH, bool PRE>
__global__ __launch_bounds__(256, (DH == 128 ? 2 : 3)) void k_attn(const _Float16* __restrict__ Q, int ldq, const _Float16* __restrict__ Kp, int ldk,
                                              const _Float16* __restrict__ Vp, int ldv, _Float16* __restrict__ O, int ldo, int Sq,
                                              int Skv, int Hq, int Hkv, float scale_log2, const int* __restrict__ kv_len,
                                              float* __restrict__ part, int nsplit, int B, int q_hs, int o_hs, long q_bs, long o_bs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = KVT * DH * 2;            // bytes per K (or V) tile
    constexpr int STAGE = 2 * TILE;
    constexpr int GPT = TILE / 16 / 256;          // 16-byte granules per thread per operand tile (2 or 4)
    constexpr int DS = DH / 32;                   // MFMA k-steps over the head dim
    constexpr int DB = DH / 16;                   // 16-row blocks of O^T

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware (block i runs on XCD i % 8): each XCD walks its own queue of (batch, kv-head, q-block) "pairs"; the
    // G = Hq/Hkv query heads of a pair are CONSECUTIVE entries of the same XCD's queue, so they run together on that XCD and
    // stream the same K/V tiles through its L2 once (PMC before: 69.5 GB fetched per 49k-token GQA launch, heads of a group
    // scattered over 7 XCDs).  q-blocks are taken in DESCENDING order: under a causal mask the longest blocks go first.
    const int nqb = (Sq + 4 * QB * 16 - 1) / (4 * QB * 16);
    const int G = Hq / Hkv;
    const int split = blockIdx.x % nsplit, bid = blockIdx.x / nsplit;            // split-KV (flash-decoding) index fastest
    const int xcd = bid & 7, j = bid >> 3;
    const int pair = (j / G) * 8 + xcd;
    const int npairs = nqb * Hkv * B;
    if (pair >= npairs) return;                                                  // padding of the per-XCD queues (whole block exits)
    const int hk = pair % Hkv, h = hk * G + j % G;
    const int qi = (pair / Hkv) % nqb, b = pair / (Hkv * nqb);
    const int qblk0 = (nqb - 1 - qi) * (4 * QB * 16);
    const int qw0 = qblk0 + wave * (QB * 16);
    const int kv_valid = kv_len ? min(kv_len[b], Skv) : Skv;
    const int coff = Skv - Sq;                    // causal: query i sits at kv position i + coff
    const bool wave_live = qw0 < Sq;              // false: every query row of this wave is padding of the last q-block
    const int qw_last = min(qw0 + QB * 16, Sq) - 1;

    int nt = (kv_valid + KVT - 1) / KVT;
    if (CAUSAL) {
        const int last_q = min(qblk0 + 4 * QB * 16, Sq) - 1;
        const int lim = (last_q + coff) / KVT + 1;
        nt = nt < lim ? nt : lim;
    }

    // ---- Q fragments (registers, loaded once) ----
    sc_h8 qf[QB][DS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int qr = qw0 + qb * 16 + rl;
        qr = qr < Sq ? qr : Sq - 1;
        const _Float16* qp = Q + (size_t)b * (size_t)q_bs + (size_t)qr * (size_t)ldq + h * q_hs + g * 8;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            qf[qb][ds] = *reinterpret_cast<const sc_h8*>(qp + ds * 32);
        }
    }

    // ---- staging sources ----
    const _Float16* kbase = Kp + (size_t)b * Skv * (size_t)ldk + hk * DH;
    const _Float16* vbase = Vp + (size_t)b * Skv * (size_t)ldv + hk * DH;
    // granule q (16 bytes) of a staged tile -> source row and swizzled 16-byte column slot, for K and for V
    auto gran = [&](int q, int& kr, int& ks, int& vr, int& vs) {
        if (DH == 32) {                         // 64-byte rows: 4 granules
            kr = q >> 2;
            ks = (q & 3) ^ ((0x78 >> (2 * ((kr >> 2) & 3))) & 3);      // f = {0,2,3,1}[(row>>2)&3]
            vr = q >> 2;
            vs = (q & 3) ^ (((vr >> 2) & 1) << 1);
        } else if (DH == 64) {
            kr = 2 * (q >> 4) + ((q & 15) >> 3);
            ks = (q & 7) ^ ((q >> 4) & 7);
            vr = q >> 3;
            vs = (q & 7) ^ (((vr >> 1) & 3) << 1);
        } else {
            kr = q >> 4;
            ks = (q & 15) ^ (kr & 15);
            vr = q >> 4;
            vs = (q & 15) ^ ((vr & 7) << 1);
        }
    };
    // K/V tiles are fetched with buffer_load ... lds: uniform tile offset in an SGPR (soffset) + a constant per-lane 32-bit
    // offset (voffset), so the steady state has no per-tile VALU address arithmetic at all, and the hardware bounds check of
    // the buffer resource returns zeros for rows >= Skv of the ragged last tile (K = 0 scores are masked below; V = 0 is finite).
    unsigned k_lo[GPT], v_lo[GPT];
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        int kr, ks, vr, vs;
        gran(j * 256 + tid, kr, ks, vr, vs);
        k_lo[j] = ((unsigned)kr * (unsigned)ldk + (unsigned)ks * 8u) * 2u;
        v_lo[j] = ((unsigned)vr * (unsigned)ldv + (unsigned)vs * 8u) * 2u;
    }
    // extent = the VALID rows only: rows in [kv_valid, Skv) of a cache hold whatever the allocator left there, and a NaN in V
    // would survive the multiplication by p = 0 in the P.V MFMA; beyond the extent the DMA writes zeros instead
    const int k_bytes = kv_valid > 0 ? (int)(((unsigned)(kv_valid - 1) * (unsigned)ldk + DH) * 2u) : 0;
    const int v_bytes = kv_valid > 0 ? (int)(((unsigned)(kv_valid - 1) * (unsigned)ldv + DH) * 2u) : 0;
    auto stage = [&](int buf, int t) {
        char* base = smem + buf * STAGE;
        const int k_so = t * KVT * ldk * 2, v_so = t * KVT * ldv * 2;
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            lds_load16(kbase, k_bytes, base + (j * 256 + wave * 64) * 16, k_lo[j], k_so);
            lds_load16(vbase, v_bytes, base + TILE + (j * 256 + wave * 64) * 16, v_lo[j], v_so);
        }
    };

    // ---- read addresses ----
    int k_off[DS];   // K fragment byte offset of (row rl, d-step ds) inside a 16-row block
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
        if (DH == 32) k_off[ds] = rl * 64 + ((g ^ ((0x78 >> (2 * ((rl >> 2) & 3))) & 3)) << 4);
        else if (DH == 64) k_off[ds] = (rl >> 1) * 256 + ((((rl & 1) << 3) | ((ds * 4 + g) ^ ((rl >> 1) & 7))) << 4);
        else k_off[ds] = rl * 256 + (((ds * 4 + g) ^ rl) << 4);
    }
    constexpr int KBLK = 16 * DH * 2;             // bytes per 16 kv rows
    // V transpose-read: this lane supplies row (4g + (rl>>2)) [+16 for the second read] of a 32-row chunk,
    // 8-byte column chunk (rl&3) of the 16-column d-block db
    const int vrow = 4 * g + (rl >> 2);
    int v_off[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        if (DH == 32) v_off[db] = vrow * 64 + ((db ^ (g & 1)) << 5) + (rl & 3) * 8;
        else if (DH == 64) v_off[db] = vrow * 128 + ((db ^ ((vrow >> 1) & 3)) << 5) + (rl & 3) * 8;
        else v_off[db] = vrow * 256 + ((db ^ (vrow & 7)) << 5) + (rl & 3) * 8;
    }
    constexpr int VROW = DH * 2;

    sc_f4 o[DB][QB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int j = 0; j < QB; ++j) o[i][j] = sc_f4{0.f, 0.f, 0.f, 0.f};
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; }
    // Row sums on the matrix pipe at head dim <= 64: one extra MFMA per (q-block, 32 kv rows) with an all-ones A operand gives
    // sum_kv P[kv][q] in every row of a 16 x 16 accumulator (the fp16-rounded P, exactly what P.V uses; complete over the four lane
    // groups, so no shuffles at the end).  At Dh = 64 the loop is bound by the VALU port (2.8 VALU per MFMA, the pipe 25-40 % busy):
    // this trades 8 v_add per lane and chunk for half an MFMA.  At Dh = 128 the pipe (and the board's power) was the limit and the VALU adds
    // stayed - until the pre-scaled-q mode took the FMA out of the loop: with it the adds are the next VALU instructions to go and the ones-MFMA
    // wins there too (+1 % at 49 k: 1262 -> 1274 TF, profiles/r03_run9), so PRE takes the matrix-pipe row sums at every head dim.
    constexpr bool LSUM_MFMA = (DH <= 64) || PRE;
    sc_f4 ol[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) ol[qb] = sc_f4{0.f, 0.f, 0.f, 0.f};
    const sc_h8 ones8 = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};

    // split-KV: this block covers tiles [t_lo, t_hi)
    const int per = (nt + nsplit - 1) / nsplit;
    const int t_lo = split * per;
    const int t_hi = (t_lo + per) < nt ? (t_lo + per) : nt;
    if (t_lo < t_hi) stage(0, t_lo);
    __syncthreads();
    // m_run is the REFERENCE max of a row in the scaled log2 domain: p = 2^(s - m_run); O and l are relative to the same reference,
    // so the final O / l does not depend on it.  Two tile bodies share the loop:
    //   GENERAL: masks, moves the reference to the true running max and rescales O (exact online softmax step);
    //   STEADY STATE: valid while every kv of the tile is visible to every query of the wave.  The reference stays where the last
    //         general tile left it, O is only ever touched
    //         by the accumulating MFMAs (a rescale branch inside one shared body made the compiler keep two copies of O: +64
    //         VGPRs, 32..66 moves per tile), and there is NO row max at all: fp16 P holds 2^-24 .. 2^16 around the reference, i.e.
    //         it only fails when a later score exceeds the first tile's row max by a factor e^11 - that is detected at the end
    //         (inf / NaN in O or l) and the block redoes its rows with the general body on every tile.
    //         Per tile that removes 28 max + 2 permlane swaps per q-block from an issue-bound loop.  (Pre-scaling Q by scale*log2 e
    //         and starting the accumulators at -m_ref would also remove the 16 packed FMAs, but the extra fp16 rounding of q*c moves a
    //         score by |s| * 2^-11 / sqrt(Dh): fine for |s| < 50, 9 % on p at the |s| ~ 2000 of the overflow test.)
    // The unit of S / softmax / P.V work is a CHUNK of CH kv rows (CH = 64: the whole tile; CH = 32: half tiles).  With half tiles S
    // and P of only 32 rows are live at a time, which is what lets a wave carry QB = 3 q-blocks (48 queries) in 256 VGPRs at
    // Dh = 128: every K / V fragment read from LDS then feeds three MFMAs instead of two.  The loop is LDS-bandwidth bound (a
    // fragment is 1 KB; at QB = 2 the LDS pipe needs as many cycles per tile as the MFMA pipe - tools/probes/probe_phases2/3.hip:
    // 12.1 -> 10.0 ns per MFMA going from 2 to 3 MFMAs per fragment); on the real kernel: +6 % at 26 k tokens, but half tiles alone
    // cost 3 % (shorter MFMA bursts) and 192-query blocks waste a quarter of a 577-token ViT sequence, so short sequences keep
    // CH = 64 / QB = 2.
    constexpr int NCH = KVT / CH, KVB = CH / 16, PC = CH / 32;
    // nkvb: the first nkvb 16-row blocks of the chunk hold a key this wave can see (KVB in the steady state: folds away); the MFMAs,
    // exponentials and P.V k-steps of the other blocks are skipped (ragged last tile: 577 = 9 x 64 + 1 keys per ViT frame; causal diagonal)
    sc_f4 negm[QB];                               // PRE, steady state: (-m, -m, -m, -m) of every q-block = the C operand the score chains start from
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) negm[qb] = sc_f4{0.f, 0.f, 0.f, 0.f};
    auto s_part = [&](const char* sk, int c, sc_f4 (&s)[KVB][QB], int nkvb, auto steady) {
        constexpr int NKF = KVB * DS;
        const char* skc = sk + c * KVB * KBLK;
        sc_h8 kfr[2];
        kfr[0] = *reinterpret_cast<const sc_h8*>(skc + k_off[0]);
#pragma unroll
        for (int i = 0; i < NKF; ++i) {
            const int kvb = i / DS, ds = i % DS;
            if (i + 1 < NKF) kfr[(i + 1) & 1] = *reinterpret_cast<const sc_h8*>(skc + ((i + 1) / DS) * KBLK + k_off[(i + 1) % DS]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const sc_f4 acc = ds == 0 ? ((PRE && decltype(steady)::value) ? negm[qb] : sc_f4{0.f, 0.f, 0.f, 0.f}) : s[kvb][qb];
                if (kvb < nkvb) s[kvb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[i & 1], qf[qb][ds], acc, 0, 0, 0);
                else s[kvb][qb] = sc_f4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto row_max = [&](const sc_f4 (&s)[KVB][QB], int qb) {
        float tmax = -INFINITY;
#pragma unroll
        for (int kvb = 0; kvb < KVB; ++kvb)
#pragma unroll
            for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[kvb][qb][r]);
        // max over the 4 lane groups holding this query column: two VALU row swaps (v_permlane16_swap / v_permlane32_swap with the
        // value as both operands give x and its xor-16 / xor-32 partner) instead of two ~100-cycle ds_bpermute round trips
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1])) * (PRE ? 1.0f : scale_log2);      // scale > 0: max commutes with scaling
    };
    // P = 2^(s*scale - m) -> fp16 MFMA operand + row sums (one packed FMA per pair of scores)
    // scalar fp32 math on purpose (the file is built with -fno-slp-vectorize): v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (and
    // v_dot2_f32_f16) do not execute under an MFMA of the same SIMD - each costs ~11 cycles of the matrix pipe, while up to four
    // plain v_fma_f32 / v_add_f32 per MFMA are free (tools/probes/probe_fat.hip, profiles/r02_run30_probe_fat.log)
    auto p_part = [&](const sc_f4 (&s)[KVB][QB], int qb, float m_sub, sc_h8 (&pf)[QB][PC], int nkvb, auto steady) {
        const float nm = -m_sub;
        constexpr bool DIRECT = PRE && decltype(steady)::value;       // the reference is already inside s
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kvb = 0; kvb < KVB; ++kvb)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                float p0 = 0.f, p1 = 0.f;
                if (kvb < nkvb) {
                    p0 = __builtin_amdgcn_exp2f(DIRECT ? s[kvb][qb][r] : (PRE ? s[kvb][qb][r] + nm : __builtin_fmaf(s[kvb][qb][r], scale_log2, nm)));
                    p1 = __builtin_amdgcn_exp2f(DIRECT ? s[kvb][qb][r + 1] : (PRE ? s[kvb][qb][r + 1] + nm : __builtin_fmaf(s[kvb][qb][r + 1], scale_log2, nm)));
                }
                if (!LSUM_MFMA) { ps0 += p0; ps1 += p1; }
                pf[qb][kvb >> 1][(kvb & 1) * 4 + r] = (_Float16)p0;
                pf[qb][kvb >> 1][(kvb & 1) * 4 + r + 1] = (_Float16)p1;
            }
        if (!LSUM_MFMA) l_run[qb] += ps0 + ps1;
    };
    auto pv_part = [&](const char* sv, int c, const sc_h8 (&pf)[QB][PC], int nkvb) {
#pragma unroll
        for (int pc = 0; pc < PC; ++pc) {
            if (2 * pc >= nkvb) continue;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const char* vp = sv + (c * PC + pc) * 32 * VROW + v_off[db];
                const sc_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sc_s4*)(vp));
                const sc_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sc_s4*)(vp + 16 * VROW));
                typedef short sc_s8 __attribute__((ext_vector_type(8)));
                const sc_s8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const sc_h8 vf = __builtin_bit_cast(sc_h8, v8);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qb][pc], o[db][qb], 0, 0, 0);
            }
            if (LSUM_MFMA)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) ol[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[qb][pc], ol[qb], 0, 0, 0);
        }
    };
    auto masked = [&](int t) {
        bool m = (t * KVT + KVT > kv_valid);
        if (CAUSAL) m = m || (t * KVT + KVT - 1 > qw0 + coff);                        // wave-uniform
        return m;
    };

    // pass 0: general tiles + the steady-state loop (no row max, no reference check: see the note above).  If any P of the block
    // overflowed fp16 (a score more than 2^16 above its row's reference) O or l is inf / NaN at the end; the WHOLE block then
    // redoes its rows in pass 1 with the general body only (exact online softmax on every chunk).
    for (int pass = 0; pass < 2; ++pass) {
        int t = t_lo;
        while (t < t_hi) {
            // ---------------- general tile ----------------
            {
                const int cur = (t - t_lo) & 1;
                if (t + 1 < t_hi) stage(cur ^ 1, t + 1);
                const char* sk = smem + cur * STAGE;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (!wave_live) break;                                            // no valid query row in this wave: staging and barriers only
                    int nkvb = (kv_valid - (t * KVT + c * CH) + 15) >> 4;                // 16-row blocks with a valid key ...
                    if (CAUSAL) nkvb = min(nkvb, ((qw_last + coff - (t * KVT + c * CH)) >> 4) + 1);      // ... that the wave's last query still sees
                    nkvb = max(0, min(nkvb, KVB));
                    sc_f4 s[KVB][QB];
                    s_part(sk, c, s, nkvb, std::false_type{});
                    sc_h8 pf[QB][PC];
                    const int kv_t0 = t * KVT + c * CH + g * 4;
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const int qpos = qw0 + qb * 16 + rl + coff;
#pragma unroll
                        for (int kvb = 0; kvb < KVB; ++kvb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int kv = kv_t0 + kvb * 16 + r;
                                const bool dead = (kv >= kv_valid) || (CAUSAL && kv > qpos);
                                s[kvb][qb][r] = dead ? -INFINITY : s[kvb][qb][r];
                            }
                        const float m_new = fmaxf(m_run[qb], row_max(s, qb));
                        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_use);       // 1 when the reference did not move, 0 at the start
                        m_run[qb] = m_new;
                        l_run[qb] *= alpha;
                        if (LSUM_MFMA) ol[qb] *= alpha;
#pragma unroll
                        for (int db = 0; db < DB; ++db) o[db][qb] *= alpha;
                        p_part(s, qb, m_use, pf, nkvb, std::false_type{});
                    }
                    pv_part(sk + TILE, c, pf, nkvb);
                }
                __syncthreads();
                ++t;
            }
            if (pass == 1) continue;
            // ---------------- steady state ----------------
            bool ok = true;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) ok = ok && (m_run[qb] > -INFINITY);
            if (!__all(ok)) continue;                                                 // a row without any visible key yet
            if (PRE) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) negm[qb] = sc_f4{-m_run[qb], -m_run[qb], -m_run[qb], -m_run[qb]};
            }
            while (t < t_hi && !masked(t)) {
                const int cur = (t - t_lo) & 1;
                if (t + 1 < t_hi) stage(cur ^ 1, t + 1);
                const char* sk = smem + cur * STAGE;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    sc_f4 s[KVB][QB];
                    s_part(sk, c, s, KVB, std::true_type{});
                    sc_h8 pf[QB][PC];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) p_part(s, qb, m_run[qb], pf, KVB, std::true_type{});
                    pv_part(sk + TILE, c, pf, KVB);
                }
                __syncthreads();
                ++t;
            }
        }
        if (pass == 1) break;
        float chk = 0.f;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            chk += (LSUM_MFMA ? ol[qb][0] : l_run[qb]) * 0.f;
#pragma unroll
            for (int db = 0; db < DB; ++db) chk += (o[db][qb][0] + o[db][qb][1] + o[db][qb][2] + o[db][qb][3]) * 0.f;        // inf, NaN -> NaN
        }
        if (!__syncthreads_or(chk != chk)) break;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int jj = 0; jj < QB; ++jj) o[i][jj] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; ol[qb] = sc_f4{0.f, 0.f, 0.f, 0.f}; }
        if (t_lo < t_hi) stage(0, t_lo);
        __syncthreads();
    }

    // ---- normalise + store: lane holds O[q][db*16 + g*4 .. +3] ----
    // In this layout adjacent lanes are adjacent QUERY ROWS: an 8-byte store per lane is 64 separate requests (8 % of the ViT launch,
    // profiles/r03_run39_40_attn_output_store.md).  When the output rows allow 16-byte stores the fp16 tile goes through LDS instead (the K / V stages are free
    // now; each wave its own slab, rows padded by 16 B) and leaves along rows: 64 / (DH / 8) rows of DH * 2 bytes per instruction.
    const bool rows16 = !part && (((unsigned)ldo | (unsigned)o_hs | (unsigned)(o_bs & 0xffff)) & 7u) == 0 && (reinterpret_cast<size_t>(O) & 15) == 0;
    constexpr int ORS = DH * 2 + 16;              // slab row stride
    static_assert(4 * QB * 16 * ORS <= 2 * STAGE, "the output slabs must fit in the K / V stages");
    if (rows16) __syncthreads();                  // every wave is done with the last tile's K / V fragments
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        if (LSUM_MFMA) l = ol[qb][0];
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qr = qw0 + qb * 16 + rl;
        if (part) {            // split-KV partial: unnormalised O (fp32), running max (scaled log2 domain) and sum
            if (qr < Sq) {
                float* pp = part + ((((size_t)b * Hq + h) * Sq + qr) * nsplit + split) * (DH + 2);
#pragma unroll
                for (int db = 0; db < DB; ++db) *reinterpret_cast<sc_f4*>(pp + db * 16 + g * 4) = o[db][qb];
                if (g == 0) { pp[DH] = m_run[qb]; pp[DH + 1] = l; }
            }
        } else if (rows16) {
            char* slab = smem + (wave * QB + qb) * 16 * ORS;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const sc_h4 v = {(_Float16)(o[db][qb][0] * inv), (_Float16)(o[db][qb][1] * inv), (_Float16)(o[db][qb][2] * inv),
                                 (_Float16)(o[db][qb][3] * inv)};
                *reinterpret_cast<sc_h4*>(slab + rl * ORS + db * 32 + g * 8) = v;
            }
            constexpr int CPR = DH / 8, RPI = 64 / CPR;               // 16-byte chunks per row, rows per store instruction
            const int sr = lane / CPR, sc = lane % CPR;
#pragma unroll
            for (int i = 0; i < 16 / RPI; ++i) {
                const int row = i * RPI + sr, qrow = qw0 + qb * 16 + row;
                const sc_u4 d = *reinterpret_cast<const sc_u4*>(slab + row * ORS + sc * 16);          // (same wave wrote it: LDS keeps a wave's accesses in order)
                if (qrow < Sq) *reinterpret_cast<sc_u4*>(O + (size_t)b * (size_t)o_bs + (size_t)qrow * (size_t)ldo + h * o_hs + sc * 8) = d;
            }
        } else if (qr < Sq) {
            _Float16* op = O + (size_t)b * (size_t)o_bs + (size_t)qr * (size_t)ldo + h * o_hs + g * 4;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const sc_h4 v = {(_Float16)(o[db][qb][0] * inv), (_Float16)(o[db][qb][1] * inv), (_Float16)(o[db][qb][2] * inv),
                                 (_Float16)(o[db][qb][3] * inv)};
                *reinterpret_cast<sc_h4*>(op + db * 16) = v;
            }
        }
    }
}

template <int DH, int QB, int CH = 64>
int launch_attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B, int Sq, int Skv,
                int Hq, int Hkv, float scale, int causal, const int32_t* kv_len, float* part, int nsplit, int q_hs, int o_hs, long q_bs, long o_bs, hipStream_t s,
                bool pre) {
    const int nqb = (Sq + 4 * QB * 16 - 1) / (4 * QB * 16), G = Hq / Hkv;
    const int npairs = nqb * Hkv * B;
    const dim3 grid((unsigned)(((npairs + 7) / 8) * 8 * G * nsplit)), block(256);
    const size_t lds = 2 * 2 * KVT * DH * 2;
    const float sl2 = scale * 1.4426950408889634f;
#define SC_LA(C, P) hipLaunchKernelGGL((k_attn<DH, QB, C, CH, P>), grid, block, lds, s, (const _Float16*)q, ldq, (const _Float16*)k, ldk, (const _Float16*)v, \
                                       ldv, (_Float16*)out, ldo, Sq, Skv, Hq, Hkv, sl2, kv_len, part, nsplit, B, q_hs, o_hs, q_bs, o_bs)
    if (causal) { if (pre) SC_LA(true, true); else SC_LA(true, false); }
    else { if (pre) SC_LA(false, true); else SC_LA(false, false); }
#undef SC_LA
    if (part) sc_attn_combine_launch(DH, part, out, ldo, B, Sq, Hq, nsplit, o_hs, o_bs, s);
    SC_CHECK_LAUNCH("sc_attention_f16");
    return SC_OK;
}

}  // namespace

static bool attn_decode_enabled() {
    static const bool on = [] { const char* e = getenv("SC_ATTN_DECODE"); return !(e && e[0] == '0'); }();
    return on;
}

// which machine scheduler attention.hip was built with (Makefile: iterative-ilp, or the default one if that compile failed) - sc_build_info()
const char* sc_attn_build_tag() {
#ifdef SC_ATTN_SCHED_FALLBACK
    return "attention=default-sched(FALLBACK)";
#else
    return "attention=iterative-ilp";
#endif
}

extern "C" int sc_attention_variant(int Dh, int Sq, int nsplit) {
    if (Dh == 128 && nsplit == 1 && Sq >= 2048) return 1;
    if (Dh == 128 && nsplit > 1 && Sq <= 16 && attn_decode_enabled()) return 3;      // (non-causal calls; a causal call of this shape runs k_attn)
    return 0;
}

extern "C" int sc_attention_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* out, int ldo, int B,
                                int Sq, int Skv, int Hq, int Hkv, int Dh, float scale, int causal_flags, const int32_t* kv_len,
                                int nsplit, void* ws, size_t ws_bytes, int q_head_stride, int o_head_stride, int64_t q_batch_stride,
                                int64_t o_batch_stride, sc_stream_t stream) {
    SC_REQUIRE(q && k && v && out, "sc_attention_f16: null pointer argument");
    SC_REQUIRE((causal_flags & ~(1 | SC_ATTN_Q_PRESCALED)) == 0, "sc_attention_f16: unknown bits in the causal / flags argument");
    const int causal = causal_flags & 1;
    SC_REQUIRE(B > 0 && Sq > 0 && Skv > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "sc_attention_f16: bad sizes");
    SC_REQUIRE(Dh == 64 || Dh == 128 || Dh == 32, "sc_attention_f16: head dim %d unsupported (32, 64, 128)", Dh);
    SC_REQUIRE(!causal || Skv >= Sq, "sc_attention_f16: causal needs Skv >= Sq");
    SC_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "sc_attention_f16: leading dims must be multiples of 8 (out: 4)");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(out) & 7) == 0, "sc_attention_f16: q/k/v must be 16-byte aligned, out 8-byte");
    hipStream_t s = (hipStream_t)stream;
    SC_REQUIRE(nsplit >= 1 && nsplit <= 1024, "sc_attention_f16: nsplit must be in [1, 1024]");
    float* part = nullptr;
    if (nsplit > 1) {
        const size_t need = (size_t)B * Hq * Sq * nsplit * (Dh + 2) * sizeof(float);
        if (!ws || ws_bytes < need) return sc_fail(SC_ERR_WORKSPACE, "sc_attention_f16: split-KV workspace %zu < required %zu", ws_bytes, need);
        part = (float*)ws;
    }
    const int qhs = q_head_stride > 0 ? q_head_stride : Dh, ohs = o_head_stride > 0 ? o_head_stride : Dh;
    SC_REQUIRE(qhs % 8 == 0 && ohs % 4 == 0, "sc_attention_f16: head strides must be multiples of 8 (q) / 4 (out)");
    const long qbs = q_batch_stride > 0 ? (long)q_batch_stride : (long)Sq * ldq, obs = o_batch_stride > 0 ? (long)o_batch_stride : (long)Sq * ldo;
    SC_REQUIRE(qbs % 8 == 0 && obs % 4 == 0, "sc_attention_f16: batch strides must be multiples of 8 (q) / 4 (out)");
    // few query rows against a long cache with split-KV = a decode step: the per-wave streaming kernel (SC_ATTN_DECODE=0: the tile kernel)
    const bool pre = (causal_flags & SC_ATTN_Q_PRESCALED) != 0;
    const float sl2_dec = pre ? 1.0f : scale * 1.4426950408889634f;
    if (sc_attention_variant(Dh, Sq, nsplit) == 3 && !causal) {
        sc_attn_decode_launch(q, ldq, k, ldk, v, ldv, B, Sq, Skv, Hq, Hkv, sl2_dec, kv_len, part, nsplit, qhs, qbs, s);
        sc_attn_combine_launch(128, part, out, ldo, B, Sq, Hq, nsplit, ohs, obs, s);
        SC_CHECK_LAUNCH("sc_attention_f16");
        return SC_OK;
    }
    if (Dh == 64) return launch_attn<64, 2>(q, ldq, k, ldk, v, ldv, out, ldo, B, Sq, Skv, Hq, Hkv, scale, causal, kv_len, part, nsplit, qhs, ohs, qbs, obs, s, pre);
    // long prefill: 48 queries per wave in half-tile chunks (three MFMAs per LDS fragment); everything else: 32 queries, whole tiles.
    // (The hand-scheduled one-wave-per-SIMD k_attn_fat of round 2 kept the matrix pipe busier and delivered the same TFLOP/s under the
    // board's power cap; it was removed in round 3 - see DESIGN.md section 4.)
    if (sc_attention_variant(Dh, Sq, nsplit) == 1) return launch_attn<128, 3, 32>(q, ldq, k, ldk, v, ldv, out, ldo, B, Sq, Skv, Hq, Hkv, scale, causal, kv_len, part, nsplit, qhs, ohs, qbs, obs, s, pre);
    if (Dh == 128) return launch_attn<128, 2>(q, ldq, k, ldk, v, ldv, out, ldo, B, Sq, Skv, Hq, Hkv, scale, causal, kv_len, part, nsplit, qhs, ohs, qbs, obs, s, pre);
    return launch_attn<32, 2>(q, ldq, k, ldk, v, ldv, out, ldo, B, Sq, Skv, Hq, Hkv, scale, causal, kv_len, part, nsplit, qhs, ohs, qbs, obs, s, pre);
}
