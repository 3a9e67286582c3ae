// fp16 GEMM with fused epilogues for gfx950 (CDNA4):  C[M,N] = epi(A[M,K] * W[N,K]^T + bias) (+ residual)
//
// This is the op that carries the encoders: the ViT-L/14-336 tower + mlp2x_gelu projector the reference
// runs through transformers' CLIPVisionModel / nn.Sequential (clip_encoder.py:76,
// multimodal_projector/builder.py:41-48), the BERT text encoders (utiles.py:707,728) and the Qwen2
// linears (llava_qwen.py:155).  Weights keep the torch.nn.Linear layout W[N,K] ("B^T input"), so both
// MFMA operands are read along K with 16-byte ds_read_b128.
//
// Three generations live here: k_gemm_fat (v3, below: the default for the large GEMMs), k_gemm256 (v2: 8 waves, K = 32 ring; now the
// fallback for K % 128 != 0, fp32 output and the row-mapped A operand), k_gemm128 (v1: small M / N % 256 != 0) and k_gemm_skinny (M <= 32).
//
// Structure (v1): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 fragments of
// v_mfma_f32_16x16x32_f16 (fp32 accumulate).  Tiles are staged HBM->LDS with 16-byte
// global_load_lds (no VGPR round trip); the LDS image is lane-linear, so the bank-conflict-free
// XOR swizzle is applied on the per-lane SOURCE address and again on the ds_read address
// (cdna guide §5.4 rule 21).  Two 32 KiB LDS stages: tile t+1 streams in while tile t is multiplied.
// Workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous run of tiles.
// The MFMA is issued with swapped operands (W fragment first) so that every lane ends up holding
// 4 consecutive output columns of one row: bias / activation / residual / fp16 pack happen in
// registers and leave as 8-byte stores.
#include "sc_common.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W

// internal: SC_EPI_ROPE where EVERY column is a rotary head (the q projection: lead_cols == N).  Round 5: with the plain-column branch compiled in next
// to the rotary one, hipcc spills 101 VGPRs (280 B of scratch per lane) in the rotary instantiation; without it 9 / 0.
constexpr int SC_EPI_ROPE_ALL = 6;

__device__ __forceinline__ float epi_apply(float x, int epi) {
    if (epi == SC_EPI_QUICK_GELU) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
    if (epi == SC_EPI_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    return x;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void k_gemm128(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                    const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                    void* __restrict__ Cout, int ldc, int M, int N, int K, int tilesN, int a_grp,
                                                    int a_grp_stride, int a_grp_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- XCD-aware, bijective block remap (block b runs on XCD b % 8) ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tm = swz / tilesN, tn = swz - tm * tilesN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- staging: thread owns granules q = i*256 + tid (i < 4) of each operand tile ----
    // granule q holds logical (row r = 2*(q>>4) + ((q&15)>>3), k-slot s = (q&7) ^ ((q>>4)&7))
    const int srow = 2 * (tid >> 4) + ((tid >> 3) & 1);
    const int sslot = (tid & 7) ^ ((tid >> 4) & 7);
    const _Float16* a_src[4];
    const _Float16* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ar = tm * BM + i * 32 + srow;
        ar = ar < M ? ar : M - 1;                       // rows past M re-read the last row (never stored)
        if (a_grp > 0) ar = (ar / a_grp) * a_grp_stride + a_grp_off + (ar % a_grp);   // logical row -> storage row
        a_src[i] = A + (size_t)ar * (size_t)lda + sslot * 8;
        w_src[i] = W + (size_t)(tn * BN + i * 32 + srow) * (size_t)K + sslot * 8;
    }
    auto stage = [&](int buf, int k0) {
        char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[i] + k0), (lds_ptr_t)(base + (i * 256 + wave * 64) * 16), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(w_src[i] + k0), (lds_ptr_t)(base + TILE_BYTES + (i * 256 + wave * 64) * 16), 16, 0, 0);
        }
    };

    // ---- fragment read addresses (bytes inside an operand tile) ----
    const int rl = lane & 15, g = lane >> 4;
    const int sw = (rl >> 1) & 7;
    const int slot0 = ((sw >> 2) << 2) | (g ^ (sw & 3));
    const int frag_off = (rl >> 1) * 256 + ((rl & 1) << 7) + slot0 * 16;
    const int a_base = wm * 32 * 256 + frag_off;
    const int b_base = TILE_BYTES + wn * 32 * 256 + frag_off;

    sc_f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sc_f4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / BK;
    stage(0, 0);
    __syncthreads();                                     // (carries the vmcnt(0) for the LDS-DMA)
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, (t + 1) * BK);
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            sc_h8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const sc_h8*>(sb + ((a_base ^ (kk * 64)) + i * 2048));
                b[i] = *reinterpret_cast<const sc_h8*>(sb + ((b_base ^ (kk * 64)) + i * 2048));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n .. n+3] for m = m0 + mi*16 + rl, n = n0 + ni*16 + g*4 ----
    const int m0 = tm * BM + wm * 64, n0 = tn * BN + wn * 64;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int n = n0 + nj * 16 + g * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            sc_h4 b4 = *reinterpret_cast<const sc_h4*>(bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (float)b4[e];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + mi * 16 + rl;
            if (m < M) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = epi_apply(acc[mi][nj][e] + bv[e], EPI);
                if (R) {
                    sc_h4 r4 = *reinterpret_cast<const sc_h4*>(R + (size_t)m * (size_t)ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                }
                if (EPI == SC_EPI_SWIGLU) {
                    // columns n..n+3 hold (gate_{j}, gate_{j+1}, up_{j}, up_{j+1}), j = n/2: write silu(gate)*up to [m][j..j+1]
                    const float g0 = acc[mi][nj][0] + bv[0], g1 = acc[mi][nj][1] + bv[1];
                    const float u0 = acc[mi][nj][2] + bv[2], u1 = acc[mi][nj][3] + bv[3];
                    const sc_h2 o = {(_Float16)(g0 / (1.0f + __expf(-g0)) * u0), (_Float16)(g1 / (1.0f + __expf(-g1)) * u1)};
                    *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
                } else if (OUT_F32) {
                    *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
                } else {
                    sc_h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// v2: 256x256 block tile, 8 waves (2 x 4, each 128x64 = 8x4 MFMA fragments, 128 accumulator VGPRs), K consumed in 32-deep
// steps through a 4-slot LDS ring (4 x (A 16 KiB + W 16 KiB) = 128 KiB, one workgroup per CU).
//   * LDS-DMA (16-byte global_load_lds) of step k+3 is issued right after the ONE barrier of step k — the barrier that
//     (a) publishes step k+1 (every wave did a counted `s_waitcnt vmcnt(4)` for its own share first: never vmcnt(0) in the
//     steady state) and (b) proves nobody still reads slot (k+3)&3 = (k-1)&3.  Raw s_barrier: __syncthreads() would drain
//     the DMA queue.
//   * register pipelining: all 12 fragments of step k+1 are read while the second MFMA cluster of step k runs, into a
//     second register set (sets ping-pong between steps: no copies); the 12 ds_read_b128 and 4 DMA issues are interleaved
//     one-per-MFMA with sched_group_barrier, so a wave never sits in a load-issue burst while its SIMD's matrix pipe idles
//     (PMC before: MFMA pipe 56 % busy with both waves of a SIMD issuing loads right after the barrier).
//   * LDS rows are 64 B (4 granules); granule g of row r sits at r*64 + ((g ^ F[(r>>2)&3]) << 4), F = {0,2,3,1}:
//     conflict-free for the 16-lane service groups of ds_read_b128 (SQ_LDS_BANK_CONFLICT = 0 measured).
//   * tile order: XCD-contiguous chunks, inside a chunk groups of GM = 8 tile-rows with the row index fastest.
// Measured (random operands): 1.09-1.18 PF at K >= 3584, 0.71-0.83 PF at K = 1024 (prologue/epilogue share).
// ------------------------------------------------------------------------------------------------------------------
#ifndef SC_GEMM_GM
#define SC_GEMM_GM 8          // tile-rows per raster group
#endif
#ifndef SC_GEMM_SETPRIO
#define SC_GEMM_SETPRIO 0     // s_setprio(1) around the MFMA clusters
#endif
#ifndef SC_GEMM_NS
#define SC_GEMM_NS 4          // LDS ring slots of the 256x256 kernel (5 x 32 KiB = the whole 160 KiB LDS)
#endif
#ifndef SC_GEMM_MPL
#define SC_GEMM_MPL 1         // MFMAs per interleaved load in the second cluster
#endif
constexpr int BM2 = 256, BN2 = 256, BK2 = 32;
constexpr int HALF2 = BM2 * BK2 * 2;            // 16 KiB per operand per stage
constexpr int STAGE2 = 2 * HALF2;               // 32 KiB

__device__ __forceinline__ int swzF(int x) { return (0x78 >> (2 * (x & 3))) & 3; }

// 16-byte LDS-DMA through a raw buffer resource (base, extent in bytes): lane address = base + voff + soff, destination = the
// wave-uniform LDS pointer + lane * 16; lanes past the extent write zeros.  (A free function: an opaque __amdgpu_buffer_rsrc_t
// inside a lambda of the kernel silently drops the kernel's host stub.)
// buffer resource from values the compiler may have computed on the vector ALU (scalar registers are scarce in the persistent GEMM):
// force base and extent into SGPRs so that buffer instructions need no waterfall loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, int extent) {
    const unsigned long long b = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(extent), 0x00020000);
}

__device__ __forceinline__ void lds_load16(const void* base, unsigned extent, char* lds, unsigned voff, unsigned soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)extent, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, (int)soff, 0, 0);
}

// WR = wave rows: 2 -> 256x256 tile, 8 waves, 4-slot ring (128 KiB, one workgroup per CU, DMA 3 steps ahead);
//                  1 -> 128x256 tile, 4 waves, 3-slot ring (72 KiB, TWO workgroups per CU, DMA 2 steps ahead): for short-K GEMMs
//                       (ViT, K = 1024: 32 steps) one workgroup's prologue / epilogue then overlaps the other's main loop.
// PERSIST: the grid is one workgroup per CU and every workgroup walks tiles vb = blockIdx.x, + gridDim.x, ... (same tile order and
//          XCD affinity as the one-tile-per-workgroup launch).  The K-step ring simply continues across the tile boundary: the DMA
//          issued DIST steps ahead in the last steps of a tile fetches the first steps of the NEXT tile, so its data lands while the
//          epilogue (no LDS) stores the finished tile, and the next tile starts without the ~2 us cold prologue (K = 1024: 32 steps
//          per tile, the prologue was ~9 % of a tile).
template <int EPI, bool OUT_F32, int WR, bool PERSIST = false>
__global__ __launch_bounds__(WR * 256, 2) void k_gemm256(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                         const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                         void* __restrict__ Cout, int ldc, int M, int N, int K, int tilesN, int a_grp,
                                                         int a_grp_stride, int a_grp_off, int GM, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = WR * 256;                    // threads
    constexpr int NS = WR == 2 ? SC_GEMM_NS : 3;    // ring slots
    constexpr int DIST = NS - 1;                    // DMA issue distance in K-steps
    constexpr int BMx = 128 * WR;
    constexpr int AH = BMx * BK2 * 2;               // A bytes per stage
    constexpr int STG = AH + HALF2;                 // + W bytes per stage (256 rows)
    constexpr int GW = 4 / WR;                      // W granules per thread per stage (A: always 2)
    const int nwg = PERSIST ? ntiles : (int)gridDim.x;
    auto tile_of = [&](int bid, int& tm_, int& tn_) {
        const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        // grouped raster: GM tile-rows per group, tile-row index fastest, so the tiles an XCD runs concurrently form a compact patch
        const int tilesM = nwg / tilesN;
        const int grp = swz / (GM * tilesN), within = swz - grp * (GM * tilesN);
        const int gm = (tilesM - grp * GM) < GM ? (tilesM - grp * GM) : GM;
        tm_ = grp * GM + within % gm; tn_ = within / gm;
    };
    int vb = blockIdx.x, tm, tn;
    tile_of(vb, tm, tn);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // staging: granule q = j*NT + tid -> row q>>2, physical slot q&3
    const _Float16* a_src[2];
    const _Float16* w_src[GW];
    // PERSIST: no per-lane pointers at all - constant per-lane 32-bit offsets inside a tile + uniform (SGPR) tile / K-step offsets
    // through buffer_load ... lds; rows >= M are out of the resource's extent and arrive as zeros (they are never stored).
    unsigned a_vo[2], w_vo[GW];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int q = j * NT + tid, r = q >> 2; a_vo[j] = ((unsigned)r * (unsigned)lda + (unsigned)(((q & 3) ^ swzF(r >> 2)) * 8)) * 2u; }
#pragma unroll
    for (int j = 0; j < GW; ++j) { const int q = j * NT + tid, r = q >> 2; w_vo[j] = ((unsigned)r * (unsigned)K + (unsigned)(((q & 3) ^ swzF(r >> 2)) * 8)) * 2u; }
    const unsigned a_ext = (unsigned)M * (unsigned)lda * 2u, w_ext = (unsigned)N * (unsigned)K * 2u;
    int tm_nx = 0, tn_nx = 0;                       // PERSIST: next tile of this workgroup
    auto set_src = [&](int tm_, int tn_, const _Float16* (&as)[2], const _Float16* (&ws)[GW]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = j * NT + tid;
            const int r = q >> 2;
            const int sl = (q & 3) ^ swzF(r >> 2);
            int ar = tm_ * BMx + r;
            ar = ar < M ? ar : M - 1;
            if (a_grp > 0) ar = (ar / a_grp) * a_grp_stride + a_grp_off + (ar % a_grp);
            as[j] = A + (size_t)ar * (size_t)lda + sl * 8;
        }
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            const int q = j * NT + tid;
            const int r = q >> 2;
            const int sl = (q & 3) ^ swzF(r >> 2);
            ws[j] = W + (size_t)(tn_ * BN2 + r) * (size_t)K + sl * 8;
        }
    };
    if (!PERSIST) set_src(tm, tn, a_src, w_src);
    const int nk = K / BK2;
    int sb = 0;                                     // ring slot of K-step 0 of the current tile (PERSIST: the ring runs on across tiles)
    bool has_next = false;
    auto slot_of = [&](int ks) { return NS == 4 ? ((sb + ks) & 3) : ((sb + ks) % NS); };
    auto issue_w = [&](int ks) {
        char* base = smem + slot_of(ks) * STG + AH;
        if (PERSIST) {
            const bool nx = ks >= nk;                // wave-uniform: the step belongs to the next tile
            const unsigned so = ((unsigned)(nx ? tn_nx : tn) * (unsigned)(BN2 * K) + (unsigned)((nx ? ks - nk : ks) * BK2)) * 2u;
#pragma unroll
            for (int j = 0; j < GW; ++j) lds_load16(W, w_ext, base + (j * NT + wave * 64) * 16, w_vo[j], so);
        } else {
#pragma unroll
            for (int j = 0; j < GW; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(w_src[j] + ks * BK2), (lds_ptr_t)(base + (j * NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto issue_a = [&](int ks) {
        char* base = smem + slot_of(ks) * STG;
        if (PERSIST) {
            const bool nx = ks >= nk;
            const unsigned so = ((unsigned)(nx ? tm_nx : tm) * (unsigned)(BMx * lda) + (unsigned)((nx ? ks - nk : ks) * BK2)) * 2u;
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_load16(A, a_ext, base + (j * NT + wave * 64) * 16, a_vo[j], so);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[j] + ks * BK2), (lds_ptr_t)(base + (j * NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    // wait until at most `steps_in_flight` K-steps of this thread's DMA are outstanding (2 + GW issues per step)
    auto wait_steps = [&](int steps_in_flight) {
        if (steps_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (steps_in_flight == 1) { if (GW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else { if (GW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
    };

    // PERSIST, first two K-steps after an epilogue: the loads being waited for (issued before the epilogue) are OLDER than the
    // epilogue's 16 (SwiGLU: 8) global stores, which share vmcnt on gfx9 - allow those stores (and the one younger DMA step) to stay
    // in flight, otherwise the wave stalls until its C tile has reached memory and the stores never overlap the next tile's MFMAs.
    auto wait_after_epilogue = [&]() {
        constexpr int CNT = (EPI == SC_EPI_SWIGLU ? 8 : 16) + (DIST - 2) * (2 + GW);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
    };
    int relax = 0;                                  // K-steps left that may use it (0 after an edge tile: fewer stores were issued)

    const int rl = lane & 15, g = lane >> 4;
    const int frag = rl * 64 + ((g ^ swzF(rl >> 2)) << 4);
    const int a_off = wr * 128 * 64 + frag;                 // + half*4096 + mi*1024
    const int b_off = AH + wc * 64 * 64 + frag;             // + ni*1024

    sc_f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sc_f4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: steps 0..DIST-1 in flight, step 0 landed for every wave, first fragments in registers ----
    issue_w(0); issue_a(0);
    if (nk > 1) { issue_w(1); issue_a(1); }
    if (DIST > 2 && nk > 2) { issue_w(2); issue_a(2); }
    if (DIST > 3 && nk > 3) { issue_w(3); issue_a(3); }
    wait_steps((nk < DIST ? nk : DIST) - 1);
    __builtin_amdgcn_s_barrier();
    // fragment sets A / B ping-pong between consecutive K-steps (no register copies)
    sc_h8 aloA[4], ahiA[4], bcA[4], aloB[4], ahiB[4], bcB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bcA[i] = *reinterpret_cast<const sc_h8*>(smem + b_off + i * 1024);
        aloA[i] = *reinterpret_cast<const sc_h8*>(smem + a_off + i * 1024);
        ahiA[i] = *reinterpret_cast<const sc_h8*>(smem + a_off + 4096 + i * 1024);
    }
    // one K-step: multiply this step's fragments (alo, ahi, bc).  After the barrier that publishes step ks+1, its 12 fragment
    // reads and the DMA issues of step ks+DIST are INTERLEAVED one-per-MFMA into the second MFMA cluster
    // (sched_group_barrier), so no wave ever sits in a load-issue burst while its SIMD's matrix pipe idles.
    auto kstep = [&](auto sid, int ks, bool steady, sc_h8(&alo)[4], sc_h8(&ahi)[4], sc_h8(&bc)[4], sc_h8(&alo_n)[4], sc_h8(&ahi_n)[4], sc_h8(&bn)[4]) {
        if (SC_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[j], alo[i], acc[i][j], 0, 0, 0);
        if (SC_GEMM_SETPRIO) __builtin_amdgcn_s_setprio(0);
        // step ks+1 must have landed for everyone (steps ks+2 .. ks+DIST-1 may stay in flight); slot of step ks-1 is free afterwards
        if (steady) { if (decltype(sid)::value >= 7) wait_after_epilogue(); else wait_steps(DIST - 2); }
        else { const int newest = (ks + DIST - 1) < (nk - 1) ? (ks + DIST - 1) : (nk - 1); wait_steps(newest - (ks + 1)); }
        __builtin_amdgcn_s_barrier();
        if (steady) { issue_w(ks + DIST); issue_a(ks + DIST); }
        else if (ks + DIST < nk) { issue_w(ks + DIST); issue_a(ks + DIST); }
        {   // unconditional (after the last step the fragments are simply unused): a branch here makes hipcc drain lgkmcnt(0)
            const char* sn = smem + slot_of(ks + 1) * STG;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bn[i] = *reinterpret_cast<const sc_h8*>(sn + b_off + i * 1024);
                alo_n[i] = *reinterpret_cast<const sc_h8*>(sn + a_off + i * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) ahi_n[i] = *reinterpret_cast<const sc_h8*>(sn + a_off + 4096 + i * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[j], ahi[i], acc[4 + i][j], 0, 0, 0);
        if (steady) {
            constexpr int SID = decltype(sid)::value;
#pragma unroll
            for (int i = 0; i < 2 + GW; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, SID); __builtin_amdgcn_sched_group_barrier(0x020, 1, SID); }
#pragma unroll
            for (int i = 0; i < (12 - (GW - 2)) / SC_GEMM_MPL; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, SID); __builtin_amdgcn_sched_group_barrier(0x100, SC_GEMM_MPL, SID); }
        }
    };
    for (;;) {                                           // PERSIST: tiles of this workgroup; otherwise exactly one pass
    if (PERSIST) {
        has_next = vb + (int)gridDim.x < ntiles;
        if (has_next) tile_of(vb + gridDim.x, tm_nx, tn_nx);
    }
    int ks = 0;
    if (PERSIST && has_next) {
        if (relax) {                                     // first pair after a full tile's epilogue: its stores may stay in flight
            kstep(std::integral_constant<int, 7>{}, 0, true, aloA, ahiA, bcA, aloB, ahiB, bcB);
            kstep(std::integral_constant<int, 8>{}, 1, true, aloB, ahiB, bcB, aloA, ahiA, bcA);
            ks = 2;
        }
        for (; ks + 1 < nk; ks += 2) {                   // every step is steady: steps >= nk of the issue stream are the next tile's
            kstep(std::integral_constant<int, 5>{}, ks, true, aloA, ahiA, bcA, aloB, ahiB, bcB);
            kstep(std::integral_constant<int, 6>{}, ks + 1, true, aloB, ahiB, bcB, aloA, ahiA, bcA);
        }
    } else {
    for (; ks + 1 + DIST < nk; ks += 2) {                // steady state: both steps of the pair still issue DMA
        kstep(std::integral_constant<int, 0>{}, ks, true, aloA, ahiA, bcA, aloB, ahiB, bcB);
        kstep(std::integral_constant<int, 1>{}, ks + 1, true, aloB, ahiB, bcB, aloA, ahiA, bcA);
    }
    for (; ks + 1 < nk; ks += 2) {
        kstep(std::integral_constant<int, 2>{}, ks, false, aloA, ahiA, bcA, aloB, ahiB, bcB);
        kstep(std::integral_constant<int, 3>{}, ks + 1, false, aloB, ahiB, bcB, aloA, ahiA, bcA);
    }
    if (ks < nk) kstep(std::integral_constant<int, 4>{}, ks, false, aloA, ahiA, bcA, aloB, ahiB, bcB);
    }

    // ---- epilogue.  Each lane owns C[m = m0 + mi*16 + rl][n0 + nj*16 + g*4 .. +3]; bias / activation / residual are applied in
    // fp32 in that ownership.  The fp16 results are then exchanged between the four 16-lane rows of the wave with
    // v_permlane16_swap (and v_permlane32_swap for SwiGLU) so that every lane holds 16 contiguous bytes of one output row and the
    // tile leaves as dwordx4 stores: 16 (8 for SwiGLU) store instructions per lane instead of 32 — the store tail of a
    // one-workgroup-per-CU GEMM is store-ISSUE bound (guide T21; K-sweep: ~15 us fixed per tile before).
    const int m0 = tm * BMx + wr * 128, n0 = tn * BN2 + wc * 64;
    auto pack2 = [](float a, float b) -> unsigned { const sc_h2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); };
    if (OUT_F32) {
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int n = n0 + nj * 16 + g * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) { const sc_h4 b4 = *reinterpret_cast<const sc_h4*>(bias + n); for (int e = 0; e < 4; ++e) bv[e] = (float)b4[e]; }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m = m0 + mi * 16 + rl;
                if (m < M) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = epi_apply(acc[mi][nj][e] + bv[e], EPI);
                    if (R) { const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(R + (size_t)m * (size_t)ldr + n); for (int e = 0; e < 4; ++e) v[e] += (float)r4[e]; }
                    *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    } else if (EPI == SC_EPI_SWIGLU) {
        float bv[4][4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int n = n0 + nj * 16 + g * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nj][e] = bias ? (float)bias[n + e] : 0.f;
        }
        _Float16* Ch = reinterpret_cast<_Float16*>(Cout);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + mi * 16 + rl;
            unsigned d[4];                                      // d[nj] = (silu(g0)*u0, silu(g1)*u1) of output columns (n0>>1) + nj*8 + g*2
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const float g0 = acc[mi][nj][0] + bv[nj][0], g1 = acc[mi][nj][1] + bv[nj][1];
                const float u0 = acc[mi][nj][2] + bv[nj][2], u1 = acc[mi][nj][3] + bv[nj][3];
                d[nj] = pack2(g0 / (1.0f + __expf(-g0)) * u0, g1 / (1.0f + __expf(-g1)) * u1);
            }
            // level 1 (rows of 16 lanes): lane row g -> 4 contiguous columns of nj = (g&1) + {0,2}
            const auto p0 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
            const auto p1 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
            // level 2 (wave halves): lane row g -> the 8 contiguous columns of nj = g
            const auto q0 = __builtin_amdgcn_permlane32_swap(p0[0], p1[0], false, false);
            const auto q1 = __builtin_amdgcn_permlane32_swap(p0[1], p1[1], false, false);
            if (m < M) {
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                const u4v o = {q0[0], q1[0], q0[1], q1[1]};
                *reinterpret_cast<u4v*>(Ch + (size_t)m * (size_t)ldc + (n0 >> 1) + g * 8) = o;
            }
        }
    } else {
        float bv[4][4];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
            const int n = n0 + nj * 16 + g * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[nj][e] = bias ? (float)bias[n + e] : 0.f;
        }
        _Float16* Ch = reinterpret_cast<_Float16*>(Cout);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + mi * 16 + rl;
            const bool live = m < M;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                unsigned lo[2], hi[2];                          // packed halves of nj = 2pr (lo) and 2pr+1 (hi): [0] = cols +0,+1  [1] = cols +2,+3
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nj = 2 * pr + h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = epi_apply(acc[mi][nj][e] + bv[nj][e], EPI);
                    if (R && live) {
                        const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(R + (size_t)m * (size_t)ldr + n0 + nj * 16 + g * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                    }
                    (h ? hi : lo)[0] = pack2(v[0], v[1]);
                    (h ? hi : lo)[1] = pack2(v[2], v[3]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
                if (live) {                                     // lane row g: 8 contiguous columns of nj = 2pr + (g&1), starting at (g>>1)*8
                    typedef unsigned u4v __attribute__((ext_vector_type(4)));
                    const u4v o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u4v*>(Ch + (size_t)m * (size_t)ldc + n0 + (2 * pr + (g & 1)) * 16 + (g >> 1) * 8) = o;
                }
            }
        }
    }
    if (!PERSIST || !has_next) break;
    // next tile of this workgroup: its first DIST steps are already in flight and its step-0 fragments are in set A (nk is even)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = sc_f4{0.f, 0.f, 0.f, 0.f};
    relax = (tm * BMx + BMx <= M) ? 2 : 0;          // full tile: every lane issued all its stores
    vb += gridDim.x;
    tm = tm_nx; tn = tn_nx;
    sb = (sb + nk) % NS;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// v3, k_gemm_fat - the DEFAULT kernel for fp16 output with K % 128 == 0 (every large GEMM of the ViT and the LLM):
// the same 256x256 tile, but FOUR waves (2 x 2), one per SIMD with the whole 512-register file, each owning 128 x 128 of the tile:
// 64 accumulator tiles (all 256 AGPRs), K = 64 per iteration out of two 64-KiB LDS buffers, 32 ds_read_b128 per 128 MFMAs (0.25 KiB
// of LDS fragment traffic per MFMA against 0.375 for the 128x64 wave tiles of k_gemm256).  The K loop is scheduled by hand (asm MFMAs
// with the accumulator tied in place, asm fragment reads, explicit waits, DMA rounds spread one per 7 MFMAs); what that bought, what
// it needed (whole 128-byte lines per DMA pair, no bursts) and the hazards it runs into are written up in DESIGN.md section 4.
// ------------------------------------------------------------------------------------------------------------------
// (the per-tile phase trace of tools/trace_fat.py hooks in at three points of the tile loop; the product build defines the hook as nothing - the
// diagnostic build force-includes tools/diag/fat_trace.h, which holds the trace buffer, the stamp and the read-back entry point)
#ifndef FAT_STAMP
#define FAT_STAMP(slot)
#endif
#ifndef SC_FAT_SNAKE
#define SC_FAT_SNAKE 1
#endif
template <int EPI, bool PERSIST, bool RES = false>
__global__ __launch_bounds__(256, 1) void k_gemm_fat(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                     const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                     void* __restrict__ Cout, int ldc, int M, int N, int K, int tilesN, int GM, int ntiles,
                                                     const float* __restrict__ tab, int pos0, int lead_cols, float col_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // RES: the launch has a residual operand (R != null) - a template parameter, not a run-time branch: with both sides of the epilogue compiled into
    // one kernel the live ranges of the residual side count against the 256 VGPRs of the other one (the k|v rotary instantiation carried 280 B of
    // scratch per lane for a residual branch its launcher can never take; the plain ones 20 - 36 B)
    // (the column-scale instantiation - CLIP's fused q|k|v projection, 46 launches per step - keeps the never-taken residual side compiled in: without it
    //  hipcc allocates its epilogue differently and the launch is 0.7 % SLOWER, 5 + 5 interleaved rounds in both orders, profiles/r06_run_m_*; the rotary
    //  and plain instantiations do not care)
    constexpr bool HAS_R = RES || EPI == SC_EPI_COLSCALE;
    static_assert(!RES || EPI == SC_EPI_NONE || EPI == SC_EPI_QUICK_GELU || EPI == SC_EPI_GELU_ERF, "residual: plain / activation epilogues only");
    // virtual block vb -> tile (same XCD-aware grouped order for the one-tile-per-workgroup launch and the persistent walk
    // vb = blockIdx.x, + gridDim.x, ...; gridDim.x is a multiple of 8 there, so a workgroup stays on the XCD slice of its tiles)
    const int tilesM = ntiles / tilesN;
    auto tile_of = [&](int vb, int& tm_, int& tn_) {
        const int xcd = vb & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (vb >> 3);
        const int grp = swz / (GM * tilesN), within = swz - grp * (GM * tilesN);
        const int gm = (tilesM - grp * GM) < GM ? (tilesM - grp * GM) : GM;
        tm_ = grp * GM + within % gm; tn_ = within / gm;
    };
    int vb = blockIdx.x, tm, tn;
    tile_of(vb, tm, tn);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // LDS: two buffers of [A | W], each operand as two K-half planes of 256 rows x 64 B (K = 64 per iteration); within a plane the
    // 16-byte chunk c of row r sits at r * 64 + ((c ^ swzF(r >> 2)) << 4) (conflict-free ds_read_b128 fragments, lane-linear DMA).
    // One DMA instruction moves 64 rows x 64 B of one plane; the two planes of the same rows are issued back to back, so both
    // halves of every 128-byte line of the operand are consumed while the line is in flight / in the L1 (fetching the halves a
    // K-step apart, as with K = 32 stages, costs twice the L2 -> L1 line traffic).
    constexpr int PL = 256 * 64, OPB = 2 * PL, BUF = 2 * OPB, SLAB_OFF = 2 * BUF;        // + 8 KiB per wave behind the two buffers: epilogue slabs (below)
    const int dr = tid >> 2, dc = (tid & 3) ^ swzF(tid >> 4);
    // buffer resources are rebased per tile (base = first row of the tile, extent = its valid rows), so operands of any size work
    // with 32-bit offsets and rows past M / N read as zeros; the four 64-row groups of a DMA round set get their own lane offsets
    unsigned a_vo[4], w_vo[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        a_vo[rg] = ((unsigned)(rg * 64 + dr) * (unsigned)lda + (unsigned)(dc * 8)) * 2u;
        w_vo[rg] = ((unsigned)(rg * 64 + dr) * (unsigned)K + (unsigned)(dc * 8)) * 2u;
    }
    const _Float16 *At, *Wt, *At_nx = A, *Wt_nx = W;
    unsigned a_ext, w_ext, a_ext_nx = 0, w_ext_nx = 0;
    auto tile_src = [&](int tm_, int tn_, const _Float16*& At_, const _Float16*& Wt_, unsigned& ae, unsigned& we) {
        At_ = A + (size_t)tm_ * (size_t)BM2 * (size_t)lda;
        Wt_ = W + (size_t)tn_ * (size_t)BN2 * (size_t)K;
        const int ar = (M - tm_ * BM2) < BM2 ? (M - tm_ * BM2) : BM2, wrw = (N - tn_ * BN2) < BN2 ? (N - tn_ * BN2) : BN2;
        ae = (unsigned)ar * (unsigned)lda * 2u; we = (unsigned)wrw * (unsigned)K * 2u;
    };
    tile_src(tm, tn, At, Wt, a_ext, w_ext);
    int tm_nx = 0, tn_nx = 0;
    bool has_nx = PERSIST && vb + (int)gridDim.x < ntiles;
    if (has_nx) { tile_of(vb + (int)gridDim.x, tm_nx, tn_nx); tile_src(tm_nx, tn_nx, At_nx, Wt_nx, a_ext_nx, w_ext_nx); }
    const int nk = K >> 6;
    const int rl = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    // fragment addresses: [K half h][buffer X]; + i * 1024 per 16-row tile
    unsigned a_ad[2][2], b_ad[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const unsigned frag = (unsigned)(rl * 64 + ((g ^ swzF(rl >> 2)) << 4));
            a_ad[h][X] = lds0 + (unsigned)(X * BUF + h * PL) + (unsigned)(wr * 128 * 64) + frag;
            b_ad[h][X] = lds0 + (unsigned)(X * BUF + OPB + h * PL) + (unsigned)(wc * 128 * 64) + frag;
        }

    sc_f4 acc[8][8];
    int tcount = 0;                                             // tiles done by this workgroup

    // The main loop is laid out by hand: MFMAs (accumulators tied in place in AGPRs), fragment reads and waits are volatile asm
    // in program order, because hipcc's allocator otherwise rotates the 256 accumulation registers through VGPRs every step.
#define FAT_RD(dst, ad, I) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"((I) * 1024))
#define FAT_MM(I, J, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(b[J]), "v"(a[I]))
    // first K half of a tile's FIRST iteration: C = 0 as an inline constant, the accumulator is only written - no 256 v_accvgpr_write per
    // tile to zero it (they sat in the epilogue's shadow-less tail: ~0.6 us of a ~30 us tile at K = 1024)
#define FAT_MM0(I, J, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(b[J]), "v"(a[I]))
    // DMA round t (0..7: W, 8..15: A) of iteration k into buffer X.  Iterations past the end of the tile fetch the first
    // iterations of the workgroup's NEXT tile (persistent walk: its data lands while this tile's epilogue stores), or nothing
    // (extent 0 -> zeros) after the last tile: every loop iteration is identical (no tail code: hipcc shuffles all 256
    // accumulators around any conditional tail).
    auto dma = [&](int t, int X, int k) {
        const bool nx = k >= nk;
        const unsigned ko = (unsigned)(nx ? k - nk : k) * 128u + (unsigned)(t & 1) * 64u;          // K half = t & 1
        const int rg = (t & 7) >> 1, lo = (t & 1) * PL + (rg * 256 + wave * 64) * 16;           // row group = (t & 7) >> 1
        if (t < 8) lds_load16(nx ? Wt_nx : Wt, nx ? w_ext_nx : w_ext, smem + X * BUF + OPB + lo, w_vo[rg], ko);
        else lds_load16(nx ? At_nx : At, nx ? a_ext_nx : a_ext, smem + X * BUF + lo, a_vo[rg], ko);
    };
#pragma unroll
    for (int t = 0; t < 16; ++t) dma(t, 0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) dma(t, 1, 1);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    sc_h8 a0[8], b0[8], a1[8], b1[8];                           // fragment sets of the two K halves
    FAT_RD(b0[0], b_ad[0][0], 0); FAT_RD(b0[1], b_ad[0][0], 1); FAT_RD(b0[2], b_ad[0][0], 2); FAT_RD(b0[3], b_ad[0][0], 3);
    FAT_RD(b0[4], b_ad[0][0], 4); FAT_RD(b0[5], b_ad[0][0], 5); FAT_RD(b0[6], b_ad[0][0], 6); FAT_RD(b0[7], b_ad[0][0], 7);
    FAT_RD(a0[0], a_ad[0][0], 0); FAT_RD(a0[1], a_ad[0][0], 1); FAT_RD(a0[2], a_ad[0][0], 2); FAT_RD(a0[3], a_ad[0][0], 3);
    FAT_RD(a0[4], a_ad[0][0], 4); FAT_RD(a0[5], a_ad[0][0], 5); FAT_RD(a0[6], a_ad[0][0], 6); FAT_RD(a0[7], a_ad[0][0], 7);
    // iteration k on buffer X, 128 MFMAs t = 0..127 (t < 64: K half 0 from a0/b0, then K half 1 from a1/b1):
    //   t = 0..15   + the 16 fragment reads of K half 1                      | t = RB: all reads of buffer X are done -> barrier
    //   t = RB..    + the 16 DMA rounds of iteration k + 2 into buffer X     | t = RC: iteration k + 1 has landed -> wait + barrier
    //   t = RC..    + the 16 fragment reads of (k + 1, K half 0) from the other buffer (a0/b0 are free after t = 63)
    constexpr int RB = 20, RC = 100, DS = 7, RS = 1;                     // DS: MFMAs per DMA round, RS: MFMAs per fragment read (swept in round 1: profiles/r01_run143)
    constexpr int DMA_BEFORE_RC = (RC - RB + DS - 1) / DS < 16 ? (RC - RB + DS - 1) / DS : 16;
    static_assert(RB + 15 * DS < 128 && RC + 15 * RS < 128 && 15 * RS < RB, "schedule does not fit the iteration");
    auto iter = [&](auto Xc, auto Fc, int k) {
        constexpr int X = decltype(Xc)::value;
        constexpr bool FIRST = decltype(Fc)::value;
        // (nk made opaque here: for the peeled iterations `fin` is the same for every tile, hipcc computed it once in the kernel prologue,
        // kept it in a VGPR across everything, spilled it - and reloaded it here behind a vmcnt(0) that drained the DMA pipeline once per
        // tile in the SwiGLU and rotary kernels)
        int nk_here = nk;
        asm volatile("" : "+s"(nk_here));
        const int fin = (k + 1 >= nk_here) ? 1 : 0;
        const int relax = (PERSIST && k == 0 && tcount > 0) ? (HAS_R && R ? 2 : 1) : 0;       // stores of the previous tile's epilogue may still be in flight
        // s_nop: the accumulators are zeroed by VALU writes in the loop preheader and the hazard recognizer does not know that the asm
        // below reads them as MFMA SrcC; everything after this statement is in the loop body, so three wait states are guaranteed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1" ::: "memory");
#pragma unroll
        for (int hi = 0; hi < 16; ++hi)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            // snake order over the wave's 8 x 8 accumulator tiles: odd rows walk the columns backwards, so that exactly ONE operand register changes
            // from every MFMA to the next (at a row boundary too).  Same sums per accumulator; +1.1 % on a register-resident MFMA loop at the
            // package power cap (tools/probes/probe_mfma_energy.hip, profiles/r05_run_aa_*)
            const int t = hi * 8 + jj, i = hi & 7, j = SC_FAT_SNAKE && (hi & 1) ? 7 - jj : jj;
            if (t == RB) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            if (t == RC) {
                // iteration k + 1 (issued one iteration ago) has landed when at most the DMA rounds of this iteration are in flight; in
                // the first iteration after an epilogue its C stores (younger than that DMA, same in-order counter) may stay in flight too
                // (relax = 2: the tile also issued the 4 DMA rounds of its first residual row tile behind that DMA)
                if (X == 0) asm volatile("v_cmp_ne_u32 vcc, 0, %3\n\ts_cbranch_vccz .Lfat_w%=\n\tv_cmp_ne_u32 vcc, 1, %3\n\ts_cbranch_vccz .Lfat_v%=\n\ts_waitcnt vmcnt(%2)\n\ts_branch .Lfat_x%=\n"
                                        ".Lfat_v%=:\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lfat_x%=\n.Lfat_w%=:\n\ts_waitcnt vmcnt(%0)\n.Lfat_x%=:"
                                        ::"n"(DMA_BEFORE_RC), "n"(DMA_BEFORE_RC + 1 + (EPI == SC_EPI_SWIGLU ? 16 : 32)), "n"(DMA_BEFORE_RC + 1 + 4 + (EPI == SC_EPI_SWIGLU ? 16 : 32)),
                                        "v"(relax) : "vcc", "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_BEFORE_RC) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (t < 64) { if (FIRST) FAT_MM0(i, j, a0, b0); else FAT_MM(i, j, a0, b0); }
            else if (X == 0 || t < 127) FAT_MM(i, j, a1, b1);
            else    // last MFMA of an iteration pair: after the tile's final one, drain the MFMA pipe INSIDE the same asm statement
                    // (the asm MFMAs are invisible to the hazard recognizer, and the compiler is free to put accumulator moves
                    // right behind any separate drain statement)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_cmp_ne_u32 vcc, 0, %3\n\ts_cbranch_vccz .Lfat_nodrain%=\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n"
                             ".Lfat_nodrain%=:" : "+a"(acc[7][j]) : "v"(b1[j]), "v"(a1[7]), "v"(fin) : "vcc");
            if (t < 16 * RS && t % RS == 0) { const int u = (t / RS) & 7; if (t / RS < 8) FAT_RD(b1[u], b_ad[1][X], u); else FAT_RD(a1[u], a_ad[1][X], u); }
            if (t >= RB && t < RB + 16 * DS && (t - RB) % DS == 0) dma((t - RB) / DS, X, k + 2);
            if (t >= RC && t < RC + 16 * RS && (t - RC) % RS == 0) {
                const int u = ((t - RC) / RS) & 7;
                if ((t - RC) / RS < 8) FAT_RD(b0[u], b_ad[0][X ^ 1], u); else FAT_RD(a0[u], a_ad[0][X ^ 1], u);
            }
        }
    };
    for (;;) {                                                  // tiles of this workgroup (one unless PERSIST)
    // The tile's bias goes to LDS by DMA now (each wave its own 128 columns into the first KiB of its residual slab, which is idle
    // during the K loop; lanes >= 16 and a null bias are out of range -> zeros): a vector load in the epilogue would make hipcc wait
    // (vmcnt is in order) for the DMA prefetch of the next tile and for the C stores issued before it.  The wait at RC of iteration 0
    // covers this DMA; the previous tile's epilogue has consumed everything it read from the slab.
    {
        const int n0b = __builtin_amdgcn_readfirstlane(tn * BN2 + wc * 128);
        lds_load16(bias ? bias + n0b : W, bias ? 256u : 0u, smem + SLAB_OFF + wave * 8192 + 4096, (unsigned)lane * 16u, 0u);
        // ... and the first 16-row tile of the wave's residual block to its C slab (4 rounds of 4 rows x 256 B; the counted wait of the
        // next iteration 0 knows about them, `relax` = 2): its ~2.6 us of HBM latency pass under the K loop instead of at the head of
        // the epilogue.  Slab layout applied at the source: LDS position (row, chunk p) receives chunk p ^ row.
        if (HAS_R && R) {
            const int m0b = __builtin_amdgcn_readfirstlane(tm * BM2 + wr * 128);
            const int rvb = (M - m0b) < 0 ? 0 : ((M - m0b) > 16 ? 16 : (M - m0b));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned row = (unsigned)(i * 4 + (lane >> 4));
                lds_load16(R + (size_t)m0b * (size_t)ldr + n0b, (unsigned)rvb * (unsigned)ldr * 2u, smem + SLAB_OFF + wave * 8192 + i * 1024,
                           row * (unsigned)ldr * 2u + ((((unsigned)lane ^ row) & 15u) << 4), 0u);
            }
        }
    }
    FAT_STAMP(0);
    iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);    // (writes every accumulator: nothing to zero)
    iter(std::integral_constant<int, 1>{}, std::false_type{}, 1);
    for (int k = 2; k < nk; k += 2) {                           // nk is even (dispatch)
        iter(std::integral_constant<int, 0>{}, std::false_type{}, k);
        iter(std::integral_constant<int, 1>{}, std::false_type{}, k + 1);
    }
    // the asm MFMAs are invisible to the hazard recognizer: drain before reading acc
    // after the last tile: the (zero-fill) DMA rounds of the last two iterations must not land in the LDS of the workgroup that
    // follows this one on the CU
    FAT_STAMP(1);
    if (!has_nx) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: lane owns C[m0 + mi*16 + rl][n0 + nj*16 + g*4 .. +3], 8 x 8 tiles per wave.  Bias, residual and the stores
    // go through per-wave buffer resources (null operand / rows >= M -> extent says out of range -> reads 0 / store dropped), so
    // the code is branch-free and the residual loads of a whole column pair are in flight together. ----
    {
    const int m0 = __builtin_amdgcn_readfirstlane(tm * BM2 + wr * 128), n0 = __builtin_amdgcn_readfirstlane(tn * BN2 + wc * 128);
    const int rv = (M - m0) < 0 ? 0 : ((M - m0) > 128 ? 128 : (M - m0));                  // valid rows of this wave's sub-tile
    // The range check of a buffer access covers the VGPR offset only (not the scalar offset), so the row part of every address
    // stays in the VGPR offset and only column steps go to the scalar one.  The row steps are made opaque per tile: otherwise
    // hipcc precomputes all 64 row offsets outside the tile loop and spills them.
    int rstep_c = 32 * ldc, rstep_r = 32 * ldr, rl_e = rl, g_e = g;     // (lane coordinates too: three hoisted offsets were spilled
    asm volatile("" : "+v"(rstep_c), "+v"(rstep_r), "+v"(rl_e), "+v"(g_e));  //  and their reload waited for the whole DMA prefetch)
    auto pack2 = [](float x, float y) -> unsigned { const sc_h2 h = {(_Float16)x, (_Float16)y}; return __builtin_bit_cast(unsigned, h); };
    auto h4f = [](sc_u2 v, float (&f)[4]) {
        const unsigned x0 = v.x, x1 = v.y;                      // (bit_cast straight from v[1] miscompiles to element 0 with this hipcc)
        const sc_h2 lo = __builtin_bit_cast(sc_h2, x0), hi = __builtin_bit_cast(sc_h2, x1);
        f[0] = (float)lo[0]; f[1] = (float)lo[1]; f[2] = (float)hi[0]; f[3] = (float)hi[1];
    };
    _Float16* Ch = reinterpret_cast<_Float16*>(Cout);
    // Epilogue slabs, per wave 8 KiB behind the K buffers (2 x 64 KiB + 4 x 8 KiB = all 160 KiB of the CU): a 4-KiB C slab and a 4-KiB
    // residual slab whose first KiB receives the tile's bias during the K loop.  The accumulator layout (lane = row rl, 4 columns) is
    // the wrong shape for global memory - adjacent lanes are adjacent ROWS, so a store or load in that layout is 64 separate 8 / 16-byte
    // requests (measured with tools/trace_fat.py: 2 us of a 5-6 us epilogue for the half-line stores, 8 us for the residual loads).  One
    // step = one 16-row tile x the wave's columns: fp16 results are written to the C slab in the accumulator layout and read back along
    // rows, so every store instruction writes whole 128-byte lines (4 rows x 256 B); residual rows arrive the same way (row-major
    // 16-byte loads -> residual slab -> 8-byte reads in the accumulator layout; added in fp32 before the one rounding, as always).
    // Slab rows are 256 B, 16-byte chunk c of row r at c ^ r: conflict-free for both access shapes.
    char* const cslab = smem + SLAB_OFF + wave * 8192;
    char* const rslab = cslab + 4096;
    const char* bslot = rslab + g_e * 8;                                            // + nj * 32: this lane's 4 bias values
    if (EPI == SC_EPI_SWIGLU) {
        // 64 output columns per wave: slab rows are 128 B (two per bank row), chunk c of row r at c ^ (r >> 1); one store = 8 rows x 128 B
        const __amdgpu_buffer_rsrc_t rs_c = uniform_rsrc(Ch + (size_t)m0 * (size_t)ldc + (n0 >> 1), rv * ldc * 2);
        const int srow = g_e * 2 + (rl_e >> 3), sch = rl_e & 7;                     // row-major view: lane -> row srow (+ 8), chunk sch
        char* cw = cslab + rl_e * 128 + g_e * 4;                                    // + ((nj ^ (rl >> 1)) << 4)
        const char* cr = cslab + srow * 128 + ((sch ^ (srow >> 1)) << 4);           // second half: + 1024, chunk ^ 4
        const char* cr1 = cslab + (srow + 8) * 128 + ((sch ^ ((srow + 8) >> 1)) << 4);
        int st_vo = (srow * ldc + sch * 8) * 2, rstep8_c = 16 * ldc;
        asm volatile("" : "+v"(st_vo), "+v"(rstep8_c));
        const int swz = rl_e >> 1;
        float bv[8][4];
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) h4f(*reinterpret_cast<const sc_u2*>(bslot + nj * 32), bv[nj]);
        auto smath = [&](int mi) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) {
                const sc_f4 c = acc[mi][nj];
                const float g0 = c[0] + bv[nj][0], g1 = c[1] + bv[nj][1], u0 = c[2] + bv[nj][2], u1 = c[3] + bv[nj][3];
                *reinterpret_cast<unsigned*>(cw + ((nj ^ swz) << 4)) = pack2(g0 * __builtin_amdgcn_rcpf(1.0f + __expf(-g0)) * u0, g1 * __builtin_amdgcn_rcpf(1.0f + __expf(-g1)) * u1);
            }
        };
        auto sslab = [&](sc_u4 (&d)[2]) {
            d[0] = *reinterpret_cast<const sc_u4*>(cr);
            d[1] = *reinterpret_cast<const sc_u4*>(cr1);
        };
        auto sstore = [&](int mi, const sc_u4 (&d)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned w0 = d[i][0], w1 = d[i][1], w2 = d[i][2], w3 = d[i][3];      // (store / s_nop pairing: see the note in cstore below)
                __builtin_amdgcn_raw_buffer_store_b128(sc_u4{w0, w1, w2, w3}, rs_c, st_vo + mi * rstep_c + i * rstep8_c, 0, 0);
                asm volatile("s_nop 1" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
            }
        };
        sc_u4 d[2];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            smath(mi);
            if (mi > 0) sstore(mi - 1, d);
            sslab(d);
        }
        sstore(7, d);
    } else {
        const __amdgpu_buffer_rsrc_t rs_c = uniform_rsrc(Ch + (size_t)m0 * (size_t)ldc + n0, rv * ldc * 2);
        const __amdgpu_buffer_rsrc_t rs_r = uniform_rsrc(HAS_R && R ? R + (size_t)m0 * (size_t)ldr + n0 : W, HAS_R && R ? rv * ldr * 2 : 0);
        // SC_EPI_COLSCALE: the wave's 128 columns are scaled if they lie in [0, lead_cols) (the query third of a fused q|k|v projection
        // carries the softmax scale * log2 e into sc_attention_f16's pre-scaled mode: applied to the fp32 sum, ONE rounding)
        const float cscale = (EPI == SC_EPI_COLSCALE && n0 < lead_cols) ? col_scale : 1.0f;
        // slab offsets of this lane: accumulator layout (8 bytes of row rl, column tile nj: XOR nj * 32 into the offset) and row-major
        // view (16 bytes of row g + 4 i at chunk rl: + i * 1024 after XORing i * 64)
        const int acc_o = rl_e * 256 + ((((g_e >> 1) ^ rl_e) & 15) << 4) + (g_e & 1) * 8;
        const int row_o = g_e * 256 + (((rl_e ^ g_e) & 15) << 4);
        int st_vo = (g_e * ldc + rl_e * 8) * 2, rstep4_c = 8 * ldc, ld_vo = (g_e * ldr + rl_e * 8) * 2, rstep4_r = 8 * ldr;
        asm volatile("" : "+v"(st_vo), "+v"(rstep4_c), "+v"(ld_vo), "+v"(rstep4_r));
        auto cmath = [&](int mi, const float (&bv)[8][4], const sc_u2 (&rr)[8], bool has_r) {   // math of row tile mi -> C slab (LDS runs a wave's accesses in order: these writes stay behind the slab reads of tile mi - 1)
            __builtin_amdgcn_sched_barrier(0);                  // keep the steps apart: hipcc otherwise pulls all accumulator reads up front and spills
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = EPI == SC_EPI_COLSCALE ? (acc[mi][nj][e] + bv[nj][e]) * cscale : epi_apply(acc[mi][nj][e] + bv[nj][e], EPI);
                if (has_r) {
                    float r4[4];
                    h4f(rr[nj], r4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                }
                // erf-GELU: keep the fp32 result a value of its own before the fp16 rounding - left alone hipcc fuses the last multiply-add
                // with the conversion (v_fma_mixlo_f16: ONE rounding), which k_gemm256 / k_gemm128 do not: a frame's features would then
                // depend on which kernel its micro-batch size selects (caught by the batching-independence tests)
                if (EPI == SC_EPI_GELU_ERF) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                *reinterpret_cast<sc_u2*>(cslab + (acc_o ^ (nj * 32))) = sc_u2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        };
        auto cmath_r = [&](int mi, const sc_u2 (&bq)[8], const sc_u2 (&rr)[8]) {             // cmath with a residual, bias as fp16 pairs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) {
                float v[4], b4[4], r4[4];
                h4f(bq[nj], b4);
                h4f(rr[nj], r4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (EPI == SC_EPI_COLSCALE ? (acc[mi][nj][e] + b4[e]) * cscale : epi_apply(acc[mi][nj][e] + b4[e], EPI)) + r4[e];
                if (EPI == SC_EPI_GELU_ERF) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                *reinterpret_cast<sc_u2*>(cslab + (acc_o ^ (nj * 32))) = sc_u2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        };
        auto cslab_rd = [&](sc_u4 (&d)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const sc_u4*>(cslab + (row_o ^ (i * 64)) + i * 1024);
        };
        auto cstore = [&](int mi, const sc_u4 (&d)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // gfx950: a VALU write to the data registers of a 16-byte buffer store in the very next instruction corrupts the stored
                // value (profiles/r01_run161) and hipcc does not guard it: the asm keeps the four registers allocated past the store
                unsigned w0 = d[i][0], w1 = d[i][1], w2 = d[i][2], w3 = d[i][3];
                __builtin_amdgcn_raw_buffer_store_b128(sc_u4{w0, w1, w2, w3}, rs_c, st_vo + mi * rstep_c + i * rstep4_c, 0, 0);
                asm volatile("s_nop 1" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
            }
        };
        auto load_res = [&](int mi, sc_u4 (&rg)[4]) {            // residual rows of row tile mi, row-major (4 rows x 256 B per instruction)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ld_vo + mi * rstep_r + i * rstep4_r, 0, 0);
        };
        auto rslab_rw = [&](const sc_u4 (&rg)[4], sc_u2 (&rr)[8]) {  // -> this lane's residual values in the accumulator layout
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<sc_u4*>(rslab + (row_o ^ (i * 64)) + i * 1024) = rg[i];
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) rr[nj] = *reinterpret_cast<const sc_u2*>(rslab + (acc_o ^ (nj * 32)));
        };
        if (EPI == SC_EPI_ROPE_ALL || (EPI == SC_EPI_ROPE && n0 < lead_cols)) {
            // Rotary epilogue (Qwen2 q / k projections): the wave's 128 columns are ONE head; rotate-half pairs column j with j + 64, i.e.
            // the 16-column tile nj with nj + 4 of the SAME lane.  x' = (x_j cos - x_{j+64} sin, x_{j+64} cos + x_j sin) on the fp32
            // accumulators + bias with the fp32 table row of the token's position (row m -> position pos0 + m; [cos(64) | sin(64)], the
            // query table pre-multiplied by the softmax scale * log2 e), rounded to fp16 once.  Same arithmetic as k_rope_f32in
            // (llm_ops.hip) and k_decode_qkv<true> (gemv.hip).
            const __amdgpu_buffer_rsrc_t rs_t = uniform_rsrc(tab + (size_t)(pos0 + m0) * 128, rv * 512);
            int rstep_t = 16 * 512;
            asm volatile("" : "+v"(rstep_t));
            auto load_tab = [&](int st, sc_u4 (&cs)[2], sc_u4 (&sn)[2]) {          // one step = ONE row tile x (two column tiles + their partners)
                const int pr = st >> 3, mi = st & 7;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    cs[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (rl_e * 128 + g_e * 4) * 4 + mi * rstep_t, (2 * pr + h) * 64, 0);
                    sn[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_t, (rl_e * 128 + g_e * 4) * 4 + mi * rstep_t, (2 * pr + h) * 64 + 256, 0);
                }
            };
            auto rstep = [&](int st, const sc_u4 (&cs)[2], const sc_u4 (&sn)[2]) {
                const int pr = st >> 3, mi = st & 7;
                __builtin_amdgcn_sched_barrier(0);
                unsigned la[2], ha[2], lb[2], hb[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int nj = 2 * pr + h;
                    float ba[4], bb[4];
                    h4f(*reinterpret_cast<const sc_u2*>(bslot + nj * 32), ba);
                    h4f(*reinterpret_cast<const sc_u2*>(bslot + (nj + 4) * 32), bb);
                    const sc_f4 c4 = __builtin_bit_cast(sc_f4, cs[h]), s4 = __builtin_bit_cast(sc_f4, sn[h]);
                    float oa[4], ob[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = acc[mi][nj][e] + ba[e], b = acc[mi][nj + 4][e] + bb[e];
                        oa[e] = __builtin_fmaf(-b, s4[e], a * c4[e]);
                        ob[e] = __builtin_fmaf(a, s4[e], b * c4[e]);
                        asm volatile("" : "+v"(oa[e]), "+v"(ob[e]));      // fp32 fma, THEN the fp16 rounding: never one v_fma_mix (see k_decode_qkv)
                    }
                    (h ? ha : la)[0] = pack2(oa[0], oa[1]); (h ? ha : la)[1] = pack2(oa[2], oa[3]);
                    (h ? hb : lb)[0] = pack2(ob[0], ob[1]); (h ? hb : lb)[1] = pack2(ob[2], ob[3]);
                }
                const auto s0 = __builtin_amdgcn_permlane16_swap(la[0], ha[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(la[1], ha[1], false, false);
                unsigned w0 = s0[0], w1 = s1[0], w2 = s0[1], w3 = s1[1];          // (store / s_nop pairing: see the note in step())
                __builtin_amdgcn_raw_buffer_store_b128(sc_u4{w0, w1, w2, w3}, rs_c, (rl_e * ldc + (g_e & 1) * 16 + (g_e >> 1) * 8) * 2 + mi * rstep_c, 2 * pr * 32, 0);
                asm volatile("s_nop 1" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));
                const auto t0 = __builtin_amdgcn_permlane16_swap(lb[0], hb[0], false, false);
                const auto t1 = __builtin_amdgcn_permlane16_swap(lb[1], hb[1], false, false);
                unsigned x0 = t0[0], x1 = t1[0], x2 = t0[1], x3 = t1[1];
                __builtin_amdgcn_raw_buffer_store_b128(sc_u4{x0, x1, x2, x3}, rs_c, (rl_e * ldc + (g_e & 1) * 16 + (g_e >> 1) * 8) * 2 + mi * rstep_c, 2 * pr * 32 + 128, 0);
                asm volatile("s_nop 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            };
            sc_u4 ca[2], sa[2], cb[2], sb[2];
            load_tab(0, ca, sa);
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                load_tab(st + 1, cb, sb); rstep(st, ca, sa);
                if (st + 2 < 16) load_tab(st + 2, ca, sa);
                rstep(st + 1, cb, sb);
            }
        } else if (HAS_R && R) {
            // (the bias values are made opaque on each side of this branch: hipcc otherwise hoists the `acc + bias` adds common to both
            // sides above the branch and parks the sums in AGPRs and scratch)
            // residual rows are requested three row tiles ahead (row tile 0: by DMA at the top of the tile), and before the stores of the step in
            // between (vmcnt retires in order: a load issued behind stores could only be consumed after those stores have completed)
            // (bias kept as fp16 pairs on this side and widened per use: 16 registers less across the steps - this side spilled loaded
            // residual rows behind vmcnt(0) otherwise; made opaque so that hipcc does not hoist the `acc + bias` adds common to both sides
            // of the branch above it)
            sc_u2 bq[8];
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) {
                bq[nj] = *reinterpret_cast<const sc_u2*>(bslot + nj * 32);
                asm volatile("" : "+v"(bq[nj]));
            }
            sc_u4 rg[2][4];
            sc_u2 rr[8];
            load_res(1, rg[1]); load_res(2, rg[0]);
            // row tile 0 came with the tile's DMA (see the top of the tile loop): its values in the accumulator layout, read before the C
            // slab takes the first results
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) rr[nj] = *reinterpret_cast<const sc_u2*>(cslab + (acc_o ^ (nj * 32)));
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                // (the stores are not deferred by a step here - the residual slab traffic of the next row tile covers most of the C slab's
                // read latency, and 16 registers fewer are live across the math: this side of the branch is the tight one)
                cmath_r(mi, bq, rr);
                sc_u4 d[4];
                cslab_rd(d);
                if (mi + 1 < 8) rslab_rw(rg[(mi + 1) & 1], rr);
                if (mi + 3 < 8) load_res(mi + 3, rg[(mi + 1) & 1]);     // (into the registers the slab write above has just read)
                cstore(mi, d);
            }
        } else {
            float bv[8][4];
#pragma unroll
            for (int nj = 0; nj < 8; ++nj) {
                h4f(*reinterpret_cast<const sc_u2*>(bslot + nj * 32), bv[nj]);
                asm volatile("" : "+v"(bv[nj][0]), "+v"(bv[nj][1]), "+v"(bv[nj][2]), "+v"(bv[nj][3]));     // see the note at the branch
            }
            sc_u2 rz[8] = {};
            sc_u4 d[4];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                cmath(mi, bv, rz, false);
                if (mi > 0) cstore(mi - 1, d);
                cslab_rd(d);
            }
            cstore(7, d);
        }
    }
    }
    FAT_STAMP(2);
    ++tcount;
    if (!has_nx) break;
    vb += (int)gridDim.x;
    tm = tm_nx; tn = tn_nx; At = At_nx; Wt = Wt_nx; a_ext = a_ext_nx; w_ext = w_ext_nx;
    has_nx = vb + (int)gridDim.x < ntiles;
    a_ext_nx = 0; w_ext_nx = 0;
    if (has_nx) { tile_of(vb + (int)gridDim.x, tm_nx, tn_nx); tile_src(tm_nx, tn_nx, At_nx, Wt_nx, a_ext_nx, w_ext_nx); }
    }
    // The last iteration of the last tile still issued its 16 fragment reads into a0/b0 (nothing consumes them).  The compiler
    // cannot see asm reads in flight, so keep those registers allocated to the end of the kernel: it must not hand them to the
    // epilogue while the reads can still land.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(a0[i]), "v"(b0[i]));
#undef FAT_RD
#undef FAT_MM
#undef FAT_MM0
}

// ------------------------------------------------------------------------------------------------------------------
// skinny M (<= 32 rows): batched decode / few-token projections.  HBM-bound: W is streamed exactly once, straight from global
// memory into MFMA A-operand registers (every weight element is used once, so LDS staging would only add traffic); the few
// activation rows are the B operand and come out of L2.  One workgroup = one strip of 16 W rows (= 16 output columns); its 4
// waves split K four ways and reduce their 16x16 partial tiles through LDS; wave 0 applies bias / activation / residual /
// SwiGLU and stores.  C/D layout of v_mfma_f32_16x16x32_f16: lane holds D[n = (lane>>4)*4 + r][m = lane&15], i.e. one
// (g0, g1, u0, u1) quad of the interleaved SwiGLU weight per lane.
// ------------------------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32, int MG>
__global__ __launch_bounds__(256) void k_gemm_skinny(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                     const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                     void* __restrict__ Cout, int ldc, int M, int N, int K) {
    __shared__ float red[3][MG][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kc = K / 4, k_lo = wave * kc;
    const _Float16* wp = W + (size_t)(n0 + rl) * (size_t)K + k_lo + g * 8;
    const _Float16* xp[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        int m = mg * 16 + rl;
        m = m < M ? m : M - 1;
        xp[mg] = A + (size_t)m * (size_t)lda + k_lo + g * 8;
    }
    sc_f4 acc[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;                          // K-steps in flight per wave: 8 x 16 B of W per lane
    int k = 0;
    for (; k + U * 32 <= kc; k += U * 32) {
        sc_h8 wf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wf[u] = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp + k + u * 32));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
                acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u], *reinterpret_cast<const sc_h8*>(xp[mg] + k + u * 32), acc[mg], 0, 0, 0);
    }
    for (; k < kc; k += 32) {
        const sc_h8 wf = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp + k));
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
            acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, *reinterpret_cast<const sc_h8*>(xp[mg] + k), acc[mg], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][mg][r][lane] = acc[mg][r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int n = n0 + g * 4;
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mg][r] + red[0][mg][r][lane] + red[1][mg][r][lane] + red[2][mg][r][lane] + (bias ? (float)bias[n + r] : 0.f);
        const int m = mg * 16 + rl;
        if (m >= M) continue;
        if (EPI == SC_EPI_SWIGLU) {
            const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[2], o1 = v[1] / (1.0f + __expf(-v[1])) * v[3];
            const sc_h2 o = {(_Float16)o0, (_Float16)o1};
            *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = epi_apply(v[r], EPI);
            if (R) { const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(R + (size_t)m * (size_t)ldr + n); for (int r = 0; r < 4; ++r) v[r] += (float)r4[r]; }
            if (OUT_F32) *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
            else *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
}

// Round 5: k_gemm_skinny for K known at compile time (the Qwen2-7B widths 3584 and 18944): the K-quarter of a wave is straight-line
// code in batches of BT k-steps with two register buffers.  In the run-time loop above hipcc requests the activation fragments BEHIND
// the weights they multiply (vmcnt retires in order: the first MFMA of a batch waits for the whole batch and then for an L2 round
// trip), consumes a batch before it requests the next one, and peels the trip % 8 remainder into one-k-step trips - the batched
// caption decode (26 sequences) ran its projections at 1.7 - 2.5 TB/s (profiles/r05_*).  Here the queue of a wave is
//   [x fragments of batch 0] [W of batch 0] | per batch b: [x of b + 1] [W of b + 1]  ->  MFMAs of batch b
// (x first: it comes out of L2 and is back long before the weights), same k order per accumulator as the loop above: the sums are
// bit-identical to k_gemm_skinny's.
template <int EPI, bool OUT_F32, int MG, int KW, int BT>      // KW = k-steps (of 32) per wave = K / 4 / 32
__global__ __launch_bounds__(256) void k_gemm_skinny_u(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                       const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                       void* __restrict__ Cout, int ldc, int M, int N) {
    constexpr int K = KW * 32 * 4, NB = (KW + BT - 1) / BT;
    __shared__ float red[3][MG][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int k_lo = wave * (K / 4);
    const _Float16* wp = W + (size_t)(n0 + rl) * (size_t)K + k_lo + g * 8;
    const _Float16* xp[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        int m = mg * 16 + rl;
        m = m < M ? m : M - 1;
        xp[mg] = A + (size_t)m * (size_t)lda + k_lo + g * 8;
    }
    sc_f4 acc[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
    sc_h8 wf[2][BT], xf[2][BT][MG];
    auto load = [&](auto bc) {
        constexpr int B = decltype(bc)::value;
#pragma unroll
        for (int u = 0; u < BT; ++u)
            if (B * BT + u < KW) {
#pragma unroll
                for (int mg = 0; mg < MG; ++mg) xf[B & 1][u][mg] = *reinterpret_cast<const sc_h8*>(xp[mg] + (B * BT + u) * 32);
            }
#pragma unroll
        for (int u = 0; u < BT; ++u)
            if (B * BT + u < KW) wf[B & 1][u] = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp + (B * BT + u) * 32));
    };
    auto run = [&](auto self, auto bc) -> void {
        constexpr int B = decltype(bc)::value;
        if constexpr (B < NB) {
            if constexpr (B + 1 < NB) load(std::integral_constant<int, B + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < BT; ++u)
                if (B * BT + u < KW) {
#pragma unroll
                    for (int mg = 0; mg < MG; ++mg) acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[B & 1][u], xf[B & 1][u][mg], acc[mg], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            self(self, std::integral_constant<int, B + 1>{});
        }
    };
    load(std::integral_constant<int, 0>{});
    run(run, std::integral_constant<int, 0>{});
    if (wave > 0) {
#pragma unroll
        for (int mg = 0; mg < MG; ++mg)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][mg][r][lane] = acc[mg][r];
    }
    __syncthreads();
    if (wave > 0) return;
    const int n = n0 + g * 4;
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[mg][r] + red[0][mg][r][lane] + red[1][mg][r][lane] + red[2][mg][r][lane] + (bias ? (float)bias[n + r] : 0.f);
        const int m = mg * 16 + rl;
        if (m >= M) continue;
        if (EPI == SC_EPI_SWIGLU) {
            const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[2], o1 = v[1] / (1.0f + __expf(-v[1])) * v[3];
            const sc_h2 o = {(_Float16)o0, (_Float16)o1};
            *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = epi_apply(v[r], EPI);
            if (R) { const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(R + (size_t)m * (size_t)ldr + n); for (int r = 0; r < 4; ++r) v[r] += (float)r4[r]; }
            if (OUT_F32) *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
            else *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
}

// Round 5: the batched-decode projection for K = 3584 (Qwen2-7B hidden width) with the ACTIVATIONS RESIDENT IN REGISTERS.
// k_gemm_skinny(_u) re-reads the M x K activation block from L2 for every strip of 16 weight rows: at 26 rows that is twice the bytes of
// the weights themselves through the texture path (gate/up: 543 MB of x for 271 MB of W), and the weight stream stood at 2.3 - 2.9
// TB/s.  Here a workgroup is persistent over a contiguous range of strips (even shares of N / 16 over the CUs); wave w keeps the
// MFMA B fragments of x[:, w * 896 : (w + 1) * 896] - 28 k-steps x MG fragments = 224 VGPRs at MG = 2, one wave per SIMD owns the
// whole 512-register file - for its lifetime, and per strip streams ONLY weights: 28 x 16 B per lane, all requested at once, two
// register buffers so that strip s + 1 is in flight while strip s is multiplied and reduced (4 waves = 4 K-quarters, partial tiles
// through a double-buffered LDS scratch, one barrier per strip; wave 0 applies the epilogue).  Everything a strip's epilogue reads
// (bias, residual) is requested BEFORE the next strip's weights, so that no wait ever drains the weight queue.  Same k order per
// accumulator and same cross-wave sum as k_gemm_skinny: bit-identical results.
template <int EPI, bool OUT_F32, int MG>
__global__ __launch_bounds__(256) void k_gemm_skinny_x(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                       const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                       void* __restrict__ Cout, int ldc, int M, int N, int strips_q, int strips_r) {
    constexpr int KW = 28, K = KW * 32 * 4;
    __shared__ float red[2][3][MG][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rl = lane & 15, g = lane >> 4;
    const int k_lo = wave * (K / 4);
    const int b = blockIdx.x;
    const int s_lo = b * strips_q + min(b, strips_r), s_hi = s_lo + strips_q + (b < strips_r ? 1 : 0);      // the first strips_r workgroups take one strip more
    if (s_lo >= s_hi) return;
    sc_h8 xf[KW][MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        int m = mg * 16 + rl;
        m = m < M ? m : M - 1;
        const _Float16* xp = A + (size_t)m * (size_t)lda + k_lo + g * 8;
#pragma unroll
        for (int u = 0; u < KW; ++u) xf[u][mg] = *reinterpret_cast<const sc_h8*>(xp + u * 32);
    }
    const _Float16* wbase = W + (size_t)rl * (size_t)K + k_lo + g * 8;
    sc_h8 wa[KW], wb[KW];
    sc_h4 ba, bb, ra[MG], rb[MG];
    auto load_w = [&](sc_h8 (&buf)[KW], int strip) {
        const _Float16* wp = wbase + (size_t)strip * (size_t)(16 * K);
#pragma unroll
        for (int u = 0; u < KW; ++u) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const sc_h8*>(wp + u * 32));
    };
    auto load_e = [&](sc_h4& bv, sc_h4 (&rv)[MG], int strip) {          // what the epilogue of `strip` adds (null pointers read W / A: dropped)
        const int n = strip * 16 + g * 4;
        bv = *reinterpret_cast<const sc_h4*>(bias ? bias + n : W);
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            int m = mg * 16 + rl;
            m = m < M ? m : M - 1;
            rv[mg] = *reinterpret_cast<const sc_h4*>(R ? R + (size_t)m * (size_t)ldr + n : A);
        }
    };
    auto compute = [&](const sc_h8 (&buf)[KW], const sc_h4& bv, const sc_h4 (&rv)[MG], int strip) {
        sc_f4 acc[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KW; ++u)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(buf[u], xf[u][mg], acc[mg], 0, 0, 0);
        float (*rd)[MG][4][64] = red[strip & 1];
        if (wave > 0) {
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r) rd[wave - 1][mg][r][lane] = acc[mg][r];
        }
        __syncthreads();
        if (wave > 0) return;
        const int n = strip * 16 + g * 4;
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mg][r] + rd[0][mg][r][lane] + rd[1][mg][r][lane] + rd[2][mg][r][lane] + (bias ? (float)bv[r] : 0.f);
            const int m = mg * 16 + rl;
            if (m >= M) continue;
            if (EPI == SC_EPI_SWIGLU) {
                const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[2], o1 = v[1] / (1.0f + __expf(-v[1])) * v[3];
                const sc_h2 o = {(_Float16)o0, (_Float16)o1};
                *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = epi_apply(v[r], EPI);
                if (R) { for (int r = 0; r < 4; ++r) v[r] += (float)rv[mg][r]; }
                if (OUT_F32) *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
                else *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
    };
    int st = s_lo;
    load_e(ba, ra, st);
    load_w(wa, st);
    while (st + 2 < s_hi) {                // strips st and st + 1 are processed in this trip; st + 2 exists: no load sits behind a condition
        // (the scheduling fences keep the next strip's loads AHEAD of this strip's MFMAs: left alone, hipcc sinks them below the MFMAs to
        //  reuse the registers they free - one buffer instead of two, and every strip starts with a full memory round trip)
        load_e(bb, rb, st + 1); load_w(wb, st + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(wa, ba, ra, st);
        __builtin_amdgcn_sched_barrier(0);
        load_e(ba, ra, st + 2); load_w(wa, st + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(wb, bb, rb, st + 1);
        __builtin_amdgcn_sched_barrier(0);
        st += 2;
    }
    if (st + 1 < s_hi) {
        load_e(bb, rb, st + 1); load_w(wb, st + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(wa, ba, ra, st);
        __builtin_amdgcn_sched_barrier(0);
        compute(wb, bb, rb, st + 1);
    } else {
        compute(wa, ba, ra, st);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 5: the skinny GEMM with BOTH operands staged through per-wave LDS rings by LDS-DMA (k_gemm_skinny_l).
// What bounds the register-operand kernels above is not HBM and not occupancy (8 / 16 waves per strip change nothing,
// profiles/r05_run_k_*) but the CU's address path: the MFMA A / B operand layout makes lane (rl, g) read 16 bytes of ROW rl at k-chunk g, so
// the 16 lanes of a quarter-wave touch 16 different rows - every `global_load_dwordx4` is 64 separate cache-line lookups of 16 bytes
// each, against 8 for a coalesced 1 KiB.  At one line per clock that IS the measured time of the 3584-row projections of the batched
// decode: o 84 instructions x 64 lines x 4 waves = 21.5 k clocks = 10.2 us of 12.2; q|k|v the same in two rounds of workgroups (20.6);
// down 444 x 64 x 4 = 113.7 k clocks = 54 us (50.7 measured).  Here every global access is an LDS-DMA instruction that fetches four rows
// x 256 contiguous bytes (8 lines), and the operand fragments come out of LDS with conflict-free ds_read_b128 (rows of 256 B, the
// 16-byte piece index XORed with the row on the SOURCE side - the layout of k_attn_decode's K ring):
//   * one workgroup = 4 waves = the 4 K-quarters of a strip of 16 weight rows, persistent over a contiguous range of strips;
//   * per wave a ring of PD stages; a stage = 128 k of its K-quarter: W 16 rows x 256 B (4 DMA instructions) + x MG*16 rows x 256 B
//     (4 MG instructions; rows >= M lie outside the buffer extent: zeros, no traffic); the ring runs on across strip boundaries;
//   * no block barrier in the stream: every wait is a hand-counted `s_waitcnt vmcnt` on the wave's own DMA queue, every LDS read inline
//     asm (behind builtin LDS reads hipcc drains the DMA queue with vmcnt(0): see attention_decode.hip);
//   * per strip ONE pair of barriers (without vmcnt(0)) around the exchange of the partial tiles; wave 0 applies the epilogue.  What the
//     epilogue adds (bias, residual tile) reaches wave 0 by two more DMA instructions issued with the strip's first stage - an ordinary
//     load there would make hipcc drain wave 0's queue once per strip;
//   * MFMA operands, k order per accumulator and the cross-wave sum are those of k_gemm_skinny: bit-identical results.
// K % 512 == 0 (K / 4 in whole stages), M <= 16 MG, lda / ldr multiples of 8.
// ------------------------------------------------------------------------------------------------------------------
template <int EPI, bool OUT_F32, int MG, int PD>
__global__ __launch_bounds__(256) void k_gemm_skinny_l(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                       const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                       void* __restrict__ Cout, int ldc, int M, int N, int K, int strips_q, int strips_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WST = 4096, XST = MG * 4096, STB = WST + XST, NOPS = 4 + 4 * MG;
    constexpr int CNT = (PD - 1) * NOPS;                       // DMA ops younger than the stage being waited for
    static_assert(PD >= 2 && CNT < 64, "vmcnt is a 6-bit counter");
    constexpr int RING = 4 * PD * STB, RED = 3 * MG * 4 * 64 * 4, EB = 2048;      // rings | partial tiles | 2 x {residual tile 1 KiB, bias 1 KiB}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int s_lo = b * strips_q + min(b, strips_r), s_hi = s_lo + strips_q + (b < strips_r ? 1 : 0);
    if (s_lo >= s_hi) return;
    const int kq = K >> 2, k_lo = wave * kq, nst = kq >> 7;
    const int total = (s_hi - s_lo) * nst;
    char* ring = smem + wave * (PD * STB);
    const unsigned ring_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)(wave * (PD * STB));
    float* red = reinterpret_cast<float*>(smem + RING);        // [3][MG][4][64]
    char* ebuf = smem + RING + RED;                            // [2][R tile | bias]
    // per-lane source offsets of a stage's granules: granule j lands lane-linear = row j*4 + (lane >> 4), 16-byte slot lane & 15
    unsigned w_vo[4], x_vo[4 * MG];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = j * 4 + (lane >> 4); w_vo[j] = ((unsigned)r * (unsigned)K + (unsigned)(((lane & 15) ^ (r & 15)) * 8)) * 2u; }
#pragma unroll
    for (int j = 0; j < 4 * MG; ++j) { const int r = j * 4 + (lane >> 4); x_vo[j] = ((unsigned)r * (unsigned)lda + (unsigned)(((lane & 15) ^ (r & 15)) * 8)) * 2u; }
    const unsigned w_ext = (15u * (unsigned)K + 128u) * 2u, x_ext = ((unsigned)(M - 1) * (unsigned)lda + 128u) * 2u;
    unsigned f_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) f_off[t] = (unsigned)(rl * 256 + (((t * 4 + g) ^ rl) << 4));
    // running (strip, stage) of the NEXT stage to request
    int is_strip = s_lo, is_st = 0, iq = 0;
    auto issue = [&]() {
        const bool live = iq < total;
        char* dst = ring + (iq % PD) * STB;
        if (wave == 0 && is_st == 0) {                          // what this strip's epilogue adds, into the E buffer of the strip's parity
            char* e = ebuf + ((is_strip - s_lo) & 1) * EB;
            const int n0 = is_strip * 16;
            lds_load16(R ? R + n0 : W, (live && R) ? ((unsigned)(M - 1) * (unsigned)ldr + 16u) * 2u : 0u, e, ((unsigned)(lane >> 1) * (unsigned)ldr + (unsigned)(lane & 1) * 8u) * 2u, 0u);
            lds_load16(bias ? bias + n0 : W, (live && bias) ? 32u : 0u, e + 1024, (unsigned)lane * 16u, 0u);
        }
        const _Float16* wb = W + (size_t)is_strip * (size_t)(16 * K) + k_lo + is_st * 128;
        const _Float16* xb = A + k_lo + is_st * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_load16(wb, live ? w_ext : 0u, dst + j * 1024, w_vo[j], 0u);
#pragma unroll
        for (int j = 0; j < 4 * MG; ++j) lds_load16(xb, live ? x_ext : 0u, dst + WST + j * 1024, x_vo[j], 0u);
        ++iq;
        if (++is_st == nst) { is_st = 0; if (live) ++is_strip; }
    };
    sc_f4 acc[MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < PD; ++u) issue();
    int st = 0, strip = s_lo;
    for (int q = 0; q < total; ++q) {
        const unsigned sa = ring_lds + (unsigned)((q % PD) * STB);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
        sc_u4 wf[4], xf[MG][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(wf[t]) : "v"(sa + f_off[t]));
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[mg][t]) : "v"(sa + f_off[t]), "n"(WST + mg * 4096));
        }
        if constexpr (MG == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[0][3]),
                                                  "+v"(xf[1][0]), "+v"(xf[1][1]), "+v"(xf[1][2]), "+v"(xf[1][3]) :: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf[0][0]), "+v"(xf[0][1]), "+v"(xf[0][2]), "+v"(xf[0][3]) :: "memory");
        issue();                                               // this stage's slot is free: its fragments are in registers
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
                acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_h8, wf[t]), __builtin_bit_cast(sc_h8, xf[mg][t]), acc[mg], 0, 0, 0);
        if (++st < nst) continue;
        // ---- strip done: exchange the partial tiles (barrier pair without vmcnt(0): the DMA ring stays in flight), wave 0 stores ----
        st = 0;
        asm volatile("s_barrier" ::: "memory");                // wave 0 has read the previous strip's partials
        if (wave > 0) {
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wave - 1) * MG + mg) * 4 + r) * 64 + lane] = acc[mg][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave == 0) {
            const int n = strip * 16 + g * 4;
            const char* e = ebuf + ((strip - s_lo) & 1) * EB;   // landed: issued before this strip's first stage, which has been waited for
            const sc_h4 bv = *reinterpret_cast<const sc_h4*>(e + 1024 + g * 8);
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[mg][r];
#pragma unroll
                    for (int w = 0; w < 3; ++w) t += red[((w * MG + mg) * 4 + r) * 64 + lane];
                    v[r] = t + (bias ? (float)bv[r] : 0.f);
                }
                const int m = mg * 16 + rl;
                if (m >= M) continue;
                if (EPI == SC_EPI_SWIGLU) {
                    const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[2], o1 = v[1] / (1.0f + __expf(-v[1])) * v[3];
                    const sc_h2 o = {(_Float16)o0, (_Float16)o1};
                    *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_apply(v[r], EPI);
                    if (R) { const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(e + m * 32 + g * 8); for (int r = 0; r < 4; ++r) v[r] += (float)r4[r]; }
                    if (OUT_F32) *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
                    else *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                }
            }
        }
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
        ++strip;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may be in flight when the workgroup ends (hazard 3 of k_gemm_fat)
}

// k_gemm_skinny_l for the MANY-strip shapes of K = 3584 (gate/up, lm_head): the activations stay resident in registers for the lifetime of the
// workgroup as in k_gemm_skinny_x (one fetch per WORKGROUP instead of one per strip), only the weights run through the per-wave LDS ring -
// 8 stages of 4 KiB: 28 KiB of weight requests in flight per wave.  Same operands, same order: bit-identical to the other skinny kernels.
template <int EPI, bool OUT_F32, int MG>
__global__ __launch_bounds__(256) void k_gemm_skinny_lx(const _Float16* __restrict__ A, int lda, const _Float16* __restrict__ W,
                                                        const _Float16* __restrict__ bias, const _Float16* __restrict__ R, int ldr,
                                                        void* __restrict__ Cout, int ldc, int M, int N, int strips_q, int strips_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = 3584, NST = 7, PD = 8, STB = 4096, CNT = (PD - 1) * 4;
    constexpr int RING = 4 * PD * STB, RED = 3 * MG * 4 * 64 * 4, EB = 2048, NEB = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & 15, g = lane >> 4;
    const int b = blockIdx.x;
    const int s_lo = b * strips_q + min(b, strips_r), s_hi = s_lo + strips_q + (b < strips_r ? 1 : 0);
    if (s_lo >= s_hi) return;
    const int k_lo = wave * (K / 4);
    const int total = (s_hi - s_lo) * NST;
    char* ring = smem + wave * (PD * STB);
    const unsigned ring_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem) + (unsigned)(wave * (PD * STB));
    float* red = reinterpret_cast<float*>(smem + RING);
    char* ebuf = smem + RING + RED;
    unsigned w_vo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = j * 4 + (lane >> 4); w_vo[j] = ((unsigned)r * (unsigned)K + (unsigned)(((lane & 15) ^ (r & 15)) * 8)) * 2u; }
    const unsigned w_ext = (15u * (unsigned)K + 128u) * 2u;
    unsigned f_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) f_off[t] = (unsigned)(rl * 256 + (((t * 4 + g) ^ rl) << 4));
    int is_strip = s_lo, is_st = 0, iq = 0;
    auto issue = [&]() {
        const bool live = iq < total;
        char* dst = ring + (iq % PD) * STB;
        if (wave == 0 && is_st == 0) {
            char* e = ebuf + ((is_strip - s_lo) % NEB) * EB;
            const int n0 = is_strip * 16;
            lds_load16(R ? R + n0 : W, (live && R) ? ((unsigned)(M - 1) * (unsigned)ldr + 16u) * 2u : 0u, e, ((unsigned)(lane >> 1) * (unsigned)ldr + (unsigned)(lane & 1) * 8u) * 2u, 0u);
            lds_load16(bias ? bias + n0 : W, (live && bias) ? 32u : 0u, e + 1024, (unsigned)lane * 16u, 0u);
        }
        const _Float16* wb = W + (size_t)is_strip * (size_t)(16 * K) + k_lo + is_st * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_load16(wb, live ? w_ext : 0u, dst + j * 1024, w_vo[j], 0u);
        ++iq;
        if (++is_st == NST) { is_st = 0; if (live) ++is_strip; }
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) issue();                      // the weight stream starts before the activations are fetched
    sc_h8 xf[NST * 4][MG];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
        int m = mg * 16 + rl;
        m = m < M ? m : M - 1;
        const _Float16* xp = A + (size_t)m * (size_t)lda + k_lo + g * 8;
#pragma unroll
        for (int u = 0; u < NST * 4; ++u) xf[u][mg] = *reinterpret_cast<const sc_h8*>(xp + u * 32);
    }
    int q = 0;
    for (int strip = s_lo; strip < s_hi; ++strip) {
        sc_f4 acc[MG];
#pragma unroll
        for (int mg = 0; mg < MG; ++mg) acc[mg] = sc_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st, ++q) {
            const unsigned sa = ring_lds + (unsigned)((q % PD) * STB);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
            sc_u4 wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[t]) : "v"(sa + f_off[t]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) :: "memory");
            issue();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mg = 0; mg < MG; ++mg)
                    acc[mg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(sc_h8, wf[t]), xf[st * 4 + t][mg], acc[mg], 0, 0, 0);
        }
        asm volatile("s_barrier" ::: "memory");                // wave 0 has read the previous strip's partials
        if (wave > 0) {
#pragma unroll
            for (int mg = 0; mg < MG; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wave - 1) * MG + mg) * 4 + r) * 64 + lane] = acc[mg][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave == 0) {
            const int n = strip * 16 + g * 4;
            const char* e = ebuf + ((strip - s_lo) % NEB) * EB;
            const sc_h4 bv = *reinterpret_cast<const sc_h4*>(e + 1024 + g * 8);
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[mg][r];
#pragma unroll
                    for (int w = 0; w < 3; ++w) t += red[((w * MG + mg) * 4 + r) * 64 + lane];
                    v[r] = t + (bias ? (float)bv[r] : 0.f);
                }
                const int m = mg * 16 + rl;
                if (m >= M) continue;
                if (EPI == SC_EPI_SWIGLU) {
                    const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[2], o1 = v[1] / (1.0f + __expf(-v[1])) * v[3];
                    const sc_h2 o = {(_Float16)o0, (_Float16)o1};
                    *reinterpret_cast<sc_h2*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + (n >> 1)) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = epi_apply(v[r], EPI);
                    if (R) { const sc_h4 r4 = *reinterpret_cast<const sc_h4*>(e + m * 32 + g * 8); for (int r = 0; r < 4; ++r) v[r] += (float)r4[r]; }
                    if (OUT_F32) *reinterpret_cast<sc_f4*>(reinterpret_cast<float*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_f4{v[0], v[1], v[2], v[3]};
                    else *reinterpret_cast<sc_h4*>(reinterpret_cast<_Float16*>(Cout) + (size_t)m * (size_t)ldc + n) = sc_h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI>
int launch_gemm(const void* A, int lda, const void* W, const void* bias, const void* R, int ldr, void* C, int ldc, int M, int N,
                int K, int out_f32, int a_grp, int a_grp_stride, int a_grp_off, hipStream_t s) {
    // hipFuncSetAttribute and the CU count are PER DEVICE: the flags below are indexed by the current device, so a process that drives
    // several GPUs (the reference puts its two model replicas on cuda:0 / cuda:1) raises the LDS limit on each of them
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    static std::atomic<int> force{-1};                       // SC_GEMM_KERNEL=128|256 pins the variant (A/B benchmarking)
    if (force < 0) { const char* e = getenv("SC_GEMM_KERNEL"); force = e ? atoi(e) : 0; }
    if (force == 0 && M <= 32 && K % 128 == 0 && a_grp == 0 && ldc % 4 == 0 && (!R || ldr % 4 == 0)) {      // few rows: stream W once
        const dim3 grid((unsigned)(N / 16)), block(256);
        // both operands through LDS rings (k_gemm_skinny_l): K in whole 512-element quarters-of-stages; SC_SKINNY_LDS=0 pins the register-operand kernels
        static std::atomic<int> lds_on{-1};
        if (lds_on < 0) { const char* e = getenv("SC_SKINNY_LDS"), *gnr = getenv("SC_SKINNY_GENERIC"); lds_on = ((e && e[0] == '0') || (gnr && gnr[0] == '1')) ? 0 : 1; }
        // (K >= 2048: a strip has at least PD stages per wave, so the epilogue buffer of strip s + 2 is requested after strip s has been stored)
        // FEW strips only (N / 16 below two per CU: o, down, q|k|v, kv of the batched decode, the text encoders' K = 4096 projections).  Measured
        // at M = 26, same box, us per launch LDS / register operands (profiles/r05_run_o_skinny_lds.md): q 7.9 / 11.8, kv 6.9 / 11.1, o 8.0 / 12.1,
        // q|k|v 11.5 / 20.5, down 38.8 / 50.6 - but gate/up 67.4 / 59.6 and lm_head 268 / 231: with many strips per workgroup the x-resident
        // register kernel (k_gemm_skinny_x) reads x once per WORKGROUP, this one once per strip (543 MB of x through LDS for 271 MB of weights)
        static std::atomic<int> n_cu_l[16];
        if (n_cu_l[dev] == 0) { int n = 0; n_cu_l[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
        if constexpr (EPI == SC_EPI_NONE || EPI == SC_EPI_SWIGLU) if (lds_on && N / 16 < 2 * n_cu_l[dev] && K % 512 == 0 && K >= 2048 && lda % 8 == 0 && (!R || ldr % 8 == 0) &&
                                                                        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0) {
            const int strips = N / 16, cus = sc_launch_cu_count(n_cu_l[dev], s), gx = strips < cus ? strips : cus;
            static std::atomic<bool> attr_done[16][8];
#define SC_LSL(F32, MGV, PDV, SLOT) do { constexpr int LDSB = 4 * PDV * (4096 + MGV * 4096) + 3 * MGV * 4 * 64 * 4 + 2 * 2048;                                   \
                SC_REQUIRE(K / 512 >= PDV, "sc_gemm_f16: k_gemm_skinny_l needs K / 512 >= %d stages per strip (its epilogue tile is double-buffered by strip parity)", PDV); \
                if (!attr_done[dev][SLOT]) { (void)hipFuncSetAttribute((const void*)k_gemm_skinny_l<EPI, F32, MGV, PDV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); attr_done[dev][SLOT] = true; } \
                hipLaunchKernelGGL((k_gemm_skinny_l<EPI, F32, MGV, PDV>), dim3((unsigned)gx), block, LDSB, s, (const _Float16*)A, lda, (const _Float16*)W,        \
                                   (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N, K, strips / gx, strips % gx); } while (0)
            if (M <= 16) { if (out_f32) SC_LSL(true, 1, 4, 0); else SC_LSL(false, 1, 4, 1); }
            else { if (out_f32) SC_LSL(true, 2, 3, 2); else SC_LSL(false, 2, 3, 3); }
#undef SC_LSL
            SC_CHECK_LAUNCH("sc_gemm_f16");
            return SC_OK;
        }
        static std::atomic<int> unrolled{-1};                 // SC_SKINNY_GENERIC=1 pins the run-time-loop kernel (A/B runs, bit-identity test)
        if (unrolled < 0) { const char* e = getenv("SC_SKINNY_GENERIC"); unrolled = (e && e[0] == '1') ? 0 : 1; }
        if constexpr (EPI == SC_EPI_NONE || EPI == SC_EPI_SWIGLU) if (unrolled && (K == 3584 || K == 18944)) {
#define SC_LSU(F32, MGV, KWV, BTV) hipLaunchKernelGGL((k_gemm_skinny_u<EPI, F32, MGV, KWV, BTV>), grid, block, 0, s, (const _Float16*)A, lda, (const _Float16*)W, \
                                                      (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N)
            static std::atomic<int> xres{-1};                     // SC_SKINNY_XREG=0: the x-from-L2 kernel for K = 3584 too (A/B runs)
            if (xres < 0) { const char* e = getenv("SC_SKINNY_XREG"); xres = (e && e[0] == '0') ? 0 : 1; }
            static int n_cu_x[16] = {};
            if (n_cu_x[dev] == 0) { int n = 0; n_cu_x[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }
            // activations resident in registers, persistent over strips (k_gemm_skinny_x): where a workgroup gets at least two strips (gate/up,
            // lm_head); with one strip per workgroup (the 3584-row projections) loading x first only delays the weights: k_gemm_skinny_u
            if (K == 3584 && xres && N / 16 >= 2 * n_cu_x[dev]) {
                const int strips = N / 16, gx = sc_launch_cu_count(n_cu_x[dev], s);
                static std::atomic<int> lx{-1};                   // SC_SKINNY_LX=0: weights as register operands (k_gemm_skinny_x) instead of the LDS ring (k_gemm_skinny_lx)
                if (lx < 0) { const char* e = getenv("SC_SKINNY_LX"); lx = (e && e[0] == '0') ? 0 : 1; }
                if (lx && lda % 8 == 0 && (!R || ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(R) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0) {
                    static std::atomic<bool> lx_attr[16][4];
#define SC_LLX(F32, MGV, SLOT) do { constexpr int LDSB = 4 * 8 * 4096 + 3 * MGV * 4 * 64 * 4 + 4 * 2048;                                                        \
                        if (!lx_attr[dev][SLOT]) { (void)hipFuncSetAttribute((const void*)k_gemm_skinny_lx<EPI, F32, MGV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); lx_attr[dev][SLOT] = true; } \
                        hipLaunchKernelGGL((k_gemm_skinny_lx<EPI, F32, MGV>), dim3((unsigned)gx), block, LDSB, s, (const _Float16*)A, lda, (const _Float16*)W,           \
                                           (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N, strips / gx, strips % gx); } while (0)
                    if (M <= 16) { if (out_f32) SC_LLX(true, 1, 0); else SC_LLX(false, 1, 1); }
                    else { if (out_f32) SC_LLX(true, 2, 2); else SC_LLX(false, 2, 3); }
#undef SC_LLX
                    SC_CHECK_LAUNCH("sc_gemm_f16");
                    return SC_OK;
                }
#define SC_LSX(F32, MGV) hipLaunchKernelGGL((k_gemm_skinny_x<EPI, F32, MGV>), dim3((unsigned)gx), block, 0, s, (const _Float16*)A, lda, (const _Float16*)W, \
                                            (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N, strips / gx, strips % gx)
                if (M <= 16) { if (out_f32) SC_LSX(true, 1); else SC_LSX(false, 1); }
                else { if (out_f32) SC_LSX(true, 2); else SC_LSX(false, 2); }
#undef SC_LSX
                SC_CHECK_LAUNCH("sc_gemm_f16");
                return SC_OK;
            }
            // (one strip per workgroup and at most ~one workgroup per CU: the whole K-quarter of a wave in flight at once - 28 weight + 28 MG
            //  activation fragments per lane)
            if (K == 3584) { if (M <= 16) { if (out_f32) SC_LSU(true, 1, 28, 28); else SC_LSU(false, 1, 28, 28); }
                             else { if (out_f32) SC_LSU(true, 2, 28, 7); else SC_LSU(false, 2, 28, 7); } }
            else { if (M <= 16) { if (out_f32) SC_LSU(true, 1, 148, 8); else SC_LSU(false, 1, 148, 8); }
                   else { if (out_f32) SC_LSU(true, 2, 148, 8); else SC_LSU(false, 2, 148, 8); } }
#undef SC_LSU
            SC_CHECK_LAUNCH("sc_gemm_f16");
            return SC_OK;
        }
#define SC_LSK(F32, MGV) hipLaunchKernelGGL((k_gemm_skinny<EPI, F32, MGV>), grid, block, 0, s, (const _Float16*)A, lda, (const _Float16*)W, \
                                              (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N, K)
        if (M <= 16) { if (out_f32) SC_LSK(true, 1); else SC_LSK(false, 1); }
        else { if (out_f32) SC_LSK(true, 2); else SC_LSK(false, 2); }
#undef SC_LSK
        SC_CHECK_LAUNCH("sc_gemm_f16");
        return SC_OK;
    }
    const bool big = force == 256 || force == 2561 || (force != 128 && M >= 1024);
    const bool wide_ok = out_f32 || (ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0);      // 16-byte epilogue stores
    if (big && wide_ok && N % BN2 == 0 && K % BK2 == 0) {
        // 256x256 tiles; the 128x256 / two-workgroups-per-CU variant is kept for A/B runs (SC_GEMM_KERNEL=2561)
        static std::atomic<int> gmw{-1};                      // raster group for wide N (SC_GEMM_GMW overrides; A/B in profiles/)
        if (gmw < 0) { const char* e = getenv("SC_GEMM_GMW"); gmw = e ? atoi(e) : 4; }
        const int gm_sel = (N / BN2) > 16 ? gmw.load() : SC_GEMM_GM;
        const bool half = force == 2561;        // measured: the 256x256 tile wins at every K once the store tail is widened (K-sweep in profiles)
        const int bm = half ? 128 : 256;
        const int tM = (M + bm - 1) / bm, tN = N / BN2;
        const dim3 grid2((unsigned)(tM * tN)), block2(half ? 256 : 512);
        const size_t lds2 = half ? 3 * (128 * BK2 * 2 + HALF2) : SC_GEMM_NS * STAGE2;
        const void* fn = half ? (out_f32 ? (const void*)k_gemm256<EPI, true, 1> : (const void*)k_gemm256<EPI, false, 1>)
                              : (out_f32 ? (const void*)k_gemm256<EPI, true, 2> : (const void*)k_gemm256<EPI, false, 2>);
        static std::atomic<bool> attr_done[16][16];
        const int ai = EPI * 4 + (out_f32 ? 2 : 0) + (half ? 1 : 0);
        if (!attr_done[dev][ai]) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); attr_done[dev][ai] = true; }
        // persistent walk (one workgroup per CU) when a workgroup gets more than one tile; needs an even number of K-steps (the
        // fragment register sets ping-pong in pairs) and at least DIST + 1 of them.  SC_GEMM_PERSIST=0 switches it off (A/B runs).
        static std::atomic<int> persist{-1}, n_cu_dev[16];
        if (persist < 0) { const char* e = getenv("SC_GEMM_PERSIST"); persist = e ? atoi(e) : 1; }
        if (n_cu_dev[dev] == 0) {
            int cur = 0, n = 0;
            if (hipGetDevice(&cur) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, cur) == hipSuccess) n_cu_dev[dev] = n;
            if (n_cu_dev[dev] <= 0) n_cu_dev[dev] = 256;
        }
        const int n_cu = sc_launch_cu_count(n_cu_dev[dev], s);          // (a CU-masked stream gets a persistent grid of its own size)
        const int nt_all = tM * tN, nk2 = K / BK2;
        // (measured, profiles/r01_run99: +2..4 % at K = 1024, -1..3.5 % at K >= 3584 where the per-tile epilogue is a small share)
        static std::atomic<int> fat{-1};
        if (fat < 0) { const char* e = getenv("SC_GEMM_FAT"); fat = e ? atoi(e) : 1; }
        if (fat && !half && !out_f32 && a_grp == 0 && K % 128 == 0 && (size_t)lda * 512 < (1ull << 31) && lda % 8 == 0 &&
            (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&          // 16-byte DMA granules
            (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) && (!R || (ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0))) {       // (16-byte residual rows: loads along rows + the DMA of the first row tile)
            // persistent walk (one workgroup per CU) once there are more tiles than CUs: the next tile's first iterations are
            // fetched under the epilogue of the current one
            const bool fp = persist && fat != 2 && nt_all > n_cu;
            constexpr bool CAN_R = EPI == SC_EPI_NONE || EPI == SC_EPI_QUICK_GELU || EPI == SC_EPI_GELU_ERF;
            SC_REQUIRE(CAN_R || !R, "sc_gemm_f16: this epilogue takes no residual");
            static std::atomic<bool> fattr[16][8][2][2];
            const bool res = CAN_R && R != nullptr;
            const void* kfn = res ? (fp ? (const void*)k_gemm_fat<EPI, true, CAN_R> : (const void*)k_gemm_fat<EPI, false, CAN_R>)
                                  : (fp ? (const void*)k_gemm_fat<EPI, true, false> : (const void*)k_gemm_fat<EPI, false, false>);
            if (!fattr[dev][EPI][fp][res]) {
                (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
                fattr[dev][EPI][fp][res] = true;
            }
#define SC_LFAT(P, RS, GRID)                                                                                                              \
            hipLaunchKernelGGL((k_gemm_fat<EPI, P, RS>), GRID, dim3(256), 163840, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias, \
                               (const _Float16*)R, ldr, C, ldc, M, N, K, tN, gm_sel, nt_all, (const float*)nullptr, 0, 0, 1.0f)
            if (res) { if (fp) SC_LFAT(true, CAN_R, dim3(n_cu)); else SC_LFAT(false, CAN_R, grid2); }
            else { if (fp) SC_LFAT(true, false, dim3(n_cu)); else SC_LFAT(false, false, grid2); }
#undef SC_LFAT
            SC_CHECK_LAUNCH("sc_gemm_f16");
            return SC_OK;
        }
        const bool pers = persist && !half && !out_f32 && nt_all > n_cu && (nk2 % 2 == 0) && nk2 >= 4 && nk2 <= 64 && a_grp == 0 &&
                          (size_t)M * (size_t)lda * 2 < (1ull << 31) && (size_t)N * (size_t)K * 2 < (1ull << 31);
#define SC_L256(F32, WRV)                                                                                                                 \
        hipLaunchKernelGGL((k_gemm256<EPI, F32, WRV>), grid2, block2, lds2, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias, \
                           (const _Float16*)R, ldr, C, ldc, M, N, K, tN, a_grp, a_grp_stride, a_grp_off, gm_sel, nt_all)
        if (pers) {
            static std::atomic<bool> pattr[16][8];
            if (!pattr[dev][EPI]) { (void)hipFuncSetAttribute((const void*)k_gemm256<EPI, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); pattr[dev][EPI] = true; }
            hipLaunchKernelGGL((k_gemm256<EPI, false, 2, true>), dim3((unsigned)n_cu), block2, lds2, s, (const _Float16*)A, lda, (const _Float16*)W,
                               (const _Float16*)bias, (const _Float16*)R, ldr, C, ldc, M, N, K, tN, a_grp, a_grp_stride, a_grp_off, gm_sel, nt_all);
        }
        else if (half) { if (out_f32) SC_L256(true, 1); else SC_L256(false, 1); }
        else { if (out_f32) SC_L256(true, 2); else SC_L256(false, 2); }
#undef SC_L256
        SC_CHECK_LAUNCH("sc_gemm_f16");
        return SC_OK;
    }
    const int tilesM = (M + BM - 1) / BM, tilesN = N / BN;
    const dim3 grid((unsigned)(tilesM * tilesN)), block(256);
    const size_t lds = 2 * STAGE_BYTES;
    if (out_f32)
        hipLaunchKernelGGL((k_gemm128<EPI, true>), grid, block, lds, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias,
                           (const _Float16*)R, ldr, C, ldc, M, N, K, tilesN, a_grp, a_grp_stride, a_grp_off);
    else
        hipLaunchKernelGGL((k_gemm128<EPI, false>), grid, block, lds, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias,
                           (const _Float16*)R, ldr, C, ldc, M, N, K, tilesN, a_grp, a_grp_stride, a_grp_off);
    SC_CHECK_LAUNCH("sc_gemm_f16");
    return SC_OK;
}

}  // namespace

namespace {
template <int EPI>
int launch_headed(const void* A, int lda, const void* W, const void* bias, void* C, int ldc, int M, int N, int K, const float* tab, int pos0, int lead_cols,
                  float col_scale, hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 15;
    static std::atomic<int> n_cu_dev[16];
    if (n_cu_dev[dev] == 0) {
        int cur = 0, n = 0;
        if (hipGetDevice(&cur) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, cur) == hipSuccess) n_cu_dev[dev] = n;
        if (n_cu_dev[dev] <= 0) n_cu_dev[dev] = 256;
    }
    const int n_cu = sc_launch_cu_count(n_cu_dev[dev], s), tM = (M + BM2 - 1) / BM2, tN = N / BN2, nt_all = tM * tN;
    const bool fp = nt_all > n_cu;                                   // persistent walk once there are more tiles than CUs (as in launch_gemm)
    const int gm_sel = tN > 16 ? 4 : SC_GEMM_GM;
    static std::atomic<bool> attr[16][2];
    if (!attr[dev][fp]) {
        (void)hipFuncSetAttribute(fp ? (const void*)k_gemm_fat<EPI, true> : (const void*)k_gemm_fat<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr[dev][fp] = true;
    }
    if (fp) hipLaunchKernelGGL((k_gemm_fat<EPI, true>), dim3(n_cu), dim3(256), 163840, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias,
                               (const _Float16*)nullptr, 0, C, ldc, M, N, K, tN, gm_sel, nt_all, tab, pos0, lead_cols, col_scale);
    else hipLaunchKernelGGL((k_gemm_fat<EPI, false>), dim3((unsigned)nt_all), dim3(256), 163840, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias,
                            (const _Float16*)nullptr, 0, C, ldc, M, N, K, tN, gm_sel, nt_all, tab, pos0, lead_cols, col_scale);
    SC_CHECK_LAUNCH("sc_gemm_headed_f16");
    return SC_OK;
}
}  // namespace

// C = headed_epilogue(A W^T + bias): the GEMM whose epilogue knows that the output columns are heads of width 128 (hand-scheduled kernel only)
extern "C" int sc_gemm_headed_f16(const void* A, int lda, const void* W, const void* bias, void* C, int ldc, int M, int N, int K, int mode,
                                  const float* rope_tab, int rope_tab_rows, int pos0, int lead_cols, float col_scale, sc_stream_t stream) {
    SC_REQUIRE(A && W && C, "sc_gemm_headed_f16: null pointer argument");
    SC_REQUIRE(M > 0 && N > 0 && K > 0 && lead_cols >= 0 && lead_cols <= N && lead_cols % 128 == 0, "sc_gemm_headed_f16: bad sizes (lead_cols: a multiple of 128 within N)");
    SC_REQUIRE(mode == SC_EPI_ROPE || mode == SC_EPI_COLSCALE, "sc_gemm_headed_f16: mode must be SC_EPI_ROPE or SC_EPI_COLSCALE");
    SC_REQUIRE(mode != SC_EPI_ROPE || (rope_tab && pos0 >= 0 && (reinterpret_cast<uintptr_t>(rope_tab) & 15) == 0), "sc_gemm_headed_f16: SC_EPI_ROPE needs a 16-byte aligned table and pos0 >= 0");
    SC_REQUIRE(mode != SC_EPI_ROPE || (long long)pos0 + M <= (long long)rope_tab_rows, "sc_gemm_headed_f16: positions %d..%lld exceed the rotary table (%d rows)", pos0, (long long)pos0 + M - 1, rope_tab_rows);
    // what the hand-scheduled kernel serves (the caller's other route: sc_gemm_f16 with out_f32 = 1, then sc_rope_f32in_f16 - same numbers)
    if (!(N % BN2 == 0 && K % 128 == 0 && lda >= K && lda % 8 == 0 && (size_t)lda * 512 < (1ull << 31) && ldc >= N && ldc % 8 == 0 &&
          ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0))
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_gemm_headed_f16: needs N %% 256 == 0, K %% 128 == 0, lda %% 8 == 0, ldc %% 8 == 0 and 16-byte aligned A / W / C / bias");
    hipStream_t s = (hipStream_t)stream;
    if (mode == SC_EPI_ROPE && lead_cols >= N) return launch_headed<SC_EPI_ROPE_ALL>(A, lda, W, bias, C, ldc, M, N, K, rope_tab, pos0, N, 1.0f, s);
    if (mode == SC_EPI_ROPE) return launch_headed<SC_EPI_ROPE>(A, lda, W, bias, C, ldc, M, N, K, rope_tab, pos0, lead_cols, 1.0f, s);
    return launch_headed<SC_EPI_COLSCALE>(A, lda, W, bias, C, ldc, M, N, K, nullptr, 0, lead_cols, col_scale, s);
}

extern "C" int sc_gemm_f16(const void* A, int lda, const void* W, const void* bias, const void* residual, int ldr, void* C, int ldc,
                           int M, int N, int K, int epilogue, int out_f32, int a_grp, int a_grp_stride, int a_grp_off, sc_stream_t stream) {
    SC_REQUIRE(A && W && C, "sc_gemm_f16: null pointer argument");
    SC_REQUIRE(M > 0 && N > 0 && K > 0, "sc_gemm_f16: M, N, K must be positive");
    SC_REQUIRE(N % BN == 0, "sc_gemm_f16: N (%d) must be a multiple of %d", N, BN);
    SC_REQUIRE(K % BK == 0, "sc_gemm_f16: K (%d) must be a multiple of %d", K, BK);
    SC_REQUIRE(lda >= K && lda % 8 == 0 && ldc >= (epilogue == SC_EPI_SWIGLU ? N / 2 : N) && ldc % 2 == 0 && (epilogue == SC_EPI_SWIGLU || ldc % 4 == 0),
               "sc_gemm_f16: bad leading dimensions");
    SC_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0), "sc_gemm_f16: bad residual leading dimension");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0, "sc_gemm_f16: A and W must be 16-byte aligned");
    SC_REQUIRE(((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 7) == 0 ||
                   (epilogue == SC_EPI_SWIGLU && (reinterpret_cast<uintptr_t>(C) & 3) == 0),
               "sc_gemm_f16: C, bias, residual must be 8-byte aligned");
    SC_REQUIRE(a_grp >= 0 && (a_grp == 0 || (a_grp_stride >= a_grp && a_grp_off >= 0)), "sc_gemm_f16: bad A row-group map");
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case SC_EPI_NONE: return launch_gemm<SC_EPI_NONE>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, out_f32, a_grp, a_grp_stride, a_grp_off, s);
        case SC_EPI_QUICK_GELU: return launch_gemm<SC_EPI_QUICK_GELU>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, out_f32, a_grp, a_grp_stride, a_grp_off, s);
        case SC_EPI_GELU_ERF: return launch_gemm<SC_EPI_GELU_ERF>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, out_f32, a_grp, a_grp_stride, a_grp_off, s);
        case SC_EPI_SWIGLU:
            SC_REQUIRE(!residual && !out_f32, "sc_gemm_f16: SwiGLU epilogue takes no residual and writes fp16");
            return launch_gemm<SC_EPI_SWIGLU>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, 0, a_grp, a_grp_stride, a_grp_off, s);
    }
    return sc_fail(SC_ERR_ARG, "sc_gemm_f16: unknown epilogue %d", epilogue);
}
