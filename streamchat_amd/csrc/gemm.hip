// fp16 GEMM with fused epilogues for gfx950 (CDNA4):  C[M,N] = epi(A[M,K] * W[N,K]^T + bias) (+ residual)
//
// This is the op that carries the encoders: the ViT-L/14-336 tower + mlp2x_gelu projector the reference
// runs through transformers' CLIPVisionModel / nn.Sequential (clip_encoder.py:76,
// multimodal_projector/builder.py:41-48), the BERT text encoders (utiles.py:707,728) and the Qwen2
// linears (llava_qwen.py:155).  Weights keep the torch.nn.Linear layout W[N,K] ("B^T input"), so both
// MFMA operands are read along K with 16-byte ds_read_b128.
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

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W

__device__ __forceinline__ float epi_apply(float x, int epi) {
    if (epi == SC_EPI_QUICK_GELU) return x / (1.0f + __expf(-1.702f * x));
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

template <int EPI>
int launch_gemm(const void* A, int lda, const void* W, const void* bias, const void* R, int ldr, void* C, int ldc, int M, int N,
                int K, int out_f32, int a_grp, int a_grp_stride, int a_grp_off, hipStream_t s) {
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
