// Frame preprocessing (HBM-bound elementwise): uint8 HWC -> normalised fp16, either CHW
// (drop-in for CLIPImageProcessor.preprocess + .to(float16); reference utiles.py:71-87,
// inference_streaming_longva_v2.py:520) or directly in the im2col "patch row" layout the
// patch-embedding GEMM consumes (so the Conv2d k=s=14 of CLIPVisionEmbeddings becomes one GEMM and
// the [N,3,336,336] fp16 intermediate is never written).
// Arithmetic = the numpy pipeline of the HF image processor, bit for bit: rescale in float64
// (`image.astype(float64) * (1/255)` -> float32), normalise in float32 (`(x - mean) / std`), then the
// reference's `.to(torch.float16)` (round-to-nearest-even).  A uint8 pixel has 256 values, so the
// library evaluates that chain once per (channel, value) on the host with IEEE arithmetic and the
// kernels are pure HBM-bound table gathers through a 1.5 KiB LDS copy of the table.
#include "sc_common.h"

namespace {

struct Lut3 { _Float16 v[3][256]; };   // passed by value in the kernarg segment (1536 B)

__device__ __forceinline__ void lut_to_lds(const Lut3& lut, _Float16* s) {
    for (int i = threadIdx.x; i < 768; i += blockDim.x) s[i] = lut.v[i >> 8][i & 255];
    __syncthreads();
}

// CHW output: one thread produces 8 consecutive pixels of one channel plane (16-byte store);
// the 24 source bytes are read as uint8 (L1/L2 absorb the 3x channel re-read).
__global__ __launch_bounds__(256) void k_pre_chw(const uint8_t* __restrict__ in, _Float16* __restrict__ out, size_t n_img,
                                                 int hw, Lut3 lut) {
    __shared__ _Float16 T[768];
    lut_to_lds(lut, T);
    const size_t groups_per_plane = (size_t)(hw + 7) / 8;
    const size_t total = n_img * 3 * groups_per_plane;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = g / groups_per_plane;
        const int p0 = (int)(g - plane * groups_per_plane) * 8;
        const size_t img = plane / 3;
        const int c = (int)(plane - img * 3);
        const uint8_t* src = in + (img * (size_t)hw + (size_t)p0) * 3 + c;
        _Float16* dst = out + plane * (size_t)hw + p0;
        const _Float16* Tc = T + c * 256;
        if (p0 + 8 <= hw && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
            sc_h8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = Tc[src[e * 3]];
            *reinterpret_cast<sc_h8*>(dst) = v;
        } else {
            for (int e = 0; e < 8 && p0 + e < hw; ++e) dst[e] = Tc[src[e * 3]];
        }
    }
}

// patch-row output: row = (img, py, px) patch, col = c*P*P + iy*P + ix, zero pad to ld.
// One thread produces 8 consecutive columns (16-byte store).
__global__ __launch_bounds__(256) void k_pre_patch(const uint8_t* __restrict__ in, _Float16* __restrict__ out, size_t n_img, int h,
                                                   int w, int P, int ld, Lut3 lut) {
    __shared__ _Float16 T[768];
    lut_to_lds(lut, T);
    const int gw = w / P, gh = h / P, kcols = 3 * P * P;
    const size_t rows = n_img * (size_t)gh * gw;
    const int gpr = ld / 8;
    const size_t total = rows * (size_t)gpr;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t row = g / gpr;
        const int c0 = (int)(g - row * gpr) * 8;
        const size_t img = row / ((size_t)gh * gw);
        const int pr = (int)(row - img * (size_t)gh * gw);
        const int py = pr / gw, px = pr - py * gw;
        sc_h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = c0 + e;
            _Float16 r = (_Float16)0.f;
            if (col < kcols) {
                const int c = col / (P * P);
                const int rem = col - c * P * P;
                const int iy = rem / P, ix = rem - iy * P;
                const size_t pix = (img * (size_t)h + (size_t)(py * P + iy)) * (size_t)w + (size_t)(px * P + ix);
                r = T[c * 256 + in[pix * 3 + c]];
            }
            v[e] = r;
        }
        *reinterpret_cast<sc_h8*>(out + row * (size_t)ld + c0) = v;
    }
}

// fp16 CHW pixel values (the reference's `encode_images` input, llava_arch.py:179) -> patch rows
__global__ __launch_bounds__(256) void k_patchify_f16(const _Float16* __restrict__ in, _Float16* __restrict__ out, size_t n_img, int h, int w,
                                                      int P, int ld) {
    const int gw = w / P, gh = h / P, kcols = 3 * P * P;
    const size_t rows = n_img * (size_t)gh * gw;
    const int gpr = ld / 8;
    const size_t total = rows * (size_t)gpr;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t row = g / gpr;
        const int c0 = (int)(g - row * gpr) * 8;
        const size_t img = row / ((size_t)gh * gw);
        const int pr = (int)(row - img * (size_t)gh * gw);
        const int py = pr / gw, px = pr - py * gw;
        sc_h8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = c0 + e;
            _Float16 r = (_Float16)0.f;
            if (col < kcols) {
                const int c = col / (P * P);
                const int rem = col - c * P * P;
                const int iy = rem / P, ix = rem - iy * P;
                r = in[((img * 3 + c) * (size_t)h + (size_t)(py * P + iy)) * (size_t)w + (size_t)(px * P + ix)];
            }
            v[e] = r;
        }
        *reinterpret_cast<sc_h8*>(out + row * (size_t)ld + c0) = v;
    }
}

Lut3 make_lut(const float* mean, const float* std) {
    Lut3 l;
    for (int c = 0; c < 3; ++c)
        for (int u = 0; u < 256; ++u) {
            const float x = (float)((double)u * (1.0 / 255.0));    // HF rescale: float64 product, then float32
            const float r = (x - mean[c]) / std[c];                  // HF normalize: float32
            l.v[c][u] = (_Float16)r;                                 // .to(torch.float16): RNE
        }
    return l;
}

}  // namespace

extern "C" int sc_preprocess_u8(const uint8_t* hwc, int n, int h, int w, const float* mean, const float* std, void* out_f16,
                                sc_stream_t stream) {
    SC_REQUIRE(hwc && mean && std && out_f16, "sc_preprocess_u8: null pointer argument");
    SC_REQUIRE(n > 0 && h > 0 && w > 0, "sc_preprocess_u8: n, h, w must be positive");
    const Lut3 nm = make_lut(mean, std);
    const size_t total = (size_t)n * 3 * (((size_t)h * w + 7) / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(k_pre_chw, dim3(grid), dim3(256), 0, (hipStream_t)stream, hwc, (_Float16*)out_f16, (size_t)n, h * w, nm);
    SC_CHECK_LAUNCH("sc_preprocess_u8");
    return SC_OK;
}

extern "C" int sc_preprocess_patchify_u8(const uint8_t* hwc, int n, int h, int w, int patch, const float* mean, const float* std,
                                         void* out_f16, int ld, sc_stream_t stream) {
    SC_REQUIRE(hwc && mean && std && out_f16, "sc_preprocess_patchify_u8: null pointer argument");
    SC_REQUIRE(n > 0 && h > 0 && w > 0 && patch > 0, "sc_preprocess_patchify_u8: sizes must be positive");
    SC_REQUIRE(h % patch == 0 && w % patch == 0, "sc_preprocess_patchify_u8: h, w must be multiples of patch");
    SC_REQUIRE(ld >= 3 * patch * patch && ld % 8 == 0, "sc_preprocess_patchify_u8: ld must be >= 3*patch^2 and a multiple of 8");
    SC_REQUIRE((reinterpret_cast<uintptr_t>(out_f16) & 15) == 0, "sc_preprocess_patchify_u8: out must be 16-byte aligned");
    const Lut3 nm = make_lut(mean, std);
    const size_t total = (size_t)n * (h / patch) * (w / patch) * (ld / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(k_pre_patch, dim3(grid), dim3(256), 0, (hipStream_t)stream, hwc, (_Float16*)out_f16, (size_t)n, h, w, patch, ld, nm);
    SC_CHECK_LAUNCH("sc_preprocess_patchify_u8");
    return SC_OK;
}

extern "C" int sc_patchify_f16(const void* chw_f16, int n, int h, int w, int patch, void* out_f16, int ld, sc_stream_t stream) {
    SC_REQUIRE(chw_f16 && out_f16, "sc_patchify_f16: null pointer argument");
    SC_REQUIRE(n > 0 && h > 0 && w > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "sc_patchify_f16: bad sizes");
    SC_REQUIRE(ld >= 3 * patch * patch && ld % 8 == 0, "sc_patchify_f16: ld must be >= 3*patch^2 and a multiple of 8");
    SC_REQUIRE((reinterpret_cast<uintptr_t>(out_f16) & 15) == 0, "sc_patchify_f16: out must be 16-byte aligned");
    const size_t total = (size_t)n * (h / patch) * (w / patch) * (ld / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(k_patchify_f16, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)chw_f16, (_Float16*)out_f16, (size_t)n, h, w,
                       patch, ld);
    SC_CHECK_LAUNCH("sc_patchify_f16");
    return SC_OK;
}
