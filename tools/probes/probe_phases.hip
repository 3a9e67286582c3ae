// Probe: phase-structured waves (attention-like): per "tile" [MFMA burst S] [VALU burst: NV v_fma + NT v_exp] [MFMA burst PV], all
// register-only, 2 workgroups of 4 waves per CU (2 waves per SIMD).  Same flops with 16x16x32 (32 + 32 MFMAs) or 32x32x16 (16 + 16).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <bool BIG, int NV, int NT>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f4 acc4[16]; f16v acc16[4];
    for (int i = 0; i < 16; ++i) acc4[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if (BIG) {
#pragma unroll
                for (int m = 0; m < 16; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc16[m & 3]) : "v"(a), "v"(b));
            } else {
#pragma unroll
                for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[m & 15]) : "v"(a), "v"(b));
            }
            if (ph == 0) {
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]));
#pragma unroll
                for (int v = 0; v < NT; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(v + 4) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc4[i][0];
    for (int i = 0; i < 4; ++i) s += acc16[i][0];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BIG, int NV, int NT>
void run(float* d) {
    const int iters = 1000;
    dim3 grid(256 * 2 * 4), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BIG, NV, NT>), grid, block, 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BIG, NV, NT>), grid, block, 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 4 rounds of blocks x iters tiles; MFMA pipe time per tile = 1024 cycles
    const double tiles_per_simd = 2.0 * 4 * iters;
    printf("%s NV=%3d NT=%2d : %.3f ms -> %.0f ns per tile per SIMD (MFMA-only floor ~%.0f ns at 2.1 GHz)\n", BIG ? "32x32x16" : "16x16x32", NV, NT, ms,
           ms * 1e6 / tiles_per_simd, 1024 / 2.1);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<false, 0, 0>(d); run<true, 0, 0>(d);
    run<false, 64, 32>(d); run<true, 64, 32>(d);
    run<false, 112, 32>(d); run<true, 112, 32>(d);
    run<false, 160, 32>(d); run<true, 160, 32>(d);
    return 0;
}
