// Probe: how much VALU / transcendental work can be issued "under" MFMAs on gfx950?
// Each wave runs ITER iterations of: 16 independent v_mfma_f32_32x32x16_f16, interleaved with NV plain VALU (v_fma_f32) and NT
// v_exp_f32 per MFMA.  Prints cycles per MFMA for several (NV, NT, waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(16)));

template <int NV, int NT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(m * NV + v) & 7]) : "v"(x[(m + 3) & 7] ));
#pragma unroll
            for (int v = 0; v < NT; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(m * NT + v + 4) & 7]));
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int NV, int NT>
void run(int waves_per_simd, float* d) {
    const int iters = 2000;
    // one block of 256 threads = 4 waves = 1 wave per SIMD on one CU; 2 blocks co-resident on a CU needs many blocks: launch
    // 256 CUs x waves_per_simd blocks and time with events
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NT>), grid, block, 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NT>), grid, block, 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 8 * waves_per_simd;
    printf("NV=%d NT=%d waves/SIMD=%d : %.3f ms, wave-clock cycles/MFMA (one wave) = %.1f, wall ns per MFMA per SIMD = %.2f\n", NV, NT, waves_per_simd,
           ms, h / (iters * 8.0), ms * 1e6 / mfma_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w, d); run<1, 0>(w, d); run<2, 0>(w, d); run<3, 0>(w, d); run<4, 0>(w, d); run<6, 0>(w, d); run<8, 0>(w, d); run<12, 0>(w, d);
        run<0, 1>(w, d); run<2, 1>(w, d); run<4, 1>(w, d);
    }
    return 0;
}
