// Probe: what does one wave per SIMD pay for VALU work issued between v_mfma_f32_32x32x16_f16 (acc in AGPRs or VGPRs, B operand in
// AGPRs or VGPRs)?  Every VALU op works on its own registers (no dependencies), NV of them after each MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int ACC_A, int B_A, int KIND, int NV>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f16x acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[16]; f2 y[8]; unsigned u[8];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.01f + i;
    for (int i = 0; i < 8; ++i) { y[i] = f2{x[i], x[i + 8]}; u[i] = threadIdx.x + i; }
    const f2 sc = {1.0001f, 1.0001f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (ACC_A && B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a), "a"(b));
            else if (ACC_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a), "v"(b));
            else if (B_A) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "a"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = (m * NV + v);
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i & 15]) : "v"(x[(i + 5) & 15]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(y[i & 7]) : "v"(sc));
                if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i & 7]) : "v"(x[i & 15]), "v"(x[(i + 3) & 15]));
                if (KIND == 3) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i & 15]) : "v"(u[i & 7]), "v"(u[(i + 1) & 7]));
                if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 15]));
                if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i & 7]) : "v"(sc));
                if (KIND == 6) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i & 7]) : "v"(u[(i + 3) & 7]));
                if (KIND == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i & 15]) : "v"(x[(i + 5) & 15]));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i & 7]) : "v"(sc));
                if (KIND == 9) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i & 15]) : "v"(x[(i + 5) & 15]), "s"(1.0001f));
                if (KIND == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i & 15]) : "v"(x[(i + 5) & 15]));
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15] + y[i][0] + y[i][1] + (float)u[i];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int ACC_A, int B_A, int KIND, int NV>
void run(float* d) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<ACC_A, B_A, KIND, NV>), dim3(256), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<ACC_A, B_A, KIND, NV>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    static const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32", "v_dot2_f32_f16", "v_exp_f32", "v_pk_add_f32", "v_pk_add_f16", "v_add_f32", "v_pk_mul_f32", "v_fma_f32 (sgpr)", "v_max_f32"};
    printf("acc %s, B %s, %d x %-18s per MFMA: %.1f cycles per MFMA\n", ACC_A ? "AGPR" : "VGPR", B_A ? "AGPR" : "VGPR", NV, names[KIND], h / (iters * 16.0));
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<1, 0, 0, 0>(d); run<0, 0, 0, 0>(d); run<1, 1, 0, 0>(d); run<0, 1, 0, 0>(d);
    run<1, 0, 0, 1>(d); run<1, 0, 0, 2>(d); run<1, 0, 0, 3>(d); run<1, 0, 0, 4>(d); run<1, 0, 0, 6>(d);
    run<0, 0, 0, 2>(d); run<0, 0, 0, 4>(d); run<0, 1, 0, 4>(d); run<1, 1, 0, 4>(d);
    run<1, 0, 1, 2>(d); run<1, 0, 1, 4>(d); run<1, 0, 2, 2>(d); run<1, 0, 2, 4>(d); run<1, 0, 3, 2>(d); run<1, 0, 3, 4>(d);
    run<1, 0, 4, 1>(d); run<1, 0, 4, 2>(d); run<1, 0, 4, 3>(d);
    run<1, 0, 5, 2>(d); run<1, 0, 6, 2>(d); run<1, 0, 6, 4>(d); run<1, 0, 7, 4>(d); run<1, 0, 8, 2>(d); run<1, 0, 9, 4>(d); run<1, 0, 10, 4>(d);
    run<1, 0, 7, 5>(d); run<1, 0, 7, 6>(d); run<1, 0, 4, 4>(d);
    return 0;
}
