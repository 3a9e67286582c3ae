// Probe: cost of LDS fragment refills under v_mfma_f32_32x32x16_f16 for one wave per SIMD.  16 MFMAs per iteration read their A
// operand from fr[m & 7]; after MFMA m one read (ds_read_b128, or two ds_read_b64_tr_b16) refills fr[(m - LAG) & 7] (LAG = 0: the
// register the MFMA just issued is reading), or a register no MFMA uses (LAG = -1).  RPM = reads per MFMA in 1/4 units (4 = every MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));

template <int LAG, int TR, int EVERY, int NV>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    h8 b, fr[8], spare[8];
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.5f + i);
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 8; ++i) { fr[j][i] = (_Float16)(threadIdx.x * 0.001f + i + j); spare[j][i] = fr[j][i]; }
    f16x acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.01f + i;
    const unsigned a0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (threadIdx.x & 63) * 16;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(fr[m & 7]), "v"(b));
            if (m % EVERY == 0) {
                const unsigned ad = a0 + (m & 7) * 1024;
                if (LAG >= 0) {
                    if (!TR) asm volatile("ds_read_b128 %0, %1" : "=v"(fr[(m - LAG) & 7]) : "v"(ad));
                    else { s4 lo, hi; asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(ad));
                           typedef short s8 __attribute__((ext_vector_type(8))); s8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; fr[(m - LAG) & 7] = __builtin_bit_cast(h8, v8); }
                } else {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(spare[m & 7]) : "v"(ad));
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(m * NV + v) & 15]) : "v"(x[(m * NV + v + 5) & 15]));
            if (m == 7 || m == 15) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15] + (float)fr[i][0] + (float)spare[i][1];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int LAG, int TR, int EVERY, int NV>
void run(float* d) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<LAG, TR, EVERY, NV>), dim3(256), dim3(256), 0, 0, d, 10);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<LAG, TR, EVERY, NV>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    float h; (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("refill lag %2d MFMAs, %s, one per %d MFMAs, %d v_fma per MFMA: %.1f cycles per MFMA\n", LAG, TR ? "2 x ds_read_b64_tr_b16" : "ds_read_b128", EVERY, NV, h / (iters * 16.0));
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 256 * 4);
    run<-1, 0, 1, 0>(d); run<0, 0, 1, 0>(d); run<1, 0, 1, 0>(d); run<2, 0, 1, 0>(d); run<4, 0, 1, 0>(d);
    run<-1, 0, 2, 0>(d); run<0, 0, 2, 0>(d); run<1, 0, 2, 0>(d); run<2, 0, 2, 0>(d);
    run<0, 1, 2, 0>(d); run<1, 1, 2, 0>(d); run<2, 1, 2, 0>(d);
    run<-1, 0, 1, 3>(d); run<1, 0, 1, 3>(d); run<2, 0, 1, 3>(d); run<2, 0, 2, 3>(d); run<2, 1, 2, 3>(d);
    return 0;
}
