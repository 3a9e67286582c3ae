// Probe: raw LDS read bandwidth per CU (8 waves per CU, ds_read_b128 / ds_read_b64_tr_b16 only, conflict-free addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    float acc = 0.f;
    h8 fr[4];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int rl = lane & 15, g = lane >> 4;
    const char* vb = smem + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 256 + (lane & 3) * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) {
                h8 f;
                const unsigned a = lds0 + rl * 256 + ((((i & 3) * 4 + g) ^ rl) << 4) + ((i >> 2) & 3) * 4096 + (it & 1) * 32768;
                asm volatile("ds_read_b128 %0, %1" : "=v"(f) : "v"(a));
                fr[i & 3] = f;
            }
            else {
                const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vb + (((i & 7) ^ ((lane >> 2) & 7)) << 5) + (it & 1) * 32768 + (i >> 3) * 8192));
                const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vb + (((i & 7) ^ ((lane >> 2) & 7)) << 5) + (it & 1) * 32768 + (i >> 3) * 8192 + 4096));
                acc += (float)(a[0] + b[0]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 0) for (int i = 0; i < 4; ++i) acc += (float)fr[i][0];
    out[blockIdx.x * 256 + tid] = acc;
}
template <int KIND>
void run(float* d) {
    const int iters = 4000;
    dim3 grid(256 * 2 * 4), block(256);
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), grid, block, 65536, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), grid, block, 65536, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: 8 resident waves x 4 rounds x iters x 16 KB
    const double bytes_per_cu = 8.0 * 4 * iters * 16 * 1024;
    printf("%s: %.3f ms -> %.1f bytes/ns per CU (= %.0f B/clk at 2.1 GHz); 16 KB per wave takes %.0f ns with 8 waves per CU\n", KIND ? "ds_read_b64_tr_b16 x32" : "ds_read_b128 x16",
           ms, bytes_per_cu / (ms * 1e6), bytes_per_cu / (ms * 1e6) / 2.1, ms * 1e6 / (4.0 * iters));
}
int main() { float* d; hipMalloc(&d, 256 * 8 * 256 * 4); run<0>(d); run<1>(d); return 0; }
