// Probe: global -> LDS DMA bandwidth (16-byte global_load_lds) out of L2 / MALL / HBM: 256 workgroups (one per CU, 512 threads)
// repeatedly stream a working set of WS bytes (shared by all CUs, each starting at a different offset) into a 64 KB LDS ring.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__global__ __launch_bounds__(512, 1) void k(const char* src, size_t ws, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    size_t off = ((size_t)blockIdx.x * 262144) % ws;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)                     // 4 x 512 x 16 B = 32 KB per iteration per workgroup
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off + (size_t)(j * 512 + tid) * 16), (lds_ptr_t)(smem + ((it & 1) * 32768) + (j * 512 + wave * 64) * 16), 16, 0, 0);
        off += 32768; if (off + 32768 > ws) off = 0;
        if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 512 + tid] = reinterpret_cast<float*>(smem)[tid];
}
int main() {
    const size_t big = 2ull << 30;
    char* src; hipMalloc(&src, big); hipMemset(src, 1, big);
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (size_t ws : {(size_t)8 << 20, (size_t)24 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        const int iters = 4000;
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, ws, 100, out);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, src, ws, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("working set %5zu MB: %.2f TB/s aggregate global->LDS (%.1f B/clk/CU at 2.1 GHz)\n", ws >> 20, 256.0 * iters * 32768 / (ms * 1e9), 256.0 * iters * 32768 / (ms * 1e6) / 256 / 2.1);
    }
    return 0;
}
