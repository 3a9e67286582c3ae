// Probe: the GEMM K-step skeleton (8 waves per CU, per step and wave: 32 MFMAs + 12 ds_read_b128) with three ways of feeding the
// 32 KB per step into LDS:  0 = nothing (stale LDS),  1 = LDS-DMA (global_load_lds 16 B, 4 per lane),  2 = global_load_dwordx4 into
// VGPRs + ds_write_b128 one step later.  Source: a 64 MB buffer streamed by all CUs (L2 / MALL hits mostly).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int FEED>
__global__ __launch_bounds__(512, 1) void k(const char* src, size_t ws, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 131072 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f4{0, 0, 0, 0};
    size_t off = ((size_t)blockIdx.x * 262144) % ws;
    const int rl = lane & 15, g = lane >> 4;
    const int frag = rl * 64 + ((g ^ ((0x78 >> (2 * ((rl >> 2) & 3))) & 3)) << 4);
    h8 stg[4] = {};
    for (int it = 0; it < iters; ++it) {
        const char* cur = smem + (it & 3) * 32768;
        char* nxt = smem + ((it + 3) & 3) * 32768;
        if (FEED == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off + (size_t)(j * 512 + tid) * 16), (lds_ptr_t)(nxt + (j * 512 + wave * 64) * 16), 16, 0, 0);
        } else if (FEED == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<h8*>(nxt + (j * 512 + tid) * 16) = stg[j];           // data loaded one step ago
#pragma unroll
            for (int j = 0; j < 4; ++j) stg[j] = __builtin_nontemporal_load(reinterpret_cast<const h8*>(src + off + (size_t)(j * 512 + tid) * 16));
        }
        off += 32768; if (off + 32768 > ws) off = 0;
        h8 a[8], b[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const h8*>(cur + (wave >> 2) * 8192 + i * 1024 + frag);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const h8*>(cur + 16384 + (wave & 3) * 4096 + i * 1024 + frag);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[jj], a[i], acc[i * 4 + jj], 0, 0, 0);
        if (FEED == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0];
    out[blockIdx.x * 512 + tid] = s + (float)stg[0][0];
}
template <int FEED, size_t WSMB>
void run(const char* src, float* out) {
    const int iters = 4000; const size_t ws = WSMB << 20;
    hipFuncSetAttribute((const void*)k<FEED>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((k<FEED>), dim3(256), dim3(512), 131072, 0, src, ws, 50, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FEED>), dim3(256), dim3(512), 131072, 0, src, ws, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * iters * 8 * 32 * 16384;
    printf("ws=%3zu MB feed=%d (%s): %.3f ms, %.0f ns per K-step, %.0f TFLOP/s\n", (size_t)WSMB, FEED, FEED == 0 ? "none" : FEED == 1 ? "LDS-DMA" : "VGPR + ds_write_b128", ms, ms * 1e6 / iters, flops / (ms * 1e9));
}
int main() {
    char* src; hipMalloc(&src, 128ull << 20); hipMemset(src, 0, 128ull << 20);
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 8>(src, out); run<1, 8>(src, out); run<2, 8>(src, out); run<1, 24>(src, out); run<2, 24>(src, out); run<1, 64>(src, out);
    return 0;
}
