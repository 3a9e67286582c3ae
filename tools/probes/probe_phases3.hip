// Probe 2: the attention tile skeleton with its LDS traffic and barrier, register-only otherwise.  Per tile and wave:
//   S phase : 16 ds_read_b128 (K fragments), each feeding 2 MFMAs        (MODE bit 0: LDS reads on)
//   softmax : NV v_fma + NT v_exp
//   PV phase: 32 ds_read_b64 (V fragments, 2 per fragment), each fragment feeding 2 MFMAs
//   barrier : __syncthreads() per tile                                    (MODE bit 1)
//   DMA     : 8 x 16-B global_load_lds per tile per lane into the other half of LDS + vmcnt(0) before the barrier (MODE bit 2)
// 2 workgroups (4 waves) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int MODE, int NV, int NT, int QBX>
__global__ __launch_bounds__(256, 2) void k(float* out, const _Float16* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    h8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(0.5f + i);
    f4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f4{0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = tid * 0.01f + i;
    const char* kbase = smem + (lane & 15) * 256 + (((lane >> 4) ^ (lane & 15)) << 4);
    const h8 areg = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int it = 0; it < iters; ++it) {
        const char* buf = smem + (it & 1) * 32768;
        if (MODE & 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + ((size_t)(it & 63) * 16384 + (blockIdx.x & 255) * 64 + j * 2048 + tid * 8)),
                                                 (lds_ptr_t)(smem + ((it + 1) & 1) * 32768 + (j * 256 + wave * 64) * 16), 16, 0, 0);
        }
        // S phase
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h8 f = areg;
            if (MODE & 1) f = *reinterpret_cast<const h8*>(kbase + (buf - smem) + (i >> 2) * 4096 + (i & 3) * 64 * 0 + ((i & 3) << 6) % 256);
#pragma unroll
            for (int q = 0; q < QBX; ++q) acc[(i * QBX + q) % 24] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, b, acc[(i * QBX + q) % 24], 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]));
#pragma unroll
        for (int v = 0; v < NT; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(v + 4) & 7]));
        // PV phase
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h8 f = areg;
            if (MODE & 1) {
                typedef short s4 __attribute__((ext_vector_type(4)));
                typedef short s8 __attribute__((ext_vector_type(8)));
                const char* vp = buf + 16384 + (i >> 3) * 8192 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 256 + (((i & 7) ^ ((lane >> 2) & 7)) << 5) + (lane & 3) * 8;
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vp));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(vp + 4096));
                const s8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                f = __builtin_bit_cast(h8, v8);
            }
#pragma unroll
            for (int q = 0; q < QBX; ++q) acc[(i * QBX + q) % 24] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, b, acc[(i * QBX + q) % 24], 0, 0, 0);
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE, int NV, int NT, int QBX>
void run(float* d, const _Float16* src) {
    const int iters = 1000;
    dim3 grid(256 * 2 * 4), block(256);
    hipFuncSetAttribute((const void*)k<MODE, NV, NT, QBX>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, NT, QBX>), grid, block, 65536, 0, d, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, NT, QBX>), grid, block, 65536, 0, d, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("QB=%d mode lds=%d barrier=%d dma=%d NV=%3d NT=%2d : %.3f ms -> %.0f ns per tile per SIMD = %.2f ns per MFMA (floor 7.6)\n", QBX, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, NV, NT,
           ms, ms * 1e6 / (2.0 * 4 * iters), ms * 1e6 / (2.0 * 4 * iters) / (32.0 * QBX));
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    _Float16* src; hipMalloc(&src, 64 * 16384 * 2 + (256 * 64 + 8 * 2048 + 256 * 8) * 2 + 4096); hipMemset(src, 0, 64 * 16384 * 2 + (256 * 64 + 8 * 2048 + 256 * 8) * 2);
    // QB = 2: 64 MFMAs per tile-wave; QB = 3: 96 MFMAs with the same 48 LDS reads (VALU scaled by 1.5)
    run<7, 112, 32, 2>(d, src); run<7, 168, 48, 3>(d, src); run<0, 112, 32, 2>(d, src); run<0, 168, 48, 3>(d, src);
    return 0;
}
