// Hardware probe (GPU box): prints the lane/element mapping of ds_read_b64_tr_b16 for a linear address pattern,
// and the C/D + A/B layouts of the f16 MFMAs used by the kernels.  hipcc --offload-arch=gfx950 probe_tr.hip -o probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k_tr(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
// A one-hot probes: A[i][k] = 1 at (i0,k0) and B[k][j] = j + 100*k  => D[i0][j] = B[k0][j]
__global__ void k_mfma16(float* out, int i0, int k0) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l >> 4) * 8 + e;          // assumed A layout: row = l&15, k = (l>>4)*8+e
        a[e] = (_Float16)(((l & 15) == i0 && k == k0) ? 1.f : 0.f);
        b[e] = (_Float16)(float)((l & 15) + 16 * k);   // assumed B layout: col = l&15, k = (l>>4)*8+e ; value = j + 16k
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void k_mfma32(float* out, int i0, int k0) {
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = (l >> 5) * 8 + e;          // assumed: row = l&31, k = (l>>5)*8+e
        a[e] = (_Float16)(((l & 31) == i0 && k == k0) ? 1.f : 0.f);
        b[e] = (_Float16)(float)((l & 31) + 32 * k);
    }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k_tr<<<1, 64>>>(d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("TR: lane -> 4 element indices (linear address lane*4 elements)\n");
    for (int l = 0; l < 64; ++l) printf("L%02d: %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    float* f; hipMalloc(&f, 64 * 16 * 4); float hf[1024];
    int ok16 = 1;
    for (int i0 = 0; i0 < 16; i0 += 5) for (int k0 = 0; k0 < 32; k0 += 7) {
        k_mfma16<<<1, 64>>>(f, i0, k0); hipMemcpy(hf, f, 64 * 4 * 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            const float want = (row == i0) ? (float)(col + 16 * k0) : 0.f;
            if (hf[l * 4 + r] != want) { if (ok16) printf("MFMA16 mismatch i0=%d k0=%d lane=%d r=%d got %g want %g\n", i0, k0, l, r, hf[l * 4 + r], want); ok16 = 0; }
        }
    }
    printf("MFMA16x16x32 layout assumption: %s\n", ok16 ? "OK" : "WRONG");
    int ok32 = 1;
    for (int i0 = 0; i0 < 32; i0 += 9) for (int k0 = 0; k0 < 16; k0 += 5) {
        k_mfma32<<<1, 64>>>(f, i0, k0); hipMemcpy(hf, f, 64 * 16 * 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            const float want = (row == i0) ? (float)(col + 32 * k0) : 0.f;
            if (hf[l * 16 + r] != want) { if (ok32) printf("MFMA32 mismatch i0=%d k0=%d lane=%d r=%d got %g want %g\n", i0, k0, l, r, hf[l * 16 + r], want); ok32 = 0; }
        }
    }
    printf("MFMA32x32x16 layout assumption: %s\n", ok32 ? "OK" : "WRONG");
    return 0;
}
