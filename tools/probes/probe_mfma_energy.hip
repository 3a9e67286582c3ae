// Probe: what does the 1400 W package cap leave of the matrix pipe, by instruction shape and by what else the wave does?
// One wave per SIMD (256 threads, 1 workgroup per CU, all 256 accumulator registers live - the shape of k_gemm_fat's K loop), operands in
// registers, no global traffic inside the loop.  Every variant runs for a few seconds; the rate, the package power and the shader clock
// (amdgpu hwmon) are averaged over the second half of that time, when the power controller has settled.
//   V0  v_mfma_f32_16x16x32_f16, random operands (8 x 8 tiles of 16 x 16: the order of k_gemm_fat - a[i] constant over 8 MFMAs, b[j] walks)
//   V1  v_mfma_f32_32x32x16_f16, random operands (4 x 4 tiles of 32 x 32: half the operand-register reads per flop)
//   V2  V0 on zero operands (nothing toggles: what the schedule delivers without the cap)
//   V3  V0 + 32 ds_read_b128 per 128 MFMAs (k_gemm_fat's fragment traffic, conflict-free addresses), results folded into the operands
//   V4  V1 + the same LDS bytes per flop
//   V5  V0 with v_mfma_f32_16x16x32_bf16 on the same bits
//   V6  V0 with the operand roles exchanged (srcA walks, srcB stationary over 8 MFMAs - k_gemm_fat's arrangement)   V7  V0 in snake order
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/probe_mfma_energy.hip -o /tmp/probe_mfma_energy ; run: /tmp/probe_mfma_energy [seconds per variant]
#include <hip/hip_runtime.h>
#include <glob.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <climits>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <unistd.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ inline h8 rnd8(unsigned s, bool zero) {
    h8 r;
    for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        r[e] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 9) & 0x7fff) * (2.0f / 32768.0f) - 1.0f);
    }
    return r;
}

template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, rl = lane & 15, g = lane >> 4;
    constexpr bool ZERO = V == 2, LDS = V == 3 || V == 4, BIG = V == 1 || V == 4;
    if (LDS) {
        for (int i = tid; i < 131072 / 16; i += 256) reinterpret_cast<h8*>(smem)[i] = rnd8(i * 2654435761u + blockIdx.x, false);
        __syncthreads();
    }
    const char* fa = smem + (tid >> 6) * 32768 + rl * 64 + ((g ^ ((rl >> 2) & 3)) << 4);        // k_gemm_fat's fragment addressing (rows of 64 B)
    float sum = 0.f;
    if constexpr (!BIG) {
        h8 a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = rnd8(tid * 977u + i * 131u + blockIdx.x * 7919u, ZERO); b[i] = rnd8(tid * 613u + i * 257u + 99991u + blockIdx.x, ZERO); }
        f4 acc[8][8];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f4{0, 0, 0, 0};
        asm volatile("s_nop 7" ::: "memory");                    // (the asm MFMAs are invisible to the hazard recognizer)
        const long long c0 = clock64(), w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");                       // (LDS is read again in every iteration)
#pragma unroll
            for (int h = 0; h < 2; ++h) {                        // two K halves = 128 MFMAs, as one iteration of k_gemm_fat
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (V == 6) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(b[j]), "v"(a[i]));        // srcA walks, srcB stationary (k_gemm_fat today)
                        else if (V == 7) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][(i & 1) ? 7 - j : j]) : "v"(a[i]), "v"(b[(i & 1) ? 7 - j : j]));   // snake: one operand changes per step
                        else if (V == 5) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
                        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
                    }     // (tied in place: hipcc rotates builtin accumulators through VGPRs)
                    if (LDS) {                                    // 2 reads per 8 MFMAs -> 32 per 128
                        const h8 x = *reinterpret_cast<const h8*>(fa + (i * 2 + h) * 1024);
                        const h8 y = *reinterpret_cast<const h8*>(fa + (16 + i * 2 + h) * 1024);
                        a[(i + 4) & 7] = x; b[(i + 4) & 7] = y;
                    }
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        if (blockIdx.x == 7 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
    } else {
        h8 a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = rnd8(tid * 977u + i * 131u + blockIdx.x * 7919u, false); b[i] = rnd8(tid * 613u + i * 257u + 99991u + blockIdx.x, false); }
        f16v acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        asm volatile("s_nop 7" ::: "memory");
        const long long c0 = clock64(), w0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int h = 0; h < 4; ++h) {                        // four K = 16 steps = the flops of the 128 small MFMAs above
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
                    if (LDS) {                                    // 8 reads per 16 MFMAs -> 32 per 64 (same bytes per flop as V3)
                        const h8 x = *reinterpret_cast<const h8*>(fa + (i * 4 + h) * 1024);
                        const h8 y = *reinterpret_cast<const h8*>(fa + (16 + i * 4 + h) * 1024);
                        a[(i + 2) & 3] = x; b[(i + 2) & 3] = y;
                    }
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        if (blockIdx.x == 7 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][15];
    }
    if (sum == 123.456f) out[blockIdx.x * 256 + tid] = sum;      // (keeps the loop alive)
}

static std::string hw;
static double rd(const char* f) {
    FILE* fp = fopen((hw + "/" + f).c_str(), "r");
    if (!fp) return 0;
    double v = 0; if (fscanf(fp, "%lf", &v) != 1) v = 0; fclose(fp); return v;
}

template <int V>
static void run(const char* name, double secs, float* out, int ncu, long long* clk, int wall_khz) {
    const int iters = 20000;
    const double flop = (double)ncu * 4 * iters * 128.0 * 16384.0;
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::steady_clock::now();
    double tf = 0, w = 0, mhz = 0, cyc = 0, eff = 0; int n = 0, ns = 0;
    std::atomic<int> phase{0};                                   // 0: settling, 1: averaging, 2: done
    std::thread sampler([&] {                                    // the package power / clock files are read WHILE the kernels run (every 20 ms)
        while (phase.load() < 2) { if (phase.load() == 1) { w += rd("power1_input") / 1e6; mhz += rd("freq1_input") / 1e6; ++ns; } usleep(20000); }
    });
    for (;;) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<V>, dim3(ncu), dim3(256), 131072, 0, out, iters, clk);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el > secs * 0.5) { phase.store(1); tf += flop / (ms * 1e-3) / 1e12; cyc += (double)clk[0] / ((double)iters * (V == 1 || V == 4 ? 64 : 128)); eff += (double)clk[0] / (double)clk[1] * wall_khz / 1e3; ++n; }
        if (el > secs) break;
    }
    phase.store(2); sampler.join();
    if (ns == 0) ns = 1;
    printf("{\"variant\": \"%s\", \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"package_w\": %.0f, \"sclk_mhz\": %.0f, \"clock64_per_mfma\": %.2f, \"clock64_mhz\": %.0f, \"launches_averaged\": %d}\n", name, tf / n, tf / n / 2500.0,
           w / ns, mhz / ns, cyc / n, eff / n, n);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 6.0;
    char pci[64] = {0};
    hipDeviceGetPCIBusId(pci, sizeof pci, 0);                    // "0000:xx:00.0": the hwmon directory of THIS device
    for (char* c = pci; *c; ++c) *c = (char)tolower(*c);
    glob_t gl;
    if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &gl) == 0)
        for (size_t i = 0; i < gl.gl_pathc; ++i) {
            std::string f = gl.gl_pathv[i], dev = f.substr(0, f.find("/hwmon/"));
            char real[4096];
            if (realpath(dev.c_str(), real) && std::string(real).find(pci) != std::string::npos) hw = f.substr(0, f.rfind('/'));
            else if (hw.empty() && gl.gl_pathc == 1) hw = f.substr(0, f.rfind('/'));
        }
    fprintf(stderr, "device %s hwmon %s\n", pci, hw.c_str());
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    float* out; hipMalloc(&out, (size_t)p.multiProcessorCount * 256 * 4);
    long long* clk; hipHostMalloc(&clk, 16); clk[0] = clk[1] = 1;
    int wall_khz = 100000; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    run<2>("16x16x32 zeros", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<0>("16x16x32 random", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<1>("32x32x16 random", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<3>("16x16x32 random + LDS fragment reads", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<4>("32x32x16 random + LDS fragment reads", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<6>("16x16x32 random, srcA walks / srcB stationary", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<7>("16x16x32 random, snake order", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<0>("16x16x32 random (srcA stationary / srcB walks, 3rd)", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<6>("16x16x32 random, srcA walks / srcB stationary (2nd)", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<5>("16x16x32 bf16, the same random bits", secs, out, p.multiProcessorCount, clk, wall_khz);
    run<0>("16x16x32 random (again)", secs, out, p.multiProcessorCount, clk, wall_khz);
    return 0;
}
