// Internal helpers shared by the HIP translation units of libstreamchat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/streamchat_hip.h"

#define SC_WAVE 64

// thread-local error text (sc_last_error)
char* sc_err_buf();
static inline int sc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(sc_err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
#define SC_REQUIRE(cond, ...)                                 \
    do {                                                      \
        if (!(cond)) return sc_fail(SC_ERR_ARG, __VA_ARGS__); \
    } while (0)
#define SC_CHECK_LAUNCH(name)                                                                   \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

int sc_launch_cu_count(int device_cus, hipStream_t stream);      // CUs a persistent launch on `stream` may count on (core.hip)

static inline size_t sc_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef _Float16 sc_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 sc_h4 __attribute__((ext_vector_type(4)));
typedef unsigned sc_u2 __attribute__((ext_vector_type(2)));
typedef unsigned sc_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 sc_h2 __attribute__((ext_vector_type(2)));
typedef float sc_f4 __attribute__((ext_vector_type(4)));
typedef float sc_f16v __attribute__((ext_vector_type(16)));
typedef unsigned short sc_us8 __attribute__((ext_vector_type(8)));

// element-type tags for templated loaders
struct ScF16 { typedef _Float16 type; };
struct ScBF16 { typedef unsigned short type; };
struct ScF32 { typedef float type; };

// load 8 consecutive elements (16-byte aligned for 2-byte types, 32-byte span for fp32) as fp32
template <typename Tag>
__device__ __forceinline__ void sc_load8(const void* base, size_t elem_off, float (&out)[8]);
template <>
__device__ __forceinline__ void sc_load8<ScF16>(const void* base, size_t elem_off, float (&out)[8]) {
    sc_h8 v = *reinterpret_cast<const sc_h8*>(reinterpret_cast<const _Float16*>(base) + elem_off);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (float)v[e];
}
template <>
__device__ __forceinline__ void sc_load8<ScBF16>(const void* base, size_t elem_off, float (&out)[8]) {
    sc_us8 v = *reinterpret_cast<const sc_us8*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = __uint_as_float(((unsigned)v[e]) << 16);
}
template <>
__device__ __forceinline__ void sc_load8<ScF32>(const void* base, size_t elem_off, float (&out)[8]) {
    const sc_f4* p = reinterpret_cast<const sc_f4*>(reinterpret_cast<const float*>(base) + elem_off);
    sc_f4 a = p[0], b = p[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { out[e] = a[e]; out[4 + e] = b[e]; }
}
// scalar element load (unaligned / tail path)
template <typename Tag>
__device__ __forceinline__ float sc_load1(const void* base, size_t elem_off);
template <>
__device__ __forceinline__ float sc_load1<ScF16>(const void* base, size_t o) { return (float)reinterpret_cast<const _Float16*>(base)[o]; }
template <>
__device__ __forceinline__ float sc_load1<ScBF16>(const void* base, size_t o) { return __uint_as_float(((unsigned)reinterpret_cast<const unsigned short*>(base)[o]) << 16); }
template <>
__device__ __forceinline__ float sc_load1<ScF32>(const void* base, size_t o) { return reinterpret_cast<const float*>(base)[o]; }

// Full-wave fp32 tree sum with the SC-KM1 pairing (lane a with a+h, h = 32,16,8,4,2,1):
// every lane returns the same value, bitwise equal to oracle/kmeans_oracle.c wave_tree().
__device__ __forceinline__ float sc_wave_tree_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}
