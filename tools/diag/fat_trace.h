// DIAGNOSTIC BUILD ONLY (never part of libstreamchat_hip.so): force-included in front of streamchat_amd/csrc/gemm.hip by tools/trace_fat.py
// (hipcc -include tools/diag/fat_trace.h).  Per workgroup and tile of k_gemm_fat the 100 MHz timestamps of: tile start (0), end of the K loop
// (1), end of the epilogue (2).  `tid`, `tcount` and `blockIdx` are the kernel's own variables at the three FAT_STAMP sites.
#pragma once
#include <hip/hip_runtime.h>
#define FAT_TRACE_TILES 96
__device__ unsigned long long g_fat_trace[256 * FAT_TRACE_TILES * 3];
extern "C" int sc_fat_trace_read(void* dst, size_t bytes) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fat_trace), bytes) == hipSuccess ? 0 : 1; }
#define FAT_STAMP(slot) do { if (tid == 0 && tcount < FAT_TRACE_TILES && blockIdx.x < 256) g_fat_trace[(blockIdx.x * FAT_TRACE_TILES + tcount) * 3 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
