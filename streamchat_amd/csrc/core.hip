// ABI housekeeping: version, thread-local error text, device probe.
#include "sc_common.h"
#include <string.h>

static thread_local char g_err[512] = "";
char* sc_err_buf() { return g_err; }

extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" const char* sc_last_error(void) { return g_err; }

const char* sc_attn_build_tag();      // attention.hip
const char* sc_decode_build_tag();    // attention_decode.hip
extern "C" const char* sc_build_info(void) {
    static char buf[160] = "";
    if (!buf[0]) snprintf(buf, sizeof(buf), "abi=%d %s %s", SC_ABI_VERSION, sc_attn_build_tag(), sc_decode_build_tag());
    return buf;
}

extern "C" int sc_device_info(int* cu_count, int* is_gfx950, size_t* hbm_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_device_info: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_device_info: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (is_gfx950) *is_gfx950 = (strncmp(p.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return SC_OK;
}

// ---- CU partitioning (round 5: the HBM-bound answer decode next to the MFMA-bound encode / prefill of the next segment) ----
// A stream restricted to the CUs [cu_first, cu_first + cu_count) of the device's CU mask.  On gfx942 / gfx950 consecutive mask bits go
// round-robin over the XCDs (KFD's symmetric CU-mask mapping), so a contiguous range of 8 n bits takes n CUs from EVERY XCD - which is what a
// launch needs: workgroups are dealt to the XCDs round-robin whatever the mask.
// The persistent launches (one workgroup per CU) size their grids for the CUs of the STREAM they go to: a masked stream carries its CU count
// (recorded when it is created), so two host threads driving two partitions never share a process-wide setting (round 5: the reader / updater
// and the QA thread of streamchat_amd/session.py).  (ABI 7 removed the process-wide sc_set_cu_budget: the table below is the only source.)
// One 64-bit word per slot - stream pointer's low 48 bits | CU count << 48 - so a reader never pairs a stream with another stream's count.
#include <atomic>
static constexpr int SC_MAX_MASKED = 32;
static std::atomic<unsigned long long> g_masked[SC_MAX_MASKED];
static inline unsigned long long sc_pack_masked(const void* s, int cus) { return ((unsigned long long)(uintptr_t)s & 0xFFFFFFFFFFFFull) | ((unsigned long long)cus << 48); }
int sc_launch_cu_count(int device_cus, hipStream_t stream) {
    int n = 0;
    if (stream)
        for (int i = 0; i < SC_MAX_MASKED; ++i) {
            const unsigned long long v = g_masked[i].load(std::memory_order_acquire);
            if (v && (v & 0xFFFFFFFFFFFFull) == ((unsigned long long)(uintptr_t)stream & 0xFFFFFFFFFFFFull)) { n = (int)(v >> 48); break; }
        }
    return (n > 0 && n < device_cus) ? n : device_cus;
}

extern "C" int sc_stream_create_masked(int cu_first, int cu_count, int high_priority, sc_stream_t* out) {
    SC_REQUIRE(out, "sc_stream_create_masked: null out pointer");
    int dev = 0, n = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_stream_create_masked: %s", hipGetErrorString(e));
    SC_REQUIRE(cu_first >= 0 && cu_count > 0 && cu_first + cu_count <= n, "sc_stream_create_masked: CU range [%d, %d) outside the device's %d CUs", cu_first, cu_first + cu_count, n);
    uint32_t mask[32] = {};
    SC_REQUIRE(n <= 32 * 32, "sc_stream_create_masked: more than 1024 CUs");
    for (int i = cu_first; i < cu_first + cu_count; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    (void)high_priority;               // (hipExtStreamCreateWithCUMask takes no priority; the CU partition is what separates the two streams)
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((n + 31) / 32), mask);
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_stream_create_masked: %s", hipGetErrorString(e));
    bool kept = false;
    for (int i = 0; i < SC_MAX_MASKED && !kept; ++i) {   // remember the partition's size for sc_launch_cu_count: stream and count claimed in ONE exchange
        unsigned long long expect = 0;
        kept = g_masked[i].compare_exchange_strong(expect, sc_pack_masked(s, cu_count), std::memory_order_acq_rel);
    }
    if (!kept) { (void)hipStreamDestroy(s); return sc_fail(SC_ERR_UNSUPPORTED, "sc_stream_create_masked: more than %d masked streams alive", SC_MAX_MASKED); }
    *out = (sc_stream_t)s;
    return SC_OK;
}

extern "C" int sc_stream_destroy(sc_stream_t s) {
    if (!s) return SC_OK;
    for (int i = 0; i < SC_MAX_MASKED; ++i) {
        unsigned long long v = g_masked[i].load(std::memory_order_acquire);
        if (v && (v & 0xFFFFFFFFFFFFull) == ((unsigned long long)(uintptr_t)s & 0xFFFFFFFFFFFFull) && g_masked[i].compare_exchange_strong(v, 0ull, std::memory_order_acq_rel)) break;
    }
    const hipError_t e = hipStreamDestroy((hipStream_t)s);
    if (e != hipSuccess) return sc_fail(SC_ERR_LAUNCH, "sc_stream_destroy: %s", hipGetErrorString(e));
    return SC_OK;
}
