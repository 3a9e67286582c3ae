// ABI housekeeping: version, thread-local error text, device probe.
#include "sc_common.h"
#include <string.h>

static thread_local char g_err[512] = "";
char* sc_err_buf() { return g_err; }

extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" const char* sc_last_error(void) { return g_err; }

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
