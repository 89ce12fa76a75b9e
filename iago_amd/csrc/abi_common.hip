// abi_common.hip -- version, device probe and error reporting of the C ABI.
#include "abi_common.hpp"

#include <stdio.h>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

int iago_fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int iago_check_launch(const char *where)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        return IAGO_OK;
    snprintf(g_err, sizeof g_err, "%s: %s", where, hipGetErrorString(e));
    return IAGO_ERR_HIP;
}

int iago_reserve_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done, const char *who)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return iago_fail(IAGO_ERR_HIP, who);
    }
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit)
        return IAGO_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return iago_fail(IAGO_ERR_HIP, who);
    }
    done.fetch_or(bit, std::memory_order_release);
    return IAGO_OK;
}

extern "C" {

int iago_abi_version(void) { return 13; }

const char *iago_last_error(void) { return g_err; }

int iago_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

} // extern "C"
