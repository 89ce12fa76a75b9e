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

extern "C" {

int iago_abi_version(void) { return 1; }

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
