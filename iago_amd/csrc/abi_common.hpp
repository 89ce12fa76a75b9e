// abi_common.hpp -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/iago_hip.h"

// Records `msg` as the thread's last error and returns `code`.
int iago_fail(int code, const char *msg);
// hipGetLastError() after a launch -> IAGO_OK / IAGO_ERR_HIP.
int iago_check_launch(const char *where);
