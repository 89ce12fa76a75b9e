// abi_common.hpp -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/iago_hip_experimental.h" // (includes iago_hip.h)

// Records `msg` as the thread's last error and returns `code`.
int iago_fail(int code, const char *msg);
// hipGetLastError() after a launch -> IAGO_OK / IAGO_ERR_HIP.
int iago_check_launch(const char *where);

#include <atomic>
// One-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) per (kernel, device): `done` is a
// per-kernel bit mask of the devices already configured.  Safe from several threads (a lost
// race only repeats an idempotent call).  Returns IAGO_OK or IAGO_ERR_HIP.
int iago_reserve_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done, const char *who);
