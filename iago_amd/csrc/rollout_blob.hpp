// rollout_blob.hpp -- layout of the RolloutPolicy table blob built by
// iago_rollout_build_table (host) and consumed by both rollout kernels.
#pragma once
#include "../../include/iago_hip.h"

namespace iago {

// E[ky][plane][half][row byte][4]: row-pattern contributions to 8 adjacent outputs
constexpr int OFF_E = 0;
constexpr int N_E = 3 * 2 * 2 * 256 * 4; // 12288 floats = 48 KiB
// bias2/b per cell (64); exp(b - max b) in product form
constexpr int OFF_BIAS = OFF_E + N_E;
// mode[0] = 1.0: product form, 0.0: log form
constexpr int OFF_MODE = OFF_BIAS + 64;
// CT[plane][9-bit 3x3 neighbourhood]: per-cell contributions (product form only);
// neighbourhood bit 3*ky + kx = cell (row + ky - 1, col + kx - 1)
constexpr int OFF_CT = OFF_MODE + 4;
constexpr int N_CT = 2 * 512;
static_assert(OFF_CT + N_CT == IAGO_ROLLOUT_TABLE_FLOATS, "blob size");

} // namespace iago
