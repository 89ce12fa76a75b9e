// rollout_row_kernel.hip -- the leaf rollout with ONE DPP ROW (16 lanes) PER BOARD.
//
// Same contract and results as rollout_kernel.hip (iago_rollout, replacing
// Simulate(state)(color), mcts_self_play.py:9-134); the launcher's choice for small
// batches in product form: BASELINE configs[1] (4096 boards per launch) and the leaf
// rollouts of a PV-MCTS playout (1024 boards).
//
// Why.  A wave alone on its SIMD issues one vector instruction per 4 cycles whatever the
// instruction (MI355X_MICROARCH.md), so a small batch is bound by the LENGTH of one
// wave's instruction stream, not by the chip's issue rate.  4096 boards at 8 lanes per
// board are 512 waves -- one wave on HALF of the 1024 SIMDs, ~235 instructions and ~135
// stall slots per turn (profiles/r02a_pmc_summary.json: 38 us per launch).  At 16 lanes
// per board the batch is exactly one wave per SIMD, and the stream per turn is shorter:
//   * lanes 0-7 of a row keep the board as it is, lanes 8-15 keep it BIT-REVERSED (their
//     own copy of the game state, updated in their own orientation: no v_bfrev / select on
//     the way in).  Lane l floods / resolves flips in direction l & 3 (shifts 1, 7, 8, 9)
//     of its orientation = 8 directions; partial masks are OR-ed inside the quad (two DPP
//     steps), exchanged with the other orientation by row_mirror and bit-reversed once:
//     every lane ends up with the full mask in ITS orientation.  Everything cross-lane is
//     DPP inside one row -- no LDS round trip, no scalar unit;
//   * the rollout policy: lane l owns the four cells 4l .. 4l+3 (lanes 8-15: cells 32..63
//     = cells 31..0 of their reversed board, so every lane works on rows 0-4 and the low
//     legality word of its orientation).  The 3x3 convolution is linear in the 0/1 planes:
//     the four logits are bias + the sum over (kernel row, plane) of a table row indexed by
//     the 5 on-board window bits of that input row: 6 ds_read_b128 from a 12 KB table built
//     from the blob (product form: exp'ed entries, 10 packed multiplies); the reversed
//     orientation reads the entries of the 180-degree rotated kernel;
//   * inverse CDF in cell order = lane order: exclusive prefix over the 16 lanes by row_shr
//     1/2/4/8, the total and the count of cells with CDF <= u * total by DPP butterflies;
//   * Philox: the 16 lanes draw 16 counter blocks = 64 turns at once.
// Measured and rejected on the way: 32 lanes per board (two waves per SIMD,
// ds_swizzle + scalar population counts): 40.8 us -- three times the replicated work per
// board, and the chip's issue rate, not the stream length, becomes the bound.
#include "rollout_row_body.hpp"

using namespace iago_row;

namespace {

template <bool DIAG>
__global__ __launch_bounds__(HW_BLOCK) void rollout_row_kernel(HwParams P)
{
    rollout_row_body<DIAG>(P, blockIdx.x);
}

} // namespace

void iago_launch_rollout_row(const iago_rollout_args *a, void *stream)
{
    const HwParams P = hw_params_of(a);
    const unsigned grid = (unsigned)((a->n + (HW_BLOCK / 16) - 1) / (HW_BLOCK / 16));
    if (P.trace || P.uniforms)
        hipLaunchKernelGGL(rollout_row_kernel<true>, dim3(grid), dim3(HW_BLOCK), 0, (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL(rollout_row_kernel<false>, dim3(grid), dim3(HW_BLOCK), 0, (hipStream_t)stream, P);
}
