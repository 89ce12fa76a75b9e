// sample_dev.hpp -- the masked inverse-CDF draw of src/rl_self_play.py:111-127 for ONE board on ONE wave (lane k =
// cell k), shared by sample_moves_wave_kernel (rules_kernels.hip) and the one-launch policy-vs-policy games
// (selfplay_policy_kernel.hip).  float64 in cell order like the reference's numpy code: the two sums stay sequential --
// every lane runs them itself over the wave's row in LDS (broadcast reads), so nothing has to be sent back -- and the
// divisions (one per cell and pass) run side by side.
#pragma once
#include "abi_common.hpp"
#include "othello_dev.hpp"

namespace iago {

// the uniform of (seed, board id, step): word step & 3 of Philox counter block (id, step >> 2, stream, 0)
__device__ __forceinline__ double sample_uniform(uint32_t key0, uint32_t key1, uint32_t id, uint32_t step, uint32_t stream_id)
{
    uint32_t c[4] = {id, step >> 2, stream_id, 0u};
    philox4x32_10(c, key0, key1);
    return (double)((float)(c[step & 3u] >> 8) * (1.0f / 16777216.0f));
}

// probs: the board's 64 probabilities; lg: its legal cells (not 0); row: 2 x 64 doubles of LDS of this wave; k: the lane.
// Returns the sampled cell in every lane, 64 when the legal probabilities hold NaN / inf or sum to nothing
// (numpy.random.choice raises "probabilities contain NaN" there, src/rl_self_play.py:122).
__device__ __forceinline__ int sample_wave(const float *__restrict__ probs, uint64_t lg, double u, double (*row)[64], int k)
{
    const double v = ((lg >> k) & 1ull) ? (double)probs[k] : 0.0;
    row[0][k] = v;
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0); the wave's lanes run in lockstep
    __builtin_amdgcn_wave_barrier();
    double s = 0.0; // np.sum(prob * valid)
    for (int j = 0; j < 64; j++)
        s += row[0][j];
    row[1][k] = v / s;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0, mine = 0.0; // cumsum(p / s): this lane's element, and cdf[-1]
    for (int j = 0; j < 64; j++) {
        acc += row[1][j];
        mine = (j == k) ? acc : mine;
    }
    const double last = acc;
    if (!(s > 0.0) || !(s <= 1.7976931348623157e308) || !(last > 0.0))
        return 64;
    const uint64_t le = __builtin_amdgcn_ballot_w64(mine / last <= u); // searchsorted(cdf, u, side='right')
    return (int)__popcll(le);
}

} // namespace iago
