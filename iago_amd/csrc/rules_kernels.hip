// rules_kernels.hip -- batched Othello rule kernels (gfx950) + their C ABI.
// Entry points and the reference interfaces they replace: include/iago_hip.h.
#include "abi_common.hpp"
#include "othello_dev.hpp"
#include "sample_dev.hpp"

using namespace iago;

namespace {

constexpr int BLOCK = 256; // 4 waves, 32 boards per block

// 8 lanes per board; lane 0 of each group stores the mask.
__global__ __launch_bounds__(BLOCK) void legal_moves_kernel(const uint64_t *__restrict__ own,
                                                            const uint64_t *__restrict__ opp,
                                                            uint64_t *__restrict__ legal, int64_t n)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t b = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const bool live = b < n;
    const uint64_t o = live ? own[b] : 0ull, p = live ? opp[b] : 0ull;
    const uint64_t mv = group8_legal(to_lane(o, L), to_lane(p, L), L);
    if (live && L.l8 == 0)
        legal[b] = mv;
}

__global__ __launch_bounds__(BLOCK) void apply_moves_kernel(uint64_t *__restrict__ own,
                                                            uint64_t *__restrict__ opp,
                                                            const int8_t *__restrict__ action,
                                                            int64_t n)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t b = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const bool live = b < n;
    const uint64_t o = live ? own[b] : 0ull, p = live ? opp[b] : 0ull;
    const int a = live ? (int)action[b] : -1;
    const uint32_t pos = (uint32_t)a & 63u;
    const uint64_t f = group8_flips(to_lane(o, L), to_lane(p, L), pos, L);
    if (live && L.l8 == 0 && a >= 0) {
        const uint64_t bit = 1ull << pos;
        own[b] = o | f | bit;
        opp[b] = p & ~f & ~bit;
    }
}

// iago_play_turn: the move, the turn bookkeeping, the swap of sides and the next mover's legal
// moves in one pass (8 lanes per game; lane 0 keeps the books).
__global__ __launch_bounds__(BLOCK) void play_turn_kernel(uint64_t *__restrict__ own, uint64_t *__restrict__ opp,
                                                          const int8_t *__restrict__ action,
                                                          const uint8_t *__restrict__ active_in,
                                                          int32_t *__restrict__ stone_num, uint8_t *__restrict__ pass_flg,
                                                          uint8_t *__restrict__ done, int close_pair,
                                                          uint64_t *__restrict__ legal, uint8_t *__restrict__ active_out,
                                                          int64_t n)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t b = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const bool live = b < n;
    uint64_t o = live ? own[b] : 0ull, p = live ? opp[b] : 0ull;
    const bool placed = live && active_in[b] != 0; // the mover had a legal move (and is not done)
    const int a = placed ? (int)action[b] : -1;
    const uint32_t pos = (uint32_t)a & 63u;
    const uint64_t f = group8_flips(to_lane(o, L), to_lane(p, L), pos, L);
    if (a >= 0) {
        const uint64_t bit = 1ull << pos;
        o = o | f | bit;
        p = p & ~f & ~bit;
    }
    // the sides swap: the next mover's stones are `own` from here on
    const uint64_t mv = group8_legal(to_lane(p, L), to_lane(o, L), L);
    if (!live || L.l8 != 0)
        return;
    const bool was_done = done[b] != 0;
    int stones = stone_num[b] + (placed ? 1 : 0);                // stone_num += 1 per stone placed
    const bool passing = !placed && !was_done;                   // the mover had no move
    if (passing && pass_flg[b] != 0)
        stones = 64;                                             // a pass after a pass ends the game
    if (!was_done)
        pass_flg[b] = passing ? 1 : 0;
    stone_num[b] = stones;
    bool dn = was_done;
    if (close_pair) {                                            // `while stone_num < 64` per pair of turns
        dn = dn || stones >= 64;
        done[b] = dn ? 1 : 0;
    }
    own[b] = p;
    opp[b] = o;
    legal[b] = dn ? 0ull : mv;
    active_out[b] = (mv != 0ull && !dn) ? 1 : 0;
}

// One thread per float4 of the (n,2,8,8) tensor: 32 threads cover the 512 B
// of one board, so a wave writes 1 KiB contiguous (fully coalesced stores).
// index: optional gather list (row b of the output encodes board index[b]).
__global__ __launch_bounds__(BLOCK) void encode_planes_kernel(const uint64_t *__restrict__ own,
                                                              const uint64_t *__restrict__ opp,
                                                              const int64_t *__restrict__ index,
                                                              float4 *__restrict__ planes,
                                                              int64_t n,
                                                              const int32_t *__restrict__ n_dev)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t row = gtid >> 5;
    if (row >= n || (n_dev && row >= (int64_t)*n_dev))
        return;
    const int64_t b = index ? index[row] : row;
    const uint32_t k = (uint32_t)gtid & 31u;
    // channel 0 = opponent of the side to move, channel 1 = side to move
    // (game.py:168-174: the colour-1 swap puts the mover in `state==2`)
    const uint64_t bits = (k < 16u) ? opp[b] : own[b];
    const uint32_t nib = (uint32_t)(bits >> ((k & 15u) * 4u)) & 15u;
    float4 v;
    v.x = (nib & 1u) ? 1.0f : 0.0f;
    v.y = (nib & 2u) ? 1.0f : 0.0f;
    v.z = (nib & 4u) ? 1.0f : 0.0f;
    v.w = (nib & 8u) ? 1.0f : 0.0f;
    planes[gtid] = v;
}

__global__ __launch_bounds__(BLOCK) void judge_kernel(const uint64_t *__restrict__ own,
                                                      const uint64_t *__restrict__ opp,
                                                      int8_t *__restrict__ z, int64_t n)
{
    const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= n)
        return;
    const int d = __popcll(own[b]) - __popcll(opp[b]);
    z[b] = (int8_t)((d > 0) - (d < 0));
}


// One thread per board; float64 in cell order like the reference's numpy code.
__global__ __launch_bounds__(BLOCK) void sample_moves_kernel(
    const float *__restrict__ probs, const uint64_t *__restrict__ legal,
    const double *__restrict__ uniforms, uint32_t key0, uint32_t key1, uint32_t id_base,
    uint32_t step, uint32_t stream_id, int8_t *__restrict__ action, int64_t n)
{
    const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= n)
        return;
    const uint64_t lg = legal[b];
    if (lg == 0ull) {
        action[b] = -1;
        return;
    }
    double u;
    if (uniforms) {
        u = uniforms[b];
    } else {
        uint32_t c[4] = {id_base + (uint32_t)b, step >> 2, stream_id, 0u};
        philox4x32_10(c, key0, key1);
        u = (double)((float)(c[step & 3u] >> 8) * (1.0f / 16777216.0f));
    }
    const float *pr = probs + b * 64;
    double s = 0.0; // np.sum(prob * valid)
    for (int k = 0; k < 64; k++)
        s += ((lg >> k) & 1ull) ? (double)pr[k] : 0.0;
    double last = 0.0; // cdf[-1] of cumsum(p / s)
    for (int k = 0; k < 64; k++)
        last += (((lg >> k) & 1ull) ? (double)pr[k] : 0.0) / s;
    // all-zero, NaN or infinite legal probabilities: p / s holds NaNs and numpy.random.choice
    // raises "probabilities contain NaN" (src/rl_self_play.py:122).  Every comparison below
    // would be false (idx 0, a possibly illegal cell): report 64 instead, the host raises.
    if (!(s > 0.0) || !(s <= 1.7976931348623157e308) || !(last > 0.0)) {
        action[b] = 64;
        return;
    }
    double acc = 0.0;
    int idx = 0;
    for (int k = 0; k < 64; k++) {
        acc += (((lg >> k) & 1ull) ? (double)pr[k] : 0.0) / s;
        idx += (acc / last <= u) ? 1 : 0; // searchsorted(cdf, u, side='right')
    }
    action[b] = (int8_t)idx;
}


// The same arithmetic with one WAVE per board (sample_dev.hpp), for the small batches of the self-play loop (64 games of
// src/rl_self_play.py step through a turn in lockstep: one thread per board is 192 dependent float64 divisions on
// one lane, 26 us per turn): bit for bit the kernel above.
__global__ __launch_bounds__(BLOCK) void sample_moves_wave_kernel(
    const float *__restrict__ probs, const uint64_t *__restrict__ legal,
    const double *__restrict__ uniforms, uint32_t key0, uint32_t key1, uint32_t id_base,
    uint32_t step, uint32_t stream_id, int8_t *__restrict__ action, int64_t n)
{
    __shared__ double row[BLOCK / 64][2][64];
    const int w = threadIdx.x >> 6, k = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (BLOCK / 64) + w;
    if (b >= n)
        return; // (the whole wave: no workgroup barrier below)
    const uint64_t lg = legal[b];
    if (lg == 0ull) {
        if (k == 0)
            action[b] = -1;
        return;
    }
    const double u = uniforms ? uniforms[b] : sample_uniform(key0, key1, id_base + (uint32_t)b, step, stream_id);
    const int a = sample_wave(probs + b * 64, lg, u, row[w], k);
    if (k == 0)
        action[b] = (int8_t)a;
}


// Block epilogue: one float4 per thread, 16 threads per (board, channel) plane.
__global__ __launch_bounds__(BLOCK) void bias_relu_kernel(float4 *__restrict__ x,
                                                          const float *__restrict__ bias,
                                                          int64_t n4, int channels)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n4)
        return;
    const float b = bias[(i >> 4) % channels];
    float4 v = x[i];
    v.x = fmaxf(v.x + b, 0.0f);
    v.y = fmaxf(v.y + b, 0.0f);
    v.z = fmaxf(v.z + b, 0.0f);
    v.w = fmaxf(v.w + b, 0.0f);
    x[i] = v;
}


// Board symmetries on a bitboard (bit a = row*8+col).
__device__ __forceinline__ uint64_t bb_transpose(uint64_t x) // (y,x) -> (x,y)
{
    uint64_t t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;
    x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull;
    x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull;
    x ^= t ^ (t << 28);
    return x;
}
// np.rot90 (counter-clockwise): (y,x) -> (7-x, y) = transpose, then flip the rows
__device__ __forceinline__ uint64_t bb_rot90(uint64_t x) { return __builtin_bswap64(bb_transpose(x)); }
__device__ __forceinline__ int act_rot90(int a) { return a < 0 ? a : (7 - (a & 7)) * 8 + (a >> 3); }
__device__ __forceinline__ int act_transpose(int a) { return a < 0 ? a : (a & 7) * 8 + (a >> 3); }

__global__ __launch_bounds__(BLOCK) void augment8_kernel(
    const uint64_t *__restrict__ own, const uint64_t *__restrict__ opp,
    const int8_t *__restrict__ action, uint64_t *__restrict__ own_out,
    uint64_t *__restrict__ opp_out, int8_t *__restrict__ action_out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n)
        return;
    uint64_t o = own[i], p = opp[i];
    int a = (int)action[i];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k == 4) { // load.py:64-68
            o = bb_transpose(o);
            p = bb_transpose(p);
            a = act_transpose(a);
        } else if (k > 0) { // load.py:58-63,69-74
            o = bb_rot90(o);
            p = bb_rot90(p);
            a = act_rot90(a);
        }
        own_out[k * n + i] = o;
        opp_out[k * n + i] = p;
        action_out[k * n + i] = (int8_t)a;
    }
}

inline unsigned grid_for(int64_t threads) { return (unsigned)((threads + BLOCK - 1) / BLOCK); }

} // namespace

extern "C" {

int iago_legal_moves(const uint64_t *own, const uint64_t *opp, uint64_t *legal, int64_t n,
                     void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !legal)))
        return iago_fail(IAGO_ERR_INVALID, "iago_legal_moves: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(legal_moves_kernel, dim3(grid_for(n * 8)), dim3(BLOCK), 0,
                       (hipStream_t)stream, own, opp, legal, n);
    return iago_check_launch("iago_legal_moves");
}

int iago_apply_moves(uint64_t *own, uint64_t *opp, const int8_t *action, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !action)))
        return iago_fail(IAGO_ERR_INVALID, "iago_apply_moves: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(apply_moves_kernel, dim3(grid_for(n * 8)), dim3(BLOCK), 0,
                       (hipStream_t)stream, own, opp, action, n);
    return iago_check_launch("iago_apply_moves");
}

int iago_play_turn(uint64_t *own, uint64_t *opp, const int8_t *action, const uint8_t *active, int32_t *stone_num,
                   uint8_t *pass_flg, uint8_t *done, int close_pair, uint64_t *legal, uint8_t *active_next, int64_t n,
                   void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !action || !active || !stone_num || !pass_flg || !done || !legal ||
                            !active_next)))
        return iago_fail(IAGO_ERR_INVALID, "iago_play_turn: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(play_turn_kernel, dim3(grid_for(n * 8)), dim3(BLOCK), 0, (hipStream_t)stream, own, opp, action,
                       active, stone_num, pass_flg, done, close_pair, legal, active_next, n);
    return iago_check_launch("iago_play_turn");
}

int iago_encode_planes(const uint64_t *own, const uint64_t *opp, float *planes, int64_t n,
                       void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !planes)))
        return iago_fail(IAGO_ERR_INVALID, "iago_encode_planes: null pointer or negative n");
    if (((uintptr_t)planes & 15u) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_encode_planes: planes must be 16-byte aligned");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(encode_planes_kernel, dim3(grid_for(n * 32)), dim3(BLOCK), 0,
                       (hipStream_t)stream, own, opp, (const int64_t *)nullptr, (float4 *)planes, n,
                       (const int32_t *)nullptr);
    return iago_check_launch("iago_encode_planes");
}

int iago_encode_planes_indexed(const uint64_t *own, const uint64_t *opp, const int64_t *index, float *planes,
                               int64_t n, const int32_t *n_dev, void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !index || !planes)))
        return iago_fail(IAGO_ERR_INVALID, "iago_encode_planes_indexed: null pointer or negative n");
    if (((uintptr_t)planes & 15u) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_encode_planes_indexed: planes must be 16-byte aligned");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(encode_planes_kernel, dim3(grid_for(n * 32)), dim3(BLOCK), 0,
                       (hipStream_t)stream, own, opp, index, (float4 *)planes, n, n_dev);
    return iago_check_launch("iago_encode_planes_indexed");
}

int iago_judge(const uint64_t *own, const uint64_t *opp, int8_t *z, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !z)))
        return iago_fail(IAGO_ERR_INVALID, "iago_judge: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(judge_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, own,
                       opp, z, n);
    return iago_check_launch("iago_judge");
}

int iago_sample_moves(const float *probs, const uint64_t *legal, const double *uniforms,
                      uint64_t seed, uint32_t id_base, uint32_t step, uint32_t stream_id,
                      int8_t *action, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!probs || !legal || !action)))
        return iago_fail(IAGO_ERR_INVALID, "iago_sample_moves: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    if (n <= 16384) // (a wave per board up to 64 waves per CU's worth of boards, a lane per board beyond)
        hipLaunchKernelGGL(sample_moves_wave_kernel, dim3((unsigned)((n + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0,
                           (hipStream_t)stream, probs, legal, uniforms, (uint32_t)seed, (uint32_t)(seed >> 32), id_base,
                           step, stream_id, action, n);
    else
        hipLaunchKernelGGL(sample_moves_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream,
                           probs, legal, uniforms, (uint32_t)seed, (uint32_t)(seed >> 32), id_base,
                           step, stream_id, action, n);
    return iago_check_launch("iago_sample_moves");
}

int iago_bias_relu(float *x, const float *bias, int64_t n, int32_t channels, void *stream)
{
    if (n < 0 || channels < 1 || (n > 0 && (!x || !bias)))
        return iago_fail(IAGO_ERR_INVALID, "iago_bias_relu: null pointer or bad size");
    if ((uintptr_t)x & 15u)
        return iago_fail(IAGO_ERR_INVALID, "iago_bias_relu: x must be 16-byte aligned");
    if (n == 0)
        return IAGO_OK;
    const int64_t n4 = n * channels * 16;
    hipLaunchKernelGGL(bias_relu_kernel, dim3(grid_for(n4)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (float4 *)x, bias, n4, (int)channels);
    return iago_check_launch("iago_bias_relu");
}

int iago_augment8(const uint64_t *own, const uint64_t *opp, const int8_t *action, uint64_t *own_out,
                  uint64_t *opp_out, int8_t *action_out, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!own || !opp || !action || !own_out || !opp_out || !action_out)))
        return iago_fail(IAGO_ERR_INVALID, "iago_augment8: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(augment8_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, own,
                       opp, action, own_out, opp_out, action_out, n);
    return iago_check_launch("iago_augment8");
}

} // extern "C"
