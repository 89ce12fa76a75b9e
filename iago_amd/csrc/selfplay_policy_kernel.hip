// selfplay_policy_kernel.hip -- whole policy-vs-policy games (src/rl_self_play.py:8-149: Game(model1, model2)()) in ONE
// launch: a workgroup plays a game from its first turn to its last.
//
// A turn of such a game is one forward of the mover's SLPolicy on ONE board (118 us: the one-board walk of
// conv_policy_body.hpp, the kernel iago_policy_forward_split3 launches), the masked draw of
// src/rl_self_play.py:111-127, the stone and the books of src/rl_self_play.py:27-31,130-145.  Turn by turn from
// the host that is five launches and their gaps for the 64 games of a REINFORCE set (142 us per turn, the walk
// 122 of them as two half-net launches); here the walk's workgroup simply goes on: sample_wave on its first wave,
// the flips and the books on its lanes, the next walk.  The games of a set finish together either way (they are
// as long as their turns are many); what the launch saves is everything between the walks.
//
// Arithmetic: policy_item (the same device function: the same probabilities bit for bit), sample_wave /
// sample_uniform (iago_sample_moves' wave form), play_turn_kernel's bookkeeping: the games are those of
// rl_self_play.play_batch's launch-per-turn loop, record for record (tests/test_mirrors_gpu.py).
#include "abi_common.hpp"
#include "othello_dev.hpp"
#include "sample_dev.hpp"
#include "conv_policy_body.hpp"

#include <atomic>

namespace {
using namespace iago;
using namespace iago_policy;

constexpr int STAGE_FLOATS = 64 * 18 + 64 + 128 + 64;          // a model's block1 weights + biases, conv9, bias10
constexpr int SELFPLAY_LDS = LDS_BYTES + 2 * STAGE_FLOATS * 4; // the walk's image + both models' small weights

struct SelfplayParams {
    uint64_t *own, *opp;          // [n] in: start (own = colour 1, the first mover); out: the final boards (own = colour 1)
    uint64_t *row_own, *row_opp;  // [n] the rows the policy walks read (= P1.own / P2.own)
    int64_t n;
    uint32_t key0, key1, id_base;
    int32_t max_turns;            // even
    uint64_t *rec_own, *rec_opp;  // [max_turns / 2][n]: the position before each of colour 1's turns
    int8_t *rec_act;              // [max_turns / 2][n]: colour 1's move, -1 = none
    int32_t *n_turns;             // [n]: the (even) turn at which `while stone_num < 64` ended the game, or max_turns
    uint32_t *bad_probs;          // raised when a draw met NaN / zero-mass probabilities
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void selfplay_policy_kernel(
    SelfplayParams S, PolicyParams P1, PolicyParams P2)
{
    __shared__ double srow[2][64];
    __shared__ int s_action;
    // The walk's hand-offs stay in LDS (the persistent search's form of policy_item): the position in, the distribution
    // out -- through the row arrays and P.probs each was a store to global memory read back by this very workgroup, a round
    // trip to L2 (~2 us) in front of and behind every walk; both models' block1 and head weights are staged once per
    // workgroup above the walk's image instead of once per walk.  (The rows and the distributions are still WRITTEN to
    // memory, as the interface says; nothing waits for those stores.)
    __shared__ uint32_t s_pos[8];
    __shared__ float s_probs[64];
    const int tid = threadIdx.x;
    float *const stage = (float *)(policy_lds + LDS_BYTES);
    float *const w1s[2] = {stage, stage + STAGE_FLOATS};
    float *const heads[2] = {stage + (64 * 18 + 64), stage + STAGE_FLOATS + (64 * 18 + 64)};
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const PolicyParams &P = m ? P2 : P1;
        for (int e = tid; e < 64 * 18 / 4; e += 256)
            ((float4 *)w1s[m])[e] = ((const float4 *)P.w1)[e];
        if (tid < 16)
            ((float4 *)(w1s[m] + 64 * 18))[tid] = ((const float4 *)P.b1)[tid];
        if (tid < 128)
            heads[m][tid] = P.w9[tid];
        if (tid < 64)
            heads[m][128 + tid] = P.b10[tid];
    }
    __syncthreads();
    const Lane8 L = make_lane8(threadIdx.x); // (every group of 8 lanes computes the game's board functions: one value in all)
    for (int64_t g = blockIdx.x; g < S.n; g += gridDim.x) {
        uint64_t own = S.own[g], opp = S.opp[g];
        int stones = 4; // src/rl_self_play.py:20 (the handicap stone is not counted there either)
        bool pass_flg = false, done = false;
        int ended = S.max_turns;
        // one turn of the game with the mover's net (two call sites: each walk reads its own parameter block)
        auto turn = [&](const PolicyParams &P, const int m, const int t) __attribute__((always_inline)) {
            const uint64_t lg = group8_legal(to_lane(own, L), to_lane(opp, L), L);
            const bool placed = lg != 0ull && !done; // the mover has a move (and the game is not over)
            int a = -1;
            if (placed) {
                if (tid == 0) {
                    S.row_own[g] = own;
                    S.row_opp[g] = opp;
                    s_pos[2] = (uint32_t)own, s_pos[3] = (uint32_t)(own >> 32);
                    s_pos[4] = (uint32_t)opp, s_pos[5] = (uint32_t)(opp >> 32);
                }
                __syncthreads();
                // model(make_state_var(state, color)), src/rl_self_play.py:113-116
                policy_item<true>(P, g, s_pos, s_probs, w1s[m], heads[m]);
                __syncthreads();
                if (tid < 64) {
                    P.probs[g * 64 + tid] = s_probs[tid];
                    const double u = sample_uniform(S.key0, S.key1, S.id_base + (uint32_t)g, (uint32_t)t, 0u);
                    const int drawn = sample_wave(s_probs, lg, u, srow, tid);
                    if (tid == 0)
                        s_action = drawn;
                }
                __syncthreads();
                a = s_action;
                if (a > 63 && tid == 0)
                    *S.bad_probs = 1u;
            }
            if (!(t & 1) && tid == 0) { // colour 1's plies are the recorded ones (src/rl_self_play.py:134-138)
                const int64_t at = (int64_t)(t >> 1) * S.n + g;
                S.rec_own[at] = own;
                S.rec_opp[at] = opp;
                S.rec_act[at] = (int8_t)a;
            }
            // iago_play_turn's arithmetic (src/rl_self_play.py:27-31,130-145)
            const uint32_t pos = (uint32_t)a & 63u;
            const uint64_t f = group8_flips(to_lane(own, L), to_lane(opp, L), pos, L);
            uint64_t o = own, p = opp;
            if (a >= 0) {
                const uint64_t bit = 1ull << pos;
                o = o | f | bit;
                p = p & ~f & ~bit;
            }
            stones += placed ? 1 : 0;
            const bool passing = !placed && !done;
            if (passing && pass_flg)
                stones = 64; // a pass after a pass ends the game
            if (!done)
                pass_flg = passing;
            if (t & 1) { // `while stone_num < 64` once per pair of turns
                if (!done && stones >= 64)
                    ended = t + 1;
                done = done || stones >= 64;
            }
            own = p; // the sides swap
            opp = o;
            __syncthreads(); // s_action / the rows are rewritten by the next turn
        };
#pragma unroll 1
        for (int t = 0; t < S.max_turns; t += 2) {
            turn(P1, 0, t);
            turn(P2, 1, t + 1);
        }
        if (tid == 0) {
            S.own[g] = own; // an even number of swaps: colour 1 again
            S.opp[g] = opp;
            S.n_turns[g] = ended;
        }
    }
}

} // namespace

extern "C" int iago_selfplay_policy(const iago_selfplay_policy_args *a, void *stream)
{
    if (!a || a->n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_selfplay_policy: null args or n < 0");
    if (a->n == 0)
        return IAGO_OK;
    if (!a->model1 || !a->model2 || !a->own || !a->opp || !a->rec_own || !a->rec_opp || !a->rec_act || !a->n_turns || !a->bad_probs)
        return iago_fail(IAGO_ERR_INVALID, "iago_selfplay_policy: null pointer");
    if (a->max_turns < 2 || (a->max_turns & 1) || a->max_turns > IAGO_MAX_TURNS)
        return iago_fail(IAGO_ERR_INVALID, "iago_selfplay_policy: max_turns is even, 2 .. IAGO_MAX_TURNS");
    PolicyParams P1, P2;
    if (const int rc = policy_params_of(a->model1, P1))
        return rc;
    if (const int rc = policy_params_of(a->model2, P2))
        return rc;
    if (a->model1->n < a->n || a->model2->n < a->n || a->model1->own != a->model2->own || a->model1->opp != a->model2->opp ||
        a->model1->index || a->model2->index || a->model1->n_dev || a->model2->n_dev)
        return iago_fail(IAGO_ERR_INVALID, "iago_selfplay_policy: both models read their rows from the SAME own / opp arrays of "
                                           ">= n rows (the launch writes them), no gather list, no device count");
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)selfplay_policy_kernel, SELFPLAY_LDS, configured,
                         "iago_selfplay_policy: cannot reserve 70 KB of LDS"))
        return IAGO_ERR_HIP;
    SelfplayParams S;
    S.own = a->own;
    S.opp = a->opp;
    S.row_own = (uint64_t *)a->model1->own;
    S.row_opp = (uint64_t *)a->model1->opp;
    S.n = a->n;
    S.key0 = (uint32_t)a->seed;
    S.key1 = (uint32_t)(a->seed >> 32);
    S.id_base = a->id_base;
    S.max_turns = a->max_turns;
    S.rec_own = a->rec_own;
    S.rec_opp = a->rec_opp;
    S.rec_act = a->rec_act;
    S.n_turns = a->n_turns;
    S.bad_probs = a->bad_probs;
    const unsigned grid = (unsigned)(a->n < 256 ? a->n : 256); // one game per workgroup, one workgroup per CU
    hipLaunchKernelGGL(selfplay_policy_kernel, dim3(grid), dim3(256), SELFPLAY_LDS, (hipStream_t)stream, S, P1, P2);
    return iago_check_launch("iago_selfplay_policy");
}
