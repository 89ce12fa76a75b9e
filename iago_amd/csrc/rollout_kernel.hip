// rollout_kernel.hip -- fused leaf rollout: rules + RolloutPolicy + masked
// sampling + pass/terminal logic + judge, played to the end in registers.
//
// Replaces Simulate(state)(color) of the reference (mcts_self_play.py:9-134);
// see include/iago_hip.h (iago_rollout) for the contract.
//
// Mapping: 8 lanes per board (othello_dev.hpp).  Per turn, per board:
//   1. legal moves: direction-per-lane flood fill, DPP OR-reduce;
//   2. RolloutPolicy logits (network.py:59-64: conv3x3 2->1, pad 1, + bias[64]):
//      lane r produces the 8 logits of board row r as
//        bias[r][:] + sum over the 3 input rows (r-1, r, r+1) and the 2 planes of
//        T[ky][plane][row byte][:]
//      where T (48 KiB, staged once per block in LDS) holds, for every possible
//      8-cell row pattern, that row's contribution to 8 adjacent outputs -- the
//      conv is linear in the 0/1 planes, so 6 table rows (12 ds_read_b128) replace
//      144 multiply-adds;
//   3. masked softmax over the legal cells, inverse-CDF sampling in cell order
//      (numpy.random.choice semantics, mcts_self_play.py:103-106): per-lane
//      running sums, a 3-step DPP scan across the 8 rows, and the sampled cell
//      is the number of cells whose CDF is <= u * total;
//   4. flips: direction-per-lane carry propagation against LDS ray masks;
//   5. pass / double pass / full board bookkeeping in the reference's
//      paired-turn loop (mcts_self_play.py:25-29,124-134).
// Uniforms come from Philox4x32-10: the 8 lanes of a group generate 8 counter
// blocks (32 turns) at once; turn t uses word t&3 of counter t>>2, fetched with
// one ds_bpermute.  HBM traffic is 16 B in + 1 B out per game (+ optional
// trace / final boards); everything else stays in VGPRs/LDS.
#include "abi_common.hpp"
#include "othello_dev.hpp"

#include <math.h>

using namespace iago;

namespace {

constexpr int TBL_FLOATS = IAGO_ROLLOUT_TABLE_FLOATS; // [3][2][256][8]
constexpr float LOG2E = 1.4426950408889634f;

struct RolloutParams {
    const uint64_t *own;
    const uint64_t *opp;
    int64_t n;
    const float *table;
    const float *bias;
    const float *uniforms;
    uint32_t key0, key1, id_base, stream_id;
    int8_t *z;
    uint64_t *final_own;
    uint64_t *final_opp;
    uint8_t *n_turns;
    uint8_t *trace;
    int uniform_policy;
};

__device__ __forceinline__ float4 lds_f4(const float *p) { return *(const float4 *)p; }

__global__ __launch_bounds__(256) void rollout_kernel(RolloutParams P)
{
    __shared__ __attribute__((aligned(16))) float tbl[TBL_FLOATS];
    __shared__ uint64_t ray[RAY_TABLE_WORDS];

    const bool use_net = P.uniform_policy == 0;
    if (use_net) {
        const float4 *src = (const float4 *)P.table;
        float4 *dst = (float4 *)tbl;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(TBL_FLOATS / 4); i += blockDim.x)
            dst[i] = src[i];
    }
    fill_ray_table(ray);
    __syncthreads();

    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const uint32_t r = L.l8;
    const bool live = b < P.n;

    uint64_t own = live ? P.own[b] : 0ull;
    uint64_t opp = live ? P.opp[b] : 0ull;
    uint32_t stones = (uint32_t)__popcll(own | opp);
    bool pass_flg = false;
    bool done = !live || stones >= 64u; // `while stone_num < 64` (mcts_self_play.py:26)
    uint32_t nt = 0;

    // this lane's 8 biases (row r of bias2/b)
    float bias[8];
#pragma unroll
    for (int x = 0; x < 8; x++)
        bias[x] = use_net ? P.bias[r * 8 + x] : 0.0f;

    // window of rows r-1, r, r+1 as 24 bits: (x >> sh_r) << sh_l
    const uint32_t sh_r = r ? 8u * (r - 1u) : 0u;
    const uint32_t sh_l = r ? 0u : 8u;

    const uint32_t rid = P.id_base + (uint32_t)b;
    uint32_t rw[4] = {0, 0, 0, 0};
    const uint32_t lane = threadIdx.x & 63u;

    for (uint32_t t = 0; t < (uint32_t)IAGO_MAX_TURNS; t++) {
        // ---- uniform for this turn
        float u;
        if (P.uniforms) {
            u = live ? P.uniforms[(int64_t)t * P.n + b] : 0.0f;
        } else {
            if ((t & 31u) == 0u) {
                rw[0] = rid;
                rw[1] = (t >> 2) + r;
                rw[2] = P.stream_id;
                rw[3] = 0u;
                philox4x32_10(rw, P.key0, P.key1);
            }
            const uint32_t k = t & 3u; // wave-uniform
            const uint32_t mine = (k == 0u) ? rw[0] : (k == 1u) ? rw[1] : (k == 2u) ? rw[2] : rw[3];
            const uint32_t src = (lane & ~7u) + ((t >> 2) & 7u);
            const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)mine);
            u = (float)(w >> 8) * (1.0f / 16777216.0f);
        }

        // ---- legal moves of the side to move
        const uint64_t o = to_lane(own, L), p = to_lane(opp, L);
        const uint64_t legal = group8_legal(o, p, L);
        const bool has = legal != 0ull;

        // ---- RolloutPolicy logits of row r
        float l[8];
#pragma unroll
        for (int x = 0; x < 8; x++)
            l[x] = bias[x];
        if (use_net) {
            const uint32_t wo = ((uint32_t)(own >> sh_r) << sh_l) & 0xFFFFFFu;
            const uint32_t wp = ((uint32_t)(opp >> sh_r) << sh_l) & 0xFFFFFFu;
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const uint32_t bo = (wo >> (8 * ky)) & 0xFFu; // own stones: plane 1
                const uint32_t bp = (wp >> (8 * ky)) & 0xFFu; // opp stones: plane 0
                const float *tp = tbl + ((ky * 2 + 0) * 256 + bp) * 8;
                const float *to = tbl + ((ky * 2 + 1) * 256 + bo) * 8;
                const float4 a0 = lds_f4(tp), a1 = lds_f4(tp + 4);
                const float4 c0 = lds_f4(to), c1 = lds_f4(to + 4);
                l[0] += a0.x; l[1] += a0.y; l[2] += a0.z; l[3] += a0.w;
                l[4] += a1.x; l[5] += a1.y; l[6] += a1.z; l[7] += a1.w;
                l[0] += c0.x; l[1] += c0.y; l[2] += c0.z; l[3] += c0.w;
                l[4] += c1.x; l[5] += c1.y; l[6] += c1.z; l[7] += c1.w;
            }
        }

        // ---- masked softmax + inverse-CDF sample
        const uint32_t lr = (uint32_t)(legal >> (8u * r)) & 0xFFu;
        float m = -INFINITY;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            l[x] = ((lr >> x) & 1u) ? l[x] : -INFINITY;
            m = fmaxf(m, l[x]);
        }
        m = group8_max(m);
        float c[8];
        float acc = 0.0f;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const float e = ((lr >> x) & 1u) ? __builtin_amdgcn_exp2f((l[x] - m) * LOG2E) : 0.0f;
            acc += e;
            c[x] = acc;
        }
        float start, total;
        group8_scan(acc, r, start, total);
        const float thr = u * total;
        uint32_t cnt = 0;
#pragma unroll
        for (int x = 0; x < 8; x++)
            cnt += ((start + c[x]) <= thr) ? 1u : 0u;
        cnt = group8_add(cnt);
        // first legal cell at or above the sampled index; the last legal cell if
        // rounding pushed the threshold past the total
        const uint64_t rem = (cnt < 64u) ? (legal & (~0ull << cnt)) : 0ull;
        const uint32_t action =
            rem ? (uint32_t)__builtin_ctzll(rem) : (63u - (uint32_t)__builtin_clzll(legal | 1ull));

        // ---- flips and board update
        const uint64_t f = group8_flips(o, p, action & 63u, L, ray);
        const bool play = has && !done;
        if (play) {
            const uint64_t bit = 1ull << (action & 63u);
            own = own | f | bit;
            opp = opp & ~f;
            stones += 1u;
            pass_flg = false;
        } else if (!done) {
            if (pass_flg)
                stones = 64u; // double pass ends the game (mcts_self_play.py:131-133)
            pass_flg = true;
        }
        if (!done) {
            if (P.trace && r == 0u)
                P.trace[(int64_t)t * P.n + b] = play ? (uint8_t)action : (uint8_t)IAGO_TRACE_PASS;
            const uint64_t tmp = own; // the other side moves next
            own = opp;
            opp = tmp;
            nt += 1u;
        }
        // `while stone_num < 64` is evaluated once per pair of turns
        if (t & 1u)
            done = done || stones >= 64u;
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull)
            break;
    }

    if (live && r == 0u) {
        // nt is even: `own` is the side that was to move at the leaf again
        const int d = __popcll(own) - __popcll(opp);
        P.z[b] = (int8_t)((d > 0) - (d < 0));
        if (P.final_own)
            P.final_own[b] = own;
        if (P.final_opp)
            P.final_opp[b] = opp;
        if (P.n_turns)
            P.n_turns[b] = (uint8_t)nt;
    }
}

} // namespace

extern "C" {

int iago_rollout_build_table(const float *w18, float *table)
{
    if (!w18 || !table)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout_build_table: null pointer");
    // T[ky][plane][byte][x] = sum_kx W[plane][ky][kx] * bit(byte, x + kx - 1)
    for (int ky = 0; ky < 3; ky++)
        for (int pl = 0; pl < 2; pl++)
            for (int byte = 0; byte < 256; byte++)
                for (int x = 0; x < 8; x++) {
                    float acc = 0.0f;
                    for (int kx = 0; kx < 3; kx++) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx > 7)
                            continue;
                        if ((byte >> xx) & 1)
                            acc += w18[pl * 9 + ky * 3 + kx];
                    }
                    table[((ky * 2 + pl) * 256 + byte) * 8 + x] = acc;
                }
    return IAGO_OK;
}

int iago_rollout(const iago_rollout_args *a, void *stream)
{
    if (!a)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: null args");
    if (a->n < 0 || a->n > 0x7fffffffll)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: n out of range");
    if (a->n == 0)
        return IAGO_OK;
    if (!a->own || !a->opp || !a->z)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: own/opp/z must not be null");
    if (!a->uniform_policy && (!a->table || !a->bias))
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: table/bias required unless uniform_policy");
    if (a->table && ((uintptr_t)a->table & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: table must be 16-byte aligned");
    RolloutParams P;
    P.own = a->own;
    P.opp = a->opp;
    P.n = a->n;
    P.table = a->table;
    P.bias = a->bias;
    P.uniforms = a->uniforms;
    P.key0 = (uint32_t)a->seed;
    P.key1 = (uint32_t)(a->seed >> 32);
    P.id_base = a->id_base;
    P.stream_id = a->stream_id;
    P.z = a->z;
    P.final_own = a->final_own;
    P.final_opp = a->final_opp;
    P.n_turns = a->n_turns;
    P.trace = a->trace;
    P.uniform_policy = a->uniform_policy;
    // 8 lanes per board; 4 waves (32 boards) per block share one staging of the
    // 48 KiB table.  The kernel is latency-bound per wave, so 4 waves on the 4
    // SIMDs of one CU run as fast as on 4 CUs.
    const int64_t threads = a->n * 8;
    const int block = (threads >= 256) ? 256 : 64;
    const unsigned grid = (unsigned)((threads + block - 1) / block);
    hipLaunchKernelGGL(rollout_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, P);
    return iago_check_launch("iago_rollout");
}

} // extern "C"
