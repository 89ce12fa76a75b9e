// rollout_kernel.hip -- fused leaf rollout: rules + RolloutPolicy + masked
// sampling + pass/terminal logic + judge, played to the end in registers.
//
// Replaces Simulate(state)(color) of the reference (mcts_self_play.py:9-134);
// see include/iago_hip.h (iago_rollout) for the contract.
//
// Mapping: 8 lanes per board (othello_dev.hpp).  Per turn, per board:
//   1. legal moves: direction-per-lane flood fill, DPP OR-reduce;
//   2. RolloutPolicy (network.py:59-64: conv3x3 2->1, pad 1, + bias[64], softmax):
//      lane r evaluates the 8 cells of board row r.  The conv is linear in the
//      0/1 planes, so a row's 8 logits are
//        bias[r][:] + sum over the 3 input rows (r-1, r, r+1) and the 2 planes of
//        T[ky][plane][row byte][:]
//      with T precomputed for all 256 row patterns and staged in LDS.  In the
//      default PRODUCT form the table holds exp() of those contributions, so
//      the unnormalised softmax numerators are a product of 7 factors: no
//      v_exp_f32, no max pass (the host checks the dynamic range and selects
//      the LOG form -- sum, max, exp2 -- when a product could under/overflow);
//   3. masked inverse-CDF sampling in cell order (numpy.random.choice
//      semantics, mcts_self_play.py:103-106): illegal cells are zeroed with
//      bit-field masks, per-lane running sums, a 3-step DPP scan
//      across the 8 rows; the sampled cell index is the number of cells whose
//      CDF is <= u * total;
//   4. flips: direction-per-lane carry propagation against computed ray masks;
//   5. pass / double pass / full board bookkeeping in the reference's
//      paired-turn loop (mcts_self_play.py:25-29,124-134).
// Uniforms come from Philox4x32-10: the 8 lanes of a group generate 8 counter
// blocks (32 turns) at once; turn t uses word t&3 of counter t>>2; the 4 words
// of a counter are fetched with 4 back-to-back ds_bpermute every 4 turns.
// HBM traffic is 16 B in + 1 B out per game (+ optional trace / final boards);
// everything else stays in VGPRs/LDS.
#include "abi_common.hpp"
#include "othello_dev.hpp"
#include "rollout_blob.hpp"

#include <math.h>

using namespace iago;

namespace {

constexpr int LDS_FLOATS = N_E; // the legal-cell multipliers are computed on the VALU
constexpr float LOG2E = 1.4426950408889634f;

struct RolloutParams {
    const uint64_t *own;
    const uint64_t *opp;
    int64_t n;
    const float *blob;
    const float *uniforms;
    uint32_t key0, key1, id_base, stream_id;
    const uint32_t *stream_id_dev;
    int8_t *z;
    uint64_t *final_own;
    uint64_t *final_opp;
    uint8_t *n_turns;
    uint8_t *trace;
};

__device__ __forceinline__ float4 lds_f4(const float *p) { return *(const float4 *)p; }

typedef float f2 __attribute__((ext_vector_type(2)));
struct Row8 {
    f2 a, b, c, d; // cells (0,1) (2,3) (4,5) (6,7)
};
__device__ __forceinline__ void mul8(Row8 &e, const float4 lo, const float4 hi)
{
    e.a *= (f2){lo.x, lo.y};
    e.b *= (f2){lo.z, lo.w};
    e.c *= (f2){hi.x, hi.y};
    e.d *= (f2){hi.z, hi.w};
}
__device__ __forceinline__ void add8(Row8 &e, const float4 lo, const float4 hi)
{
    e.a += (f2){lo.x, lo.y};
    e.b += (f2){lo.z, lo.w};
    e.c += (f2){hi.x, hi.y};
    e.d += (f2){hi.z, hi.w};
}


// Per-board state carried across turns (replicated in the 8 lanes of the group).
struct Game {
    uint64_t own, opp; // own = side to move
    uint32_t stones;   // stone_num (mcts_self_play.py:15)
    uint32_t nt;       // turns played
    // pass_flg / done as 0/1 integers in VGPRs: the per-turn bookkeeping stays on
    // the VALU (compare results routed through SGPR masks and SALU logic cost a
    // VALU->SALU->VALU round trip per term)
    uint32_t pass_flg, done;
};

template <bool PRODUCT>
__device__ __forceinline__ void play_turn(Game &G, const float u, const uint32_t t,
                                          const Lane8 &L, const float *tbl,
                                          const float (&bias)[8], const uint32_t sh_r,
                                          const uint32_t sh_l, const RolloutParams &P,
                                          const int64_t b)
{
    const uint32_t r = L.l8;
    // ---- table rows of this lane's 3x8 window: issue the LDS reads first
    const uint32_t wo = ((uint32_t)(G.own >> sh_r) << sh_l) & 0xFFFFFFu; // own: plane 1
    const uint32_t wp = ((uint32_t)(G.opp >> sh_r) << sh_l) & 0xFFFFFFu; // opp: plane 0
    float4 ta[6], tb[6];
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
        const uint32_t bp = (wp >> (8 * ky)) & 0xFFu;
        const uint32_t bo = (wo >> (8 * ky)) & 0xFFu;
        const float *tp = tbl + OFF_E + ((ky * 2 + 0) * 2) * 1024 + bp * 4;
        const float *to = tbl + OFF_E + ((ky * 2 + 1) * 2) * 1024 + bo * 4;
        ta[2 * ky] = lds_f4(tp);
        tb[2 * ky] = lds_f4(tp + 1024);
        ta[2 * ky + 1] = lds_f4(to);
        tb[2 * ky + 1] = lds_f4(to + 1024);
    }

    // ---- legal moves of the side to move
    const uint64_t o = to_lane(G.own, L), p = to_lane(G.opp, L);
    const uint64_t legal = group8_legal(o, p, L);
    const uint32_t has = min(1u, (uint32_t)legal | (uint32_t)(legal >> 32)); // 0/1
    const uint32_t lr = (uint32_t)(legal >> (8u * r)) & 0xFFu;

    // ---- unnormalised probabilities e[x] of row r, zero on illegal cells
    Row8 E;
    E.a = (f2){bias[0], bias[1]};
    E.b = (f2){bias[2], bias[3]};
    E.c = (f2){bias[4], bias[5]};
    E.d = (f2){bias[6], bias[7]};
    if (PRODUCT) {
#pragma unroll
        for (int k = 0; k < 6; k++)
            mul8(E, ta[k], tb[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++)
            add8(E, ta[k], tb[k]);
    }
    float e[8] = {E.a.x, E.a.y, E.b.x, E.b.y, E.c.x, E.c.y, E.d.x, E.d.y};
    // legality of the row's cells as all-ones / zero words (v_bfe_i32 of one bit)
    uint32_t lmask[8];
#pragma unroll
    for (int x = 0; x < 8; x++)
        lmask[x] = (uint32_t)((int32_t)(lr << (31 - x)) >> 31);
    if (PRODUCT) {
#pragma unroll
        for (int x = 0; x < 8; x++)
            e[x] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, e[x]) & lmask[x]);
    } else {
        float m = -INFINITY;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            e[x] = lmask[x] ? e[x] : -INFINITY;
            m = fmaxf(m, e[x]);
        }
        m = group8_max(m);
#pragma unroll
        for (int x = 0; x < 8; x++)
            e[x] = lmask[x] ? __builtin_amdgcn_exp2f((e[x] - m) * LOG2E) : 0.0f;
    }

    // ---- inverse-CDF sample in cell order
    float c[8];
    c[0] = e[0];
#pragma unroll
    for (int x = 1; x < 8; x++)
        c[x] = c[x - 1] + e[x];
    float start, total;
    group8_scan(c[7], r, start, total);
    // u * total is rounded BEFORE the subtraction (no FMA contraction): the oracle's
    // floor(u * n) for the uniform policy is reproduced bit for bit
    float thr;
    {
#pragma clang fp contract(off)
        const float ut = u * total;
        thr = ut - start;
    }
    // cnt = #cells with c[x] <= thr: funnel the sign bits of thr - c[x] (set iff
    // c[x] > thr; thr - c == +0 on equality) into one word, one v_alignbit each
    uint32_t over = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
        over = __builtin_amdgcn_alignbit(over, __builtin_bit_cast(uint32_t, thr - c[x]), 31);
    uint32_t cnt = 8u - (uint32_t)__popc(over & 0xFFu);
    cnt = group8_add(cnt);
    uint32_t action = cnt;
    // Rounding can leave the count one cell off a legal one (or at 64): take
    // the next legal cell, else the last one.  Rare, so branch per wave.
    const bool bad = (has != 0u) && (cnt > 63u || ((legal >> (cnt & 63u)) & 1ull) == 0ull);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
        const uint64_t rem = (cnt < 64u) ? (legal & (~0ull << cnt)) : 0ull;
        const uint32_t fix =
            rem ? (uint32_t)__builtin_ctzll(rem) : (63u - (uint32_t)__builtin_clzll(legal | 1ull));
        action = bad ? fix : action;
    }
    action &= 63u;

    // ---- flips and board update (branch-free)
    const uint64_t f = group8_flips(o, p, action, L);
    const uint32_t live_turn = G.done ^ 1u;
    const uint32_t play = has & live_turn;
    const uint32_t passing = (has ^ 1u) & live_turn;
    const uint32_t pm = 0u - play; // all ones iff a stone is placed
    const uint64_t fm = f & (((uint64_t)pm << 32) | pm);
    const uint64_t bit = (uint64_t)play << action;
    const uint64_t nown = G.own | fm | bit;
    const uint64_t nopp = G.opp & ~fm;
    // stone_num += 1 on a move; a second consecutive pass sets it to 64
    // (mcts_self_play.py:126-133); stones never exceeds 64 otherwise
    G.stones = max(G.stones + play, (passing & G.pass_flg) << 6);
    G.pass_flg = (G.pass_flg & G.done) | passing;
    if (P.trace && live_turn && r == 0u)
        P.trace[(int64_t)t * P.n + b] = play ? (uint8_t)action : (uint8_t)IAGO_TRACE_PASS;
    // The other side moves next.  Finished boards keep swapping too: `done` is
    // only raised, and the turn loop only left, after an odd turn, so a finished
    // board is swapped an even number of times and ends in its final orientation.
    G.own = nopp;
    G.opp = nown;
    G.nt += live_turn;
    // `while stone_num < 64` is evaluated once per pair of turns (mcts_self_play.py:26-28)
    if (t & 1u)
        G.done |= G.stones >> 6;
}

template <bool PRODUCT>
__global__ __launch_bounds__(256) void rollout_kernel(RolloutParams P)
{
    __shared__ __attribute__((aligned(16))) float tbl[LDS_FLOATS];
    {
        const float4 *src = (const float4 *)P.blob;
        float4 *dst = (float4 *)tbl;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(LDS_FLOATS / 4); i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();

    const Lane8 L = make_lane8(threadIdx.x);
    const uint32_t r = L.l8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t stream_id = P.stream_id + (P.stream_id_dev ? *P.stream_id_dev : 0u);

    float bias[8]; // this lane's row of bias2/b (exp'ed in product form)
#pragma unroll
    for (int x = 0; x < 8; x++)
        bias[x] = P.blob[OFF_BIAS + r * 8 + x];

    // window of rows r-1, r, r+1 as 24 bits: (x >> sh_r) << sh_l
    const uint32_t sh_r = r ? 8u * (r - 1u) : 0u;
    const uint32_t sh_l = r ? 0u : 8u;

    // Grid-stride over groups of 32 boards (normally one group per block).
    const int64_t per_block = blockDim.x >> 3;
    const int64_t n_groups = (P.n + per_block - 1) / per_block;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t b = grp * per_block + (threadIdx.x >> 3);
        const bool live = b < P.n;

        Game G;
        G.own = live ? P.own[b] : 0ull;
        G.opp = live ? P.opp[b] : 0ull;
        G.stones = (uint32_t)__popcll(G.own | G.opp);
        G.pass_flg = 0u;
        G.done = (!live || G.stones >= 64u) ? 1u : 0u; // `while stone_num < 64` (mcts_self_play.py:26)
        G.nt = 0;
        const uint32_t rid = P.id_base + (uint32_t)b;
        uint32_t rw[4] = {0, 0, 0, 0};

        for (uint32_t t4 = 0; t4 < (uint32_t)IAGO_MAX_TURNS; t4 += 4) {
            float u4[4];
            if (P.uniforms) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    u4[k] = live ? P.uniforms[(int64_t)(t4 + k) * P.n + b] : 0.0f;
            } else {
                if ((t4 & 31u) == 0u) {
                    rw[0] = rid;
                    rw[1] = (t4 >> 2) + r;
                    rw[2] = stream_id;
                    rw[3] = 0u;
                    philox4x32_10(rw, P.key0, P.key1);
                }
                const int src = (int)(((lane & ~7u) + ((t4 >> 2) & 7u)) << 2);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rw[k]);
                    u4[k] = (float)(w >> 8) * (1.0f / 16777216.0f);
                }
            }
            play_turn<PRODUCT>(G, u4[0], t4 + 0, L, tbl, bias, sh_r, sh_l, P, b);
            play_turn<PRODUCT>(G, u4[1], t4 + 1, L, tbl, bias, sh_r, sh_l, P, b);
            if (__builtin_amdgcn_ballot_w64(G.done == 0u) == 0ull)
                break;
            play_turn<PRODUCT>(G, u4[2], t4 + 2, L, tbl, bias, sh_r, sh_l, P, b);
            play_turn<PRODUCT>(G, u4[3], t4 + 3, L, tbl, bias, sh_r, sh_l, P, b);
            if (__builtin_amdgcn_ballot_w64(G.done == 0u) == 0ull)
                break;
        }

        if (live && r == 0u) {
            // nt is even: `own` is the side that was to move at the leaf again
            const int d = __popcll(G.own) - __popcll(G.opp);
            P.z[b] = (int8_t)((d > 0) - (d < 0));
            if (P.final_own)
                P.final_own[b] = G.own;
            if (P.final_opp)
                P.final_opp[b] = G.opp;
            if (P.n_turns)
                P.n_turns[b] = (uint8_t)G.nt;
        }
    }
}

} // namespace

// defined in rollout_lpb_kernel.hip / rollout_row_kernel.hip
void iago_launch_rollout_lpb(const iago_rollout_args *a, void *stream);
void iago_launch_rollout_row(const iago_rollout_args *a, void *stream);

extern "C" {

int iago_rollout_build_table(const float *w18, const float *b64, float *blob)
{
    if (!blob || (w18 && !b64))
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout_build_table: null pointer");
    // T[ky][plane][byte][x] = sum_kx W[plane][ky][kx] * bit(byte, x + kx - 1), float32
    // accumulation in kx order; w18 == NULL: uniform policy (all contributions 0)
    static thread_local float T[3][2][256][8];
    double tmax[3][2], tmin[3][2];
    for (int ky = 0; ky < 3; ky++)
        for (int pl = 0; pl < 2; pl++) {
            tmax[ky][pl] = -1e300;
            tmin[ky][pl] = 1e300;
            for (int byte = 0; byte < 256; byte++)
                for (int x = 0; x < 8; x++) {
                    float acc = 0.0f;
                    for (int kx = 0; kx < 3 && w18; kx++) {
                        const int xx = x + kx - 1;
                        if (xx >= 0 && xx <= 7 && ((byte >> xx) & 1))
                            acc += w18[pl * 9 + ky * 3 + kx];
                    }
                    T[ky][pl][byte][x] = acc;
                    if (acc > tmax[ky][pl]) tmax[ky][pl] = acc;
                    if (acc < tmin[ky][pl]) tmin[ky][pl] = acc;
                }
        }
    double lmax = -1e300, lmin = 1e300;
    for (int x = 0; x < 64; x++) {
        const double bv = w18 ? (double)b64[x] : 0.0;
        if (bv > lmax) lmax = bv;
        if (bv < lmin) lmin = bv;
    }
    for (int ky = 0; ky < 3; ky++)
        for (int pl = 0; pl < 2; pl++) {
            lmax += tmax[ky][pl];
            lmin += tmin[ky][pl];
        }
    // Product form: every factor is shifted by ITS OWN maximum, so each factor is in
    // (0, 1], the full product is exp(logit - lmax) <= 1 and no partial product can
    // overflow; the smallest partial product is >= exp(-(lmax - lmin)), kept normal
    // by the range test.  (Softmax is shift-invariant, so the shifts cancel.)
    const bool product = std::isfinite(lmax) && std::isfinite(lmin) && (lmax - lmin) < 60.0;
    double bmax = -1e300;
    for (int x = 0; x < 64; x++)
        if (w18 && (double)b64[x] > bmax)
            bmax = (double)b64[x];
    if (!w18)
        bmax = 0.0;
    for (int ky = 0; ky < 3; ky++)
        for (int pl = 0; pl < 2; pl++)
            for (int byte = 0; byte < 256; byte++)
                for (int x = 0; x < 8; x++) {
                    const float tv = T[ky][pl][byte][x];
                    const float v = product ? (float)exp((double)tv - tmax[ky][pl]) : tv;
                    blob[OFF_E + (((ky * 2 + pl) * 2 + (x >> 2)) * 256 + byte) * 4 + (x & 3)] = v;
                }
    for (int x = 0; x < 64; x++) {
        const float bv = w18 ? b64[x] : 0.0f;
        blob[OFF_BIAS + x] = product ? (float)exp((double)bv - bmax) : bv;
    }
    blob[OFF_MODE] = product ? 1.0f : 0.0f;
    blob[OFF_MODE + 1] = blob[OFF_MODE + 2] = blob[OFF_MODE + 3] = 0.0f;
    // per-cell tables of the lane-per-board kernel: all 9 taps of one plane at once,
    // float32 accumulation in (ky, kx) order, each table shifted by its own maximum
    for (int pl = 0; pl < 2; pl++) {
        float ct[512];
        double cmax = -1e300;
        for (int idx = 0; idx < 512; idx++) {
            float acc = 0.0f;
            for (int k = 0; k < 9 && w18; k++)
                if ((idx >> k) & 1)
                    acc += w18[pl * 9 + k];
            ct[idx] = acc;
            if (acc > cmax) cmax = acc;
        }
        for (int idx = 0; idx < 512; idx++)
            blob[OFF_CT + pl * 512 + idx] = product ? (float)exp((double)ct[idx] - cmax) : ct[idx];
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
    if (!a->own || !a->opp || !a->z || !a->table)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: own/opp/z/table must not be null");
    if ((uintptr_t)a->table & 15u)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: table must be 16-byte aligned");
    if ((a->log_form != 0 && a->log_form != 1) || a->throughput_hint < 0 || a->throughput_hint > 2)
        return iago_fail(IAGO_ERR_INVALID, "iago_rollout: log_form must be 0 or 1, throughput_hint 0, 1 or 2");
    RolloutParams P;
    P.own = a->own;
    P.opp = a->opp;
    P.n = a->n;
    P.blob = a->table;
    P.uniforms = a->uniforms;
    P.key0 = (uint32_t)a->seed;
    P.key1 = (uint32_t)(a->seed >> 32);
    P.id_base = a->id_base;
    P.stream_id = a->stream_id;
    P.stream_id_dev = a->stream_id_dev;
    P.z = a->z;
    P.final_own = a->final_own;
    P.final_opp = a->final_opp;
    P.n_turns = a->n_turns;
    P.trace = a->trace;
    // 8 lanes per board; 4 waves (32 boards) per block share one staging of the
    // 48 KiB table.  The kernel is latency-bound per wave, so 4 waves on the 4
    // SIMDs of one CU run as fast as on 4 CUs.  One block per group of 32 boards:
    // the hardware dispatcher balances the uneven game lengths better than a
    // persistent grid (measured at 1M boards: 2.81 ms vs 3.12 ms with 768
    // persistent blocks); the kernel's group loop only matters beyond 2^31 threads.
    const int64_t threads = a->n * 8;
    const int block = (threads >= 256) ? 256 : 64;
    const int64_t n_groups = (threads + block - 1) / block;
    const unsigned grid = (unsigned)(n_groups < 0x7fffffffll ? n_groups : 0x7fffffffll);
    // Three kernels, one contract.  Product form (the usual case): the lane-per-board kernel
    // (rollout_lpb_kernel.hip) when throughput matters more than this launch's latency,
    // otherwise the 16-lanes-per-board kernel (rollout_row_kernel.hip).  The
    // 8-lanes-per-board kernel below serves the log form and throughput_hint == 2.
    if (!a->log_form && (a->throughput_hint == 1 || (a->throughput_hint == 0 && a->n >= 32768))) {
        iago_launch_rollout_lpb(a, stream);
        return iago_check_launch("iago_rollout");
    }
    if (!a->log_form && a->throughput_hint == 0) {
        iago_launch_rollout_row(a, stream);
        return iago_check_launch("iago_rollout");
    }
    if (a->log_form)
        hipLaunchKernelGGL(rollout_kernel<false>, dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           P);
    else
        hipLaunchKernelGGL(rollout_kernel<true>, dim3(grid), dim3(block), 0, (hipStream_t)stream, P);
    return iago_check_launch("iago_rollout");
}

} // extern "C"
