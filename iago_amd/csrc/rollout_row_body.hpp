// rollout_row_body.hpp -- the body of the 16-lanes-per-board rollout kernel (rollout_row_kernel.hip
// holds the description) as a device function of (parameters, workgroup index): launched on its
// own by iago_rollout and, beside the Value net's workgroups, by iago_value_rollout
// (conv_trunk_kernel.hip).
#pragma once
#include "abi_common.hpp"
#include "othello_dev.hpp"
#include "rollout_blob.hpp"

namespace iago_row {
using namespace iago;


constexpr int HW_BLOCK = 256; // 16 boards per workgroup: 4096 boards = one workgroup per CU
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_MIRROR = 0x140; // lane i <-> 15 - i inside a row of 16
constexpr int DPP_ROW_BCAST15 = 0x15F; // row_newbcast:15: every lane reads lane 15 of its row
constexpr int N_T4 = 2 * 3 * 2 * 2 * 32; // [orientation][kernel row][plane][row half][5 window bits]

struct HwParams {
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
    // game-asynchronous search steps (iago_value_rollout_async): only the boards with mask[b] != 0
    // are played (the others keep their z), board b draws from Philox stream stream_id + stream_ids[b]
    const uint8_t *mask;
    const int32_t *stream_ids;
};

typedef float f4 __attribute__((ext_vector_type(4)));


template <int TT>
__device__ __forceinline__ uint64_t bitop64(uint64_t a, uint64_t b, uint64_t c)
{
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32),
                                                    (uint32_t)(c >> 32), TT);
    return ((uint64_t)hi << 32) | lo;
}
constexpr int TT_A_OR_BC = 0xF8;  // a | (b & c)
constexpr int TT_AB_OR_C = 0xEA;  // (a & b) | c
constexpr int TT_NOR_AND = 0x02;  // ~a & ~b & c  (a=F0, b=CC, c=AA: only minterm 001)
constexpr int TT_A_OR_NB = 0xF3;  // a | ~b
constexpr int TT_A_NB_C = 0x20;   // a & ~b & c
constexpr int TT_ABC = 0x80;      // a & b & c

template <int CTRL>
__device__ __forceinline__ float dpp_or_zero(float v)
{
    // lanes without a valid source read 0
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a row, in every lane
__device__ __forceinline__ float row_sum(float x)
{
    x += dpp_f32<DPP_XOR1>(x);
    x += dpp_f32<DPP_XOR2>(x);
    x += dpp_f32<DPP_HALF_MIRROR>(x);
    x += dpp_f32<DPP_ROW_MIRROR>(x);
    return x;
}
__device__ __forceinline__ uint32_t row_sum(uint32_t x)
{
    x += dpp_u32<DPP_XOR1>(x);
    x += dpp_u32<DPP_XOR2>(x);
    x += dpp_u32<DPP_HALF_MIRROR>(x);
    x += dpp_u32<DPP_ROW_MIRROR>(x);
    return x;
}

// OR over the 8 directions: within the quad (4 directions of this orientation), then with
// the other orientation's quad result, bit-reversed into this lane's orientation.
__device__ __forceinline__ uint64_t reduce_dirs(uint64_t x)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo |= dpp_u32<DPP_XOR1>(lo);
    hi |= dpp_u32<DPP_XOR1>(hi);
    lo |= dpp_u32<DPP_XOR2>(lo);
    hi |= dpp_u32<DPP_XOR2>(hi);
    const uint32_t olo = dpp_u32<DPP_ROW_MIRROR>(lo); // quads 0,1 (as is) <-> quads 3,2 (reversed)
    const uint32_t ohi = dpp_u32<DPP_ROW_MIRROR>(hi);
    lo |= __builtin_bitreverse32(ohi); // rev64(o) = (bfrev(o.hi), bfrev(o.lo))
    hi |= __builtin_bitreverse32(olo);
    return ((uint64_t)hi << 32) | lo;
}

// Per-lane constants.
struct LaneHw {
    uint32_t l;        // lane within the row, 0..15
    uint32_t rev63;    // 63 for the lanes of the reversed orientation, else 0
    uint32_t sh, sh2;  // shift of this lane's direction (1, 7, 8, 9) and twice that, opaque to the compiler
    uint64_t wrap;     // destination mask of that shift (A/H-file wrap-around)
    uint64_t base;     // ray of the direction from cell 0, without cell 0
    uint32_t ca, cx;   // column-mask recipe (othello_dev.hpp: ray_mask)
    // policy: oriented cell quad q = cells 4q .. 4q+3 of this lane's board: row q >> 1, half q & 1
    uint32_t sh_w;     // 8 * (q >> 1) + 3 * (q & 1): funnel shift that brings the 5 window columns of rows r-1 .. r+1 to bits 8 ky + 4
    uint32_t tbase;    // byte offset of this (orientation, half)'s tables
    uint32_t bit[4];   // legality bits of the lane's cells (TRUE cell order) in its low legal word
    f4 bias;           // bias factors of those cells
};

__device__ __forceinline__ uint64_t ray_mask_hw(const LaneHw &L, uint32_t pl)
{
    const uint32_t m8 = (L.ca << (pl & 7u)) ^ L.cx;
    const uint32_t m32 = __builtin_amdgcn_perm(m8, m8, 0u);
    return (L.base << pl) & (((uint64_t)m32 << 32) | m32);
}

// game.py:210-235 for this lane's direction, on its own orientation; all 8 directions after
// reduce_dirs.  Shifts by a VGPR amount: one v_lshlrev_b64 each.
__device__ __forceinline__ uint64_t legal_hw(uint64_t o, uint64_t p, const LaneHw &L)
{
    // parallel-prefix flood through up to 6 opponent stones: 1, 2, 4, 6 (the same set as six
    // single steps, in four dependent shift + bit-op groups instead of six)
    const uint64_t pm = p & L.wrap;
    uint64_t t = (o << L.sh) & pm;
    t = bitop64<TT_A_OR_BC>(t, t << L.sh, pm);
    const uint64_t pm2 = pm & (pm << L.sh);
    t = bitop64<TT_A_OR_BC>(t, t << L.sh2, pm2);
    t = bitop64<TT_A_OR_BC>(t, t << L.sh2, pm2);
    const uint64_t em = bitop64<TT_NOR_AND>(o, p, L.wrap); // empty cells, wrap-masked
    return reduce_dirs((t << L.sh) & em);
}

// game.py:180-207 for this lane's direction (carry propagation, othello_dev.hpp), position
// in the lane's orientation.
__device__ __forceinline__ uint64_t flips_hw(uint64_t o, uint64_t p, uint32_t pos_l, const LaneHw &L)
{
    const uint64_t M = ray_mask_hw(L, pos_l);
    const uint64_t x = bitop64<TT_A_OR_NB>(p, M, 0ull);
    const uint64_t t = x + 1ull;
    const uint64_t cand = bitop64<TT_A_NB_C>(p, t, M); // the opponent run the carry went through
    const uint64_t of = bitop64<TT_ABC>(t, M, o);
    return reduce_dirs(of ? cand : 0ull); // kept iff an own stone brackets the run
}

// DIAG: the instance of the parity tests (the launch records the action of every turn and / or
// takes its uniforms from a buffer); a wave alone on its SIMD pays a full issue slot for every
// instruction, scalar ones and branches included, so the production instance does not even
// test those pointers.
// ASYNC: the per-board mask and stream ids of HwParams (prologue only; the turn loop is the same).
// INDEXED (the persistent search): the 16 boards of the call are boards idx[0..15] (-1 = no board in that row of
// lanes) instead of boards 16 block_id ..; `table_ready`: the factor table is in LDS already (an earlier call of
// this workgroup put it there).
// (INDEXED, the persistent search again) `hand`: the boards' positions and Philox stream offsets come from, and their results
// go to, LDS arrays indexed by the board's number within the workgroup (idx[..] - base) -- written / read by this very
// workgroup around the call: through global memory each was a store followed by a load of the same address, a round trip to
// L2 at the head of every pass.
struct RowHandoff {
    const uint64_t *own, *opp; // [board of the workgroup]
    const int32_t *stream;
    int8_t *z;
    int32_t base;              // idx[..] of the workgroup's board 0
};
template <bool DIAG, bool ASYNC = false, bool INDEXED = false>
__device__ __forceinline__ void rollout_row_body(const HwParams &P, const uint32_t block_id, const int32_t *idx = nullptr,
                                                 const bool table_ready = false, const RowHandoff *hand = nullptr)
{
    // T4[orientation][kernel row][plane][row half][5 window bits] -> factors of the 4 cells
    // (in TRUE cell order), 12 KB, gathered from the blob's row tables.  As-is orientation,
    // half 0: window = columns 0-4 = row byte m, cells 0-3; half 1: columns 3-7 = byte
    // m << 3, cells 4-7.  Reversed orientation = the 180-degree rotated kernel: kernel row
    // 2 - ky, the window's columns mirrored (oriented column c = true column 7 - c), the
    // other half of the true row -- whose four factors then already are in true order.
    __shared__ f4 t4[N_T4];
    // the table's global loads go out first (3 per thread, all in flight together); lane
    // constants, the board loads and the first Philox blocks are computed under their latency
    f4 stg[N_T4 / HW_BLOCK];
    if (!(INDEXED && table_ready)) {
#pragma unroll
    for (uint32_t i = 0; i < (uint32_t)(N_T4 / HW_BLOCK); i++) {
        const uint32_t e = threadIdx.x + i * HW_BLOCK;
        const uint32_t orient = e / 384u, rem = e % 384u;
        const uint32_t ky = rem >> 7, pl = (rem >> 6) & 1u, hf = (rem >> 5) & 1u, m = rem & 31u;
        uint32_t sky = ky, shf = hf, byte = hf ? (m << 3) & 0xFFu : m;
        if (orient) {
            sky = 2u - ky;
            shf = hf ^ 1u;
            // the window's columns mirrored: bit bb of m -> bit 4 - bb (half 1) / 7 - bb (half 0)
            const uint32_t r5 = __builtin_bitreverse32(m) >> 27;
            byte = hf ? r5 : (r5 << 3);
        }
        stg[i] = *(const f4 *)(P.blob + OFF_E + (((sky * 2u + pl) * 2u + shf) * 256u + byte) * 4u);
    }
    }
    const uint32_t lane = threadIdx.x & 63u;
    LaneHw L;
    L.l = lane & 15u;
    const bool rev = L.l >= 8u;
    L.rev63 = rev ? 63u : 0u;
    const uint32_t k = L.l & 3u;
    {
        uint32_t s = (k == 0u) ? 1u : (6u + k);
        asm("v_mov_b32 %0, %1" : "=v"(L.sh) : "v"(s)); // keep 64-bit shifts one instruction
        asm("v_mov_b32 %0, %1" : "=v"(L.sh2) : "v"(2u * s));
    }
    L.wrap = (k == 1u) ? ~FILE_H : ((k == 2u) ? ~0ull : ~FILE_A);
    L.base = (k == 0u)   ? 0x00000000000000FEull
             : (k == 1u) ? 0x0002040810204080ull
             : (k == 2u) ? 0x0101010101010100ull
                         : 0x8040201008040200ull;
    L.ca = (k == 1u) ? 0xFFu : ((k == 2u) ? 0u : 0xFEu);
    L.cx = (k == 0u || k == 3u) ? 0u : 0xFFu;
    const uint32_t q = rev ? 15u - L.l : L.l; // oriented quad of cells, 0..7
    const uint32_t pr = q >> 1, pp = q & 1u;
    L.sh_w = 8u * pr + 3u * pp;
    // window word w4 = ((board << 12) >> (8 pr + 3 pp)): column c of oriented row pr + ky - 1
    // at bit 8 ky + c + 4 - 3 pp, i.e. the 5 window columns (0-4 / 3-7) at bits 8 ky + 4 ..
    // 8 ky + 8 (<< 4: 16-byte table entries); the zero bits shifted in are row -1
    L.tbase = (rev ? 384u * 16u : 0u) + pp * 32u * 16u;
    // the lane's cells in TRUE order: as-is 4q .. 4q+3; reversed orientation: true cells
    // 4 l .. 4 l + 3 are oriented bits 4q+3 .. 4q
    for (uint32_t j = 0; j < 4; j++)
        L.bit[j] = rev ? 4u * q + 3u - j : 4u * q + j;
    L.bias = *(const f4 *)(P.blob + OFF_BIAS + 4u * L.l);
    uint32_t stream_id = P.stream_id + (P.stream_id_dev ? *P.stream_id_dev : 0u);
    const char *const tb = (const char *)t4;
    uint32_t m1f0;
    asm("v_mov_b32 %0, 0x1f0" : "=v"(m1f0));

    int64_t b = (int64_t)block_id * (HW_BLOCK / 16) + (threadIdx.x >> 4);
    bool live = b < P.n;
    if constexpr (INDEXED) {
        const int32_t at = idx[threadIdx.x >> 4];
        live = at >= 0;
        b = live ? at : 0;
    }
    uint64_t own, opp; // side to move, in this lane's orientation below
    if (INDEXED && hand) {
        const int slot = live ? (int)b - hand->base : 0;
        stream_id += live ? (uint32_t)hand->stream[slot] : 0u;
        own = live ? hand->own[slot] : 0ull;
        opp = live ? hand->opp[slot] : 0ull;
    } else {
        if constexpr (ASYNC) {
            live = live && P.mask[b] != 0;
            stream_id += live ? (uint32_t)P.stream_ids[b] : 0u;
        }
        own = live ? P.own[b] : 0ull;
        opp = live ? P.opp[b] : 0ull;
    }
    uint32_t stones = (uint32_t)__popcll(own | opp);
    if (rev) {
        own = rev64(own);
        opp = rev64(opp);
    }
    uint32_t pass_flg = 0u, nt = 0u;
    uint32_t done = (!live || stones >= 64u) ? 1u : 0u; // `while stone_num < 64` (mcts_self_play.py:26)

    // lane l draws Philox counter block l: the 16 lanes of a row hold the uniforms of 64 turns
    // (kept as the float32 uniforms themselves: converted once per 64 turns, not once per fetch)
    uint32_t rw[4] = {P.id_base + (uint32_t)b, L.l, stream_id, 0u};
    auto draw = [&]() __attribute__((always_inline)) {
        philox4x32_10(rw, P.key0, P.key1);
#pragma unroll
        for (int i = 0; i < 4; i++)
            rw[i] = __float_as_uint((float)(rw[i] >> 8) * (1.0f / 16777216.0f));
    };
    if (!(DIAG && P.uniforms))
        draw();
    if (!(INDEXED && table_ready)) {
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)(N_T4 / HW_BLOCK); i++)
            t4[threadIdx.x + i * HW_BLOCK] = stg[i];
        __syncthreads();
    }
    // A lone wave fetches its instruction stream in 32-byte windows, and an 8-byte instruction
    // that straddles two of them costs extra: the loop's speed moves by +-1 % with its offset
    // in that grid (measured in round 2, offsets 0..7: 25.99 .. 26.57 us).  The loop
    // head is therefore pinned to the grid and shifted by the best of the 8 offsets.
#define ROW_PAD4 1
#define ROW_PAD_STR(n) ROW_PAD_STR2(n)
#define ROW_PAD_STR2(n) ".p2align 5\n .rept " #n "\n s_nop 0\n .endr"
    asm volatile(ROW_PAD_STR(ROW_PAD4));
    for (uint32_t t4 = 0; t4 < (uint32_t)IAGO_MAX_TURNS; t4 += 4) {
        float u4[4];
        if (DIAG && P.uniforms) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                u4[i] = live ? P.uniforms[(int64_t)(t4 + i) * P.n + b] : 0.0f;
        } else {
            if (__builtin_expect((t4 & 63u) == 0u && t4 != 0u, 0)) { // the next 16 counter blocks = 64 turns
                rw[0] = P.id_base + (uint32_t)b;
                rw[1] = (t4 >> 2) + L.l;
                rw[2] = stream_id;
                rw[3] = 0u;
                draw();
            }
            const int src = (int)(((lane & 48u) + ((t4 >> 2) & 15u)) << 2);
#pragma unroll
            for (int i = 0; i < 4; i++)
                u4[i] = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rw[i]));
        }
        bool any_live = true;
        auto turn = [&](const int i) {
            const uint32_t t = t4 + (uint32_t)i;
            // ---- policy factors of this lane's four cells: table reads first
            f4 fac[6];
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const uint64_t brd = pl ? own : opp; // plane 0 = opponent, plane 1 = side to move
                const uint64_t y = brd << 12;
                const uint32_t w4 = __builtin_amdgcn_alignbit((uint32_t)(y >> 32), (uint32_t)y, L.sh_w);
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const uint32_t idx =
                        __builtin_amdgcn_bitop3_b32(w4 >> (8 * ky), m1f0, L.tbase, TT_AB_OR_C);
                    fac[ky * 2 + pl] = *(const f4 *)(tb + idx + (ky * 2 + pl) * 1024);
                }
            }
            // ---- legal moves of the side to move (this lane's orientation)
            const uint64_t legal = legal_hw(own, opp, L);
            uint32_t has; // 1 iff the side to move has a move
            asm("v_min_u32 %0, 1, %1" : "=v"(has) : "v"((uint32_t)legal | (uint32_t)(legal >> 32)));
            // ---- softmax numerators, zero on illegal cells (all four are in the low word)
            // ONE wait for the six table reads (issued a move generation ago) instead of four
            // counted ones: every instruction of a lone wave, waits included, is an issue slot
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
            f4 e = L.bias * fac[0];
#pragma unroll
            for (int j = 1; j < 6; j++)
                e *= fac[j];
            const uint32_t lw = (uint32_t)legal;
            float ev[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                ev[j] = __uint_as_float(__float_as_uint(e[j]) &
                                        (uint32_t)__builtin_amdgcn_sbfe((int)lw, L.bit[j], 1u));
            // ---- inverse CDF in cell order: running sums in the lane, then ONE dependent DPP
            // chain over the row: inclusive scan of the lane totals (4 steps); the exclusive
            // prefix is its row_shr:1 (bit for bit the exclusive scan of the shifted totals) and
            // the total its last lane, broadcast (row_newbcast:15) -- the last element of the
            // CDF itself, what numpy normalises by (mcts_self_play.py:103-106), instead of a
            // second 4-step butterfly.  A dependent DPP step costs a lone wave 5 + 8 cycles
            // (read-after-write hazard: two wait states); the pass / termination bookkeeping
            // (mcts_self_play.py:26-28,126-133; branch-free, independent of the sampling) is
            // placed into those wait states, which also puts the compare behind the loop's
            // exit test far ahead of its branch.
            float cdf[4], thr;
            uint32_t live_turn, play;
            {
#pragma clang fp contract(off)
                const float c1 = ev[0] + ev[1], c2 = c1 + ev[2], c3 = c2 + ev[3];
                float inc = c3;
                inc += dpp_or_zero<DPP_ROW_SHR1>(inc);
                live_turn = done ^ 1u;
                play = has & live_turn;
                __builtin_amdgcn_sched_barrier(0);
                inc += dpp_or_zero<DPP_ROW_SHR2>(inc);
                const uint32_t passing = (has ^ 1u) & live_turn;
                const uint32_t grown = stones + play;
                __builtin_amdgcn_sched_barrier(0);
                inc += dpp_or_zero<DPP_ROW_SHR4>(inc);
                stones = max(grown, (passing & pass_flg) << 6);
                __builtin_amdgcn_sched_barrier(0);
                inc += dpp_or_zero<DPP_ROW_SHR8>(inc);
                pass_flg = (pass_flg & done) | passing;
                nt += live_turn;
                __builtin_amdgcn_sched_barrier(0);
                const float xs = dpp_or_zero<DPP_ROW_SHR1>(inc);
                const float total = dpp_f32<DPP_ROW_BCAST15>(inc);
                if (i & 1) { // `while stone_num < 64` once per pair of turns
                    done |= stones >> 6;
                    any_live = __builtin_amdgcn_ballot_w64(done == 0u) != 0ull;
                }
                cdf[0] = xs + ev[0];
                cdf[1] = xs + c1;
                cdf[2] = xs + c2;
                cdf[3] = xs + c3;
                // u * total is rounded BEFORE any comparison: the uniform policy stays bit-exact
                thr = u4[i] * total;
            }
            // sampled cell = number of cells with CDF <= thr (numpy searchsorted 'right'): a
            // compare + add-with-carry per cell (4-byte encodings: 4 cycles each for a lone wave)
            uint32_t k4; // (one asm statement: the hazard recogniser pads every statement with an s_nop)
            asm("v_cmp_le_f32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, 0, 0, vcc\n\t"
                "v_cmp_le_f32 vcc, %2, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                "v_cmp_le_f32 vcc, %3, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                "v_cmp_le_f32 vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                : "=&v"(k4)
                : "v"(cdf[0]), "v"(cdf[1]), "v"(cdf[2]), "v"(cdf[3]), "v"(thr)
                : "vcc");
            const uint32_t cnt = row_sum(k4);
            uint32_t action = cnt & 63u;
            // Rounding can leave the count one cell off a legal one (or at 64): the next legal
            // cell, else the last one.  Rare, and a branch on a vector compare stalls a lone wave
            // for ~24 cycles when it follows the compare directly: the test is taken here, the
            // flips are resolved for the unfixed cell, and the branch -- far behind its compare
            // by then -- repeats them for a fixed one.
            const uint32_t lbit = (uint32_t)(legal >> ((cnt ^ L.rev63) & 63u)) & 1u;
            const uint32_t badw = has & ((cnt >> 6) | (lbit ^ 1u));
            const bool any_bad = __builtin_amdgcn_ballot_w64(badw != 0u) != 0ull;
            // ---- flips and board update in this lane's orientation (branch-free)
            uint32_t pos_l = action ^ L.rev63;
            uint64_t f = flips_hw(own, opp, pos_l, L);
            // (the hot path's flips are complete here: without this the compiler sinks the tail of
            // the reduction below the branch to share it with the cold block, un-fusing its DPPs)
            asm("" : "+v"(f));
            if (__builtin_expect(any_bad, 0)) {
                const uint64_t lt = rev ? rev64(legal) : legal; // TRUE orientation
                const uint64_t rem = (cnt < 64u) ? (lt & (~0ull << cnt)) : 0ull;
                const uint32_t fix = rem ? (uint32_t)__builtin_ctzll(rem)
                                         : (63u - (uint32_t)__builtin_clzll(lt | 1ull));
                uint32_t badc = badw; // (opaque: the per-lane select belongs to this block only)
                asm("" : "+v"(badc));
                action = badc ? fix : action;
                pos_l = action ^ L.rev63;
                f = flips_hw(own, opp, pos_l, L);
            }
            const uint32_t pm = 0u - play;
            const uint64_t fm = f & (((uint64_t)pm << 32) | pm);
            const uint64_t bit = (uint64_t)play << pos_l;
            const uint64_t nown = own | fm | bit;
            const uint64_t nopp = opp & ~fm;
            if (DIAG && P.trace && live_turn && L.l == 0u)
                P.trace[(int64_t)t * P.n + b] = play ? (uint8_t)action : (uint8_t)IAGO_TRACE_PASS;
            own = nopp; // the other side moves next (finished boards swap an even number of times)
            opp = nown;
        };
        turn(0);
        turn(1);
        if (!any_live)
            break;
        turn(2);
        turn(3);
        if (!any_live)
            break;
    }

    if (live && L.l == 0u) {
        const int d = __popcll(own) - __popcll(opp);
        if (INDEXED && hand)
            hand->z[(int)b - hand->base] = (int8_t)((d > 0) - (d < 0));
        P.z[b] = (int8_t)((d > 0) - (d < 0));
        if (P.final_own)
            P.final_own[b] = own;
        if (P.final_opp)
            P.final_opp[b] = opp;
        if (P.n_turns)
            P.n_turns[b] = (uint8_t)nt;
    }
}


inline HwParams hw_params_of(const iago_rollout_args *a)
{
    HwParams P;
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
    P.mask = nullptr;
    P.stream_ids = nullptr;
    return P;
}

} // namespace iago_row
