// rollout_lpb_kernel.hip -- the leaf rollout with ONE LANE PER BOARD.
//
// Same contract and results as rollout_kernel.hip (iago_rollout, replacing
// Simulate(state)(color), mcts_self_play.py:9-134); chosen by the launcher when
// the caller keeps the chip full (throughput_hint) or the batch alone does.
//
// Why a second kernel.  The 8-lanes-per-board kernel minimises the latency of a
// small batch (8x the waves, direction- and row-parallel work) but pays for it
// with cross-lane traffic (bit-reversed orientations, DPP reductions), with
// logits for all 64 cells through 1.5 KB of LDS gathers per board-turn (the LDS
// pipe saturates at 12 waves/CU), and ~27 wave-instructions per board-turn.
// Here a lane owns its board:
//   * legal moves: 8 constant-shift floods on the lane's two u64;
//   * the rollout policy is evaluated ONLY at the legal cells (8.5 on average):
//     the softmax numerator of a cell is bias[c] * CT[0][opp 3x3 pattern] *
//     CT[1][own 3x3 pattern] -- two 4-byte LDS reads from 512-entry tables;
//   * the running sums of the legal cells stay in registers (one slot per legal
//     move, the loop leaves when no lane of the wave has a move left), the
//     sampled slot is the number of sums <= u * total, its cell comes from a
//     packed byte array;
//   * flips: the 4 rays towards higher bits by carry propagation on the board,
//     the 4 others on the bit-reversed board;
//   * Philox: the lane draws the 4 uniforms of turns 4k..4k+3 itself.
// ~17 wave-instructions per board-turn and almost no LDS traffic.
#include "abi_common.hpp"
#include "othello_dev.hpp"
#include "rollout_blob.hpp"

using namespace iago;

namespace {

constexpr int BLOCK = 256; // (64- / 128- / 512-thread blocks and 5-6 waves per SIMD measured equal or slower: LABNOTES.md)
constexpr int SLOTS = 34; // an Othello position has at most 33 legal moves

struct LpbParams {
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

// Bitwise function TT of three 64-bit words, one v_bitop3_b32 per half (truth table
// with a = 0xF0, b = 0xCC, c = 0xAA).  Spelled out because it is full rate on gfx950 and
// the compiler otherwise leaves `a | (b & c)` as two instructions per half.
template <int TT>
__device__ __forceinline__ uint64_t bitop64(uint64_t a, uint64_t b, uint64_t c)
{
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32),
                                                    (uint32_t)(c >> 32), TT);
    return ((uint64_t)hi << 32) | lo;
}
constexpr int TT_A_OR_BC = 0xF8;    // a | (b & c)
constexpr int TT_ABC = 0x80;        // a & b & c
constexpr int TT_A_OR_NBC = 0xF7;   // a | ~(b & c)
constexpr int TT_A_NB_C = 0x20;     // a & ~b & c
constexpr int TT_AB_NC = 0x40;      // a & b & ~c
constexpr int TT_A_OR_B_OR_C = 0xFE; // a | b | c

// Shift amounts the compiler cannot see through: a 64-bit shift by a VGPR amount is ONE
// v_lshlrev_b64 / v_lshrrev_b64 (the issue cost of a single 32-bit left shift on
// gfx950), while a constant amount gets split into v_alignbit_b32 + a 32-bit shift.
struct ShiftAmounts {
    uint32_t s1, s7, s8, s9, s14, s16, s18;
};
__device__ __forceinline__ ShiftAmounts opaque_shift_amounts()
{
    ShiftAmounts A;
    asm("v_mov_b32 %0, 1" : "=v"(A.s1));
    asm("v_mov_b32 %0, 7" : "=v"(A.s7));
    asm("v_mov_b32 %0, 8" : "=v"(A.s8));
    asm("v_mov_b32 %0, 9" : "=v"(A.s9));
    asm("v_mov_b32 %0, 14" : "=v"(A.s14));
    asm("v_mov_b32 %0, 16" : "=v"(A.s16));
    asm("v_mov_b32 %0, 18" : "=v"(A.s18));
    return A;
}

// Parallel-prefix flood through up to 6 opponent stones (1, 2, 4, 6): the same set as six single
// steps, in four dependent shift + bit-op groups.  sh2 = 2 * sh.
__device__ __forceinline__ uint64_t moves_up(uint64_t own, uint64_t opp, uint64_t empty, uint64_t m,
                                             uint32_t sh, uint32_t sh2)
{
    const uint64_t pm = opp & m;
    uint64_t t = (own << sh) & pm;
    t = bitop64<TT_A_OR_BC>(t, t << sh, pm);
    const uint64_t pm2 = pm & (pm << sh);
    t = bitop64<TT_A_OR_BC>(t, t << sh2, pm2);
    t = bitop64<TT_A_OR_BC>(t, t << sh2, pm2);
    return (t << sh) & (empty & m);
}
__device__ __forceinline__ uint64_t moves_down(uint64_t own, uint64_t opp, uint64_t empty,
                                               uint64_t m, uint32_t sh, uint32_t sh2)
{
    const uint64_t pm = opp & m;
    uint64_t t = (own >> sh) & pm;
    t = bitop64<TT_A_OR_BC>(t, t >> sh, pm);
    const uint64_t pm2 = pm & (pm >> sh);
    t = bitop64<TT_A_OR_BC>(t, t >> sh2, pm2);
    t = bitop64<TT_A_OR_BC>(t, t >> sh2, pm2);
    return (t >> sh) & (empty & m);
}

// Moves towards the east (+1) for all 8 rows at once by carry propagation: adding the
// opponent stones that have an own stone on their west to the inner-column opponent
// mask ripples through each such run and sets the cell just past it.
__device__ __forceinline__ uint64_t moves_east(uint64_t own, uint64_t opp, uint64_t empty,
                                               uint32_t s1)
{
    const uint64_t inner = opp & 0x7E7E7E7E7E7E7E7Eull; // a run never wraps: no col 0 / 7
    const uint64_t start = (own << s1) & inner;
    return bitop64<TT_AB_NC>(start + inner, empty, inner);
}

// game.py:210-235 on one lane; the masks are the DESTINATION files a shifted stone may
// not land on (A/H-file wrap-around).  ro / rp: the bit-reversed boards (west = east
// of the reversed board).
__device__ __forceinline__ uint64_t legal_moves_1(uint64_t own, uint64_t opp, uint64_t ro,
                                                  uint64_t rp, const ShiftAmounts &A)
{
    const uint64_t e = ~(own | opp);
    const uint64_t a = bitop64<TT_A_OR_B_OR_C>(moves_east(own, opp, e, A.s1),
                                               rev64(moves_east(ro, rp, rev64(e), A.s1)),
                                               moves_up(own, opp, e, ~FILE_H, A.s7, A.s14));
    const uint64_t b = bitop64<TT_A_OR_B_OR_C>(moves_up(own, opp, e, ~0ull, A.s8, A.s16),
                                               moves_up(own, opp, e, ~FILE_A, A.s9, A.s18),
                                               moves_down(own, opp, e, ~FILE_A, A.s7, A.s14));
    return bitop64<TT_A_OR_B_OR_C>(a, b, moves_down(own, opp, e, ~0ull, A.s8, A.s16) |
                                             moves_down(own, opp, e, ~FILE_H, A.s9, A.s18));
}

// Adds the flips along the ray of direction K (0: +1, 1: +7, 2: +8, 3: +9) from `pos`
// to `acc`: othello_dev.hpp's ray_mask / carry trick with compile-time direction
// constants.  gt / lt: the columns right / left of pos, replicated to all rows.
template <int K>
__device__ __forceinline__ uint64_t flips_up(uint64_t acc, uint64_t o, uint64_t p, uint32_t pos,
                                             uint64_t gt, uint64_t lt)
{
    constexpr uint64_t base = (K == 0)   ? 0x00000000000000FEull
                              : (K == 1) ? 0x0002040810204080ull
                              : (K == 2) ? 0x0101010101010100ull
                                         : 0x8040201008040200ull;
    const uint64_t M0 = base << pos;
    const uint64_t side = (K == 1) ? lt : gt;
    const uint64_t M = (K == 2) ? M0 : (M0 & side);
    // x: every bit that lets a carry pass = opponent stones on the ray, everything off it
    const uint64_t x = (K == 2) ? (p | ~M0) : bitop64<TT_A_OR_NBC>(p, M0, side);
    const uint64_t t = x + 1ull;                  // the carry stops at the first ray cell not in p
    const uint64_t of = bitop64<TT_ABC>(t, M, o); // that cell, if it is an own stone
    const uint64_t run = bitop64<TT_A_NB_C>(x, t, M); // the opponent run the carry went through
    // of is one bit or zero: all ones iff there is a bracketing own stone
    const uint32_t z = (uint32_t)of | (uint32_t)(of >> 32);
    const uint32_t ok = (uint32_t)((int32_t)(0u - z) >> 31);
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)acc, (uint32_t)run, ok, TT_A_OR_BC);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(acc >> 32), (uint32_t)(run >> 32),
                                                    ok, TT_A_OR_BC);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t flips_4(uint64_t o, uint64_t p, uint32_t pos)
{
    const uint32_t c = pos & 7u;
    const uint32_t g8 = 0xFEu << c, l8 = (0xFFu << c) ^ 0xFFu;
    const uint32_t g32 = __builtin_amdgcn_perm(g8, g8, 0u), l32 = __builtin_amdgcn_perm(l8, l8, 0u);
    const uint64_t gt = ((uint64_t)g32 << 32) | g32, lt = ((uint64_t)l32 << 32) | l32;
    uint64_t f = flips_up<0>(0ull, o, p, pos, gt, lt);
    f = flips_up<1>(f, o, p, pos, gt, lt);
    f = flips_up<2>(f, o, p, pos, gt, lt);
    return flips_up<3>(f, o, p, pos, gt, lt);
}

// game.py:180-207 on one lane (no legality check)
__device__ __forceinline__ uint64_t flips_1(uint64_t own, uint64_t opp, uint64_t ro, uint64_t rp,
                                            uint32_t pos)
{
    return flips_4(own, opp, pos) | rev64(flips_4(ro, rp, 63u - pos));
}

// A board with a guard column between the rows and an empty row in front: cell (k, j)
// at bit 10 + 9k + j of a 96-bit string.  The 3x3 neighbourhood of cell c = 8r + x is
// then the 21-bit window at bit 9r + x with no wrap-around to mask (the guard bit is
// column -1 of one row and column 8 of the previous one).
struct Padded {
    uint64_t w01; // bits 0..63
    uint64_t w12; // bits 32..95
};

__device__ __forceinline__ uint64_t spread4(uint32_t h) // rows at bits 0, 9, 18, 27
{
    uint64_t y = (uint64_t)(h & 0xFFFFu) | ((uint64_t)(h & 0xFFFF0000u) << 2);
    return (y & 0x0003FC00FFull) | ((y & 0x03FC00FF00ull) << 1);
}

__device__ __forceinline__ Padded pad_board(uint64_t b)
{
    const uint64_t L = spread4((uint32_t)b) << 10;  // rows 0..3: bits 10..45
    const uint64_t H = spread4((uint32_t)(b >> 32)); // rows 4..7: bits 46..81 of the string
    Padded P;
    P.w01 = L | (H << 46);
    P.w12 = (L >> 32) | (H << 14);
    return P;
}

// byte offset into a 512-entry float table of the 9-bit pattern (bit 3*ky + kx = cell
// (r + ky - 1, x + kx - 1), zero outside the board) of the window at bit s = 9r + x
__device__ __forceinline__ uint32_t pattern9_off(const Padded &T, uint32_t s, uint32_t s32)
{
    const uint32_t q = ((s < 32u) ? (uint32_t)(T.w01 >> s) : (uint32_t)(T.w12 >> s32)) & 0x1C0E07u;
    // rows at bits 0, 9, 18 -> 12, 15, 18: the three partial products do not overlap
    return (__umul24(q, 0x1041u) >> 10) & 0x7FCu;
}

// index of the lowest set bit; 31 for x = 0 (v_ffbl_b32 returns -1 then)
__device__ __forceinline__ uint32_t lowest_bit(uint64_t x)
{
    uint32_t a, b;
    asm("v_ffbl_b32 %0, %1" : "=v"(a) : "v"((uint32_t)x));
    asm("v_ffbl_b32 %0, %1" : "=v"(b) : "v"((uint32_t)(x >> 32)));
    return min(a, b + 32u);
}

// Per-turn sampling state.  One template instance per slot (recursion instead of a loop
// with early exits): the running sum of a slot lives in a register of its own frame.
struct Slots {
    uint32_t cells[(SLOTS + 3) / 4]; // 4 cell indices per word
    uint64_t rem;                    // legal cells not yet visited
    float acc;                       // running sum
    uint32_t thr;                    // bits of u * total
    uint32_t above_lo, above_hi;     // one bit per visited slot: running sum > u * total
    int filled;                      // wave-uniform number of slots visited
};

// mcts_self_play.py:103-106: the threshold of the inverse CDF, rounded before any
// comparison so that the uniform policy stays bit-exact with numpy
__device__ __forceinline__ void close_slots(Slots &S, float u, int filled)
{
#pragma clang fp contract(off)
    const float thr = u * S.acc;
    S.thr = __float_as_uint(thr);
    S.above_lo = 0u;
    S.above_hi = 0u;
    S.filled = filled;
}

// Way down: softmax numerators of the legal cells in cell order, until no lane of the
// wave has a cell left.  Way back up: compare each slot's running sum with u * total.
template <int J>
__device__ __forceinline__ void fill_slots(Slots &S, const Padded &To, const Padded &Tp,
                                           const char *ct, const float *be, float u)
{
    if constexpr (J < SLOTS) {
        if (__builtin_amdgcn_ballot_w64(S.rem != 0ull) == 0ull) {
            close_slots(S, u, J);
            return;
        }
        const bool valid = S.rem != 0ull;
        const uint32_t c = lowest_bit(S.rem); // a lane without one reads cell 31's tables
        S.rem &= S.rem - 1ull;
        const uint32_t s = c + (c >> 3), s32 = s - 32u;
        const uint32_t io = pattern9_off(To, s, s32);
        const uint32_t ip = pattern9_off(Tp, s, s32);
        // plane 0 = opponent of the side to move, plane 1 = side to move (game.py:168-174)
        const float e = be[c] * *(const float *)(ct + ip) * *(const float *)(ct + 2048 + io);
        S.acc += valid ? e : 0.0f;
        const uint32_t sum = __float_as_uint(S.acc);
        S.cells[J >> 2] |= c << (8 * (J & 3));
        fill_slots<J + 1>(S, To, Tp, ct, be, u);
        // both floats are >= 0: their bit patterns order like the values, and bit 31 of
        // the difference says sum > thr; shifted into the slot bitmap (newest = bit 0)
        const uint32_t d = S.thr - sum;
        if (J < 32)
            S.above_lo = __builtin_amdgcn_alignbit(S.above_lo, d, 31);
        else
            S.above_hi = __builtin_amdgcn_alignbit(S.above_hi, d, 31);
    } else {
        close_slots(S, u, SLOTS);
    }
}

// (Round 2 measured a per-turn counting sort of a block's boards by their number of legal moves,
// so that a wave holds boards of similar mobility: bit-identical games, 751 vs 813 M games/s --
// rejected, LABNOTES.md; the variant is in the history at commit b61d6ed.)
__global__ __launch_bounds__(BLOCK) void rollout_lpb_kernel(LpbParams P)
{
    __shared__ float ct[N_CT]; // [plane][512]
    __shared__ float be[64];   // exp'ed biases
    for (uint32_t i = threadIdx.x; i < (uint32_t)N_CT; i += BLOCK)
        ct[i] = P.blob[OFF_CT + i];
    if (threadIdx.x < 64) {
        be[threadIdx.x] = P.blob[OFF_BIAS + threadIdx.x];
    }
    __syncthreads();

    const int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    bool live = b < P.n;
    uint64_t own = live ? P.own[b] : 0ull; // side to move
    uint64_t opp = live ? P.opp[b] : 0ull;
    uint32_t stones = (uint32_t)__popcll(own | opp);
    uint32_t pass_flg = 0u, nt = 0u;
    uint32_t done = (!live || stones >= 64u) ? 1u : 0u; // `while stone_num < 64`
    const uint32_t stream_id = P.stream_id + (P.stream_id_dev ? *P.stream_id_dev : 0u);
    uint32_t rw[4] = {0, 0, 0, 0};
    const ShiftAmounts SA = opaque_shift_amounts();

    Padded To = pad_board(own), Tp = pad_board(opp);
    for (uint32_t t = 0; t < (uint32_t)IAGO_MAX_TURNS; t++) {
        uint64_t ro = rev64(own), rp = rev64(opp);
        uint64_t legal = legal_moves_1(own, opp, ro, rp, SA);

        // ---- uniform of this turn: word t&3 of Philox counter (rid, t>>2, stream, 0)
        float u;
        if (P.uniforms) {
            u = live ? P.uniforms[(int64_t)t * P.n + b] : 0.0f;
        } else {
            if ((t & 3u) == 0u) {
                rw[0] = P.id_base + (uint32_t)b;
                rw[1] = t >> 2;
                rw[2] = stream_id;
                rw[3] = 0u;
                philox4x32_10(rw, P.key0, P.key1);
            }
            const uint32_t k = t & 3u; // wave-uniform
            const uint32_t w = (k == 0u) ? rw[0] : (k == 1u) ? rw[1] : (k == 2u) ? rw[2] : rw[3];
            u = (float)(w >> 8) * (1.0f / 16777216.0f);
        }

        const uint32_t has = min(1u, (uint32_t)legal | (uint32_t)(legal >> 32));

        // ---- softmax numerators of the legal cells, running sums in slot order
        Slots S;
#pragma unroll
        for (int i = 0; i < (SLOTS + 3) / 4; i++)
            S.cells[i] = 0u;
        S.rem = done ? 0ull : legal;
        S.acc = 0.0f;
        fill_slots<0>(S, To, Tp, (const char *)ct, be, u);
        // ---- inverse CDF (mcts_self_play.py:103-106): slot = #sums <= u * total
        uint32_t slot = (uint32_t)S.filled - (uint32_t)__popc(S.above_lo) - (uint32_t)__popc(S.above_hi);
        const uint32_t nlegal = (uint32_t)__popcll(legal);
        slot = min(slot, nlegal ? nlegal - 1u : 0u); // rounding past the total: last legal cell
        uint32_t word = S.cells[0];
#pragma unroll
        for (int i = 1; i < (SLOTS + 3) / 4; i++)
            word = ((slot >> 2) == (uint32_t)i) ? S.cells[i] : word;
        const uint32_t action = (word >> (8u * (slot & 3u))) & 63u;

        // ---- flips, board update, pass / termination bookkeeping (branch-free)
        const uint64_t f = flips_1(own, opp, ro, rp, action);
        const uint32_t live_turn = done ^ 1u;
        const uint32_t play = has & live_turn;
        const uint32_t passing = (has ^ 1u) & live_turn;
        const uint32_t pm = 0u - play;
        const uint64_t fm = f & (((uint64_t)pm << 32) | pm);
        const uint64_t bit = (uint64_t)play << action;
        const uint64_t nown = own | fm | bit;
        const uint64_t nopp = opp & ~fm;
        stones = max(stones + play, (passing & pass_flg) << 6); // mcts_self_play.py:126-133
        pass_flg = (pass_flg & done) | passing;
        if (P.trace && live_turn)
            P.trace[(int64_t)t * P.n + b] = play ? (uint8_t)action : (uint8_t)IAGO_TRACE_PASS;
        own = nopp; // the other side moves next (finished boards swap an even number of times)
        opp = nown;
        // the guard-column strings follow incrementally: only the changed cells are spread
        {
            const Padded D = pad_board(fm | bit);
            const Padded oldTo = To;
            To.w01 = Tp.w01 & ~D.w01; // new mover = old opponent minus the flipped stones
            To.w12 = Tp.w12 & ~D.w12;
            Tp.w01 = oldTo.w01 | D.w01;
            Tp.w12 = oldTo.w12 | D.w12;
        }
        nt += live_turn;
        if (t & 1u) { // `while stone_num < 64` once per pair of turns
            done |= stones >> 6;
            if (__builtin_amdgcn_ballot_w64(done == 0u) == 0ull)
                break;
        }
    }

    if (live) {
        const int64_t bo = b;
        const int d = __popcll(own) - __popcll(opp);
        P.z[bo] = (int8_t)((d > 0) - (d < 0));
        if (P.final_own)
            P.final_own[bo] = own;
        if (P.final_opp)
            P.final_opp[bo] = opp;
        if (P.n_turns)
            P.n_turns[bo] = (uint8_t)nt;
    }
}

} // namespace

void iago_launch_rollout_lpb(const iago_rollout_args *a, void *stream)
{
    LpbParams P;
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
    const unsigned grid = (unsigned)((a->n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(rollout_lpb_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, P);
}
