// othello_dev.hpp -- gfx950 device primitives for 8x8 Othello bitboards.
//
// Work decomposition ("group of 8"): one board is owned by 8 consecutive lanes
// of a 64-wide wavefront (8 boards per wave).  Every lane keeps the full board
// (own, opp: one u64 per colour, bit a = row*8+col) in registers.
//   * move generation and flips: lane d of the group works on ONE of the 8
//     ray directions.  Lanes 0-3 take the directions that move towards higher
//     bit indices (E, SW, S, SE = left shifts by 1, 7, 8, 9); lanes 4-7 take
//     the opposite directions by working on the bit-reversed board (v_bfrev),
//     where they become the same four left shifts.  All 8 lanes therefore
//     execute one instruction stream; the 8 partial results are OR-combined
//     with three DPP steps (quad_perm xor 1, xor 2, row_half_mirror).
//   * the rollout policy: lane r evaluates the 8 cells of board row r.
// Rules follow the reference exactly (game.py:180-235, rl_env.py:88-138); the
// bit-exact parity tests are tests/test_rules_gpu.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace iago {

constexpr uint64_t FILE_A = 0x0101010101010101ull;
constexpr uint64_t FILE_H = 0x8080808080808080ull;

// DPP controls (cdna4 ISA: DPP_QUAD_PERM, DPP_ROW_HALF_MIRROR)
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each group of 8

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, dpp_u32<CTRL>(__builtin_bit_cast(uint32_t, v)));
}

// OR over the 8 lanes of a group; every lane receives the result.
__device__ __forceinline__ uint64_t group8_or(uint64_t x)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo |= dpp_u32<DPP_XOR1>(lo);
    hi |= dpp_u32<DPP_XOR1>(hi);
    lo |= dpp_u32<DPP_XOR2>(lo);
    hi |= dpp_u32<DPP_XOR2>(hi);
    lo |= dpp_u32<DPP_HALF_MIRROR>(lo);
    hi |= dpp_u32<DPP_HALF_MIRROR>(hi);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float group8_max(float x)
{
    x = fmaxf(x, dpp_f32<DPP_XOR1>(x));
    x = fmaxf(x, dpp_f32<DPP_XOR2>(x));
    x = fmaxf(x, dpp_f32<DPP_HALF_MIRROR>(x));
    return x;
}
__device__ __forceinline__ uint32_t group8_add(uint32_t x)
{
    x += dpp_u32<DPP_XOR1>(x);
    x += dpp_u32<DPP_XOR2>(x);
    x += dpp_u32<DPP_HALF_MIRROR>(x);
    return x;
}
// Sum over the group and the exclusive prefix (sum of the lower lanes).
__device__ __forceinline__ void group8_scan(float x, uint32_t l8, float &excl, float &total)
{
    float blk = x, pre = 0.0f;
    float o = dpp_f32<DPP_XOR1>(blk);
    pre = (l8 & 1) ? pre + o : pre;
    blk += o;
    o = dpp_f32<DPP_XOR2>(blk);
    pre = (l8 & 2) ? pre + o : pre;
    blk += o;
    o = dpp_f32<DPP_HALF_MIRROR>(blk);
    pre = (l8 & 4) ? pre + o : pre;
    blk += o;
    excl = pre;
    total = blk;
}

__device__ __forceinline__ uint64_t rev64(uint64_t x) { return __builtin_bitreverse64(x); }

// Per-lane constants of the group-of-8 decomposition.
struct Lane8 {
    uint32_t l8;    // lane index inside the group, 0..7
    uint32_t s;     // left-shift of this lane's direction: 1, 7, 8, 9
    uint64_t mask;  // destination mask that kills A/H-file wrap-around
    uint64_t base;  // ray of this direction from cell 0, without cell 0 (see ray_mask)
    uint32_t ca, cx; // column-mask recipe of ray_mask
    bool rev;       // lanes 4..7 work on the bit-reversed board
};

__device__ __forceinline__ Lane8 make_lane8(uint32_t tid)
{
    Lane8 L;
    L.l8 = tid & 7u;
    const uint32_t k = L.l8 & 3u; // 0: E (+1), 1: SW (+7), 2: S (+8), 3: SE (+9)
    L.s = (k == 0) ? 1u : (6u + k);
    L.mask = (k == 1) ? ~FILE_H : ((k == 2) ? ~0ull : ~FILE_A);
    L.base = (k == 0)   ? 0x00000000000000FEull
             : (k == 1) ? 0x0002040810204080ull
             : (k == 2) ? 0x0101010101010100ull
                        : 0x8040201008040200ull;
    // columns a ray cell may have: E, SE: > col(pos); SW: < col(pos); S: any
    L.ca = (k == 1) ? 0xFFu : ((k == 2) ? 0u : 0xFEu);
    L.cx = (k == 0 || k == 3) ? 0u : 0xFFu;
    L.rev = L.l8 >= 4u;
    return L;
}

// Cells strictly beyond `pl` (a cell index in the LANE's orientation) along the
// lane's direction, up to the board edge: the direction's ray from cell 0
// shifted to pl, with the cells that wrapped around the A/H files removed by a
// column mask ((ca << col) ^ cx, replicated to the 8 rows with v_perm_b32).
__device__ __forceinline__ uint64_t ray_mask(const Lane8 &L, uint32_t pl)
{
    const uint32_t m8 = (L.ca << (pl & 7u)) ^ L.cx;
    const uint32_t m32 = __builtin_amdgcn_perm(m8, m8, 0u); // byte 0 of m8 in all 4 bytes
    return (L.base << pl) & (((uint64_t)m32 << 32) | m32);
}

// Board in the lane's own orientation.
__device__ __forceinline__ uint64_t to_lane(uint64_t x, const Lane8 &L)
{
    uint64_t r = rev64(x);
    return L.rev ? r : x;
}

// Legal moves of `own` (reference: game.py:210-235).  o, p are own/opp in the
// lane orientation.  Each lane floods its direction through contiguous
// opponent stones (<= 6 of them fit on a ray) and lands on an empty cell.
// Written on 32-bit halves so that every flood step is v_lshlrev_b32 +
// v_alignbit_b32 + 2 x v_and_or_b32 (the shift is 1..9, never >= 32).
__device__ __forceinline__ uint64_t group8_legal(uint64_t o, uint64_t p, const Lane8 &L)
{
    const uint32_t s = L.s, rs = 32u - L.s;
    const uint32_t ml = (uint32_t)L.mask, mh = (uint32_t)(L.mask >> 32);
    const uint32_t ol = (uint32_t)o, oh = (uint32_t)(o >> 32);
    const uint32_t pl = (uint32_t)p, ph = (uint32_t)(p >> 32);
    const uint32_t pml = pl & ml, pmh = ph & mh;
    uint32_t tl = (ol << s) & pml;
    uint32_t th = __builtin_amdgcn_alignbit(oh, ol, rs) & pmh;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint32_t nl = tl << s, nh = __builtin_amdgcn_alignbit(th, tl, rs);
        tl = (nl & pml) | tl;
        th = (nh & pmh) | th;
    }
    const uint32_t el = ~(ol | pl) & ml, eh = ~(oh | ph) & mh; // empty cells, wrap-masked
    const uint32_t vl = (tl << s) & el, vh = __builtin_amdgcn_alignbit(th, tl, rs) & eh;
    return group8_or(to_lane(((uint64_t)vh << 32) | vl, L));
}

// Stones flipped by `own` playing at `pos` (reference: game.py:180-207, no
// legality check).  Each lane resolves its ray with one carry propagation:
// adding 1 to (opp | ~ray) ripples through the contiguous opponent stones next
// to `pos` and stops on the first ray cell that is not an opponent stone; the
// run is flipped iff that cell holds an own stone.
__device__ __forceinline__ uint64_t group8_flips(uint64_t o, uint64_t p, uint32_t pos,
                                                 const Lane8 &L)
{
    const uint64_t M = ray_mask(L, L.rev ? 63u - pos : pos);
    const uint64_t x = p | ~M;
    const uint64_t t = x + 1ull;
    const uint64_t cand = (t ^ x) & M & p;
    const uint64_t f = ((t & M & o) != 0ull) ? cand : 0ull;
    return group8_or(to_lane(f, L));
}

// ---------------------------------------------------------------- Philox
// Philox4x32-10 (Salmon et al., SC'11); bit-identical to oracle/othello_oracle.c.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t lo0 = 0xD2511F53u * c[0], hi0 = __umulhi(0xD2511F53u, c[0]);
        const uint32_t lo1 = 0xCD9E8D57u * c[2], hi1 = __umulhi(0xCD9E8D57u, c[2]);
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0;
        c[1] = lo1;
        c[2] = n2;
        c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

} // namespace iago
