// conv_policy_body.hpp -- the one-board LDS-resident walk of the SLPolicy net (network.py:15-47) as a device
// function (policy_item) with its parameter block, shared by policy_resident_kernel (conv_policy_kernel.hip)
// and the persistent search kernel (search_kernel.hip).  The description of the arithmetic is at the head of
// conv_policy_kernel.hip.
#pragma once
#include "abi_common.hpp"

#include <hip/hip_fp16.h>

#include <cstdlib>

#include <atomic>

namespace iago_policy {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_f16, accumulator in place (conv_trunk_body.hpp: IAGO_MFMA16)
#define IAGO_POLICY_MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int RS = 800;             // bytes of a cell row: 128 ch hi | mid | lo (256 B each) | 32 B (bank skew: rows 2 x 16 B apart mod 256)
constexpr int ZB = 1280;            // zero bytes behind the 64 rows: the target of every out-of-board tap
constexpr int BS = 64 * RS + ZB;    // 52,480
constexpr int HEAD_FLOATS = 64;     // logits
constexpr int W1_LDS = (64 * 18 + 64) * 4;  // block1's weights and biases, staged per walk
constexpr int LDS_BYTES = BS + 1024 + HEAD_FLOATS * 4 + W1_LDS; // (+1024: the operand prefetch of the last k-step reads past T)
constexpr float S1 = 1.0f / 2048.0f, S2 = 1.0f / (2048.0f * 2048.0f);

struct PolicyParams {
    const uint64_t *own, *opp;  // own = side to move (plane 1), opp = plane 0 (game.py:168-174)
    const int64_t *index;       // optional gather list
    const int32_t *n_dev;       // optional device-side row count
    int64_t n;
    const float *w1, *b1;       // block1 [64][2][3][3], [64]
    const uint4 *w_hi[7], *w_mid[7], *w_lo[7]; // blocks 2..8: [cin/16][3][3][128][16] f16
    const float *bias[7];
    const float *w9, *b10;      // conv9 [128] (1x1, no bias), bias10 [64]
    float *probs;               // [n][64]
    uint32_t *overflow;
    // a launch may run only the layers [layer_lo, layer_hi) of blocks 2..8 (0..7): block1 comes
    // with layer_lo == 0, the head with layer_hi == 7, a board's LDS image (64 rows of RS bytes)
    // travels between the launches through `scratch` [n][64 * RS]
    int layer_lo, layer_hi;
    uint4 *scratch;
    // a launch covers the rows [row_lo, row_hi) of the batch; row r parks its image in scratch slot
    // r - row_lo (the host runs a long batch as chunks of `scratch_rows` rows: bounded scratch)
    int64_t row_lo, row_hi;
};

// a -> (hi, mid, lo) with a == hi + mid 2^-11 + lo 2^-22 exactly (every difference is exact)
__device__ __forceinline__ void split3(const f2 v, h2 &hi, h2 &mid, h2 &lo)
{
    hi = __builtin_convertvector(v, h2);
    const f2 r1 = (v - __builtin_convertvector(hi, f2)) * 2048.0f;
    mid = __builtin_convertvector(r1, h2);
    const f2 r2 = (r1 - __builtin_convertvector(mid, f2)) * 2048.0f;
    lo = __builtin_convertvector(r2, h2);
}

extern __shared__ __align__(16) char policy_lds[];

// pos / res / w1s: the persistent search's hand-offs through LDS (conv_trunk_body.hpp, Piece): the position (word 2 / 3 =
// own lo / hi, 4 / 5 = opp lo / hi), the distribution [64] instead of P.probs, block1's weights staged already
// head_w (LDS): conv9's weights [128] and bias10 [64] staged there already
// SRCH: the persistent search's form (all four given; a compile-time choice: conv_trunk_body.hpp, trunk_item)
template <bool SRCH = false>
__device__ __forceinline__ void policy_item(const PolicyParams &P, const int64_t row_id, const uint32_t *pos = nullptr,
                                            float *res = nullptr, const float *w1_staged = nullptr,
                                            const float *head_w = nullptr)
{
    char *const T = policy_lds;
    // (SRCH: an opaque copy of the thread id, so that the per-thread address tables are this walk's own values and not
    // hoisted out of the net workgroups' loop into spilled registers: conv_trunk_body.hpp, trunk_item)
    int tid_ = threadIdx.x;
    if constexpr (SRCH)
        asm volatile("" : "+v"(tid_));
    const int tid = tid_, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int64_t b = P.index ? P.index[row_id] : row_id;

    // ---- the zero area, then block1 (3x3, 2 -> 64, bias, ReLU; network.py:17-19) in float32
    if (tid < ZB / 16)
        *(uint4 *)(T + 64 * RS + tid * 16) = make_uint4(0, 0, 0, 0);
    bool saturated = false;
    if (P.layer_lo > 0) {
        const uint4 *src = P.scratch + (row_id - P.row_lo) * (64 * RS / 16);
        for (int e = tid; e < 64 * RS / 16; e += 256)
            *(uint4 *)(T + e * 16) = src[e];
    } else {
        const int cell = tid & 63, y = cell >> 3, x = cell & 7;
        const uint64_t bits0 = SRCH ? (((uint64_t)pos[5] << 32) | pos[4]) : P.opp[b];
        const uint64_t bits1 = SRCH ? (((uint64_t)pos[3] << 32) | pos[2]) : P.own[b];
        float in[18];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
                    const int a = (yy * 8 + xx) & 63;
                    in[c * 9 + ky * 3 + kx] = (ok && (((c ? bits1 : bits0) >> a) & 1ull)) ? 1.0f : 0.0f;
                }
        // block1's weights and biases (4.9 KB) through LDS: one round trip to L2 for the workgroup, then broadcast reads
        // (conv_trunk_body.hpp says what the channel-by-channel loads from global memory cost)
        const float *w1s = w1_staged; // [64][18] weights, [64] biases
        if constexpr (!SRCH) {
            float *const st = (float *)(T + BS + 1024 + HEAD_FLOATS * 4);
            for (int e = tid; e < 64 * 18 / 4; e += 256)
                ((float4 *)st)[e] = ((const float4 *)P.w1)[e];
            if (tid < 16)
                ((float4 *)(st + 64 * 18))[tid] = ((const float4 *)P.b1)[tid];
            __syncthreads();
            w1s = st;
        }
#pragma unroll 1
        for (int g2 = 0; g2 < 2; g2++) {
            const int grp = __builtin_amdgcn_readfirstlane(g2 * 4 + wv); // channel block * 2 + half: wave-uniform
            const int co0 = grp * 8;
            _Float16 p0[8], p1[8], p2[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                // two output channels per packed FMA (a lone wave pays per instruction, not per lane-operation): each
                // channel's chain of 18 fused multiply-adds in the same order as one by one
                const float *wa = w1s + (co0 + k) * 18, *wb = wa + 18; // [co][ci][ky][kx]
                f2 acc = (f2){w1s[64 * 18 + co0 + k], w1s[64 * 18 + co0 + k + 1]};
#pragma unroll
                for (int j = 0; j < 18; j++)
                    acc = __builtin_elementwise_fma((f2){wa[j], wb[j]}, (f2){in[j], in[j]}, acc);
                saturated |= !(acc.x <= 65000.0f) || !(acc.y <= 65000.0f);
                f2 v;
                v.x = fminf(fmaxf(acc.x, 0.0f), 65000.0f);
                v.y = fminf(fmaxf(acc.y, 0.0f), 65000.0f);
                h2 a0, a1, a2;
                split3(v, a0, a1, a2);
                p0[k] = a0.x, p0[k + 1] = a0.y;
                p1[k] = a1.x, p1[k + 1] = a1.y;
                p2[k] = a2.x, p2[k + 1] = a2.y;
            }
            char *dst = T + cell * RS + (grp >> 1) * 32 + (grp & 1) * 16;
            *(uint4 *)dst = *(const uint4 *)p0;
            *(uint4 *)(dst + 256) = *(const uint4 *)p1;
            *(uint4 *)(dst + 512) = *(const uint4 *)p2;
        }
    }
    __syncthreads();

    // ---- per-lane addresses of the B operand.  The K loop runs on v_mfma_f32_16x16x32_f16 (conv_trunk_body.hpp says
    // why): lane = (column c16 = lane & 15, k quarter kq = lane >> 4); a k-step covers 32 input channels = two
    // 16-channel chunks at one tap, a B tile is 16 cells (quarter q of the board) x 32 channels: this lane reads cell
    // 16 q + c16, 16 bytes at + 16 kq of the chunk pair, tap (ky, kx); an out-of-board tap reads zeros from the slot with
    // the bank offset its row would have had (rows are 2 x 16 B apart mod 256 B)
    const int c16 = lane & 15, kq = lane >> 4;
    uint32_t addr[4][9];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int cell = 16 * q + c16;
            const int yy = (cell >> 3) + tap / 3 - 1, xx = (cell & 7) + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
            const int lin = (cell + (tap / 3 - 1) * 8 + (tap % 3 - 1)) & 7;
            addr[q][tap] = (uint32_t)((ok ? (yy * 8 + xx) * RS : 64 * RS + 32 * lin) + kq * 16);
        }
    uint32_t wrow[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        wrow[q] = (uint32_t)((16 * q + c16) * RS);

    for (int L = P.layer_lo; L < P.layer_hi; L++) {
        const int n_pairs = L == 0 ? 2 : 4; // chunk pairs of 32 input channels
        // this lane's A operands: output channels 32 wv + c16 (M tile 0) and + 16 (M tile 1), input channels
        // 8 (kq & 1) .. + 7 of chunk 2 cp + (kq >> 1); a chunk is 9 x 128 x 32 B, a tap 128 x 32 B further
        // (a wave-uniform base -- the layer's weights + the k-step's offset: scalar registers -- + this lane's byte offset)
        const char *const wh = (const char *)P.w_hi[L], *const wm = (const char *)P.w_mid[L], *const wl = (const char *)P.w_lo[L];
        const uint32_t a_lane = (uint32_t)(((32 * wv + c16) * 2 + (kq & 1) + (kq >> 1) * (9 * 256)) * 16);
        auto a_load = [&](const char *base, uint32_t step_bytes, int m) -> u32x4 {
            return *(const u32x4 *)(base + step_bytes + a_lane + (uint32_t)m * 512u);
        };
        // (acc0: the hi x hi products, in two halves by tap parity -- two chains of 18 additions instead of one of 36: the
        // accumulator's rounding is what the net's 1e-5 bound on the move distribution is spent on)
        float4v acc0[2][4], acc0b[2][4], acc1[2][4], acc2[2][4];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    acc0[m][n][v] = 0.0f;
                    acc0b[m][n][v] = 0.0f;
                    acc1[m][n][v] = 0.0f;
                    acc2[m][n][v] = 0.0f;
                }
        u32x4 a_hi[3][2], a_mid[3][2], a_lo[3][2]; // k-steps s, s + 1, s + 2 (ring index = tap % 3) x the two M tiles
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int m = 0; m < 2; m++) {
                a_hi[i][m] = a_load(wh, (uint32_t)i * 4096u, m);
                a_mid[i][m] = a_load(wm, (uint32_t)i * 4096u, m);
                a_lo[i][m] = a_load(wl, (uint32_t)i * 4096u, m);
            }
        // B operands two tiles ahead of their MFMAs (three register sets); a tile = 12 MFMAs here.  The three reads of
        // a tile go out one per MFMA gap (an MFMA of this shape leaves 8 of its 16 cycles to other instructions), and
        // the MFMAs that add into one accumulator stand at least four apart (conv_trunk_body.hpp)
        half8 bh[3], bm[3], bl[3];
        auto b_addr = [&](int tile) -> const char * {
            // tile = tap * 4 + q of the running chunk pair; 36, 37 = the first two tiles of the next pair
            const int over = tile >= 36 ? 64 : 0, tt = tile % 36;
            return T + addr[tt & 3][tt >> 2] + over;
        };
        {
            const char *p0 = b_addr(0), *p1 = b_addr(1);
            bh[0] = *(const half8 *)p0, bm[0] = *(const half8 *)(p0 + 256), bl[0] = *(const half8 *)(p0 + 512);
            bh[1] = *(const half8 *)p1, bm[1] = *(const half8 *)(p1 + 256), bl[1] = *(const half8 *)(p1 + 512);
        }
        for (int cp = 0; cp < n_pairs; cp++) {
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                // k-step s + 2 (the last two prefetches repeat the last k-step): weights at (18 cp + tap) x 256
                int cp2 = cp, tp2 = tap + 2;
                if (tp2 >= 9) {
                    tp2 -= 9;
                    cp2 += 1;
                }
                if (cp2 >= n_pairs) {
                    cp2 = n_pairs - 1;
                    tp2 = 8;
                }
                const uint32_t w2 = (uint32_t)(18 * cp2 + tp2) * 4096u; // (its six loads go out in the tiles below)
                half8 ah[2], am[2], al[2];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    ah[m] = __builtin_bit_cast(half8, a_hi[tap % 3][m]);
                    am[m] = __builtin_bit_cast(half8, a_mid[tap % 3][m]);
                    al[m] = __builtin_bit_cast(half8, a_lo[tap % 3][m]);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int tile = tap * 4 + q, cur = tile % 3, nxt = (tile + 2) % 3;
                    const char *p = b_addr(tile + 2);
                    // (per accumulator the same order of additions as before: acc1 = ah bm + am bh, acc2 = ah bl + al bh + am bm)
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_POLICY_MFMA16(acc2[0][q], ah[0], bl[cur]);
                    bh[nxt] = *(const half8 *)p;
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_POLICY_MFMA16(acc2[1][q], ah[1], bl[cur]);
                    bm[nxt] = *(const half8 *)(p + 256);
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_POLICY_MFMA16(acc1[0][q], ah[0], bm[cur]);
                    bl[nxt] = *(const half8 *)(p + 512);
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_POLICY_MFMA16(acc1[1][q], ah[1], bm[cur]);
                    // the A operands of k-step s + 2: one 16-byte load per gap, six over the step's first three tiles
                    if (q == 0)
                        a_hi[(tap + 2) % 3][0] = a_load(wh, w2, 0);
                    else if (q == 1)
                        a_mid[(tap + 2) % 3][0] = a_load(wm, w2, 0);
                    else if (q == 2)
                        a_lo[(tap + 2) % 3][0] = a_load(wl, w2, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap & 1)
                        IAGO_POLICY_MFMA16(acc0b[0][q], ah[0], bh[cur]);
                    else
                        IAGO_POLICY_MFMA16(acc0[0][q], ah[0], bh[cur]);
                    if (q == 0)
                        a_hi[(tap + 2) % 3][1] = a_load(wh, w2, 1);
                    else if (q == 1)
                        a_mid[(tap + 2) % 3][1] = a_load(wm, w2, 1);
                    else if (q == 2)
                        a_lo[(tap + 2) % 3][1] = a_load(wl, w2, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap & 1)
                        IAGO_POLICY_MFMA16(acc0b[1][q], ah[1], bh[cur]);
                    else
                        IAGO_POLICY_MFMA16(acc0[1][q], ah[1], bh[cur]);
                    IAGO_POLICY_MFMA16(acc2[0][q], al[0], bh[cur]);
                    IAGO_POLICY_MFMA16(acc2[1][q], al[1], bh[cur]);
                    IAGO_POLICY_MFMA16(acc1[0][q], am[0], bh[cur]);
                    IAGO_POLICY_MFMA16(acc1[1][q], am[1], bh[cur]);
                    IAGO_POLICY_MFMA16(acc2[0][q], am[0], bm[cur]);
                    IAGO_POLICY_MFMA16(acc2[1][q], am[1], bm[cur]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // next chunk pair of 32 input channels: 64 B further in every row
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int tap = 0; tap < 9; tap++)
                    addr[q][tap] += 64u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++)
                addr[q][tap] -= 64u * (uint32_t)n_pairs;

        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T.
        // D row 4 kq + v of M tile m, column c16: channel 32 wv + 16 m + 4 kq + v of cell 16 q + c16
        f2 bia[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const float4 bq = *(const float4 *)(P.bias[L] + 32 * wv + 16 * m + 4 * kq);
            bia[m][0] = (f2){bq.x, bq.y};
            bia[m][1] = (f2){bq.z, bq.w};
        }
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); // (inline-asm MFMAs: their results land 4 passes after issue)
        __syncthreads();
        float vmax = 0.0f;
        f2 vsum = (f2){0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            char *row = T + wrow[q] + (32 * wv + 4 * kq) * 2;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                h2 p0[2], p1[2], p2[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; t2++) {
                    const f2 m0 = (f2){acc0[m][q][2 * t2], acc0[m][q][2 * t2 + 1]} + (f2){acc0b[m][q][2 * t2], acc0b[m][q][2 * t2 + 1]};
                    const f2 m1 = (f2){acc1[m][q][2 * t2], acc1[m][q][2 * t2 + 1]};
                    const f2 m2 = (f2){acc2[m][q][2 * t2], acc2[m][q][2 * t2 + 1]};
                    f2 v = (m2 * S2 + m1 * S1) + m0 + bia[m][t2];
                    vmax = fmaxf(fmaxf(vmax, v.x), v.y);
                    vsum += v;
                    v.x = __builtin_amdgcn_fmed3f(v.x, 0.0f, 65000.0f);
                    v.y = __builtin_amdgcn_fmed3f(v.y, 0.0f, 65000.0f);
                    split3(v, p0[t2], p1[t2], p2[t2]);
                }
                *(uint2 *)(row + 32 * m) = (uint2){__builtin_bit_cast(uint32_t, p0[0]), __builtin_bit_cast(uint32_t, p0[1])};
                *(uint2 *)(row + 32 * m + 256) =
                    (uint2){__builtin_bit_cast(uint32_t, p1[0]), __builtin_bit_cast(uint32_t, p1[1])};
                *(uint2 *)(row + 32 * m + 512) =
                    (uint2){__builtin_bit_cast(uint32_t, p2[0]), __builtin_bit_cast(uint32_t, p2[1])};
            }
        }
        // beyond the f16 range, or NaN (the clamp would hide it)
        saturated |= !(vmax <= 65000.0f) || !(vsum.x + vsum.y == vsum.x + vsum.y);
        __syncthreads();
    }
    if (P.overflow && saturated)
        *P.overflow = 1u;
    if (P.layer_hi < 7) { // the next launch goes on from this image
        uint4 *dst = P.scratch + (row_id - P.row_lo) * (64 * RS / 16);
        for (int e = tid; e < 64 * RS / 16; e += 256)
            dst[e] = *(const uint4 *)(T + e * 16);
        return;
    }

    // ---- head (network.py:29-47): conv9 (1x1, 128 -> 1, no bias), + bias10 per cell, softmax over
    // the 64 cells.  One wave: lane = cell, float32 on the exact values hi + mid 2^-11 + lo 2^-22.
    if (wv == 0) {
        const char *row = T + lane * RS;
        float acc = 0.0f;
#pragma unroll 4
        for (int c8 = 0; c8 < 16; c8++) {
            const half8 xh = *(const half8 *)(row + c8 * 16), xm = *(const half8 *)(row + 256 + c8 * 16),
                        xl = *(const half8 *)(row + 512 + c8 * 16);
            const float *w9 = SRCH ? head_w : P.w9;
            const float4 wa = *(const float4 *)(w9 + c8 * 8), wb = *(const float4 *)(w9 + c8 * 8 + 4);
            const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float xv = ((float)xl[e] * S2 + (float)xm[e] * S1) + (float)xh[e];
                acc = fmaf(w[e], xv, acc);
            }
        }
        const float logit = acc + (SRCH ? head_w[128 + lane] : P.b10[lane]);
        float mx = logit;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = expf(logit - mx);
        float sum = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            sum += __shfl_xor(sum, o);
        if constexpr (SRCH)
            res[lane] = e / sum;
        else
            P.probs[row_id * 64 + lane] = e / sum;
    }
}

// validates `a` and fills the kernel parameters except the launch's layer / row window (host side)
inline int policy_params_of(const iago_policy_split3_args *a, PolicyParams &P)
{
    if (!a || a->n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_forward_split3: null args or n < 0");
    if (!a->own || !a->opp || !a->w1 || !a->b1 || !a->w9 || !a->b10 || !a->probs)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_forward_split3: null pointer");
    for (int L = 0; L < 7; L++) {
        if (!a->w_hi[L] || !a->w_mid[L] || !a->w_lo[L] || !a->bias[L] || ((uintptr_t)a->w_hi[L] & 15u) ||
            ((uintptr_t)a->w_mid[L] & 15u) || ((uintptr_t)a->w_lo[L] & 15u) || ((uintptr_t)a->bias[L] & 15u))
            return iago_fail(IAGO_ERR_INVALID, "iago_policy_forward_split3: weights and biases of blocks 2..8 must be "
                                               "non-null and 16-byte aligned");
        P.w_hi[L] = (const uint4 *)a->w_hi[L];
        P.w_mid[L] = (const uint4 *)a->w_mid[L];
        P.w_lo[L] = (const uint4 *)a->w_lo[L];
        P.bias[L] = a->bias[L];
    }
    if (((uintptr_t)a->w9 & 15u) || ((uintptr_t)a->w1 & 15u) || ((uintptr_t)a->b1 & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_forward_split3: w1, b1, w9 must be 16-byte aligned");
    P.own = a->own;
    P.opp = a->opp;
    P.index = a->index;
    P.n_dev = a->n_dev;
    P.n = a->n;
    P.w1 = a->w1;
    P.b1 = a->b1;
    P.w9 = a->w9;
    P.b10 = a->b10;
    P.probs = a->probs;
    P.overflow = a->overflow;
    P.layer_lo = 0;
    P.layer_hi = 7;
    P.scratch = nullptr;
    P.row_lo = 0;
    P.row_hi = a->n;
    return IAGO_OK;
}

} // namespace iago_policy
