// conv_trunk_body.hpp -- the LDS-resident walk of the Value net (network.py:66-96) as device functions:
// trunk_item / trunk_walk and their parameter block, shared by the kernels of conv_trunk_kernel.hip (batched
// forwards, the leaf evaluation of a playout) and by the persistent search kernel (search_kernel.hip).
// The description of the arithmetic is at the head of conv_trunk_kernel.hip.
#pragma once
#include "abi_common.hpp"
#include "rollout_row_body.hpp"

#include <hip/hip_fp16.h>

#include <atomic>

namespace iago_trunk {


typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int RS = 544;            // bytes of a cell row: 128 ch hi | 128 ch lo | 32 B (bank skew: rows 2 x 16 B apart mod 256)
constexpr int ZB = 1024;           // zero bytes behind a board's 64 rows: the target of every out-of-board tap
constexpr int BS = 64 * RS + ZB;   // bytes of a board
// boards per workgroup TB = 4 (138,240 B of LDS) or 2 (small batches: half the latency per workgroup)
constexpr int lds_alloc(int tb) { return tb * BS + 1024; } // the operand prefetch of the last k-step reads up to 48 B past T
constexpr int MAX_LAYERS = 8;

// v_mfma_f32_16x16x32_f16 with the accumulator in place (AGPRs).  Why this shape: on random data the chip holds a
// higher clock under the 16x16x32 form than under 32x32x16 at equal FLOPs per cycle (MI355X_MICROARCH.md, DVFS item 7;
// tools/exp_mfma_shape2.hip).  Why inline asm: through the builtin hipcc gives a float4 accumulator a destination that
// is not its source C and moves the accumulators around every MFMA (7 v_accvgpr_* per MFMA: LABNOTES.md, round 5).
#define IAGO_MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

struct TrunkRParams {
    const uint4 *x_hi, *x_lo;   // input of layer 0  [n][cin0/16][64][16] f16
    uint4 *y_hi, *y_lo;         // output of the last layer [n][8][64][16] f16
    const uint4 *w_hi[MAX_LAYERS], *w_lo[MAX_LAYERS]; // [cin/16][3][3][128][16] f16
    const float *bias[MAX_LAYERS];
    int64_t n;
    int32_t cin0, n_layers;
    uint32_t *overflow;
    // FUSED (iago_value_forward_split): block1 in the prologue, block9 + fc10 + fc11 after block 8
    const float *planes;        // [n][2][8][8] float32, or NULL: the boards themselves
    const uint64_t *own, *opp;  // own = side to move (plane 1), opp = plane 0 (game.py:168-174)
    const float *w1, *b1;       // block1 [64][2][3][3], [64]
    const uint4 *w9_hi, *w9_lo; // block9 as MFMA A operand [8 chunks][32 rows = taps, 9 used][16] f16
    const float *b9, *w10, *w11;
    float *out;                 // [n] (scattered by index when given)
    const int64_t *index;       // optional gather list: row b = board index[b] of own / opp, value to out[index[b]]
    const int32_t *n_dev;       // optional device-side row count: only the first min(n, *n_dev) rows
    int32_t count_lo, count_hi; // this launch runs iff count_lo < rows <= count_hi (variant choice on the device)
    // FUSED only: a launch may run the layers [layer_lo, layer_hi) of blocks 2..8 alone -- block1
    // comes with layer_lo == 0, the head with layer_hi == n_layers; in between a board's 64 cell rows
    // (IMG bytes) travel through scratch [row][IMG] (the game-asynchronous steps' pieces)
    int32_t layer_lo, layer_hi;
    char *scratch;
};
constexpr int IMG = 64 * RS; // == IAGO_VALUE_IMAGE_BYTES

// What a workgroup's walk takes its rows from and how far it goes (FUSED): by value, in registers --
// the kernel parameters themselves stay untouched in the kernarg segment (a kernel that edits its
// TrunkRParams gets a private copy in scratch memory: 384 bytes per lane).
struct Piece {
    const int64_t *index;
    const int32_t *n_dev;
    char *scratch;
    int layer_lo, layer_hi;
    // the persistent search's hand-offs inside the workgroup, through LDS instead of global memory (a store to global
    // memory read back by the same workgroup is a round trip to L2: ~2 us under load, twice per walk):
    const uint32_t *pos; // optional, LDS: the boards' positions, six words per board (word 2 / 3 = own lo / hi, 4 / 5 = opp lo / hi)
    float *res;          // optional, LDS: the walk's values [TB] instead of P.out
    const float *w1s;    // optional, LDS: block1's weights [64][18] and biases [64] staged there already
    const char *head_w;  // optional, LDS: the head's weights staged there already (HEAD_W_* below)
};
// the head's weights as the persistent search keeps them in LDS: w9_hi | w9_lo (as in global memory) | w10 transposed
// ([16 float4 columns][128 rows]: a thread's row at stride 16 B between threads) | w11 [128] | b9
constexpr int HEAD_W_W9LO = 8192, HEAD_W_W10 = 16384, HEAD_W_W11 = 16384 + 32768, HEAD_W_B9 = HEAD_W_W11 + 512,
              HEAD_W_LDS = HEAD_W_B9 + 16;
__device__ __forceinline__ Piece whole_walk(const TrunkRParams &P)
{
    return Piece{P.index, P.n_dev, P.scratch, P.layer_lo, P.layer_hi, nullptr, nullptr, nullptr, nullptr};
}

constexpr int head_lds(int tb) { return (9 * 64 * tb + 64 * tb + 128 * tb) * 4; } // tap maps, block9 output, fc terms
constexpr int W1_LDS = (64 * 18 + 64) * 4;   // block1's weights and biases, staged per walk
constexpr int lds_alloc_fused(int tb) { return lds_alloc(tb) + head_lds(tb) + W1_LDS; }

// ds_read_b128 serves lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (and the same + 32)
// in separate LDS cycles: the first group holds cells 0-15 of a 32-cell block, the second
// cells 16-31, so that with rows 4 banks apart (RS = 16 mod 256) the 16 lanes of a cycle hit
// 16 different 4-bank groups for every tap.
__device__ __forceinline__ int cell_of_lane(int r)
{
    return r < 4 ? r : r < 12 ? 16 + (r - 4) : r < 16 ? 4 + (r - 12) : r < 20 ? 24 + (r - 16) : r < 28 ? 8 + (r - 20)
                                                                                              : 28 + (r - 28);
}

extern __shared__ __align__(16) char trunk_lds[];

// The work of one workgroup on the TB boards (rows) b0 .. b0 + TB - 1 of n_rows.
// SRCH: the persistent search's form -- W.pos, W.res, W.w1s and W.head_w are all given (decided at compile time: as
// run-time choices both forms' registers were live in the head and the kernel spilled 103 of them)
template <bool FUSED, int TB, bool SRCH = false>
__device__ __forceinline__ void trunk_item(const TrunkRParams &P, const Piece &W, const int64_t b0, const int64_t n_rows)
{
    constexpr int NN = 4 * TB;      // 16-cell B tiles of a k-step: TB boards x 4 quarters
    char *const T = trunk_lds;
    // (SRCH: the walk is called from the net workgroups' loop.  Everything below that depends on the thread alone -- the
    // 36 B-operand addresses, the rows of the epilogue, the A operands' lane offset -- would be hoisted out of that loop,
    // kept live across both nets' walks and, there being no registers for it, spilled to scratch memory and reloaded
    // walk after walk.  An opaque copy of the thread id makes them this walk's own values: recomputed, never spilled)
    int tid_ = threadIdx.x;
    if constexpr (SRCH)
        asm volatile("" : "+v"(tid_));
    const int tid = tid_, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;

    // ---- the zero areas, then the input of the first layer
    if (tid < TB * (ZB / 16))
        *(uint4 *)(T + (tid / (ZB / 16)) * BS + 64 * RS + (tid % (ZB / 16)) * 16) = make_uint4(0, 0, 0, 0);
    bool saturated = false;
    constexpr bool PIECES = FUSED && TB <= 2; // (the 4-board variant only ever runs whole walks)
    if (PIECES && W.layer_lo > 0) {
        // a later piece of the walk: the boards' rows as the previous piece left them (all of a
        // thread's loads in flight together, then its LDS stores)
        constexpr int PER = (TB * (IMG / 16) + 255) / 256;
        uint4 img[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = min(tid + i * 256, TB * (IMG / 16) - 1);
            const int board = e / (IMG / 16), off = e - board * (IMG / 16);
            const int64_t row = min(b0 + board, n_rows - 1);
            img[i] = ((const uint4 *)(W.scratch + row * IMG))[off];
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * 256;
            const int board = e / (IMG / 16), off = e - board * (IMG / 16);
            if (e < TB * (IMG / 16))
                *(uint4 *)(T + board * BS + off * 16) = img[i];
        }
    } else if constexpr (FUSED) {
        // block1 (3x3, 2 -> 64, bias, ReLU; network.py:66-70) straight into T: the arithmetic of
        // value_stem_kernel (conv_kernels.hip) -- same FMA order, same split -- per (board,
        // cell = lane, 8 output channels = 16 bytes of a row); the channel group is wave-uniform
        const int cell = tid & 63, y = cell >> 3, x = cell & 7;
        float in[TB][18]; // the 3x3 neighbourhoods of this lane's cell on both planes, 4 boards
#pragma unroll
        for (int board = 0; board < TB; board++) {
            const int64_t row = min(b0 + board, n_rows - 1);
            const int64_t b = W.index ? W.index[row] : row;
            const float *pl = P.planes + b * 128;
            uint64_t bits0, bits1;
            if constexpr (SRCH) {
                const uint32_t *w = W.pos + 6 * board;
                bits1 = ((uint64_t)w[3] << 32) | w[2];
                bits0 = ((uint64_t)w[5] << 32) | w[4];
            } else {
                bits0 = P.planes ? 0ull : P.opp[b];
                bits1 = P.planes ? 0ull : P.own[b];
            }
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const int yy = y + ky - 1, xx = x + kx - 1;
                        const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
                        const int a = (yy * 8 + xx) & 63;
                        float v;
                        if (P.planes)
                            v = ok ? pl[c * 64 + a] : 0.0f;
                        else
                            v = (ok && (((c ? bits1 : bits0) >> a) & 1ull)) ? 1.0f : 0.0f;
                        in[board][c * 9 + ky * 3 + kx] = v;
                    }
        }
        // block1's weights and biases (4.9 KB) through LDS: ONE round trip to L2 for the workgroup, then broadcast reads
        // per output channel (fetched from global memory channel by channel they were sixteen dependent round trips per
        // wave: 5.2 us of a pair's walk under load, LABNOTES.md round 5)
        const float *w1s = W.w1s; // [64][18] weights, [64] biases
        if constexpr (!SRCH) {
            float *const st = (float *)(T + lds_alloc(TB) + head_lds(TB));
            for (int e = tid; e < 64 * 18 / 4; e += 256)
                ((float4 *)st)[e] = ((const float4 *)P.w1)[e];
            if (tid < 16)
                ((float4 *)(st + 64 * 18))[tid] = ((const float4 *)P.b1)[tid];
            __syncthreads();
            w1s = st;
        }
        // the weights of an output channel are wave-uniform: each is fetched once and used for the boards
#pragma unroll 1
        for (int g2 = 0; g2 < 2; g2++) {
            const int grp = __builtin_amdgcn_readfirstlane(g2 * 4 + wv); // channel block * 2 + half
            const int co0 = grp * 8;
            _Float16 h8[TB][8], l8[TB][8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float *wk = w1s + (co0 + k) * 18; // [co][ci][ky][kx]
                const float bk = w1s[64 * 18 + co0 + k];
                float wreg[18];
#pragma unroll
                for (int j = 0; j < 18; j++)
                    wreg[j] = wk[j];
                // boards in pairs per packed FMA (a lone wave pays per instruction, not per lane-operation): each board's
                // chain of 18 fused multiply-adds in the same order as one by one
                float accs[TB];
                if constexpr (TB % 2 == 0) {
#pragma unroll
                    for (int bp = 0; bp < TB; bp += 2) {
                        f2 acc = (f2){bk, bk};
#pragma unroll
                        for (int j = 0; j < 18; j++)
                            acc = __builtin_elementwise_fma((f2){wreg[j], wreg[j]}, (f2){in[bp][j], in[bp + 1][j]}, acc);
                        accs[bp] = acc.x;
                        accs[bp + 1] = acc.y;
                    }
                } else {
#pragma unroll
                    for (int board = 0; board < TB; board++) {
                        float acc = bk;
#pragma unroll
                        for (int j = 0; j < 18; j++)
                            acc = fmaf(wreg[j], in[board][j], acc);
                        accs[board] = acc;
                    }
                }
#pragma unroll
                for (int board = 0; board < TB; board++) {
                    const float acc = accs[board];
                    saturated |= !(acc <= 65000.0f);
                    const float v = fminf(fmaxf(acc, 0.0f), 65000.0f);
                    const _Float16 vh = (_Float16)v;
                    h8[board][k] = vh;
                    l8[board][k] = (_Float16)((v - (float)vh) * 2048.0f);
                }
            }
#pragma unroll
            for (int board = 0; board < TB; board++) {
                char *dst = T + board * BS + cell * RS + (grp >> 1) * 32 + (grp & 1) * 16;
                *(uint4 *)dst = *(const uint4 *)h8[board];
                *(uint4 *)(dst + 256) = *(const uint4 *)l8[board];
            }
        }
    } else {
        const int chunks0 = P.cin0 >> 4;
        const int pieces = TB * chunks0 * 128; // 16-byte pieces per hi / lo
        for (int e = tid; e < pieces; e += 256) {
            const int board = e / (chunks0 * 128), rem = e - board * chunks0 * 128;
            const int chunk = rem >> 7, cell = (rem >> 1) & 63, hp = rem & 1;
            // boards past the end of a ragged batch read the last board (results not stored)
            const int64_t b = min(b0 + board, P.n - 1);
            const int64_t src = (b * chunks0 + chunk) * 128 + (rem & 127);
            char *dst = T + board * BS + cell * RS + chunk * 32 + hp * 16;
            *(uint4 *)dst = P.x_hi[src];
            *(uint4 *)(dst + 256) = P.x_lo[src];
        }
    }
    __syncthreads();

    // ---- per-lane addresses of the B operand.  The K loop runs on v_mfma_f32_16x16x32_f16: lane = (column c16 =
    // lane & 15, k quarter kq = lane >> 4); a k-step covers 32 input channels = two 16-channel chunks at one tap,
    // a B tile is 16 cells (quarter q of a board) x 32 channels: this lane reads cell 16 q + c16, channels 8 kq ..
    // 8 kq + 7 of the chunk pair (16 bytes at + 16 kq), tap (ky, kx).  Boards 0 / 1 through addr (+ an immediate BS
    // for the odd board), boards 2 / 3 by adding 2 BS; the hi / lo halves of a row are 256 B apart (immediate)
    const int c16 = lane & 15, kq = lane >> 4;
    uint32_t addr[4][9]; // [quarter][tap]
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int cell = 16 * q + c16;
            const int yy = (cell >> 3) + tap / 3 - 1, xx = (cell & 7) + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
            // an out-of-board tap reads zeros from the slot with the bank offset its row would have had (rows are
            // 2 x 16 B apart mod 256 B: 32 B per row index mod 8): the 16 lanes of an LDS cycle keep 16 different
            // 4-bank groups whether or not some of them are redirected
            const int lin = (cell + (tap / 3 - 1) * 8 + (tap % 3 - 1)) & 7;
            addr[q][tap] = (uint32_t)((ok ? (yy * 8 + xx) * RS : 64 * RS + 32 * lin) + kq * 16);
        }
    // rows this lane writes in the epilogue (its cell of every quarter)
    uint32_t wrow[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        wrow[q] = (uint32_t)((16 * q + c16) * RS);

    const int L_lo = PIECES ? W.layer_lo : 0, L_hi = PIECES ? W.layer_hi : P.n_layers;
    for (int L = L_lo; L < L_hi; L++) {
        const int n_pairs = L == 0 ? (P.cin0 >> 5) : 4; // chunk pairs: 32 input channels each
        // this lane's A operands: output channels 32 wv + c16 (M tile 0) and + 16 (M tile 1), input channels
        // 8 (kq & 1) .. + 7 of chunk 2 cp + (kq >> 1); a chunk is 9 x 128 x 32 B, a tap 128 x 32 B further.  Addressed as
        // a wave-uniform base (the layer's weights + the k-step's offset: scalar registers) + this lane's byte offset
        const char *const wh = (const char *)P.w_hi[L], *const wl = (const char *)P.w_lo[L];
        const uint32_t a_lane = (uint32_t)(((32 * wv + c16) * 2 + (kq & 1) + (kq >> 1) * (9 * 256)) * 16);
        auto a_load = [&](const char *base, uint32_t step_bytes, int m) -> u32x4 {
            return *(const u32x4 *)(base + step_bytes + a_lane + (uint32_t)m * 512u);
        };
        float4v acc_main[2][NN], acc_cross[2][NN];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int n = 0; n < NN; n++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    acc_main[m][n][v] = 0.0f;
                    acc_cross[m][n][v] = 0.0f;
                }

        // k-steps of 32 input channels: s = 9 cp + tap, 9 n_pairs of them
        u32x4 a_hi[3][2], a_lo[3][2]; // k-steps s, s + 1, s + 2 (ring index = tap % 3) x the two M tiles
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int m = 0; m < 2; m++) {
                a_hi[i][m] = a_load(wh, (uint32_t)i * 4096u, m);
                a_lo[i][m] = a_load(wl, (uint32_t)i * 4096u, m);
            }
        // B operands THREE tiles ahead of the MFMAs that use them (four register sets): an LDS read issued now has
        // eighteen MFMAs (288 cycles) to arrive.  A v_mfma_f32_16x16x32_f16 leaves 8 of its 16 cycles to the wave's
        // other instructions: the two reads of a tile go out one per MFMA gap, and the two MFMAs that add into the
        // same accumulator stand four apart (measured with tools/exp_walk_stamps.py on the stamped build: with the
        // reads bunched between two tiles and one MFMA between the dependent pair the K loops of a pair took 99.6 us
        // against 75.9 us of MFMA issue; LABNOTES.md, round 5)
        half8 bh[4], bl[4];
        auto b_addr = [&](int tile) -> const char * {
            // tile = tap * NN + n of the running chunk pair; 9 NN .. 9 NN + 2 = the first three tiles of the next pair
            const int over = tile >= 9 * NN ? 64 : 0, tt = tile % (9 * NN), tp = tt / NN, n = tt % NN;
            return T + addr[n & 3][tp] + (n >> 2) * BS + over;
        };
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const char *p0 = b_addr(i);
            bh[i] = *(const half8 *)p0;
            bl[i] = *(const half8 *)(p0 + 256);
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
                const uint32_t w2 = (uint32_t)(18 * cp2 + tp2) * 4096u; // (its four loads go out one per tile below)
                const half8 ah0 = __builtin_bit_cast(half8, a_hi[tap % 3][0]), ah1 = __builtin_bit_cast(half8, a_hi[tap % 3][1]);
                const half8 al0 = __builtin_bit_cast(half8, a_lo[tap % 3][0]), al1 = __builtin_bit_cast(half8, a_lo[tap % 3][1]);
#pragma unroll
                for (int n = 0; n < NN; n++) {
                    const int tile = tap * NN + n, cur = tile % 4, nxt = (tile + 3) % 4;
                    // (past the last chunk pair: harmless reads 64 B further in the same rows)
                    const char *p = b_addr(tile + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_MFMA16(acc_cross[0][n], ah0, bl[cur]);
                    bh[nxt] = *(const half8 *)p;
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_MFMA16(acc_cross[1][n], ah1, bl[cur]);
                    bl[nxt] = *(const half8 *)(p + 256);
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_MFMA16(acc_main[0][n], ah0, bh[cur]);
                    // the A operands of k-step s + 2: one 16-byte load in this gap of each of the step's first four tiles
                    if (n == 0)
                        a_hi[(tap + 2) % 3][0] = a_load(wh, w2, 0);
                    else if (n == 1)
                        a_hi[(tap + 2) % 3][1] = a_load(wh, w2, 1);
                    else if (n == 2)
                        a_lo[(tap + 2) % 3][0] = a_load(wl, w2, 0);
                    else if (n == 3)
                        a_lo[(tap + 2) % 3][1] = a_load(wl, w2, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    IAGO_MFMA16(acc_main[1][n], ah1, bh[cur]);
                    IAGO_MFMA16(acc_cross[0][n], al0, bh[cur]);
                    IAGO_MFMA16(acc_cross[1][n], al1, bh[cur]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // next chunk pair of 32 input channels: 64 B further in every row (the zero rows are RS bytes of
            // zeros and more: their addresses move along)
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

        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T
        // bias of the 8 channels this lane finishes: 32 wv + 16 m + 4 kq + v
        f2 bia[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const float4 bq = *(const float4 *)(P.bias[L] + 32 * wv + 16 * m + 4 * kq);
            bia[m][0] = (f2){bq.x, bq.y};
            bia[m][1] = (f2){bq.z, bq.w};
        }
        // (the MFMAs above are inline asm: the compiler does not know that their results land 4 passes after issue;
        // the barrier alone is hundreds of cycles, the s_nop makes the read-after-MFMA distance explicit)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __syncthreads();
        // One wave per SIMD pays 4 cycles per instruction: packed float32 math, v_med3 for the
        // clamp, v_cvt_pk_f16_f32; the range check is a running maximum and a running sum (a
        // NaN survives in the sum) instead of a compare per value.
        float vmax = 0.0f;
        f2 vsum = (f2){0.0f, 0.0f};
#pragma unroll
        for (int n = 0; n < NN; n++) {
            // D row 4 kq + v of M tile m, column c16: channel 32 wv + 16 m + 4 kq + v of cell 16 (n & 3) + c16, board n >> 2
            char *row = T + wrow[n & 3] + (n >> 2) * BS + (32 * wv + 4 * kq) * 2;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                h2 hi[2], lo[2];
#pragma unroll
                for (int t2 = 0; t2 < 2; t2++) {
                    const f2 mm = (f2){acc_main[m][n][2 * t2], acc_main[m][n][2 * t2 + 1]};
                    const f2 c = (f2){acc_cross[m][n][2 * t2], acc_cross[m][n][2 * t2 + 1]};
                    f2 v = c * (1.0f / 2048.0f) + mm + bia[m][t2];
                    vmax = fmaxf(fmaxf(vmax, v.x), v.y);
                    vsum += v;
                    v.x = __builtin_amdgcn_fmed3f(v.x, 0.0f, 65000.0f);
                    v.y = __builtin_amdgcn_fmed3f(v.y, 0.0f, 65000.0f);
                    hi[t2] = __builtin_convertvector(v, h2);
                    lo[t2] = __builtin_convertvector((v - __builtin_convertvector(hi[t2], f2)) * 2048.0f, h2);
                }
                *(uint2 *)(row + 32 * m) = (uint2){__builtin_bit_cast(uint32_t, hi[0]), __builtin_bit_cast(uint32_t, hi[1])};
                *(uint2 *)(row + 32 * m + 256) =
                    (uint2){__builtin_bit_cast(uint32_t, lo[0]), __builtin_bit_cast(uint32_t, lo[1])};
            }
        }
        // beyond the f16 range, or NaN (the clamp would hide it)
        saturated |= !(vmax <= 65000.0f) || !(vsum.x + vsum.y == vsum.x + vsum.y);
        __syncthreads();
    }
    if (P.overflow && saturated)
        *P.overflow = 1u;

    if (PIECES && W.layer_hi < P.n_layers) {
        // the next piece of the walk goes on from these rows
        for (int e = tid; e < TB * (IMG / 16); e += 256) {
            const int board = e / (IMG / 16), off = e - board * (IMG / 16);
            if (b0 + board < n_rows)
                ((uint4 *)(W.scratch + (b0 + board) * IMG))[off] = *(const uint4 *)(T + board * BS + off * 16);
        }
        return;
    }
    if constexpr (FUSED) {
        // ---- block9 (3x3, 128 -> 1, bias, ReLU) + fc10 + fc11 (network.py:78-96, train=False) on
        // the activations still in T.  The 3x3 convolution with ONE output channel as a 1x1
        // convolution with 9: tap map M[tap][cell'] = sum_c w9[c][tap] x[c][cell'] on the MFMA
        // units (A = the 9 tap rows of w9, zero-padded to 32; B = the centre tap's operand of
        // the layers above; same split arithmetic), then out[cell] = sum_tap M[tap][cell + off(tap)].
        constexpr int NC = 64 * TB;                      // cells of the workgroup's boards
        float *const Dm = (float *)(T + lds_alloc(TB));  // [9][TB boards * 64 cells]
        float *const h9s = Dm + 9 * NC;                  // [TB][64]
        float *const hid = h9s + NC;                     // [TB][128]
        // this thread's fc10 row (tid & 127), fetched under the MFMAs below
        // (from LDS where the persistent search keeps the head's weights: fetched from global memory per walk they were
        // 96 KB through the CU's L2 port and, streamed past by the trunks' weights, mostly L2 misses -- 6.6 us of a
        // pair's walk under load: LABNOTES.md, round 5)
        // (SRCH: the row comes out of LDS right before fc10 -- sixteen reads in flight under the tap sums; held from
        // here on, its 64 registers and the 64 of block9's weights below were more than the search kernel has: the
        // weights went to scratch memory and came back one dword at a time in front of every MFMA of block9, a chain of
        // dependent global-memory round trips in every walk's head: LABNOTES.md, round 6)
        float4 w10row[16];
        if constexpr (!SRCH) {
#pragma unroll
            for (int c = 0; c < 16; c++)
                w10row[c] = ((const float4 *)(P.w10 + (tid & 127) * 64))[c];
        }
        const float w11j = SRCH ? ((const float *)(W.head_w + HEAD_W_W11))[tid & 127] : P.w11[tid & 127];
        const float b9 = SRCH ? *(const float *)(W.head_w + HEAD_W_B9) : P.b9[0];
        const int lane_cell = cell_of_lane(r); // (this block's 32x32x16 lane map: output row r, k half h)
        if (wv < TB) { // wave wv takes board wv
            const u32x4 *w9h = (const u32x4 *)P.w9_hi + r * 2 + h;
            const u32x4 *w9l = (const u32x4 *)P.w9_lo + r * 2 + h;
            float16v hm[2], hc[2];
#pragma unroll
            for (int jt = 0; jt < 2; jt++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    hm[jt][v] = 0.0f;
                    hc[jt][v] = 0.0f;
                }
            // (all sixteen 16-byte loads in flight together: one L2 round trip instead of eight dependent ones -- the
            // head of a pair took 8.4 us under load with a load, a wait and three MFMAs per chunk: LABNOTES.md, round 5)
            // (SRCH: the weights are in LDS -- a chunk's two reads go out with its B operands, nothing is held)
            u32x4 w9a[SRCH ? 1 : 8], w9b[SRCH ? 1 : 8];
            if constexpr (!SRCH) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    w9a[c] = w9h[c * 64];
                    w9b[c] = w9l[c * 64];
                }
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const half8 ah = SRCH ? *(const half8 *)(W.head_w + (c * 64 + r * 2 + h) * 16)
                                      : __builtin_bit_cast(half8, w9a[SRCH ? 0 : c]);
                const half8 al = SRCH ? *(const half8 *)(W.head_w + HEAD_W_W9LO + (c * 64 + r * 2 + h) * 16)
                                      : __builtin_bit_cast(half8, w9b[SRCH ? 0 : c]);
#pragma unroll
                for (int jt = 0; jt < 2; jt++) {
                    const int jj = 2 * wv + jt; // wave-uniform: board jj >> 1, cell half jj & 1
                    const char *p = T + (jj >> 1) * BS + (32 * (jj & 1) + lane_cell) * RS + h * 16 + c * 32; // centre tap
                    const half8 bh9 = *(const half8 *)p, bl9 = *(const half8 *)(p + 256);
                    hm[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh9, hm[jt], 0, 0, 0);
                    hc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl9, hc[jt], 0, 0, 0);
                    hc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh9, hc[jt], 0, 0, 0);
                }
            }
            // D row m = 8 (v >> 2) + 4 h + (v & 3): taps 4 h .. 4 h + 3 in v = 0..3, tap 8 in v = 4 (h = 0)
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                const int jj = 2 * wv + jt;
                const int base = (jj >> 1) * 64 + 32 * (jj & 1) + lane_cell;
#pragma unroll
                for (int t = 0; t < 4; t++)
                    Dm[(4 * h + t) * NC + base] = hm[jt][t] + hc[jt][t] * (1.0f / 2048.0f);
                if (h == 0)
                    Dm[8 * NC + base] = hm[jt][4] + hc[jt][4] * (1.0f / 2048.0f);
            }
        }
        __syncthreads();
        if constexpr (SRCH) {
#pragma unroll
            for (int c = 0; c < 16; c++)
                w10row[c] = ((const float4 *)(W.head_w + HEAD_W_W10))[c * 128 + (tid & 127)];
        }
        if (tid < NC) {
            const int cell = tid & 63, y = cell >> 3, x = cell & 7;
            float s9 = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8)
                    s9 += Dm[tap * NC + (tid & ~63) + yy * 8 + xx];
            }
            h9s[tid] = fmaxf(s9 + b9, 0.0f);
        }
        __syncthreads();
        if ((tid >> 7) * 2 < TB) {
            // fc10 row j for two boards, then its fc11 term (no bias, no activation in between)
            const int j = tid & 127, pb = (tid >> 7) * 2;
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const float4 wq = w10row[c];
                const float4 x0 = *(const float4 *)(h9s + pb * 64 + 4 * c);
                const float4 x1 = *(const float4 *)(h9s + (pb + 1 < TB ? pb + 1 : pb) * 64 + 4 * c);
                s0 = fmaf(wq.x, x0.x, s0);
                s0 = fmaf(wq.y, x0.y, s0);
                s0 = fmaf(wq.z, x0.z, s0);
                s0 = fmaf(wq.w, x0.w, s0);
                s1 = fmaf(wq.x, x1.x, s1);
                s1 = fmaf(wq.y, x1.y, s1);
                s1 = fmaf(wq.z, x1.z, s1);
                s1 = fmaf(wq.w, x1.w, s1);
            }
            hid[pb * 128 + j] = s0 * w11j;
            if (pb + 1 < TB)
                hid[(pb + 1) * 128 + j] = s1 * w11j;
        }
        __syncthreads();
        if (tid < TB && b0 + tid < n_rows) {
            // fixed order j = 0..127 (the result does not depend on the launch shape); the 32
            // LDS reads are issued together, the additions are one dependent chain
            float4 hv[32];
#pragma unroll
            for (int j4 = 0; j4 < 32; j4++)
                hv[j4] = *(const float4 *)(hid + tid * 128 + 4 * j4);
            float v = 0.0f;
#pragma unroll
            for (int j4 = 0; j4 < 32; j4++) {
#pragma clang fp reassociate(off)
                v += hv[j4].x;
                v += hv[j4].y;
                v += hv[j4].z;
                v += hv[j4].w;
            }
            const int64_t row = b0 + tid;
            if constexpr (SRCH)
                W.res[tid] = v;
            else
                P.out[W.index ? W.index[row] : row] = v;
        }
        return;
    }
    // ---- the last layer's activations: coalesced 16-byte stores, [n][8][64][16] hi and lo
    for (int e = tid; e < TB * 1024; e += 256) {
        const int board = e >> 10, cb = (e >> 7) & 7, cell = (e >> 1) & 63, hp = e & 1;
        const int64_t b = b0 + board;
        if (b < P.n) {
            const char *src = T + board * BS + cell * RS + cb * 32 + hp * 16;
            P.y_hi[b * 1024 + (e & 1023)] = *(const uint4 *)src;
            P.y_lo[b * 1024 + (e & 1023)] = *(const uint4 *)(src + 256);
        }
    }
}

// Workgroup `bid` of `nb` walks its rows with that stride (one pass unless the grid was capped:
// the device-counted launch of the value cache, iago_value_forward_split).
template <bool FUSED, int TB>
__device__ __forceinline__ void trunk_walk(const TrunkRParams &P, const Piece &W, const int64_t bid, const int64_t nb)
{
    int64_t n_rows = P.n;
    if constexpr (FUSED) {
        // device-side row count and variant choice: uniform over the launch, before any barrier
        if (W.n_dev)
            n_rows = min(P.n, (int64_t)*W.n_dev);
        if (n_rows <= P.count_lo || n_rows > P.count_hi)
            return;
    }
    for (int64_t b0 = bid * TB; b0 < n_rows; b0 += nb * TB) {
        trunk_item<FUSED, TB>(P, W, b0, n_rows);
        __syncthreads(); // the next pass re-stages the LDS image the head just read
    }
}

// validates `a` and fills the kernel parameters (host side; every entry point that walks the Value net)
// validates `a` and fills the kernel parameters; `who` names the entry point in error messages
inline int value_params_of(const iago_value_split_args *a, TrunkRParams &P)
{
    if (!a || a->n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_split: null args or n < 0");
    if ((!a->planes && (!a->own || !a->opp)) || !a->w1 || !a->b1 || !a->w9_hi || !a->w9_lo || !a->b9 || !a->w10 ||
        !a->w11 || !a->out)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_split: null pointer");
    for (int L = 0; L < 7; L++) {
        if (!a->w_hi[L] || !a->w_lo[L] || !a->bias[L] || ((uintptr_t)a->w_hi[L] & 15u) || ((uintptr_t)a->w_lo[L] & 15u) ||
            ((uintptr_t)a->bias[L] & 15u))
            return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_split: weights and biases of blocks 2..8 must be "
                                               "non-null and 16-byte aligned");
        P.w_hi[L] = (const uint4 *)a->w_hi[L];
        P.w_lo[L] = (const uint4 *)a->w_lo[L];
        P.bias[L] = a->bias[L];
    }
    P.w_hi[7] = P.w_hi[6];
    P.w_lo[7] = P.w_lo[6];
    P.bias[7] = P.bias[6];
    if (((uintptr_t)a->w9_hi & 15u) || ((uintptr_t)a->w9_lo & 15u) || ((uintptr_t)a->w10 & 15u) || ((uintptr_t)a->w1 & 15u) ||
        ((uintptr_t)a->b1 & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_split: w1, b1, w9_hi, w9_lo, w10 must be 16-byte aligned");
    P.x_hi = P.x_lo = nullptr;
    P.y_hi = P.y_lo = nullptr;
    P.n = a->n;
    P.cin0 = 64;
    P.n_layers = 7;
    P.overflow = a->overflow;
    P.planes = a->planes;
    P.own = a->own;
    P.opp = a->opp;
    P.w1 = a->w1;
    P.b1 = a->b1;
    P.w9_hi = (const uint4 *)a->w9_hi;
    P.w9_lo = (const uint4 *)a->w9_lo;
    P.b9 = a->b9;
    P.w10 = a->w10;
    P.w11 = a->w11;
    P.out = a->out;
    if (a->index && a->planes)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_split: a gather list needs the boards, not planes");
    P.index = a->index;
    P.n_dev = a->n_dev;
    P.layer_lo = 0;
    P.layer_hi = 7;
    P.scratch = nullptr;
    return IAGO_OK;
}

} // namespace iago_trunk
