// conv_kernels.hip -- the 3x3 convolution + bias + ReLU of Block.__call__
// (network.py:5-13) for 8x8 boards on the MFMA units, for the inference path of the
// Value net (network.py:66-96), which is >95 % of the net evaluations of the PV-MCTS
// loop (every playout evaluates one leaf, MCTS.py:110-131).
//
// Arithmetic: "split f16".  gfx950 has no reduced-precision fast path for f32 MFMA
// inputs (v_mfma_f32_32x32x2_f32 runs at the VALU rate), but f16 MFMA is 16x faster.
// Every f32 operand a is carried as two f16 numbers
//     a_hi = f16(a),   a_lo = f16((a - a_hi) * 2^11)        (a = a_hi + a_lo * 2^-11,
//                                                             22 significant bits)
// and a product sum is three MFMAs into two f32 accumulators:
//     main  += w_hi * x_hi
//     cross += w_hi * x_lo + w_lo * x_hi                     result = main + cross * 2^-11
// (the w_lo * x_lo term, 2^-22 relative, is dropped).  Accumulation is f32.  Measured
// deviation of the whole Value forward from an f64 evaluation: 3.7e-7 with the shipped
// weights (plain f32: 2.5e-7); tests/test_conv_gpu.py holds it to the 1e-5 bar.
//
// Activation format between layers ("split channel blocks"): two f16 tensors
// hi[b][C/16][64][16] and lo[...] -- 16 input channels of one cell are 32 contiguous
// bytes (one MFMA k-step of a lane pair), one channel block of a board 2 KB.
//
// Kernel: one workgroup = 4 boards x all 128 output channels; CS waves per board (1: one
// wave per SIMD with the whole register file, 4 x 2 tiles of v_mfma_f32_32x32x16_f16 and
// 2 x 128 accumulator registers per wave; 2, the default: two waves per SIMD, 2 x 2 tiles
// each -- same speed at 1024 boards, no spills, shorter epilogue).  The K loop runs over
// stages of (16 input channels) x (one kernel row = 3 taps); the padded 10x10 planes of a
// channel block and the weights of a stage are double-buffered in LDS (159 KB); the data of
// stage s + 1 are fetched into registers a stage ahead and written to LDS right after the
// barrier that opens stage s.
//
// The same file holds the float32 kernels around it: the Value stem and head, and the
// small-batch float32 stack (conv3x3_f32_kernel, stem_f32_kernel, policy_head_kernel) that
// serves the policy net on the expansions of a playout and single-game play.
#include "abi_common.hpp"

#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4))); // a 16-byte piece in registers
typedef float f32x4 __attribute__((ext_vector_type(4)));        // (HIP's float4 / uint4 structs end up in scratch)

constexpr int TB = 4;           // boards per workgroup
constexpr int CS = 2;            // waves per board: each owns 128 / CS output channels
constexpr int NI = 4 / CS;       // 32-channel blocks per wave
constexpr int THREADS = 64 * TB * CS;
constexpr int XQ = 512 / THREADS;                 // X pieces per thread and hi/lo
constexpr int WQ = (768 + THREADS - 1) / THREADS; // W pieces per thread and hi/lo (the last may repeat)
static_assert(XQ * THREADS == 512 && (CS == 1 || CS == 2), "piece counts");
constexpr int COUT = 128;
constexpr int ROW = 48;         // bytes per LDS row: 16 halfs + 16 B padding (conflict-free b128 reads)
constexpr int PP = 100;         // padded 10x10 plane
// Rows of the padded planes in the X buffers start at xrow(Y) = 24 (Y >> 1) + 10 (Y & 1)
// (in 48-byte cell rows) instead of 10 Y: with the lane -> cell map of cell_of_lane() the
// 16 lanes that one LDS cycle of a ds_read_b128 serves then always sit on 16 different
// 4-bank groups, for every tap (xrow(Y + 2) = xrow(Y) + 24 = 8 mod 16).
constexpr int XPP = 116;                       // cell rows per padded plane
__host__ __device__ constexpr int xrow(int Y)
{
    return 24 * (Y >> 1) + 10 * (Y & 1);
}
constexpr int X_HALF = TB * XPP * ROW;         // one of hi / lo: 22,272 B
constexpr int X_BUF = 2 * X_HALF;              // 44,544 B
constexpr int W_HALF = 3 * COUT * ROW;         // 18,432 B
constexpr int W_BUF = 2 * W_HALF;              // 36,864 B
constexpr int LDS_BYTES = 2 * X_BUF + 2 * W_BUF; // 162,816 B of 163,840
constexpr int T_ROW = (COUT + 4) * 2;          // epilogue image: 132 halfs per cell
constexpr int T_HALF = TB * 64 * T_ROW;        // 67,584 B
static_assert(2 * T_HALF + COUT * 4 <= LDS_BYTES, "epilogue image must fit in the staging buffers");

struct ConvParams {
    const uint4 *x_hi; // [n][cin/16][64][16] f16
    const uint4 *x_lo;
    const uint4 *w_hi; // [cin/16][3][3][128][16] f16
    const uint4 *w_lo;
    const float *bias; // [128]
    uint4 *y_hi;       // [n][8][64][16] f16
    uint4 *y_lo;
    int64_t n;
    int32_t n_chunks; // cin / 16
    uint32_t *overflow; // optional: set to 1 when an output left [0, 65000] or is NaN
    // MODE_BWD (backward-data of a block, csrc/policy_grad_kernels.hip): x = the gradient at the block's
    // pre-activations times 2^*scale_exp, w = the block's weights transposed and flipped, no bias; the output --
    // times 2^-*scale_exp, zeroed where the saved activation of the block below (mask_hi / mask_lo, out_blocks
    // channel blocks) is not positive -- goes out as float32 channel blocks, its largest magnitude to *max_bits
    const int32_t *scale_exp;
    const uint2 *mask_hi, *mask_lo; // [n][out_blocks][64][16] f16
    f32x4 *y_f32;                   // [n][out_blocks][64][16] float32
    uint32_t *max_bits;
    int32_t out_blocks;
};
constexpr int MODE_FWD = 0, MODE_BWD = 1;
constexpr int T_ROW32 = (COUT + 4) * 4;        // MODE_BWD's epilogue image: 132 floats per cell
static_assert(TB * 64 * T_ROW32 <= LDS_BYTES, "float32 epilogue image must fit in the staging buffers");

__device__ __forceinline__ half8 lds_half8(const char *p)
{
    return *(const half8 *)p;
}

// MFMA operand fragments of one k-step (one tap, 16 input channels) of a wave
struct Frags {
    half8 a_hi[NI], a_lo[NI]; // weights, NI blocks of 32 output channels
    half8 b_hi[2], b_lo[2]; // activations, 2 blocks of 32 cells
};

// global data on its way to LDS
struct StagedW {
    u32x4 w[2 * WQ]; // hi + lo pieces of the weights of a stage
};
struct StagedX {
    u32x4 x[2 * XQ]; // hi + lo pieces of a channel block of the 4 boards
};

// ds_read_b128 serves the lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (and the same
// + 32) in separate LDS cycles.  Lane r of a 32-cell block holds board rows 0 and 2 of the
// block in the first group and rows 1 and 3 in the second: (row << 3) | column.
__device__ __forceinline__ int cell_of_lane(int r)
{
    return r < 4 ? r : r < 12 ? 8 + (r - 4) : r < 16 ? r - 8 : r < 20 ? 24 + (r - 16) : r < 28 ? 16 + (r - 20)
                                                                                            : 24 + (r - 24);
}

__device__ __forceinline__ void load_frags(Frags &F, const char *wb, const char *xb, int a_off, int b_off0,
                                           int b_off1, int kx)
{
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int off = a_off + (kx * COUT + 32 * i) * ROW;
        F.a_hi[i] = lds_half8(wb + off);
        F.a_lo[i] = lds_half8(wb + W_HALF + off);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int off = (j ? b_off1 : b_off0) + kx * ROW;
        F.b_hi[j] = lds_half8(xb + off);
        F.b_lo[j] = lds_half8(xb + X_HALF + off);
    }
}

__device__ __forceinline__ void mfma_step(const Frags &F, float16v (&acc_main)[NI][2],
                                          float16v (&acc_cross)[NI][2])
{
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            acc_main[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[i], F.b_hi[j], acc_main[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            acc_cross[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[i], F.b_lo[j], acc_cross[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
            acc_cross[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_lo[i], F.b_hi[j], acc_cross[i][j], 0, 0, 0);
}

extern __shared__ __align__(16) char conv_lds[]; // the dynamic LDS of the two kernels below

// One layer for the 4 boards of this workgroup (the whole kernel when launched per layer).
template <int MODE = MODE_FWD>
__device__ __forceinline__ void conv_layer(const ConvParams &P)
{
    char *const lds = conv_lds;
    char *const xbuf = lds;
    char *const wbuf = lds + 2 * X_BUF;

    const int tid = threadIdx.x;
    const int w = (tid >> 6) & (TB - 1), cs = tid >> 8, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int64_t b0 = (int64_t)blockIdx.x * TB;

    const float my_bias = MODE == MODE_FWD ? P.bias[tid & (COUT - 1)] : 0.0f; // parked in a register until the epilogue
    // zero both X buffers once: the border cells of the padded planes stay zero
    for (int i = tid; i < 2 * X_BUF / 16; i += THREADS)
        ((uint4 *)xbuf)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // ---- staging: the 16-byte pieces this thread moves per stage
    // X channel block: per hi/lo TB*64*2 = 512 pieces (board = e >> 7, cell = (e >> 1) & 63,
    // half-piece = e & 1); W stage: 3*128*2 = 768 pieces (kx, co, half-piece).  With 512
    // threads the upper half repeats its first W piece instead of branching.
    int64_t x_src[XQ];
    int x_dst[XQ];
#pragma unroll
    for (int q = 0; q < XQ; q++) {
        const int e = tid + q * THREADS;
        // boards past the end of a ragged batch read the last board (their results are not stored)
        const int64_t b = min(b0 + (e >> 7), P.n - 1);
        x_src[q] = b * P.n_chunks * 128 + (e & 127); // + chunk * 128, in 16-B pieces
        const int cell = (e >> 1) & 63;
        x_dst[q] = ((e >> 7) * XPP + xrow((cell >> 3) + 1) + (cell & 7) + 1) * ROW + (e & 1) * 16;
    }
    int w_piece[WQ], w_dst[WQ];
#pragma unroll
    for (int q = 0; q < WQ; q++) {
        const int e = (tid + q * THREADS < 768) ? tid + q * THREADS : tid;
        w_piece[q] = e;
        w_dst[q] = (e >> 1) * ROW + (e & 1) * 16;
    }
    const int n_chunks = P.n_chunks, n_stages = 3 * n_chunks;

    const u32x4 *const gw_hi = (const u32x4 *)P.w_hi, *const gw_lo = (const u32x4 *)P.w_lo;
    const u32x4 *const gx_hi = (const u32x4 *)P.x_hi, *const gx_lo = (const u32x4 *)P.x_lo;
    auto fetch_w = [=](int stage) {
        StagedW G;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            const int64_t src = (int64_t)stage * 768 + w_piece[q];
            G.w[q] = gw_hi[src];
            G.w[WQ + q] = gw_lo[src];
        }
        return G;
    };
    auto fetch_x = [=](int chunk) {
        StagedX G;
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            const int64_t src = x_src[q] + (int64_t)chunk * 128;
            G.x[q] = gx_hi[src];
            G.x[XQ + q] = gx_lo[src];
        }
        return G;
    };
    auto commit_w = [=](const StagedW &G, int wsel) {
        char *wb = wbuf + wsel * W_BUF;
#pragma unroll
        for (int q = 0; q < WQ; q++) {
            *(u32x4 *)(wb + w_dst[q]) = G.w[q];
            *(u32x4 *)(wb + W_HALF + w_dst[q]) = G.w[WQ + q];
        }
    };
    auto commit_x = [=](const StagedX &G, int xsel) {
        char *xb = xbuf + xsel * X_BUF;
#pragma unroll
        for (int q = 0; q < XQ; q++) {
            *(u32x4 *)(xb + x_dst[q]) = G.x[q];
            *(u32x4 *)(xb + X_HALF + x_dst[q]) = G.x[XQ + q];
        }
    };

    float16v acc_main[NI][2], acc_cross[NI][2];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                acc_main[i][j][v] = 0.0f;
                acc_cross[i][j][v] = 0.0f;
            }

    // lane-constant parts of the operand addresses
    const int a_off = (cs * 32 * NI + r) * ROW + h * 16;                 // + (kx*128 + 32i) * ROW
    const int lane_cell = cell_of_lane(r); // (row << 3) | column within a 32-cell block
    // B operand: padded cell (4j + row + ky, column + kx) of board w
    auto b_off = [=](int j, int ky) {
        return (w * XPP + xrow(4 * j + (lane_cell >> 3) + ky) + (lane_cell & 7)) * ROW + h * 16;
    };

    commit_w(fetch_w(0), 0);
    commit_x(fetch_x(0), 0);
    __syncthreads();

    // One stage = one kernel row (3 taps) x 16 input channels; its three k-steps alternate
    // between two fragment sets, the next k-step's LDS reads are issued before the
    // current one's MFMAs.  The data of stage s + 1 (its weights and the channel block of
    // the boards it reads) are fetched global -> registers at the start of stage s - 1 and
    // written registers -> LDS at the start of stage s, right AFTER the barrier that freed
    // the other buffers (so the LDS writes run beside this stage's MFMAs instead of in
    // front of a barrier), one register set; the third k-step's MFMAs run behind the
    // closing barrier together with the first fragment reads of the next stage.  Every
    // stage moves the same number of pieces (the channel block is re-written with
    // identical bytes while it is in use): the vmcnt bookkeeping stays exact.
    Frags F0, F1;
    StagedW GW = fetch_w(min(1, n_stages - 1)); // in flight: the data of the NEXT stage
    StagedX GX = fetch_x(0);
    load_frags(F0, wbuf, xbuf, a_off, b_off(0, 0), b_off(1, 0), 0);
    auto stage = [&](int s, Frags &Fa, Frags &Fb) {
        const int chunk = s / 3, ky = s - 3 * chunk; // Fa holds k-step 0 of stage s
        const int s1 = min(s + 1, n_stages - 1), s2 = min(s + 2, n_stages - 1);
        const int chunk1 = s1 / 3, ky1 = s1 - 3 * chunk1;
        const char *xb = xbuf + (chunk & 1) * X_BUF;
        const char *wb = wbuf + (s & 1) * W_BUF;
        const int bo0 = b_off(0, ky), bo1 = b_off(1, ky);
        // the barrier that ended stage s - 1 freed the other buffers: write the data of
        // stage s + 1 (fetched a stage ago) there and re-issue the loads for stage s + 2
        commit_w(GW, (s + 1) & 1);
        commit_x(GX, chunk1 & 1);
        GW = fetch_w(s2);
        GX = fetch_x(s2 / 3);
        load_frags(Fb, wb, xb, a_off, bo0, bo1, 1);
        mfma_step(Fa, acc_main, acc_cross);
        load_frags(Fa, wb, xb, a_off, bo0, bo1, 2);
        mfma_step(Fb, acc_main, acc_cross);
        __syncthreads();
        load_frags(Fb, wbuf + ((s + 1) & 1) * W_BUF, xbuf + (chunk1 & 1) * X_BUF, a_off, b_off(0, ky1),
                   b_off(1, ky1), 0);
        mfma_step(Fa, acc_main, acc_cross);
    };
    for (int s = 0; s < n_stages; s += 2) { // n_stages is even (cin a multiple of 32)
        stage(s, F0, F1);
        stage(s + 1, F1, F0);
    }
    __syncthreads();

    if (MODE == MODE_BWD) {
        // ---- epilogue of the backward-data form: scale, transpose through LDS as float32, mask, coalesced stores
        const float unscale = ldexpf(1.0f, -*P.scale_exp);
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int co = 32 * (NI * cs + i) + 8 * q + 4 * h;
                    f32x4 v4;
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        v4[t] = (acc_main[i][j][4 * q + t] + acc_cross[i][j][4 * q + t] * (1.0f / 2048.0f)) * unscale;
                    *(f32x4 *)(lds + (w * 64 + 32 * j + lane_cell) * T_ROW32 + co * 4) = v4;
                }
        __syncthreads();
        float big = 0.0f;
        const int per_board = P.out_blocks * 256; // 16-byte pieces of 4 channels
        // (eight pieces per thread at a time, their sixteen mask loads in flight together: one piece after the other --
        // a loop whose bound the compiler does not know -- every piece waited for its own two loads, 32 dependent round
        // trips per thread: the 36 us this kernel took longer than the forward form; LABNOTES.md, round 6)
        constexpr int EU = 8;
        for (int e0 = tid; e0 < TB * per_board; e0 += THREADS * EU) {
            uint2 mh[EU], ml[EU];
#pragma unroll
            for (int u = 0; u < EU; u++) {
                const int e = e0 + u * THREADS;
                const int board = e / per_board, f = e - board * per_board;
                const bool ok = e < TB * per_board && b0 + board < P.n;
                const int64_t at = ok ? (b0 + board) * per_board + f : 0;
                mh[u] = P.mask_hi[at];
                ml[u] = P.mask_lo[at];
            }
#pragma unroll
            for (int u = 0; u < EU; u++) {
                const int e = e0 + u * THREADS;
                const int board = e / per_board, f = e - board * per_board;
                const int cb = f >> 8, cell = (f >> 2) & 63, qt = f & 3;
                const int64_t b = b0 + board;
                if (e < TB * per_board && b < P.n) {
                    f32x4 v4 = *(const f32x4 *)(lds + (board * 64 + cell) * T_ROW32 + (cb * 16 + qt * 4) * 4);
                    const int64_t at = b * per_board + f;
                    const __half2 *h2 = (const __half2 *)&mh[u], *l2 = (const __half2 *)&ml[u];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float xh = __half2float(t & 1 ? h2[t >> 1].y : h2[t >> 1].x);
                        const float xl = __half2float(t & 1 ? l2[t >> 1].y : l2[t >> 1].x);
                        if (!(xh + xl * (1.0f / 2048.0f) > 0.0f)) // the ReLU of the block below was off (network.py:13)
                            v4[t] = 0.0f;
                        big = fmaxf(big, fabsf(v4[t]));
                    }
                    P.y_f32[at] = v4;
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            big = fmaxf(big, __shfl_xor(big, d));
        if (lane == 0 && big > 0.0f)
            atomicMax(P.max_bits, __float_as_uint(big));
        return;
    }
    // ---- epilogue: bias, ReLU, split, transpose through LDS, coalesced stores
    // D tile (i, j): lane holds cell 32j + cell_of_lane(r), channels 32i + 8(v>>2) + 4h + (v&3)
    char *const t_hi = lds, *const t_lo = lds + T_HALF;
    float *const bias_lds = (float *)(lds + 2 * T_HALF);
    if (tid < COUT)
        bias_lds[tid] = my_bias;
    __syncthreads();
    bool saturated = false;
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int co = 32 * (NI * cs + i) + 8 * q + 4 * h;
                __half hi4[4], lo4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    float v = acc_main[i][j][4 * q + t] + acc_cross[i][j][4 * q + t] * (1.0f / 2048.0f) +
                              bias_lds[co + t];
                    saturated |= !(v <= 65000.0f); // beyond the f16 range, or NaN (fmaxf would hide it)
                    v = fminf(fmaxf(v, 0.0f), 65000.0f);
                    const __half vh = __float2half_rn(v);
                    hi4[t] = vh;
                    lo4[t] = __float2half_rn((v - __half2float(vh)) * 2048.0f);
                }
                const int off = (w * 64 + 32 * j + lane_cell) * T_ROW + co * 2;
                *(uint2 *)(t_hi + off) = *(const uint2 *)hi4;
                *(uint2 *)(t_lo + off) = *(const uint2 *)lo4;
            }
    // (boards past the end of a ragged batch recompute the last real board: no false alarm)
    if (P.overflow && saturated)
        *P.overflow = 1u;
    __syncthreads();
    // output pieces: per hi/lo TB * 8 blocks * 64 cells * 2 = 4096
#pragma unroll 4
    for (int q = 0; q < 4096 / THREADS; q++) {
        const int e = tid + q * THREADS; // (board, block, cell, half-piece) in output order
        const int board = e >> 10, cb = (e >> 7) & 7, cell = (e >> 1) & 63, hp = e & 1;
        const int64_t b = b0 + board;
        if (b < P.n) {
            const int off = (board * 64 + cell) * T_ROW + (cb * 16 + hp * 8) * 2;
            const int64_t dst = b * 1024 + (e & 1023);
            P.y_hi[dst] = *(const uint4 *)(t_hi + off);
            P.y_lo[dst] = *(const uint4 *)(t_lo + off);
        }
    }
}

__global__ __launch_bounds__(THREADS) void conv3x3_split_kernel(ConvParams P)
{
    conv_layer(P);
}

__global__ __launch_bounds__(THREADS) void conv3x3_bwd_data_kernel(ConvParams P)
{
    conv_layer<MODE_BWD>(P);
}

// float32 channel blocks -> split channel blocks times 2^e, e from the tensor's largest magnitude (13 - its exponent:
// the largest element lands in [2^13, 2^14)); thread 0 publishes e.  A thread = 8 channels of a cell, a workgroup =
// two (board, channel block) pairs; bias_part (optional) [workgroups][2][16]: the sums over the 64 cells of each pair
// -- the bias gradient of the block, one more pass over these bytes saved (fixed order: deterministic)
__global__ __launch_bounds__(256) void split_scaled_kernel(const f32x4 *x, const uint32_t *max_bits, uint4 *hi, uint4 *lo,
                                                           int32_t *scale_exp, int64_t pieces, float *bias_part)
{
    __shared__ float wave_sum[4][16];
    const uint32_t mb = *max_bits;
    const int e = mb == 0u ? 0 : 13 - ((int)(mb >> 23) - 127);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0)
        *scale_exp = e;
    float raw[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        raw[k] = 0.0f;
    if (t < pieces) {
        const f32x4 a = x[2 * t], b = x[2 * t + 1];
        __half h8[8], l8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            raw[k] = k < 4 ? a[k] : b[k - 4];
            const float v = ldexpf(raw[k], e);
            const __half vh = __float2half_rn(v);
            h8[k] = vh;
            l8[k] = __float2half_rn((v - __half2float(vh)) * 2048.0f);
        }
        hi[t] = *(const uint4 *)h8;
        lo[t] = *(const uint4 *)l8;
    }
    if (!bias_part)
        return;
    // lanes of one parity hold the same 8 channels of 32 cells; wave w = cells 32 (w & 1) .. + 31 of pair w >> 1
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int d = 2; d <= 32; d <<= 1)
            raw[k] += __shfl_xor(raw[k], d);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 2)
#pragma unroll
        for (int k = 0; k < 8; k++)
            wave_sum[wv][8 * lane + k] = raw[k];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int pr = threadIdx.x >> 4, c = threadIdx.x & 15;
        bias_part[(int64_t)blockIdx.x * 32 + threadIdx.x] = wave_sum[2 * pr][c] + wave_sum[2 * pr + 1][c];
    }
}

// bias gradient: channel ch of `channels` = sum over the boards of the pair sums above; one wave per channel, lane l
// takes boards l, l + 64, ..
__global__ __launch_bounds__(64) void bias_reduce_kernel(const float *bias_part, int64_t n, int channels, float *db)
{
    const int ch = blockIdx.x, ncb = channels >> 4, cb = ch >> 4, c = ch & 15, lane = threadIdx.x;
    float s = 0.0f;
    for (int64_t b = lane; b < n; b += 64) {
        const int64_t pair = b * ncb + cb;
        s += bias_part[(pair >> 1) * 32 + (pair & 1) * 16 + c];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        s += __shfl_xor(s, d);
    if (lane == 0)
        db[ch] = s;
}

// Up to 8 consecutive layers in one launch.  A workgroup owns all 128 channels of its 4
// boards, so layer k + 1 of a workgroup depends on nothing but its own output of layer k:
// the activations take the usual round trip through global memory (L2), but without
// kernel boundaries in between -- no launch gaps, and the output stores of a layer overlap
// the first loads of the next.  Every layer writes its own buffer (an address is written
// once, by this workgroup, before it is read once: no stale L1 lines).  The layer loop is
// unrolled at compile time: indexing the by-value parameter array with a run-time layer
// number sends it (and every pointer in it) through scratch memory.
constexpr int MAX_TRUNK = 8;
struct TrunkParams {
    ConvParams layer[MAX_TRUNK];
    int32_t n_layers;
};

__global__ __launch_bounds__(THREADS) void conv3x3_split_trunk_kernel(TrunkParams T)
{
#pragma unroll
    for (int L = 0; L < MAX_TRUNK; L++) {
        if (L < T.n_layers) {
            conv_layer(T.layer[L]);
            // the next layer of THIS workgroup reads what it just stored: workgroup scope
            // (an agent-scope fence would write the whole L2 back, ~50 us per layer)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

// float32 NCHW planes -> split channel blocks
__global__ __launch_bounds__(256) void split_nchw_kernel(const float *x, __half *hi, __half *lo, int64_t n,
                                                         int channels, uint32_t *overflow)
{
    // one thread per (board, block, cell, 8 channels): reads 8 floats (stride 64), writes 16 B + 16 B
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nb = channels / 16;
    const int64_t total = n * nb * 128;
    if (t >= total)
        return;
    const int hp = (int)(t & 1), cell = (int)((t >> 1) & 63);
    const int64_t bb = t >> 7; // board * nb + block
    const int64_t b = bb / nb;
    const int cb = (int)(bb - b * nb);
    const float *src = x + (b * channels + cb * 16 + hp * 8) * 64 + cell;
    __half h8[8], l8[8];
    bool saturated = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float raw = src[k * 64];
        saturated |= !(fabsf(raw) <= 65000.0f);
        const float v = fminf(fmaxf(raw, -65000.0f), 65000.0f);
        const __half vh = __float2half_rn(v);
        h8[k] = vh;
        l8[k] = __float2half_rn((v - __half2float(vh)) * 2048.0f);
    }
    if (overflow && saturated)
        *overflow = 1u;
    ((uint4 *)hi)[t] = *(const uint4 *)h8;
    ((uint4 *)lo)[t] = *(const uint4 *)l8;
}

// split channel blocks -> float32 NCHW planes
__global__ __launch_bounds__(256) void merge_nchw_kernel(const __half *hi, const __half *lo, float *y, int64_t n,
                                                         int channels)
{
    // one thread per (board, channel, cell): coalesced writes, 2-byte gathers (L2-resident)
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = n * channels * 64;
    if (t >= total)
        return;
    const int cell = (int)(t & 63);
    const int64_t bc = t >> 6;
    const int64_t b = bc / channels;
    const int c = (int)(bc - b * channels);
    const int64_t src = ((b * (channels / 16) + (c >> 4)) * 64 + cell) * 16 + (c & 15);
    y[t] = __half2float(hi[src]) + __half2float(lo[src]) * (1.0f / 2048.0f);
}

// ---- Value stem: block1 = 3x3 convolution 2 -> 64 + bias + ReLU (network.py:66-70),
// float32 planes in, split channel blocks out.  K = 18: plain float32 FMAs.  One thread
// per (board, channel block, half block, cell): 8 output channels of a cell.
// planes: float32 [n][2][8][8], or NULL with the boards themselves in own / opp (plane 0 =
// opponent of the side to move, plane 1 = side to move, game.py:168-174)
__global__ __launch_bounds__(256) void value_stem_kernel(const float *planes, const uint64_t *own,
                                                         const uint64_t *opp, const float *w, const float *bias,
                                                         uint4 *y_hi, uint4 *y_lo, int64_t n, uint32_t *overflow)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * 512)
        return;
    const int cell = (int)(t & 63), grp = (int)((t >> 6) & 7); // grp = channel block * 2 + half: wave-uniform
    const int64_t b = t >> 9;
    const int y = cell >> 3, x = cell & 7;
    const float *pl = planes + b * 128;
    const uint64_t bits0 = planes ? 0ull : opp[b], bits1 = planes ? 0ull : own[b];
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
                float v;
                if (planes)
                    v = ok ? pl[c * 64 + a] : 0.0f;
                else
                    v = (ok && (((c ? bits1 : bits0) >> a) & 1ull)) ? 1.0f : 0.0f;
                in[c * 9 + ky * 3 + kx] = v;
            }
    const int co0 = __builtin_amdgcn_readfirstlane(grp) * 8;
    __half h8[8], l8[8];
    bool saturated = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float *wk = w + (co0 + k) * 18; // [co][ci][ky][kx]
        float acc = bias[co0 + k];
#pragma unroll
        for (int j = 0; j < 18; j++)
            acc = fmaf(wk[j], in[j], acc);
        saturated |= !(acc <= 65000.0f);
        const float v = fminf(fmaxf(acc, 0.0f), 65000.0f);
        const __half vh = __float2half_rn(v);
        h8[k] = vh;
        l8[k] = __float2half_rn((v - __half2float(vh)) * 2048.0f);
    }
    if (overflow && saturated)
        *overflow = 1u;
    const int64_t dst = ((b * 4 + (grp >> 1)) * 64 + cell) * 2 + (grp & 1);
    y_hi[dst] = *(const uint4 *)h8;
    y_lo[dst] = *(const uint4 *)l8;
}

// ---- Value head: block9 = 3x3 convolution 128 -> 1 + bias + ReLU, fc10 (64 -> 128, no
// bias), fc11 (128 -> 1, no bias) (network.py:78-96 with train=False), split channel
// blocks in, one float per board out.  One workgroup per board: wave q sums channel
// blocks 2q and 2q+1, lane = cell; float32 FMAs on the exact values hi + lo * 2^-11.
__device__ __forceinline__ void fma16(float &acc, const float *wrow, uint4 a, uint4 b)
{
    const __half2 *pa = (const __half2 *)&a, *pb = (const __half2 *)&b;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float2 fa = __half22float2(pa[k]), fb = __half22float2(pb[k]);
        acc = fmaf(wrow[2 * k], fa.x, acc);
        acc = fmaf(wrow[2 * k + 1], fa.y, acc);
        acc = fmaf(wrow[8 + 2 * k], fb.x, acc);
        acc = fmaf(wrow[8 + 2 * k + 1], fb.y, acc);
    }
}

__global__ __launch_bounds__(256) void value_head_kernel(const uint4 *x_hi, const uint4 *x_lo, const float *w9,
                                                         const float *b9, const float *w10, const float *w11,
                                                         float *out, int64_t n)
{
    // padded 10x10 plane of 16-channel rows per block, hi and lo: 8 * 100 * 32 B each
    constexpr int IMG = 8 * PP * 32;
    __shared__ __align__(16) char img[2 * IMG];
    __shared__ __align__(16) float w9s[9 * 128]; // [tap][channel]
    __shared__ float part[4 * 64];
    __shared__ float h9[64];
    __shared__ float hid[128];
    const int tid = threadIdx.x, q = tid >> 6, lane = tid & 63;
    const int64_t b = blockIdx.x;
    for (int i = tid; i < 2 * IMG / 16; i += 256)
        ((uint4 *)img)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 9 * 128; i += 256) {
        const int tap = i >> 7, c = i & 127;
        w9s[i] = w9[c * 9 + tap]; // [1][128][3][3]
    }
    __syncthreads();
    // 1024 pieces per hi/lo: (block, cell, half)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int e = tid + k * 256;
        const int cb = e >> 7, cell = (e >> 1) & 63, hp = e & 1;
        const int off = ((cb * PP + ((cell >> 3) + 1) * 10 + (cell & 7) + 1) * 2 + hp) * 16;
        *(uint4 *)(img + off) = x_hi[b * 1024 + e];
        *(uint4 *)(img + IMG + off) = x_lo[b * 1024 + e];
    }
    __syncthreads();
    const int y = lane >> 3, x = lane & 7;
    float acc_hi = 0.0f, acc_lo = 0.0f;
#pragma unroll
    for (int cbi = 0; cbi < 2; cbi++) {
        const int cb = 2 * q + cbi;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int pp = (y + tap / 3) * 10 + x + tap % 3;
            const char *ph = img + (cb * PP + pp) * 32;
            const float *wrow = w9s + tap * 128 + cb * 16;
            fma16(acc_hi, wrow, *(const uint4 *)ph, *(const uint4 *)(ph + 16));
            fma16(acc_lo, wrow, *(const uint4 *)(ph + IMG), *(const uint4 *)(ph + IMG + 16));
        }
    }
    part[q * 64 + lane] = acc_hi + acc_lo * (1.0f / 2048.0f);
    __syncthreads();
    if (tid < 64)
        h9[tid] = fmaxf(((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) + b9[0], 0.0f);
    __syncthreads();
    if (tid < 128) { // fc10 row tid, then its fc11 term
        const float4 *row = (const float4 *)(w10 + tid * 64);
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const float4 wv = row[c];
            s = fmaf(wv.x, h9[4 * c], s);
            s = fmaf(wv.y, h9[4 * c + 1], s);
            s = fmaf(wv.z, h9[4 * c + 2], s);
            s = fmaf(wv.w, h9[4 * c + 3], s);
        }
        hid[tid] = s * w11[tid];
    }
    __syncthreads();
    if (tid == 0) {
        float v = 0.0f;
        for (int j = 0; j < 128; j++) // fixed order: the result does not depend on the launch shape
            v += hid[j];
        out[b] = v;
    }
}

// ---- float32 convolution for SMALL batches (the policy net on the expansions of a
// playout, MCTS.py:109-121: a few dozen boards per call).  Exact float32 products on
// v_mfma_f32_32x32x2_f32 (which runs at the VALU rate: 9.4 M MACs of one board and layer
// are 74 k cycles of one CU), so a board is spread over 4 workgroups (32 output channels
// each) and the K loop over the 4 waves of a workgroup (a quarter of the input channels
// each, all 9 taps); partial sums meet in LDS.  float32 NCHW in and out, bias + ReLU fused.
// LDS holds the padded board only (51 KB at 128 channels: three workgroups per CU); the weights
// go from L2 straight into registers, one tap ahead of their MFMAs.  The kernel runs at the
// float32 MFMA rate: 35 us per layer MFMA-bound at 256 boards, 40 measured.
constexpr int F32_CO = 32;          // output channels per workgroup
constexpr int F32_XPLANE = PP;      // padded plane, floats

struct ConvF32Params {
    const float *x;    // [n][cin][64]
    const float4 *w;   // [4][9][cin][32]
    const float *bias; // [128]
    float *y;          // [n][128][64]
    int32_t cin;
    int64_t n;            // boards (upper bound when n_dev is given)
    const int32_t *n_dev; // optional device word: only the first min(n, *n_dev) boards
};

// NJ = 2: a workgroup covers all 64 cells of its board (two 32-cell MFMA tiles per wave);
// NJ = 1: half a board (rows 0-3 or 4-7), eight workgroups per board -- for batches so
// small that four per board leave CUs idle.
template <int CIN, int NJ>
__device__ __forceinline__ void conv3x3_f32_items(const ConvF32Params &P, int64_t n_eff)
{
    constexpr int CQ = CIN / 4;              // input channels per wave
    extern __shared__ __align__(16) char lds[];
    float *const xs = (float *)lds;                       // [CIN][100]
    const int tid = threadIdx.x, q = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // Work items = (board, channel group[, board half]); a workgroup walks them with the
    // grid's stride.  With a host-side count the grid has one workgroup per item; with a
    // device-side count (n_dev: a hipGraph replays the launch without the host knowing how
    // many leaves expand) the grid is fixed and every workgroup leaves the loop as soon as
    // its next item is past the count.
    const int64_t n_items = n_eff * (NJ == 2 ? 4 : 8);
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t wg = NJ == 2 ? item : item >> 1;
    const int j0 = NJ == 2 ? 0 : (int)(item & 1); // first 32-cell tile of this workgroup
    const int64_t b = wg >> 2;
    const int cg = (int)(wg & 3);

    // The A operand of MFMA t of a tap is ONE float per lane, W[tap][ci = q CQ + 2 t + h][co = r]:
    // 256 contiguous bytes per wave and MFMA.  It comes straight from L2 into registers, a whole
    // tap (CQ / 2 values per lane) ahead of its use -- no weight slab in LDS: the workgroup's
    // footprint is the board alone (51 KB at 128 channels), three workgroups share a CU and hide
    // each other's staging and reduction phases.
    const float *wsrc = (const float *)P.w + (((int64_t)(cg * 9) * CIN + q * CQ + h) * F32_CO + r);
    constexpr int TAPF = CIN * F32_CO; // floats between two taps
    float a_reg[2][CQ / 2];
#pragma unroll
    for (int t = 0; t < CQ / 2; t++)
        a_reg[0][t] = wsrc[2 * t * F32_CO];
    constexpr int XPIECES = CIN * 16 / 256; // float4 pieces of the board per thread
    const f32x4 *xsrc = (const f32x4 *)(P.x + b * CIN * 64);
    f32x4 xv[XPIECES];
#pragma unroll
    for (int k = 0; k < XPIECES; k++)
        xv[k] = xsrc[tid + k * 256];

    // ---- stage the board: zero the borders, copy the 8x8 interiors
    for (int i = tid; i < CIN * 36; i += 256) {
        const int c = i / 36, e = i - c * 36;
        // border cells of a 10x10 plane: rows 0 and 9 (20 cells), columns 0 and 9 of rows 1..8
        const int pp = e < 10 ? e : e < 20 ? 80 + e : (e - 20) < 8 ? (e - 19) * 10 : (e - 27) * 10 + 9;
        xs[c * F32_XPLANE + pp] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < XPIECES; k++) {
        const int i = tid + k * 256;
        const int c = i >> 4, cell = (i & 15) * 4;
        float *d = xs + c * F32_XPLANE + ((cell >> 3) + 1) * 10 + (cell & 7) + 1;
        d[0] = xv[k].x;
        d[1] = xv[k].y;
        d[2] = xv[k].z;
        d[3] = xv[k].w;
    }
    __syncthreads();

    float16v acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int v = 0; v < 16; v++)
            acc[j][v] = 0.0f;

    // lane-constant operand offsets: A = W[k = 2t + h][co = r], B = X[ci = .. + 2t + h][cell]
    const float *const xq = xs + (q * CQ + h) * F32_XPLANE + (4 * j0 + (r >> 3)) * 10 + (r & 7);
    // Tap t takes its weights from register set t & 1; the loads of tap t + 1 are issued at its
    // start: they have a whole tap of MFMAs (CQ / 2 x 64 cycles) to arrive.
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        const int buf = tap & 1;
        if (tap + 1 < 9) {
#pragma unroll
            for (int t = 0; t < CQ / 2; t++)
                a_reg[buf ^ 1][t] = wsrc[(tap + 1) * TAPF + 2 * t * F32_CO];
        }
        __builtin_amdgcn_sched_barrier(0); // the loads stay HERE, ahead of this tap's MFMAs
        const float *xb = xq + (tap / 3) * 10 + tap % 3;
#pragma unroll
        for (int t = 0; t < CQ / 2; t++) {
            const float a = a_reg[buf][t];
#pragma unroll
            for (int j = 0; j < NJ; j++)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[2 * t * F32_XPLANE + 40 * j], acc[j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- the four K-quarters meet in LDS: red[q][j][v][lane]
    float *const red = (float *)lds;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int v = 0; v < 16; v++)
            red[((q * 2 + j) * 16 + v) * 64 + lane] = acc[j][v];
    __syncthreads();
    // wave q finishes registers v = 4q .. 4q+3 of both tiles:
    // channel 32 cg + 8 (v >> 2) + 4 h + (v & 3) = 32 cg + 8 q + 4 h + t, cell 32 j + r
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int v = 4 * q + t;
            float s = 0.0f;
#pragma unroll
            for (int qq = 0; qq < 4; qq++)
                s += red[((qq * 2 + j) * 16 + v) * 64 + lane];
            const int co = 32 * cg + 8 * q + 4 * h + t;
            P.y[(b * COUT + co) * 64 + 32 * (j0 + j) + r] = fmaxf(s + P.bias[co], 0.0f);
        }
    __syncthreads(); // the next item re-stages the LDS image the reduction just read
    } // item loop
}

template <int CIN, int NJ>
__global__ __launch_bounds__(256) void conv3x3_f32_kernel(ConvF32Params P)
{
    conv3x3_f32_items<CIN, NJ>(P, P.n);
}

// Device-side count: the split (half boards up to 32 boards, whole boards above) is chosen in
// the kernel, where the count is known.
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_f32_counted_kernel(ConvF32Params P)
{
    const int64_t n_eff = min(P.n, (int64_t)*P.n_dev);
    if (n_eff <= 32)
        conv3x3_f32_items<CIN, 1>(P, n_eff);
    else
        conv3x3_f32_items<CIN, 2>(P, n_eff);
}

// float32 stem: conv3x3 2 -> 64 + bias + ReLU to float32 NCHW (SLPolicy.block1,
// network.py:17-19); one thread per (board, channel, cell)
__global__ __launch_bounds__(256) void stem_f32_kernel(const float *planes, const uint64_t *own,
                                                       const uint64_t *opp, const int64_t *index, const float *w,
                                                       const float *bias, float *y, int64_t n, const int32_t *n_dev)
{
    // grid-stride over (board, channel, cell): with a device-side count the grid is capped
    // and most of it leaves at once.  planes == NULL: the boards themselves (row b = board
    // index[b]; plane 0 = opponent of the side to move, plane 1 = side to move)
    const int64_t total = (n_dev ? min(n, (int64_t)*n_dev) : n) * 4096;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int cell = (int)(t & 63), co = __builtin_amdgcn_readfirstlane((int)((t >> 6) & 63));
    const int64_t b = t >> 12;
    const int yy0 = cell >> 3, xx0 = cell & 7;
    const float *pl = planes + b * 128;
    uint64_t bits[2] = {0ull, 0ull};
    if (!planes) {
        const int64_t src = index ? index[b] : b;
        bits[0] = opp[src];
        bits[1] = own[src];
    }
    const float *wk = w + co * 18;
    float acc = bias[co];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int yy = yy0 + ky - 1, xx = xx0 + kx - 1;
                const bool ok = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
                const int a = (yy * 8 + xx) & 63;
                float v;
                if (planes)
                    v = ok ? pl[c * 64 + a] : 0.0f;
                else
                    v = (ok && ((bits[c] >> a) & 1ull)) ? 1.0f : 0.0f;
                acc = fmaf(wk[c * 9 + ky * 3 + kx], v, acc);
            }
    y[t] = fmaxf(acc, 0.0f);
    }
}

// SLPolicy head: conv9 (1x1, 128 -> 1, no bias), bias10 (64), softmax (network.py:29-47);
// one wave per board, lane = cell
__global__ __launch_bounds__(64) void policy_head_kernel(const float *x, const float *w9, const float *b10,
                                                         float *probs, int64_t n, const int32_t *n_dev)
{
    const int64_t b = blockIdx.x;
    if (n_dev && b >= (int64_t)*n_dev)
        return;
    const int lane = threadIdx.x;
    const float *xb = x + b * COUT * 64 + lane;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < COUT; c++)
        s = fmaf(w9[c], xb[c * 64], s);
    s += b10[lane];
    float m = s;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        m = fmaxf(m, __shfl_xor(m, d, 64));
    const float e = expf(s - m);
    float z = e;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        z += __shfl_xor(z, d, 64);
    probs[b * 64 + lane] = e / z;
}

} // namespace

// conv_trunk_kernel.hip: the layers in one launch with the activations resident in LDS
int iago_launch_trunk_resident(const iago_conv_split_layer *layers, int32_t n_layers, int64_t n, uint32_t *overflow,
                               void *stream);

extern "C" {

int iago_conv3x3_split(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo, const float *bias,
                       void *y_hi, void *y_lo, int64_t n, int32_t cin, int32_t cout, uint32_t *overflow,
                       void *stream)
{
    if (n < 0 || cout != COUT || cin <= 0 || (cin % 32) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split: cout must be 128 and cin a multiple of 32");
    if (n == 0)
        return IAGO_OK;
    if (!x_hi || !x_lo || !w_hi || !w_lo || !bias || !y_hi || !y_lo)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split: null pointer");
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)conv3x3_split_kernel, LDS_BYTES, configured,
                         "iago_conv3x3_split: cannot reserve 159 KB of LDS"))
        return IAGO_ERR_HIP;
    ConvParams P = {};
    P.x_hi = (const uint4 *)x_hi;
    P.x_lo = (const uint4 *)x_lo;
    P.w_hi = (const uint4 *)w_hi;
    P.w_lo = (const uint4 *)w_lo;
    P.bias = bias;
    P.y_hi = (uint4 *)y_hi;
    P.y_lo = (uint4 *)y_lo;
    P.n = n;
    P.n_chunks = cin / 16;
    P.overflow = overflow;
    const unsigned grid = (unsigned)((n + TB - 1) / TB);
    hipLaunchKernelGGL(conv3x3_split_kernel, dim3(grid), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, P);
    return iago_check_launch("iago_conv3x3_split");
}

int iago_conv3x3_bwd_data_split(const void *dy_hi, const void *dy_lo, const int32_t *scale_exp, const void *wt_hi,
                                const void *wt_lo, const void *mask_hi, const void *mask_lo, int32_t out_channels,
                                float *dx, uint32_t *max_bits, int64_t n, void *stream)
{
    if (n < 0 || (out_channels != 64 && out_channels != 128))
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_bwd_data_split: out_channels must be 64 or 128");
    if (n == 0)
        return IAGO_OK;
    if (!dy_hi || !dy_lo || !scale_exp || !wt_hi || !wt_lo || !mask_hi || !mask_lo || !dx || !max_bits)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_bwd_data_split: null pointer");
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)conv3x3_bwd_data_kernel, LDS_BYTES, configured,
                         "iago_conv3x3_bwd_data_split: cannot reserve 159 KB of LDS"))
        return IAGO_ERR_HIP;
    ConvParams P = {};
    P.x_hi = (const uint4 *)dy_hi;
    P.x_lo = (const uint4 *)dy_lo;
    P.w_hi = (const uint4 *)wt_hi;
    P.w_lo = (const uint4 *)wt_lo;
    P.n = n;
    P.n_chunks = 8;
    P.scale_exp = scale_exp;
    P.mask_hi = (const uint2 *)mask_hi;
    P.mask_lo = (const uint2 *)mask_lo;
    P.y_f32 = (f32x4 *)dx;
    P.max_bits = max_bits;
    P.out_blocks = out_channels / 16;
    const unsigned grid = (unsigned)((n + TB - 1) / TB);
    hipLaunchKernelGGL(conv3x3_bwd_data_kernel, dim3(grid), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, P);
    return iago_check_launch("iago_conv3x3_bwd_data_split");
}

int iago_split_scaled(const float *x, const uint32_t *max_bits, void *hi, void *lo, int32_t *scale_exp, int64_t n,
                      int32_t channels, float *bias_part, float *bias_grad, void *stream)
{
    if (n < 0 || channels <= 0 || (channels % 16) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_split_scaled: channels must be a multiple of 16");
    if (!x || !max_bits || !hi || !lo || !scale_exp)
        return iago_fail(IAGO_ERR_INVALID, "iago_split_scaled: null pointer");
    const int64_t pieces = n * (channels / 16) * 128;
    const int64_t blocks = pieces ? (pieces + 255) / 256 : 1;
    if (bias_grad && !bias_part)
        return iago_fail(IAGO_ERR_INVALID, "iago_split_scaled: bias_grad needs bias_part");
    hipLaunchKernelGGL(split_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4 *)x, max_bits, (uint4 *)hi, (uint4 *)lo, scale_exp, pieces, bias_part);
    // (bias_part alone: the partial sums only -- iago_policy_reinforce_grad reduces those of all blocks in one launch)
    if (bias_part && bias_grad && n > 0)
        hipLaunchKernelGGL(bias_reduce_kernel, dim3((unsigned)channels), dim3(64), 0, (hipStream_t)stream,
                           (const float *)bias_part, n, channels, bias_grad);
    return iago_check_launch("iago_split_scaled");
}

int iago_conv3x3_split_trunk(const iago_conv_split_layer *layers, int32_t n_layers, int64_t n,
                             uint32_t *overflow, void *stream)
{
    if (n < 0 || n_layers < 1 || n_layers > MAX_TRUNK || !layers)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split_trunk: 1..8 layers expected");
    if (n == 0)
        return IAGO_OK;
    TrunkParams T;
    T.n_layers = n_layers;
    for (int L = 0; L < n_layers; L++) {
        const iago_conv_split_layer &a = layers[L];
        if (a.cin <= 0 || (a.cin % 32) != 0 || !a.x_hi || !a.x_lo || !a.w_hi || !a.w_lo || !a.bias || !a.y_hi ||
            !a.y_lo)
            return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split_trunk: bad layer (cin a multiple of 32, no nulls)");
        if (L > 0 && (a.x_hi != layers[L - 1].y_hi || a.x_lo != layers[L - 1].y_lo || a.cin != COUT))
            return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split_trunk: layer k must read the output of layer k-1");
        for (int M = 0; M < L; M++)
            if (a.y_hi == layers[M].y_hi || a.y_lo == layers[M].y_lo || a.y_hi == layers[M].x_hi)
                return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split_trunk: every layer needs its own output buffer");
        ConvParams &P = T.layer[L];
        P.x_hi = (const uint4 *)a.x_hi;
        P.x_lo = (const uint4 *)a.x_lo;
        P.w_hi = (const uint4 *)a.w_hi;
        P.w_lo = (const uint4 *)a.w_lo;
        P.bias = a.bias;
        P.y_hi = (uint4 *)a.y_hi;
        P.y_lo = (uint4 *)a.y_lo;
        P.n = n;
        P.n_chunks = a.cin / 16;
        P.overflow = overflow;
    }
    if (!getenv("IAGO_TRUNK_STAGED"))
        return iago_launch_trunk_resident(layers, n_layers, n, overflow, stream);
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)conv3x3_split_trunk_kernel, LDS_BYTES, configured,
                         "iago_conv3x3_split_trunk: cannot reserve 159 KB of LDS"))
        return IAGO_ERR_HIP;
    const unsigned grid = (unsigned)((n + TB - 1) / TB);
    hipLaunchKernelGGL(conv3x3_split_trunk_kernel, dim3(grid), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, T);
    return iago_check_launch("iago_conv3x3_split_trunk");
}

int iago_value_stem(const float *planes, const float *w1, const float *b1, void *y_hi, void *y_lo, int64_t n,
                    uint32_t *overflow, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_stem: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!planes || !w1 || !b1 || !y_hi || !y_lo)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_stem: null pointer");
    hipLaunchKernelGGL(value_stem_kernel, dim3((unsigned)((n * 512 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, planes, (const uint64_t *)nullptr, (const uint64_t *)nullptr, w1, b1,
                       (uint4 *)y_hi, (uint4 *)y_lo, n, overflow);
    return iago_check_launch("iago_value_stem");
}

int iago_value_stem_boards(const uint64_t *own, const uint64_t *opp, const float *w1, const float *b1, void *y_hi,
                           void *y_lo, int64_t n, uint32_t *overflow, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_stem_boards: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!own || !opp || !w1 || !b1 || !y_hi || !y_lo)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_stem_boards: null pointer");
    hipLaunchKernelGGL(value_stem_kernel, dim3((unsigned)((n * 512 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float *)nullptr, own, opp, w1, b1, (uint4 *)y_hi,
                       (uint4 *)y_lo, n, overflow);
    return iago_check_launch("iago_value_stem_boards");
}

int iago_value_head(const void *x_hi, const void *x_lo, const float *w9, const float *b9, const float *w10,
                    const float *w11, float *out, int64_t n, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_head: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!x_hi || !x_lo || !w9 || !b9 || !w10 || !w11 || !out)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_head: null pointer");
    hipLaunchKernelGGL(value_head_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       (const uint4 *)x_hi, (const uint4 *)x_lo, w9, b9, w10, w11, out, n);
    return iago_check_launch("iago_value_head");
}

int iago_conv3x3_f32(const float *x, const float *w, const float *bias, float *y, int64_t n, int32_t cin,
                     int32_t cout, const int32_t *n_dev, void *stream)
{
    if (n < 0 || cout != COUT || (cin != 64 && cin != 128))
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_f32: cout must be 128 and cin 64 or 128");
    if (n == 0)
        return IAGO_OK;
    if (!x || !w || !bias || !y)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_f32: null pointer");
    if (n > (1 << 28))
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_f32: batch too large");
    ConvF32Params P;
    P.x = x;
    P.w = (const float4 *)w;
    P.bias = bias;
    P.y = y;
    P.cin = cin;
    P.n = n;
    P.n_dev = n_dev;
    // the padded board planes; the K-quarters' reduction buffer ([4][2][16][64] floats) overlays them
    const size_t red = (size_t)4 * 2 * 16 * 64 * sizeof(float);
    const size_t planes_b = (size_t)cin * F32_XPLANE * sizeof(float);
    const size_t lds = planes_b > red ? planes_b : red;
    const int lds128 = (int)((size_t)128 * F32_XPLANE * sizeof(float) > red ? (size_t)128 * F32_XPLANE * sizeof(float) : red);
    static std::atomic<uint64_t> configured2{0}, configured1{0}, configuredc{0};
    if (iago_reserve_lds((const void *)conv3x3_f32_kernel<128, 2>, lds128, configured2,
                         "iago_conv3x3_f32: cannot reserve LDS") ||
        iago_reserve_lds((const void *)conv3x3_f32_kernel<128, 1>, lds128, configured1,
                         "iago_conv3x3_f32: cannot reserve LDS") ||
        iago_reserve_lds((const void *)conv3x3_f32_counted_kernel<128>, lds128, configuredc,
                         "iago_conv3x3_f32: cannot reserve LDS"))
        return IAGO_ERR_HIP;
    // up to 32 boards: eight workgroups per board (half a board each) fill the CUs.
    // Device-side count: a fixed grid of at most 1024 workgroups walks the items (the usual
    // count is a few dozen boards, the rest of the grid exits at once).
    const bool half = n <= 32;
    const int64_t items = n * (half ? 8 : 4);
    if (n_dev) {
        // grid for the worst case of either split: 8 items per board up to 32 boards
        const int64_t cap = n <= 32 ? n * 8 : (n * 4 < 256 ? 256 : n * 4);
        const dim3 cgrid((unsigned)(cap < 1024 ? cap : 1024));
        if (cin == 128)
            hipLaunchKernelGGL((conv3x3_f32_counted_kernel<128>), cgrid, dim3(256), lds, (hipStream_t)stream, P);
        else
            hipLaunchKernelGGL((conv3x3_f32_counted_kernel<64>), cgrid, dim3(256), lds, (hipStream_t)stream, P);
        return iago_check_launch("iago_conv3x3_f32");
    }
    const dim3 grid((unsigned)items);
    if (cin == 128 && half)
        hipLaunchKernelGGL((conv3x3_f32_kernel<128, 1>), grid, dim3(256), lds, (hipStream_t)stream, P);
    else if (cin == 128)
        hipLaunchKernelGGL((conv3x3_f32_kernel<128, 2>), grid, dim3(256), lds, (hipStream_t)stream, P);
    else if (half)
        hipLaunchKernelGGL((conv3x3_f32_kernel<64, 1>), grid, dim3(256), lds, (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL((conv3x3_f32_kernel<64, 2>), grid, dim3(256), lds, (hipStream_t)stream, P);
    return iago_check_launch("iago_conv3x3_f32");
}

int iago_stem_f32(const float *planes, const float *w1, const float *b1, float *y, int64_t n,
                  const int32_t *n_dev, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_stem_f32: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!planes || !w1 || !b1 || !y)
        return iago_fail(IAGO_ERR_INVALID, "iago_stem_f32: null pointer");
    const int64_t blocks = (n_dev && n * 16 > 2048) ? 2048 : n * 16;
    hipLaunchKernelGGL(stem_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, planes,
                       (const uint64_t *)nullptr, (const uint64_t *)nullptr, (const int64_t *)nullptr, w1, b1, y, n,
                       n_dev);
    return iago_check_launch("iago_stem_f32");
}

int iago_stem_f32_boards(const uint64_t *own, const uint64_t *opp, const int64_t *index, const float *w1,
                         const float *b1, float *y, int64_t n, const int32_t *n_dev, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_stem_f32_boards: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!own || !opp || !w1 || !b1 || !y)
        return iago_fail(IAGO_ERR_INVALID, "iago_stem_f32_boards: null pointer");
    const int64_t blocks = (n_dev && n * 16 > 2048) ? 2048 : n * 16;
    hipLaunchKernelGGL(stem_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float *)nullptr, own, opp, index, w1, b1, y, n, n_dev);
    return iago_check_launch("iago_stem_f32_boards");
}

int iago_policy_head(const float *x, const float *w9, const float *b10, float *probs, int64_t n,
                     const int32_t *n_dev, void *stream)
{
    if (n < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_head: negative n");
    if (n == 0)
        return IAGO_OK;
    if (!x || !w9 || !b10 || !probs)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_head: null pointer");
    hipLaunchKernelGGL(policy_head_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, x, w9, b10, probs, n,
                       n_dev);
    return iago_check_launch("iago_policy_head");
}

int iago_split_nchw(const float *x, void *hi, void *lo, int64_t n, int32_t channels, uint32_t *overflow,
                    void *stream)
{
    if (n < 0 || channels <= 0 || (channels % 16) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_split_nchw: channels must be a multiple of 16");
    if (n == 0)
        return IAGO_OK;
    if (!x || !hi || !lo)
        return iago_fail(IAGO_ERR_INVALID, "iago_split_nchw: null pointer");
    const int64_t total = n * (channels / 16) * 128;
    hipLaunchKernelGGL(split_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, (__half *)hi, (__half *)lo, n, channels, overflow);
    return iago_check_launch("iago_split_nchw");
}

int iago_merge_nchw(const void *hi, const void *lo, float *y, int64_t n, int32_t channels, void *stream)
{
    if (n < 0 || channels <= 0 || (channels % 16) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_merge_nchw: channels must be a multiple of 16");
    if (n == 0)
        return IAGO_OK;
    if (!hi || !lo || !y)
        return iago_fail(IAGO_ERR_INVALID, "iago_merge_nchw: null pointer");
    const int64_t total = n * channels * 64;
    hipLaunchKernelGGL(merge_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const __half *)hi, (const __half *)lo, y, n, channels);
    return iago_check_launch("iago_merge_nchw");
}

} // extern "C"
