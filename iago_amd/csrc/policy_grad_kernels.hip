// policy_grad_kernels.hip -- the gradients of the REINFORCE update of SLPolicy (src/train_rl.py:55-66: pred =
// model(x) -- softmax probabilities --, loss = mean(softmax_cross_entropy(pred, y) * r), backward) on the f16 matrix
// units in "split f16" arithmetic: every float32 operand as two f16 numbers (22 significant bits), three MFMAs per
// product sum into float32 accumulators (csrc/conv_kernels.hip).  Together with iago_conv3x3_split (forward and,
// with transposed weights, backward-data) this replaces the MIOpen float32 convolutions of the update
// (9 ms of an 18 ms set at 1,900 rows: DESIGN.md section 5).
//
// Data: split channel blocks [n][C/16][64][16] f16 hi / lo as everywhere in conv_kernels.hip.  A gradient tensor
// carries a power-of-two scale 2^e (an int32 device word per tensor) so that its largest element sits near 2^14: the
// f16 pieces then hold 22 bits of every element down to 2^-28 of the largest (float32 keeps 24 of all of them; what
// is lost lies below the rounding of the sums the large elements dominate).
//
// wgrad_split_kernel: dW[co][ci][ky][kx] = sum over boards and cells of dY[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1].
//   GEMM M = co, N = (ci, tap), K = (board, cell).  The operands live in memory with the CHANNELS innermost -- the
//   contraction runs over cells -- so both MFMA operands come out of LDS through ds_read_b64_tr_b16 (gfx950's
//   transposed read: a lane gets 4 consecutive cells of ONE channel).  One workgroup = 64 output channels x 32 input
//   channels x 9 taps (72 tiles of v_mfma_f32_16x16x32_f16, 18 per wave, 144 accumulator registers) over a group of
//   boards; a board is one stage (K = 64 = two k-steps), double-buffered in LDS.
//   Partial sums per group of boards go to memory and are added up in a fixed order (deterministic).
#include "abi_common.hpp"

#include <hip/hip_fp16.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) short4v lds_short4v;

// (through the builtin: a first form of this loop fed MFMAs from VALU results -- v_perm windows, register copies --, and
// the compiler only keeps the wait states of that right when it knows the instruction: the asm form of the walks gave
// wrong sums there.  The ISA has no accumulator copies in the loop: 108 MFMAs, 88 transposed reads per board and wave)
#define PG_MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)

// LDS images of one board: rows of 8 (dY) or 12 (X, padded: columns 0 and 9 are the zero border, 10 and 11 filler)
// cells x 32 B (16 channels), 384 B apart: the two 16-lane groups of a 32-lane half read rows that are one board row
// apart -- 96 banks = 32 mod 64 further, so their 128-byte blocks never share a bank
constexpr int WG_ROWB = 384;
constexpr int WG_DY_CB = 8 * WG_ROWB;       // one channel block of dY: 3,072 B
constexpr int WG_DY_PIECE = 4 * WG_DY_CB;   // hi or lo of the workgroup's 64 output channels: 12,288 B
constexpr int WG_X_CB = 10 * WG_ROWB;       // one channel block of the padded input: 3,840 B
constexpr int WG_X_PIECE = 2 * WG_X_CB;     // hi or lo of the workgroup's 32 input channels: 7,680 B
constexpr int WG_X_AT = 2 * WG_DY_PIECE;    // 24,576
constexpr int WG_BUF = WG_X_AT + 2 * WG_X_PIECE; // 39,936 B per board
constexpr int WG_LDS = 2 * WG_BUF;          // 79,872 B

struct WgradParams {
    const u32x4 *dy_hi, *dy_lo; // [n][8][64][16] f16: dL / d(pre-activation), times 2^e
    const u32x4 *x_hi, *x_lo;   // [n][cin/16][64][16] f16: the layer's input
    float *part;                // [groups][9][128][cin]
    int64_t n;
    int32_t cin, groups;
};

extern __shared__ __align__(16) char pg_lds[];

__device__ __forceinline__ uint2 lds_tr(int off)
{
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)(pg_lds + off));
    return __builtin_bit_cast(uint2, v);
}

__device__ __forceinline__ half8 cat8(uint2 a, uint2 b)
{
    u32x4 v;
    v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
    return __builtin_bit_cast(half8, v);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wgrad_split_kernel(WgradParams P)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the workgroups of one group of boards sit on ONE XCD (workgroup i runs on XCD i mod 8; groups is a multiple of
    // 8): they read the same boards at about the same time, through one L2
    const int group = (int)(blockIdx.x % (unsigned)P.groups), sub = (int)(blockIdx.x / (unsigned)P.groups);
    const int co_half = sub & 1, ciq = sub >> 1;
    const int mh = wv & 1, cbl = wv >> 1; // the wave's 32 of the 64 output channels, its 16 of the 32 input channels
    const int ncb = P.cin >> 4;
    const int64_t per = (P.n + P.groups - 1) / P.groups;
    const int64_t b_lo = group * per, b_hi = b_lo + per < P.n ? b_lo + per : P.n;

    for (int i = tid; i < 2 * WG_X_PIECE / 16; i += 256) { // the borders of the padded planes stay zero
        ((uint4 *)(pg_lds + WG_X_AT))[i] = make_uint4(0, 0, 0, 0);
        ((uint4 *)(pg_lds + WG_BUF + WG_X_AT))[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads(); // (the first board's cells are written by OTHER threads than the ones that zeroed them)
    // staging: per board 1,024 16-byte pieces of dY (4 channel blocks x 64 cells x 2 halves x hi / lo) and 512 of X
    int dy_src[2], dy_dst[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int e = tid + 256 * q, cb = e >> 7, cell = (e >> 1) & 63, hp = e & 1;
        dy_src[q] = (co_half * 4 + cb) * 128 + (e & 127);
        dy_dst[q] = cb * WG_DY_CB + (cell >> 3) * WG_ROWB + (cell & 7) * 32 + hp * 16;
    }
    int x_src, x_dst;
    {
        const int e = tid, cb = e >> 7, cell = (e >> 1) & 63, hp = e & 1;
        x_src = (ciq * 2 + cb) * 128 + (e & 127);
        x_dst = WG_X_AT + cb * WG_X_CB + ((cell >> 3) + 1) * WG_ROWB + ((cell & 7) + 1) * 32 + hp * 16;
    }
    struct Staged {
        u32x4 v[6];
    };
    auto fetch = [&](int64_t b) {
        Staged G;
        G.v[0] = P.dy_hi[b * 1024 + dy_src[0]];
        G.v[1] = P.dy_hi[b * 1024 + dy_src[1]];
        G.v[2] = P.dy_lo[b * 1024 + dy_src[0]];
        G.v[3] = P.dy_lo[b * 1024 + dy_src[1]];
        G.v[4] = P.x_hi[b * ncb * 128 + x_src];
        G.v[5] = P.x_lo[b * ncb * 128 + x_src];
        return G;
    };
    auto commit = [&](const Staged &G, int buf) {
        char *at = pg_lds + buf * WG_BUF;
        *(u32x4 *)(at + dy_dst[0]) = G.v[0];
        *(u32x4 *)(at + dy_dst[1]) = G.v[1];
        *(u32x4 *)(at + WG_DY_PIECE + dy_dst[0]) = G.v[2];
        *(u32x4 *)(at + WG_DY_PIECE + dy_dst[1]) = G.v[3];
        *(u32x4 *)(at + x_dst) = G.v[4];
        *(u32x4 *)(at + WG_X_PIECE + x_dst) = G.v[5];
    };

    float4v acc_m[2][9], acc_c[2][9];
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                acc_m[mi][t][v] = 0.0f;
                acc_c[mi][t][v] = 0.0f;
            }

    // transposed reads: lane 4q + p of the 16-lane group kq supplies the address of row q (a cell), columns 4p .. 4p+3
    // (channels); lane i of the group receives channel i of the 4 cells.  k = 8 kq + j of k-step s <-> cell (row kq +
    // 4s, column j): two reads per operand (columns 0-3, 4-7).
    const int kq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int lane_off = kq * WG_ROWB + q * 32 + p * 8;

    auto compute = [&](int buf) {
        const int at = buf * WG_BUF + lane_off;
        // A: this wave's two M tiles, both k-steps (16 reads); B: one (k-step, kernel row, kernel column) at a time --
        // the padded row kq + 4s + ky of this lane's channel, cells kx .. kx + 7: two transposed reads per piece (a
        // lane's rows may start anywhere: + 32 B per cell; windows cut out of one 12-cell read by v_alignbit and
        // register copies measured slower) -- read TWO steps ahead of the six MFMAs that use it
        half8 a_hi[2][2], a_lo[2][2];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int mi = 0; mi < 2; mi++) {
                const int pa = at + (2 * mh + mi) * WG_DY_CB + 4 * s * WG_ROWB;
                a_hi[s][mi] = cat8(lds_tr(pa), lds_tr(pa + 128));
                a_lo[s][mi] = cat8(lds_tr(pa + WG_DY_PIECE), lds_tr(pa + WG_DY_PIECE + 128));
            }
        half8 b_hi[3], b_lo[3];
        auto load_b = [&](int step, int slot) {
            const int s = step / 9, ky = (step % 9) / 3, kx = step % 3;
            const int pb = at + WG_X_AT + cbl * WG_X_CB + (4 * s + ky) * WG_ROWB + 32 * kx;
            b_hi[slot] = cat8(lds_tr(pb), lds_tr(pb + 128));
            b_lo[slot] = cat8(lds_tr(pb + WG_X_PIECE), lds_tr(pb + WG_X_PIECE + 128));
        };
        load_b(0, 0);
        load_b(1, 1);
#pragma unroll
        for (int step = 0; step < 18; step++) {
            if (step + 2 < 18)
                load_b(step + 2, (step + 2) % 3);
            const int s = step / 9, t = step % 9, cur = step % 3;
            PG_MFMA16(acc_c[0][t], a_hi[s][0], b_lo[cur]);
            PG_MFMA16(acc_c[1][t], a_hi[s][1], b_lo[cur]);
            PG_MFMA16(acc_m[0][t], a_hi[s][0], b_hi[cur]);
            PG_MFMA16(acc_m[1][t], a_hi[s][1], b_hi[cur]);
            PG_MFMA16(acc_c[0][t], a_lo[s][0], b_hi[cur]);
            PG_MFMA16(acc_c[1][t], a_lo[s][1], b_hi[cur]);
        }
    };

    // Boards b + 1 .. b + 3 are on their way (registers) while board b is multiplied: one board is 0.75 us of MFMAs, a
    // load from beyond L2 takes longer than that under load.  Three register sets in rotation: the set of board b + 1
    // goes to the other LDS buffer right after board b's MFMAs, and is re-issued for board b + 4 at the next turn.
    if (b_lo < b_hi)
        commit(fetch(b_lo), 0);
    __syncthreads();
    const int64_t last = b_hi - 1;
    auto clamp = [&](int64_t b) { return b < last ? b : last; }; // (past the end: a harmless repeat, never committed)
    Staged S0, S1, S2;
    if (b_lo < b_hi) {
        S1 = fetch(clamp(b_lo + 1));
        S2 = fetch(clamp(b_lo + 2));
    }
    auto turn = [&](int64_t b, Staged &next, Staged &refill) {
        // next: board b + 1; refill: the set that held board b (committed a turn ago), now board b + 3
        refill = fetch(clamp(b + 3));
        compute((int)(b - b_lo) & 1);
        if (b + 1 < b_hi)
            commit(next, ((int)(b - b_lo) & 1) ^ 1);
        __syncthreads();
    };
    for (int64_t b = b_lo; b < b_hi; b += 3) {
        turn(b, S1, S0);
        if (b + 1 < b_hi)
            turn(b + 1, S2, S1);
        if (b + 2 < b_hi)
            turn(b + 2, S0, S2);
    }
    // tile (mi, tap): the lane holds rows 4 kq + v (output channels), column lane & 15 (input channel)
    const int ci = ciq * 32 + cbl * 16 + (lane & 15);
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int co = co_half * 64 + (2 * mh + mi) * 16 + 4 * kq + v;
                P.part[(((int64_t)group * 9 + t) * 128 + co) * P.cin + ci] =
                    acc_m[mi][t][v] + acc_c[mi][t][v] * (1.0f / 2048.0f);
            }
}

// dW[co][ci][tap] = 2^-e * sum over the groups, in group order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *part, int groups, int cin, const int32_t *scale_exp,
                                                           float *dw)
{
    const int t = blockIdx.x * 256 + threadIdx.x; // (tap, co, ci)
    const int total = 9 * 128 * cin;
    if (t >= total)
        return;
    float s = 0.0f;
    for (int g = 0; g < groups; g++)
        s += part[(int64_t)g * total + t];
    if (scale_exp)
        s = ldexpf(s, -*scale_exp);
    const int ci = t % cin, co = (t / cin) & 127, tap = t / (cin * 128);
    dw[((int64_t)co * cin + ci) * 9 + tap] = s;
}


// The partial sums of ALL blocks of an update in one launch (iago_policy_reinforce_grad): per block k the weight gradient's
// groups (wgrad_reduce_kernel's arithmetic: the groups in order, then 2^-e) and the bias gradient's pair sums
// (bias_reduce_kernel's, conv_kernels.hip: a wave per channel, lane l takes boards l, l + 64, .., then the xor tree) -- the
// same sums in the same order as the per-block launches, which were 14 launches of mostly latency (23 us per block).
__device__ __forceinline__ float wave_sum_xor(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d);
    return v;
}
struct ReduceAllParams {
    const float *wpart[7], *bpart[7];
    float *dw[7], *db[7];
    const int32_t *scale_exp[7];
    int32_t groups[7], cin[7];
    int32_t w_block0[8]; // first workgroup of block k's weights; [7] = first workgroup of the biases
    int64_t n;
};
__global__ __launch_bounds__(256) void grad_reduce_all_kernel(ReduceAllParams P)
{
    const int b = blockIdx.x;
    if (b < P.w_block0[7]) {
        int k = 0;
#pragma unroll
        for (int i = 1; i < 7; i++)
            k += b >= P.w_block0[i] ? 1 : 0;
        // (by-value parameter arrays indexed at run time go through scratch memory: select with compile-time indices)
        const float *part = P.wpart[0];
        float *dw = P.dw[0];
        const int32_t *se = P.scale_exp[0];
        int groups = P.groups[0], cin = P.cin[0], first = P.w_block0[0];
#pragma unroll
        for (int i = 1; i < 7; i++)
            if (k == i) {
                part = P.wpart[i], dw = P.dw[i], se = P.scale_exp[i];
                groups = P.groups[i], cin = P.cin[i], first = P.w_block0[i];
            }
        const int t = (b - first) * 256 + threadIdx.x; // (tap, co, ci)
        const int total = 9 * 128 * cin;
        if (t >= total)
            return;
        float s = 0.0f;
        for (int g = 0; g < groups; g++)
            s += part[(int64_t)g * total + t];
        s = ldexpf(s, -*se);
        const int ci = t % cin, co = (t / cin) & 127, tap = t / (cin * 128);
        dw[((int64_t)co * cin + ci) * 9 + tap] = s;
        return;
    }
    // biases: 4 channels per workgroup (a wave each), 128 channels per block
    const int j = (b - P.w_block0[7]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = j >> 7, ch = j & 127;
    if (k >= 7)
        return;
    const float *bp = P.bpart[0];
    float *db = P.db[0];
#pragma unroll
    for (int i = 1; i < 7; i++)
        if (k == i)
            bp = P.bpart[i], db = P.db[i];
    const int cb = ch >> 4, c = ch & 15;
    float s = 0.0f;
    for (int64_t bd = lane; bd < P.n; bd += 64) {
        const int64_t pair = bd * 8 + cb;
        s += bp[(pair >> 1) * 32 + (pair & 1) * 16 + c];
    }
    s = wave_sum_xor(s);
    if (lane == 0)
        db[ch] = s;
}

// ---- The head of SLPolicy and the loss, forward and backward in one pass (network.py:29-47, src/train_rl.py:61-64):
//   logits = conv9 (1x1, 128 -> 1, no bias) + bias10[cell];  p = softmax(logits)            (the model's output)
//   c = softmax_cross_entropy(p, a) = logsumexp(p) - p[a]     (the reference applies log-softmax to the probabilities)
//   loss = sum_b c_b z_b / n_true
// and back: g = (z / n_true) (softmax(p) - onehot(a)), dlogits = p (g - <g, p>), dW9[c] = sum dlogits x8[c],
// dbias10[cell] = sum dlogits, dY8[c][cell] = [x8 > 0] dlogits[cell] w9[c] (the gradient at block 8's pre-activations,
// float32 channel blocks + the largest magnitude).  One wave per board at a time, lane = cell; float32 arithmetic on
// the exact values hi + lo 2^-11.
struct HeadGradParams {
    const uint4 *x_hi, *x_lo;  // [n][8][64][16] f16: the output of block 8
    const float *w9, *b10;     // [128], [64]
    const int32_t *action;     // [n]
    const float *reward;       // [n]
    float inv_n;               // 1 / (rows the mean divides by)
    int64_t n;
    float4v *dy;               // [n][8][64][16] float32
    uint32_t *max_bits;
    float *part;               // [gridDim.x][193]: dW9 (128), dbias10 (64), loss (1) of each workgroup
    float *probs;              // optional [n][64]: the model's output
    uint32_t *bad;             // optional: bit 1 raised when an action is outside 0 .. 63 (F.softmax_cross_entropy would raise)
};
constexpr int HEAD_PART = 193;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v = fmaxf(v, __shfl_xor(v, d));
    return v;
}

__global__ __launch_bounds__(256) void head_grad_kernel(HeadGradParams P)
{
    __shared__ float red[4][HEAD_PART];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc9[128];
#pragma unroll
    for (int c = 0; c < 128; c++)
        acc9[c] = 0.0f;
    float acc_b = 0.0f, acc_loss = 0.0f, big = 0.0f;
    const float my_b10 = P.b10[lane];
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < P.n; b += (int64_t)gridDim.x * 4) {
        float x[128];
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
            const int64_t at = ((b * 8 + cb) * 64 + lane) * 2;
            const uint4 h0 = P.x_hi[at], h1 = P.x_hi[at + 1], l0 = P.x_lo[at], l1 = P.x_lo[at + 1];
            const __half2 *hh0 = (const __half2 *)&h0, *hh1 = (const __half2 *)&h1;
            const __half2 *ll0 = (const __half2 *)&l0, *ll1 = (const __half2 *)&l1;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float2 a = __half22float2(hh0[k]), c2 = __half22float2(ll0[k]);
                const float2 d = __half22float2(hh1[k]), e2 = __half22float2(ll1[k]);
                x[16 * cb + 2 * k] = a.x + c2.x * (1.0f / 2048.0f);
                x[16 * cb + 2 * k + 1] = a.y + c2.y * (1.0f / 2048.0f);
                x[16 * cb + 8 + 2 * k] = d.x + e2.x * (1.0f / 2048.0f);
                x[16 * cb + 8 + 2 * k + 1] = d.y + e2.y * (1.0f / 2048.0f);
            }
        }
        float logit = 0.0f;
#pragma unroll
        for (int c = 0; c < 128; c++)
            logit = fmaf(P.w9[c], x[c], logit);
        logit += my_b10;
        const float m = wave_max(logit);
        const float ex = expf(logit - m);
        const float p = ex / wave_sum(ex);                          // F.softmax, network.py:46
        if (P.probs)
            P.probs[b * 64 + lane] = p;
        const int a = P.action[b];
        const float z = P.reward[b];
        if ((uint32_t)a >= 64u && lane == 0 && P.bad)
            atomicOr(P.bad, 2u); // (the row's loss and gradient mean nothing: the caller must not use this update)
        // F.softmax_cross_entropy(pred, y): log-softmax of the probabilities (src/train_rl.py:62)
        const float m2 = wave_max(p);
        const float ex2 = expf(p - m2);
        const float s2 = wave_sum(ex2);
        const float lse = m2 + logf(s2);
        const float p_a = __shfl(p, a & 63);
        acc_loss += (lse - p_a) * z;
        const float g = z * P.inv_n * (ex2 / s2 - (lane == a ? 1.0f : 0.0f));
        const float dl = p * (g - wave_sum(g * p));                 // through the model's softmax
        acc_b += dl;
#pragma unroll
        for (int cb = 0; cb < 8; cb++) {
#pragma unroll
            for (int qt = 0; qt < 4; qt++) {
                float4v v;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int c = 16 * cb + 4 * qt + k;
                    acc9[c] = fmaf(dl, x[c], acc9[c]);
                    v[k] = x[c] > 0.0f ? dl * P.w9[c] : 0.0f;
                    big = fmaxf(big, fabsf(v[k]));
                }
                P.dy[((b * 8 + cb) * 64 + lane) * 4 + qt] = v;
            }
        }
    }
    big = wave_max(big);
    if (lane == 0 && big > 0.0f)
        atomicMax(P.max_bits, __float_as_uint(big));
#pragma unroll
    for (int c = 0; c < 128; c++) {
        const float s = wave_sum(acc9[c]);
        if (lane == 0)
            red[wv][c] = s;
    }
    red[wv][128 + lane] = acc_b;
    if (lane == 0)
        red[wv][192] = acc_loss; // (the same in every lane)
    __syncthreads();
    if (threadIdx.x < HEAD_PART)
        P.part[(int64_t)blockIdx.x * HEAD_PART + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[j] = (j == 192 ? inv_n : 1) * sum over the workgroups' partial sums: one wave per output, lane l takes the partial
// sums l, l + 64, .. (a fixed order)
__global__ __launch_bounds__(256) void head_reduce_kernel(const float *part, int n_parts, float inv_n, float *dw9, float *db10,
                                                          float *loss)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= HEAD_PART)
        return;
    float s = 0.0f;
    for (int i = lane; i < n_parts; i += 64)
        s += part[(int64_t)i * HEAD_PART + j];
    s = wave_sum(s);
    if (lane != 0)
        return;
    if (j < 128)
        dw9[j] = s;
    else if (j < 192)
        db10[j - 128] = s;
    else
        *loss = s * inv_n;
}

// ---- The weight and bias gradients of block 1 (3x3, 2 -> 64 on the planes of the board: plane 0 = the opponent's
// stones, plane 1 = the mover's, game.py:168-174): dW1[co][plane][ky][kx] = sum over boards and cells of dY1[co][y][x]
// [plane has a stone at (y + ky - 1, x + kx - 1)], db1[co] = sum dY1.  One wave per board at a time, lane = channel.
constexpr int STEM_PART = 64 * 19;
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float *dy, const uint64_t *own, const uint64_t *opp, int64_t n,
                                                         float *part)
{
    __shared__ float red[4][STEM_PART];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[19];
#pragma unroll
    for (int k = 0; k < 19; k++)
        acc[k] = 0.0f;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < n; b += (int64_t)gridDim.x * 4) {
        const uint64_t pl[2] = {opp[b], own[b]};
        const float *src = dy + ((b * 4 + (lane >> 4)) * 64) * 16 + (lane & 15);
        for (int cell = 0; cell < 64; cell++) {
            const float d = src[cell * 16];
            const int y = cell >> 3, x = cell & 7;
            acc[18] += d;
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    const bool in = yy >= 0 && yy < 8 && xx >= 0 && xx < 8;
                    const int bit = in ? yy * 8 + xx : 0;
#pragma unroll
                    for (int pn = 0; pn < 2; pn++)
                        if (in && ((pl[pn] >> bit) & 1ull))
                            acc[pn * 9 + ky * 3 + kx] += d;
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 19; k++)
        red[wv][lane * 19 + k] = acc[k];
    __syncthreads();
    for (int j = threadIdx.x; j < STEM_PART; j += 256)
        part[(int64_t)blockIdx.x * STEM_PART + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
}

__global__ __launch_bounds__(256) void stem_reduce_kernel(const float *part, int n_parts, float *dw1, float *db1)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; // (channel, 18 weights + 1 bias)
    if (j >= STEM_PART)
        return;
    float s = 0.0f;
    for (int i = lane; i < n_parts; i += 64)
        s += part[(int64_t)i * STEM_PART + j];
    s = wave_sum(s);
    if (lane != 0)
        return;
    const int co = j / 19, k = j - 19 * co;
    if (k < 18)
        dw1[co * 18 + k] = s;
    else
        db1[co] = s;
}


// ---- chainer.optimizers.Adam + the WeightDecay hook (src/train_rl.py:24-26,66) over all parameters in one launch: the
// documented rule operation by operation, each product, sum, root and quotient rounded on its own (no fused
// multiply-adds), i.e. the bits of the elementwise float32 restatement in iago_amd/train_rl.py
struct AdamParams {
    float *p[IAGO_ADAM_MAX_TENSORS];
    const float *g[IAGO_ADAM_MAX_TENSORS];
    float *m[IAGO_ADAM_MAX_TENSORS], *v[IAGO_ADAM_MAX_TENSORS], *step[IAGO_ADAM_MAX_TENSORS];
    int64_t end[IAGO_ADAM_MAX_TENSORS]; // running element counts
    int32_t n_tensors;
    float alpha_t, one_minus_beta1, one_minus_beta2, eps, weight_decay;
};

__global__ __launch_bounds__(256) void adam_chainer_kernel(AdamParams A)
{
#pragma clang fp contract(off) // (a product and the sum behind it must not fuse)
    const int64_t total = A.end[A.n_tensors - 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int s = 0;
        while (i >= A.end[s])
            s++;
        const int64_t j = i - (s ? A.end[s - 1] : 0);
        const float p = A.p[s][j], m = A.m[s][j], v = A.v[s][j];
        // (plain operators: the pragma above covers THIS function's expressions, not those inside HIP's __f*_rn inlines)
        const float wd_p = p * A.weight_decay;
        const float g = A.g[s][j] + wd_p;                        // the hook: g += rate * w
        const float dm = g - m;
        const float dm2 = dm * A.one_minus_beta1;
        const float m2 = m + dm2;                                // m += (1 - beta1)(g - m)
        const float gg = g * g;
        const float dv = gg - v;
        const float dv2 = dv * A.one_minus_beta2;
        const float v2 = v + dv2;                                // v += (1 - beta2)(g g - v)
        const float root = sqrtf(v2); // (correctly rounded; HIP's __fsqrt_rn is the native approximation)
        const float den = root + A.eps;
        A.m[s][j] = m2;
        A.v[s][j] = v2;
        const float num = m2 * A.alpha_t;
        const float step = num / den;                            // alpha_t m / (sqrt(v) + eps)
        if (A.step[s])
            A.step[s][j] = step; // (the caller subtracts it: its tensor library then knows the parameter changed)
        else
            A.p[s][j] = p - step;
    }
}

} // namespace

extern "C" {

int iago_conv3x3_wgrad_split(const void *dy_hi, const void *dy_lo, const void *x_hi, const void *x_lo, int64_t n,
                             int32_t cin, float *part, int32_t groups, const int32_t *scale_exp, float *dw, void *stream)
{
    if (n < 0 || (cin != 64 && cin != 128) || groups < 8 || (groups % 8) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_wgrad_split: cin must be 64 or 128, groups a multiple of 8");
    if (!dy_hi || !dy_lo || !x_hi || !x_lo || !part)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_wgrad_split: null pointer");
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)wgrad_split_kernel, WG_LDS, configured,
                         "iago_conv3x3_wgrad_split: cannot reserve 78 KB of LDS"))
        return IAGO_ERR_HIP;
    WgradParams P;
    P.dy_hi = (const u32x4 *)dy_hi;
    P.dy_lo = (const u32x4 *)dy_lo;
    P.x_hi = (const u32x4 *)x_hi;
    P.x_lo = (const u32x4 *)x_lo;
    P.part = part;
    P.n = n;
    P.cin = cin;
    P.groups = groups;
    const unsigned grid = (unsigned)(groups * 2 * (cin / 32));
    hipLaunchKernelGGL(wgrad_split_kernel, dim3(grid), dim3(256), WG_LDS, (hipStream_t)stream, P);
    const int total = 9 * 128 * cin;
    if (dw) // (NULL: the partial sums only -- iago_policy_reinforce_grad reduces those of all blocks in one launch)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)part, groups, cin, scale_exp, dw);
    return iago_check_launch("iago_conv3x3_wgrad_split");
}


static int64_t pg_round(int64_t b)
{
    return (b + 255) & ~(int64_t)255;
}
constexpr int PG_GRID = 256;        // workgroups of the head and block-1 kernels (grid-stride over the boards: any device)
constexpr int PG_MAX_GROUPS = 64;

// groups of boards of a weight-gradient launch: one workgroup per CU (groups x 2 x cin / 32 workgroups), a multiple of 8
// (the workgroups of a group share an XCD), at most PG_MAX_GROUPS (the partial sums' scratch)
static int pg_groups(int cin)
{
    // (asked of the runtime once per device, as iago_mcts_search_capacity does: a process may drive devices of
    // different sizes, and the partial sums' grouping -- hence the gradients' last bits -- follows the device)
    static std::atomic<int> known[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        dev = 0;
    int n = known[dev & 63].load(std::memory_order_acquire);
    if (n == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
            v = 256;
        known[dev & 63].store(v, std::memory_order_release);
        n = v;
    }
    int g = n / (2 * (cin / 32)) / 8 * 8;
    return g < 8 ? 8 : g > PG_MAX_GROUPS ? PG_MAX_GROUPS : g;
}

int64_t iago_policy_grad_workspace_bytes(int64_t n)
{
    if (n < 0)
        return -1;
    int64_t b = 0;
    b += 2 * pg_round(n * 8192) + 14 * pg_round(n * 16384);          // saved activations of blocks 1..8, hi + lo
    b += pg_round(n * 32768) + 2 * pg_round(n * 16384);              // a gradient as float32 and as scaled pieces
    b += 6 * pg_round((int64_t)PG_MAX_GROUPS * 9 * 128 * 128 * 4) +
         pg_round((int64_t)PG_MAX_GROUPS * 9 * 128 * 64 * 4);        // partial weight gradients, a buffer per block
    b += 7 * pg_round(((n * 8 + 1) / 2) * 32 * 4);                   // partial bias gradients, a buffer per block
    b += pg_round((int64_t)PG_GRID * HEAD_PART * 4) + pg_round((int64_t)PG_GRID * STEM_PART * 4);
    b += 256;                                                        // the tensors' largest magnitudes and scales
    return b;
}

int iago_policy_reinforce_grad(const iago_policy_grad_args *A, void *stream)
{
    if (!A || A->n < 0 || A->n_mean <= 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_reinforce_grad: bad arguments");
    const int64_t n = A->n;
    if (!A->own || !A->opp || !A->action || !A->reward || !A->w1 || !A->b1 || !A->w9 || !A->b10 || !A->g_w1 ||
        !A->g_b1 || !A->g_w9 || !A->g_b10 || !A->loss || !A->workspace)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_reinforce_grad: null pointer");
    for (int k = 0; k < 7; k++)
        if (!A->w_hi[k] || !A->w_lo[k] || !A->wt_hi[k] || !A->wt_lo[k] || !A->bias[k] || !A->g_w[k] || !A->g_b[k])
            return iago_fail(IAGO_ERR_INVALID, "iago_policy_reinforce_grad: null pointer (blocks 2..8)");
    if (A->workspace_bytes < iago_policy_grad_workspace_bytes(n) || ((uintptr_t)A->workspace & 255))
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_reinforce_grad: workspace too small or not 256-byte aligned "
                                           "(iago_policy_grad_workspace_bytes)");
    if (n == 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_reinforce_grad: no rows");
    hipStream_t st = (hipStream_t)stream;
    char *at = (char *)A->workspace;
    auto take = [&](int64_t bytes) {
        char *p = at;
        at += pg_round(bytes);
        return (void *)p;
    };
    void *x_hi[8], *x_lo[8];
    for (int k = 0; k < 8; k++) {
        x_hi[k] = take(n * (k ? 16384 : 8192));
        x_lo[k] = take(n * (k ? 16384 : 8192));
    }
    float *dyf = (float *)take(n * 32768);
    void *dys_hi = take(n * 16384), *dys_lo = take(n * 16384);
    // (partial sums of the weight and bias gradients: a buffer per block, reduced by ONE launch at the end)
    float *wpart[7], *bpart[7];
    for (int k = 0; k < 7; k++) {
        wpart[k] = (float *)take((int64_t)PG_MAX_GROUPS * 9 * 128 * (k ? 128 : 64) * 4);
        bpart[k] = (float *)take(((n * 8 + 1) / 2) * 32 * 4);
    }
    float *hpart = (float *)take((int64_t)PG_GRID * HEAD_PART * 4);
    float *spart = (float *)take((int64_t)PG_GRID * STEM_PART * 4);
    uint32_t *max_bits = (uint32_t *)take(256); // [0..7]: of the gradient at block k + 1's pre-activations; [16..23]: scales
    int32_t *scale_exp = (int32_t *)(max_bits + 16);
    if (hipMemsetAsync(max_bits, 0, 256, st) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_policy_reinforce_grad: hipMemsetAsync failed");

    // forward, every block's output kept (src/train_rl.py:61)
    int rc = iago_value_stem_boards(A->own, A->opp, A->w1, A->b1, x_hi[0], x_lo[0], n, A->overflow, stream);
    for (int k = 0; k < 7 && rc == IAGO_OK; k++)
        rc = iago_conv3x3_split(x_hi[k], x_lo[k], A->w_hi[k], A->w_lo[k], A->bias[k], x_hi[k + 1], x_lo[k + 1], n,
                                k ? 128 : 64, 128, A->overflow, stream);
    if (rc != IAGO_OK)
        return rc;
    // head + loss, forward and backward (src/train_rl.py:61-65)
    const float inv_n = 1.0f / (float)A->n_mean;
    HeadGradParams H;
    H.x_hi = (const uint4 *)x_hi[7];
    H.x_lo = (const uint4 *)x_lo[7];
    H.w9 = A->w9;
    H.b10 = A->b10;
    H.action = A->action;
    H.reward = A->reward;
    H.inv_n = inv_n;
    H.n = n;
    H.dy = (float4v *)dyf;
    H.max_bits = max_bits + 7;
    H.part = hpart;
    H.probs = A->probs;
    H.bad = A->overflow;
    hipLaunchKernelGGL(head_grad_kernel, dim3(PG_GRID), dim3(256), 0, st, H);
    hipLaunchKernelGGL(head_reduce_kernel, dim3((HEAD_PART + 3) / 4), dim3(256), 0, st, (const float *)hpart, PG_GRID, inv_n, A->g_w9,
                       A->g_b10, A->loss);
    // blocks 8 .. 2: the gradient at the block's pre-activations (float32 in dyf) -> its scaled pieces + the bias
    // gradient; the weight gradient; the gradient at the pre-activations of the block below
    for (int k = 6; k >= 0 && rc == IAGO_OK; k--) {
        const int cin = k ? 128 : 64;
        rc = iago_split_scaled(dyf, max_bits + k + 1, dys_hi, dys_lo, scale_exp + k + 1, n, 128, bpart[k], nullptr, stream);
        if (rc == IAGO_OK)
            rc = iago_conv3x3_wgrad_split(dys_hi, dys_lo, x_hi[k], x_lo[k], n, cin, wpart[k], pg_groups(cin),
                                          scale_exp + k + 1, nullptr, stream);
        if (rc == IAGO_OK)
            rc = iago_conv3x3_bwd_data_split(dys_hi, dys_lo, scale_exp + k + 1, A->wt_hi[k], A->wt_lo[k], x_hi[k], x_lo[k],
                                             cin, dyf, max_bits + k, n, stream);
    }
    if (rc != IAGO_OK)
        return rc;
    {
        ReduceAllParams Rp;
        int at_block = 0;
        for (int k = 0; k < 7; k++) {
            const int cin = k ? 128 : 64;
            Rp.wpart[k] = wpart[k];
            Rp.bpart[k] = bpart[k];
            Rp.dw[k] = A->g_w[k];
            Rp.db[k] = A->g_b[k];
            Rp.scale_exp[k] = scale_exp + k + 1;
            Rp.groups[k] = pg_groups(cin);
            Rp.cin[k] = cin;
            Rp.w_block0[k] = at_block;
            at_block += (9 * 128 * cin + 255) / 256;
        }
        Rp.w_block0[7] = at_block;
        Rp.n = n;
        hipLaunchKernelGGL(grad_reduce_all_kernel, dim3((unsigned)(at_block + 7 * 128 / 4)), dim3(256), 0, st, Rp);
    }
    // block 1 from the float32 gradient at its pre-activations
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(PG_GRID), dim3(256), 0, st, (const float *)dyf, A->own, A->opp, n, spart);
    hipLaunchKernelGGL(stem_reduce_kernel, dim3((STEM_PART + 3) / 4), dim3(256), 0, st, (const float *)spart, PG_GRID,
                       A->g_w1, A->g_b1);
    return iago_check_launch("iago_policy_reinforce_grad");
}


int iago_adam_chainer(const iago_adam_args *a, void *stream)
{
    if (!a || a->n_tensors < 1 || a->n_tensors > IAGO_ADAM_MAX_TENSORS)
        return iago_fail(IAGO_ERR_INVALID, "iago_adam_chainer: 1 .. IAGO_ADAM_MAX_TENSORS tensors expected");
    AdamParams A;
    int64_t total = 0;
    for (int k = 0; k < a->n_tensors; k++) {
        if (!a->p[k] || !a->g[k] || !a->m[k] || !a->v[k] || a->count[k] <= 0)
            return iago_fail(IAGO_ERR_INVALID, "iago_adam_chainer: null pointer or empty tensor");
        A.p[k] = a->p[k];
        A.g[k] = a->g[k];
        A.m[k] = a->m[k];
        A.v[k] = a->v[k];
        A.step[k] = a->step[k];
        total += a->count[k];
        A.end[k] = total;
    }
    A.n_tensors = a->n_tensors;
    A.alpha_t = a->alpha_t;
    A.one_minus_beta1 = a->one_minus_beta1;
    A.one_minus_beta2 = a->one_minus_beta2;
    A.eps = a->eps;
    A.weight_decay = a->weight_decay;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(adam_chainer_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       (hipStream_t)stream, A);
    return iago_check_launch("iago_adam_chainer");
}

} // extern "C"
