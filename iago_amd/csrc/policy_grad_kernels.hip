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
//   boards; a board is one stage (K = 64 = two k-steps), double-buffered in LDS.  The three column taps of a row are
//   windows of ONE 10-cell padded row per lane: 3 transposed reads + 4 v_alignbit instead of 6 reads.
//   Partial sums per group of boards go to memory and are added up in a fixed order (deterministic).
#include "abi_common.hpp"

#include <hip/hip_fp16.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) short4v lds_short4v;

// (through the builtin: this loop feeds MFMAs from VALU results -- v_perm windows, register copies --, and the compiler
// only keeps the wait states of that right when it knows the instruction; the asm form of the walks gave wrong sums
// here.  The ISA has no accumulator copies in the loop: 108 MFMAs, 52 transposed reads, 72 VALU per board and wave)
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
#pragma unroll
        for (int s = 0; s < 2; s++) {
            half8 a_hi[2], a_lo[2];
#pragma unroll
            for (int mi = 0; mi < 2; mi++) {
                const int pa = at + (2 * mh + mi) * WG_DY_CB + 4 * s * WG_ROWB;
                a_hi[mi] = cat8(lds_tr(pa), lds_tr(pa + 128));
                a_lo[mi] = cat8(lds_tr(pa + WG_DY_PIECE), lds_tr(pa + WG_DY_PIECE + 128));
            }
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                // the padded row kq + 4s + ky of this lane's channel: 12 cells = 6 dwords per piece
                const int pb = at + WG_X_AT + cbl * WG_X_CB + (4 * s + ky) * WG_ROWB;
                uint32_t dh[6], dl[6];
#pragma unroll
                for (int rd = 0; rd < 3; rd++) {
                    const uint2 h = lds_tr(pb + 128 * rd), l = lds_tr(pb + WG_X_PIECE + 128 * rd);
                    dh[2 * rd] = h.x, dh[2 * rd + 1] = h.y;
                    dl[2 * rd] = l.x, dl[2 * rd + 1] = l.y;
                }
                // cells kx .. kx + 7 of the padded row: kx = 0 and 2 are whole dwords, kx = 1 four v_alignbit per piece
                u32x4 w1h, w1l;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    w1h[d] = __builtin_amdgcn_alignbit(dh[d + 1], dh[d], 16);
                    w1l[d] = __builtin_amdgcn_alignbit(dl[d + 1], dl[d], 16);
                }
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    const int kx = kk == 0 ? 0 : kk == 1 ? 2 : 1;
                    u32x4 wh, wl;
                    if (kx == 1) {
                        wh = w1h, wl = w1l;
                    } else {
#pragma unroll
                        for (int d = 0; d < 4; d++) {
                            wh[d] = dh[d + (kx >> 1)];
                            wl[d] = dl[d + (kx >> 1)];
                        }
                    }
                    const half8 b_hi = __builtin_bit_cast(half8, wh), b_lo = __builtin_bit_cast(half8, wl);
                    const int t = 3 * ky + kx;
                    PG_MFMA16(acc_c[0][t], a_hi[0], b_lo);
                    PG_MFMA16(acc_c[1][t], a_hi[1], b_lo);
                    PG_MFMA16(acc_m[0][t], a_hi[0], b_hi);
                    PG_MFMA16(acc_m[1][t], a_hi[1], b_hi);
                    PG_MFMA16(acc_c[0][t], a_lo[0], b_hi);
                    PG_MFMA16(acc_c[1][t], a_lo[1], b_hi);
                }
            }
        }
    };

    if (b_lo < b_hi)
        commit(fetch(b_lo), 0);
    __syncthreads();
    for (int64_t b = b_lo; b < b_hi; b++) {
        const int buf = (int)(b - b_lo) & 1;
        const bool more = b + 1 < b_hi;
        Staged G;
        if (more)
            G = fetch(b + 1);
        compute(buf);
        if (more)
            commit(G, buf ^ 1);
        __syncthreads();
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

} // namespace

extern "C" {

int iago_conv3x3_wgrad_split(const void *dy_hi, const void *dy_lo, const void *x_hi, const void *x_lo, int64_t n,
                             int32_t cin, float *part, int32_t groups, const int32_t *scale_exp, float *dw, void *stream)
{
    if (n < 0 || (cin != 64 && cin != 128) || groups < 8 || (groups % 8) != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_wgrad_split: cin must be 64 or 128, groups a multiple of 8");
    if (!dy_hi || !dy_lo || !x_hi || !x_lo || !part || !dw)
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
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)part, groups, cin, scale_exp, dw);
    return iago_check_launch("iago_conv3x3_wgrad_split");
}

} // extern "C"
