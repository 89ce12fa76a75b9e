/*
 * iago_hip_layers.h -- the layer-level entry points of the nets (network.py:5-96): single convolution blocks, the
 * ends of the nets, format conversions and the three gradient kernels of a block.  The whole-net launches of iago_hip.h
 * (iago_policy_forward_split3, iago_value_forward_split, iago_policy_reinforce_grad; the persistent search) are built
 * from the same device code; these entry points serve the PyTorch modules' planes-fed forwards of small batches
 * (iago_amd/network.py), the supervised trainers' inference path, and the tests that hold the fused kernels to their
 * layer-by-layer forms.  Same conventions as iago_hip.h; part of the library's ABI (iago_abi_version).
 */
#ifndef IAGO_HIP_LAYERS_H
#define IAGO_HIP_LAYERS_H

#include "iago_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * In place x[b][c][:] = max(x[b][c][:] + bias[c], 0) on a float32 NCHW tensor
 * with 8x8 planes (hw = 64): the bias + ReLU epilogue of Block.__call__
 * (network.py:9-13: Convolution2D with bias, then F.relu) as ONE pass, for the
 * inference path of the PyTorch modules (MIOpen's convolution is called
 * without bias; PyTorch would otherwise run a bias-add and a ReLU kernel).
 * x: [n][channels][64] floats, 16-byte aligned; bias: [channels].
 */
IAGO_API int iago_bias_relu(float *x, const float *bias, int64_t n, int32_t channels, void *stream);

/*
 * The 3x3 convolution + bias + ReLU of Block.__call__ (network.py:5-13) on 8x8 boards
 * for the inference path of the Value net (network.py:66-96; evaluated once per
 * playout, MCTS.py:110-131), on the MFMA units in "split f16" arithmetic: every
 * float32 operand a is carried as a_hi = f16(a), a_lo = f16((a - a_hi) * 2^11), a
 * product sum is w_hi*x_hi + 2^-11 * (w_hi*x_lo + w_lo*x_hi) with float32
 * accumulation (22-bit products; the whole Value forward stays within 1e-6 of the
 * float32 one).
 *
 * Activations between layers are "split channel blocks": two f16 arrays
 * hi, lo [n][channels/16][64 cells][16 channels].  iago_split_nchw /
 * iago_merge_nchw convert from / to float32 NCHW [n][channels][8][8].
 * Weights: two f16 arrays [cin/16][3][3][cout][16] (kernel row, kernel column, output
 * channel, input channel within the block) split the same way; bias float32 [cout].
 * cout must be 128, cin a multiple of 32.  All pointers 16-byte aligned.
 *
 * Range: an f16 "hi" part holds |a| <= 65504, so activations are clamped to [0, 65000]
 * (inputs of iago_split_nchw to [-65000, 65000]) -- the float32 reference has no such
 * bound.  `overflow` (every function below that writes split channel blocks; optional, NULL
 * = no report) is a device word the kernel sets to 1 when a value was outside that range
 * or NaN BEFORE the clamp, i.e. when the results of this call are saturated and no longer
 * the reference's.  The caller zeroes it, reads it at its next synchronisation point and
 * falls back to the float32 kernels (iago_conv3x3_f32 / MIOpen) or raises.
 */
IAGO_API int iago_conv3x3_split(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo,
                                const float *bias, void *y_hi, void *y_lo, int64_t n, int32_t cin,
                                int32_t cout, uint32_t *overflow, void *stream);
/*
 * The ends of the Value net around the split-f16 convolutions, in float32 arithmetic:
 * iago_value_stem: block1 = conv3x3 2 -> 64 + bias + ReLU (network.py:68-70) from the
 *   float32 planes [n][2][8][8] of iago_encode_planes to split channel blocks
 *   [n][4][64][16]; w1 [64][2][3][3], b1 [64] as the reference stores them.
 * iago_value_head: block9 = conv3x3 128 -> 1 + bias + ReLU, fc10 (64 -> 128, no bias),
 *   fc11 (128 -> 1, no bias) with train=False (network.py:78-96; MCTS.py:86), from
 *   split channel blocks [n][8][64][16] to out [n]; w9 [1][128][3][3], b9 [1],
 *   w10 [128][64], w11 [1][128].
 */
IAGO_API int iago_value_stem(const float *planes, const float *w1, const float *b1, void *y_hi, void *y_lo,
                             int64_t n, uint32_t *overflow, void *stream);
/* iago_value_stem straight from the boards (own = side to move): iago_encode_planes fused in. */
IAGO_API int iago_value_stem_boards(const uint64_t *own, const uint64_t *opp, const float *w1, const float *b1,
                                    void *y_hi, void *y_lo, int64_t n, uint32_t *overflow, void *stream);
IAGO_API int iago_value_head(const void *x_hi, const void *x_lo, const float *w9, const float *b9,
                             const float *w10, const float *w11, float *out, int64_t n, void *stream);
/*
 * float32 convolutions for SMALL batches: the policy net on the expansions of a playout
 * (MCTS.py:109-121 evaluates SLPolicy on the few games whose leaf reached n_thr visits).
 * Exact float32 products on the matrix units, one board spread over 4 workgroups.
 * iago_conv3x3_f32: y = relu(conv3x3(x, w) + bias) (Block.__call__, network.py:9-13),
 *   x [n][cin][8][8], y [n][128][8][8] float32; cin 64 or 128, cout 128; w re-laid as
 *   [4 groups of 32 output channels][9 taps][cin][32] float32.
 * iago_stem_f32: SLPolicy.block1, conv3x3 2 -> 64 + bias + ReLU, planes [n][2][8][8] ->
 *   y [n][64][8][8]; w1 [64][2][3][3].
 * iago_policy_head: conv9 (1x1, 128 -> 1, no bias) + bias10 + softmax (network.py:29-47):
 *   x [n][128][8][8] -> probs [n][64]; w9 [128], b10 [64].
 * n_dev: optional device-side board count (see iago_encode_planes_indexed); iago_conv3x3_f32
 *   then runs a fixed grid whose workgroups walk the (board, channel group) items.
 */
IAGO_API int iago_conv3x3_f32(const float *x, const float *w, const float *bias, float *y, int64_t n,
                              int32_t cin, int32_t cout, const int32_t *n_dev, void *stream);
IAGO_API int iago_stem_f32(const float *planes, const float *w1, const float *b1, float *y, int64_t n,
                           const int32_t *n_dev, void *stream);
/* iago_stem_f32 straight from the boards: row b of y is block1 of board index[b] (int64
 * gather list, NULL = identity; own = side to move) -- iago_encode_planes_indexed fused in:
 * what a playout runs on the leaves it expands. */
IAGO_API int iago_stem_f32_boards(const uint64_t *own, const uint64_t *opp, const int64_t *index,
                                  const float *w1, const float *b1, float *y, int64_t n,
                                  const int32_t *n_dev, void *stream);
IAGO_API int iago_policy_head(const float *x, const float *w9, const float *b10, float *probs, int64_t n,
                              const int32_t *n_dev, void *stream);

/*
 * Up to 8 consecutive iago_conv3x3_split layers in ONE launch (blocks 2..8 of the Value
 * net): a workgroup owns all 128 channels of its 4 boards, so it runs the layers back to
 * back on its own intermediate activations.  Layer k reads the y buffers of layer k-1;
 * every layer writes buffers of its own.  cout = 128 throughout, cin of layer 0 a
 * multiple of 32.
 */
typedef struct iago_conv_split_layer {
    const void *x_hi, *x_lo; /* input activations  [n][cin/16][64][16] f16 */
    const void *w_hi, *w_lo; /* weights            [cin/16][3][3][128][16] f16 */
    const float *bias;       /* [128] */
    void *y_hi, *y_lo;       /* output activations [n][8][64][16] f16 */
    int32_t cin;
    int32_t reserved;
} iago_conv_split_layer;
IAGO_API int iago_conv3x3_split_trunk(const iago_conv_split_layer *layers, int32_t n_layers, int64_t n,
                                      uint32_t *overflow, void *stream);

/* float32 NCHW [n][channels][8][8] <-> split channel blocks (above) */
IAGO_API int iago_split_nchw(const float *x, void *hi, void *lo, int64_t n, int32_t channels,
                             uint32_t *overflow, void *stream);
IAGO_API int iago_merge_nchw(const void *hi, const void *lo, float *y, int64_t n, int32_t channels,
                             void *stream);

/*
 * The layer-level pieces of iago_policy_reinforce_grad (iago_hip.h).
 *
 * A gradient tensor in split channel blocks carries a power-of-two scale: its hi / lo pieces hold dY * 2^e with e an
 * int32 device word chosen so that the largest element sits near 2^14 (the f16 pieces then hold 22 bits of every
 * element down to 2^-28 of the largest).
 *
 * iago_conv3x3_wgrad_split: the weight gradient of one 3x3 block (Block.__call__, network.py:5-13),
 *   dW[co][ci][ky][kx] = 2^-e * sum over boards and cells of dY[b][co][y][x] * X[b][ci][y + ky - 1][x + kx - 1],
 *   dy_hi / dy_lo [n][8][64][16] (the gradient at the block's pre-activations, zero where its ReLU was off, times
 *   2^e), x_hi / x_lo [n][cin/16][64][16] (the block's input), cin 64 or 128; dw [128][cin][3][3] float32.
 *   part: scratch, [groups][9][128][cin] float32 -- the boards are summed in `groups` (a multiple of 8) contiguous
 *   groups, whose partial sums are added in group order (deterministic); scale_exp: the device word e, NULL = 0;
 *   dw = NULL: the partial sums only.
 *
 * iago_conv3x3_bwd_data_split: the gradient at the INPUT of a 3x3 block, through the ReLU of the block below:
 *   dx[b][ci][y][x] = [saved[b][ci][y][x] > 0] * 2^-e * sum over co and taps of dY[b][co][y - ky + 1][x - kx + 1] *
 *   W[co][ci][ky][kx] -- iago_conv3x3_split's kernel on the transposed, flipped weights (wt_hi / wt_lo: the split of
 *   Wt[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx], rows ci >= out_channels zero) with a float32 epilogue.
 *   mask_hi / mask_lo [n][out_channels/16][64][16]: the saved activations of the block below (its output = this
 *   block's input); dx [n][out_channels/16][64][16] float32 channel blocks; out_channels 64 or 128.
 *   max_bits: device word, atomicMax of the bit patterns of |dx| (zero it before the call).
 * iago_split_scaled: float32 channel blocks -> split channel blocks times 2^e with e = 13 - exponent of the largest
 *   magnitude (*max_bits, as written by the call above); writes e to *scale_exp.  bias_part (optional): scratch of
 *   ceil(n * channels / 32) * 32 floats, the sums over the cells per (board, channel); bias_grad (optional, needs
 *   bias_part): their sums over the boards per channel [channels] -- the bias gradient of the block whose
 *   pre-activation gradient x is, from the same pass (bias_part alone: the caller reduces, as
 *   iago_policy_reinforce_grad does for all blocks in one launch).
 */
IAGO_API int iago_conv3x3_bwd_data_split(const void *dy_hi, const void *dy_lo, const int32_t *scale_exp,
                                         const void *wt_hi, const void *wt_lo, const void *mask_hi,
                                         const void *mask_lo, int32_t out_channels, float *dx, uint32_t *max_bits,
                                         int64_t n, void *stream);
IAGO_API int iago_split_scaled(const float *x, const uint32_t *max_bits, void *hi, void *lo, int32_t *scale_exp,
                               int64_t n, int32_t channels, float *bias_part, float *bias_grad, void *stream);
IAGO_API int iago_conv3x3_wgrad_split(const void *dy_hi, const void *dy_lo, const void *x_hi, const void *x_lo,
                                      int64_t n, int32_t cin, float *part, int32_t groups, const int32_t *scale_exp,
                                      float *dw, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IAGO_HIP_LAYERS_H */
