// conv_policy_kernel.hip -- the WHOLE SLPolicy net (network.py:15-47) in one launch on the f16
// matrix units with float32-exact products, for the batches the search evaluates it on (a few
// dozen to a few hundred positions: the leaves queued by the policy look-ahead).
//
// Why.  The policy's move distribution must match the reference within 1e-5 and the shipped net
// is near one-hot, so its convolutions have to be float32-accurate: the two-piece split of the
// Value net (22-bit operands) measured 1.3-1.7e-5.  float32 MFMA (v_mfma_f32_32x32x2_f32) runs
// at the VALU rate, 16x below f16 MFMA, and bounds conv3x3_f32_kernel (conv_kernels.hip): 40 us
// per 256-board layer.  Here every float32 operand is THREE f16 pieces,
//     a = hi + mid * 2^-11 + lo * 2^-22     (hi = f16(a), mid = f16((a - hi) 2^11), ...)
// -- 33 bits: exact for every float32 -- and a product sum is six MFMAs into three float32
// accumulators (hi.hi | hi.mid + mid.hi | hi.lo + lo.hi + mid.mid; the dropped terms are below
// 2^-33): each f16 x f16 product is exact in float32, so the result is a float32 convolution
// with its own summation order, at 16 / 6 = 2.7x the float32 MFMA rate.
//
// Structure: conv_trunk_kernel.hip's LDS-resident walk with ONE board per workgroup: the
// activations of the board (64 cell rows of 128 channels x 3 pieces = 800 B + zero area, 52 KB)
// stay in LDS from block1 to the head, the weights go from L2 straight into the A-operand
// registers (each wave its quarter of the output channels), the head (1x1 convolution, per-cell
// bias, softmax: network.py:29-47) is computed from LDS.  Rows come from own / opp through an
// optional gather list, their number from an optional device word; a workgroup walks the rows
// with the grid's stride.
#include "conv_policy_body.hpp"

#include <cstdlib>

namespace {
using namespace iago_policy;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void policy_resident_kernel(PolicyParams P)
{
    int64_t n_rows = P.row_hi;
    if (P.n_dev)
        n_rows = min(n_rows, (int64_t)*P.n_dev);
    for (int64_t row = P.row_lo + blockIdx.x; row < n_rows; row += gridDim.x) {
        policy_item(P, row);
        __syncthreads(); // the next pass re-stages the LDS image the head just read
    }
}

} // namespace

int iago_policy_forward_split3(const iago_policy_split3_args *a, void *stream)
{
    if (a && a->n == 0)
        return IAGO_OK;
    PolicyParams P;
    if (const int rc = policy_params_of(a, P))
        return rc;
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)policy_resident_kernel, LDS_BYTES, configured,
                         "iago_policy_forward_split3: cannot reserve 52 KB of LDS"))
        return IAGO_ERR_HIP;
    // one board per workgroup; at most one workgroup per CU, walking the rows with the grid's stride
    static const int64_t cap = []() {
        const char *e = getenv("IAGO_POLICY_GRID"); // tuning knob (tools/)
        const long v = e ? atol(e) : 256;
        return (int64_t)(v < 1 ? 1 : v > 1024 ? 1024 : v);
    }();

    // parts > 1: the 7 convolution blocks as that many launches of 7 / parts blocks each (short
    // launches leave the CUs to the other stream's kernels sooner), the boards' LDS images
    // travelling through `scratch`
    const int parts = a->parts < 1 ? 1 : (a->parts > 7 ? 7 : a->parts);
    if (parts > 1 && (!a->scratch || ((uintptr_t)a->scratch & 15u)))
        return iago_fail(IAGO_ERR_INVALID, "iago_policy_forward_split3: parts > 1 needs a 16-byte aligned scratch "
                                           "buffer of n x 51,200 bytes");
    P.scratch = (uint4 *)a->scratch;
    // scratch_rows > 0: the scratch holds that many rows; a longer batch runs as chunks of
    // scratch_rows rows, each chunk its `parts` launches (launches of one stream run in order, so
    // the chunks may share the buffer).  With a device-side count the chunks past it exit at once
    const int64_t chunk = (parts > 1 && a->scratch_rows > 0 && a->scratch_rows < a->n) ? a->scratch_rows : a->n;
    for (int64_t lo = 0; lo < a->n; lo += chunk) {
        P.row_lo = lo;
        P.row_hi = lo + chunk < a->n ? lo + chunk : a->n;
        const int64_t rows = P.row_hi - P.row_lo;
        const unsigned grid = (unsigned)(rows < cap ? rows : cap);
        for (int p = 0; p < parts; p++) {
            P.layer_lo = 7 * p / parts;
            P.layer_hi = 7 * (p + 1) / parts;
            hipLaunchKernelGGL(policy_resident_kernel, dim3(grid), dim3(256), LDS_BYTES, (hipStream_t)stream, P);
        }
    }
    return iago_check_launch("iago_policy_forward_split3");
}
