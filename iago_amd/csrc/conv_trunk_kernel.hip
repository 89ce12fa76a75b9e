// conv_trunk_kernel.hip -- blocks 2..8 of the Value net (network.py:66-96) in ONE launch with
// the activations RESIDENT IN LDS between the layers.
//
// conv_kernels.hip's per-layer kernel (and the round-1 trunk built from it) stages both
// operands through LDS: 89 KB of padded activation planes + 74 KB of weight rows, double
// buffered, one barrier per stage -- and between two layers a workgroup stores its 128 KB of
// activations to global memory, fences, and fetches them back (~10 us of the 46 us a layer
// took, with nothing to overlap it: 159 KB of LDS = one workgroup per CU).  Measured with
// rocprofv3 (profiles/r02a_mcts_pmc_summary.json): MFMA pipes 56 % busy.
//
// Here the LDS holds nothing but the activations of the workgroup's 4 boards, all 128
// channels, hi and lo parts: T[board][64 cell rows of 528 B + 768 zero bytes] = 135 KB (the
// zero bytes behind a board are the target of every out-of-board tap).  The weights never touch LDS: in the layout
// [cin/16][ky][kx][cout][16] the MFMA A-operand of a lane (output channel r, 8 input
// channels) is 16 contiguous bytes and a wave's 32 channels are 1 KB, so every wave streams
// its OWN quarter of the output channels straight from L2 into registers, three k-steps
// ahead -- no two waves of a workgroup load the same bytes.  A wave (one per SIMD, the whole
// register file) owns 32 output channels x all 256 cells of the 4 boards: 8 tiles of
// v_mfma_f32_32x32x16_f16, 2 x 128 accumulator registers.  The B operands are read from T
// through 36 per-lane tap addresses (a tap that leaves the board points at the zero row),
// advanced by 32 B per 16-channel chunk.  The K loop has no barrier; a layer ends with
// barrier - epilogue (bias, ReLU, re-split, written back into T in place) - barrier.
// Same products in the same order as the per-layer kernel: bit-identical results
// (tests/test_conv_gpu.py: test_trunk_kernel_equals_layer_by_layer).
#include "conv_trunk_body.hpp"

namespace {
using namespace iago_trunk;

template <bool FUSED, int TB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void trunk_resident_kernel(TrunkRParams P)
{
    trunk_walk<FUSED, TB>(P, whole_walk(P), blockIdx.x, gridDim.x);
}

// The leaf evaluation of a playout (MCTS.py:123-125) in ONE launch: workgroups 0 .. n_ro-1 play
// the rollouts of ALL leaves (the 16-lanes-per-board kernel's body), the others run the value net
// on the leaves that have no stored value (one board per workgroup, device-side list).  With the
// value cache the net covers ~180 CUs and the rollouts' 64 workgroups the rest: two kernels on
// two streams cost more in graph edges than they hide, one launch with two kinds of
// workgroups costs nothing.  The rollout workgroups come first in the grid: they are dispatched
// first and never wait behind a value workgroup.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void value_rollout_kernel(
    TrunkRParams P, iago_row::HwParams R, uint32_t n_ro)
{
    if (blockIdx.x < n_ro) {
        iago_row::rollout_row_body<false>(R, blockIdx.x);
        return;
    }
    // one board per workgroup while the rows fit the value workgroups in one pass; more rows (a
    // few playouts in a hundred): two boards per workgroup share the weight stream, which bounds
    // the one-board walk -- one pass up to twice as many rows
    const int64_t nb = (int64_t)gridDim.x - n_ro;
    const int64_t n_rows = min(P.n, (int64_t)*P.n_dev);
    if (n_rows <= nb)
        trunk_walk<true, 1>(P, whole_walk(P), blockIdx.x - n_ro, nb);
    else
        trunk_walk<true, 2>(P, whole_walk(P), blockIdx.x - n_ro, nb);
}

// The leaf evaluation of one GAME-ASYNCHRONOUS step (iago_value_rollout_async, include/iago_hip.h)
// in one launch: workgroups 0 .. n_ro-1 play the rollouts of the games that descended in this step
// (mask, per-game Philox stream ids); then, oldest queue first, `parts` groups of NV workgroups:
// group p walks the leaves queued p steps ago through piece p of the Value net (block1 + the first
// layers | ... | the last layers + the head), a board's rows parked in the queue row's scratch
// between two pieces.  Every workgroup's work is a fraction of a whole walk, so the launch is as
// long as ONE piece (or one rollout), not as one walk: the value net has left the steps' critical
// chain.  One board per workgroup while a queue's rows fit its NV workgroups, two above (the pair
// shares the weight stream).
// value workgroups per piece (parts * NV + the rollouts' 64 <= 256 CUs + slack; tuning knob IAGO_ASYNC_NV: with
// the value look-ahead few leaves are evaluated in place, and every workgroup of this launch needs a CU
// of its own to start on, beside the look-ahead's batches)
static int iago_async_nv()
{
    static const int v = [] {
        const char *e = getenv("IAGO_ASYNC_NV");
        const int n = e ? atoi(e) : 64;
        return n < 1 ? 1 : (n > 64 ? 64 : n);
    }();
    return v;
}
struct AsyncParams {
    int32_t parts;
    const int64_t *fq_index; // [parts][n]
    const int32_t *fq_count; // [parts]
    const uint32_t *step;
    char *scratch;           // [parts][n][IMG]
    uint64_t bounds;         // byte p = first layer of piece p (byte `parts` = 7): piece p = layers [b[p], b[p + 1])
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void value_rollout_async_kernel(
    TrunkRParams P, iago_row::HwParams R, AsyncParams Y, uint32_t n_ro, uint32_t ASYNC_NV)
{
    if (blockIdx.x < n_ro) {
        iago_row::rollout_row_body<false, true>(R, blockIdx.x);
        return;
    }
    const uint32_t x = blockIdx.x - n_ro;
    const int part = Y.parts - 1 - (int)(x / ASYNC_NV); // the oldest queue's workgroups are dispatched first
    const int64_t bid = x % ASYNC_NV;
    const uint32_t row = (*Y.step + (uint32_t)(Y.parts - part)) % (uint32_t)Y.parts; // == (step - part) mod parts
    Piece W = whole_walk(P);
    W.index = Y.fq_index + (int64_t)row * P.n;
    W.n_dev = Y.fq_count + row;
    W.scratch = Y.scratch + (int64_t)row * P.n * IMG;
    W.layer_lo = (Y.bounds >> (8 * part)) & 0xFF;
    W.layer_hi = (Y.bounds >> (8 * part + 8)) & 0xFF;
    const int64_t n_rows = min(P.n, (int64_t)*W.n_dev);
    if (n_rows <= ASYNC_NV)
        trunk_walk<true, 1>(P, W, bid, ASYNC_NV);
    else
        trunk_walk<true, 2>(P, W, bid, ASYNC_NV);
}

} // namespace

// Called by iago_conv3x3_split_trunk (conv_kernels.hip) after it has validated the layers.
int iago_launch_trunk_resident(const iago_conv_split_layer *layers, int32_t n_layers, int64_t n, uint32_t *overflow,
                               void *stream)
{
    TrunkRParams P;
    P.x_hi = (const uint4 *)layers[0].x_hi;
    P.x_lo = (const uint4 *)layers[0].x_lo;
    P.y_hi = (uint4 *)layers[n_layers - 1].y_hi;
    P.y_lo = (uint4 *)layers[n_layers - 1].y_lo;
    for (int L = 0; L < MAX_LAYERS; L++) {
        const iago_conv_split_layer &a = layers[L < n_layers ? L : n_layers - 1];
        P.w_hi[L] = (const uint4 *)a.w_hi;
        P.w_lo[L] = (const uint4 *)a.w_lo;
        P.bias[L] = a.bias;
    }
    for (int L = 0; L < n_layers; L++)
        if (((uintptr_t)layers[L].bias & 15u) || ((uintptr_t)layers[L].w_hi & 15u) || ((uintptr_t)layers[L].w_lo & 15u))
            return iago_fail(IAGO_ERR_INVALID, "iago_conv3x3_split_trunk: weights and biases must be 16-byte aligned");
    P.n = n;
    P.cin0 = layers[0].cin;
    P.n_layers = n_layers;
    P.overflow = overflow;
    P.planes = nullptr;
    P.own = P.opp = nullptr;
    P.w1 = P.b1 = P.b9 = P.w10 = P.w11 = nullptr;
    P.w9_hi = P.w9_lo = nullptr;
    P.out = nullptr;
    P.index = nullptr;
    P.n_dev = nullptr;
    P.count_lo = 0;
    P.count_hi = 0x7fffffff;
    P.layer_lo = 0;
    P.layer_hi = n_layers;
    P.scratch = nullptr;
    constexpr int TB = 4; // boards per workgroup
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)trunk_resident_kernel<false, TB>, lds_alloc(TB), configured,
                         "iago_conv3x3_split_trunk: cannot reserve 134 KB of LDS"))
        return IAGO_ERR_HIP;
    const unsigned grid = (unsigned)((n + TB - 1) / TB);
    hipLaunchKernelGGL((trunk_resident_kernel<false, TB>), dim3(grid), dim3(256), lds_alloc(TB), (hipStream_t)stream, P);
    return iago_check_launch("iago_conv3x3_split_trunk");
}

namespace {
// rows up to which a workgroup takes ONE board (tuning knob: IAGO_VALUE_TINY, default 256)
int64_t iago_value_tiny_rows()
{
    static const int64_t v = [] {
        const char *e = getenv("IAGO_VALUE_TINY");
        return e ? (int64_t)atoll(e) : (int64_t)256;
    }();
    return v;
}
// device-counted launches as one persistent launch of the one-board variant (IAGO_VALUE_PERSIST=0:
// the three windowed variants)
bool iago_value_persistent()
{
    static const bool v = [] {
        const char *e = getenv("IAGO_VALUE_PERSIST");
        return !(e && e[0] == '0');
    }();
    return v;
}
} // namespace


int iago_value_forward_split(const iago_value_split_args *a, void *stream)
{
    if (a && a->n == 0)
        return IAGO_OK;
    TrunkRParams P;
    if (const int rc = value_params_of(a, P))
        return rc;
    static std::atomic<uint64_t> configured4{0}, configured2{0}, configured1{0};
    if (iago_reserve_lds((const void *)trunk_resident_kernel<true, 4>, lds_alloc_fused(4), configured4,
                         "iago_value_forward_split: cannot reserve 148 KB of LDS") ||
        iago_reserve_lds((const void *)trunk_resident_kernel<true, 2>, lds_alloc_fused(2), configured2,
                         "iago_value_forward_split: cannot reserve 75 KB of LDS") ||
        iago_reserve_lds((const void *)trunk_resident_kernel<true, 1>, lds_alloc_fused(1), configured1,
                         "iago_value_forward_split: cannot reserve 39 KB of LDS"))
        return IAGO_ERR_HIP;
    // Boards per workgroup by the number of rows: the launch is latency-bound until the chip is
    // full, so fewer boards per workgroup (more workgroups, each with a shorter chain) win for
    // small batches: one up to TINY rows, two up to SMALL, four above.  Same products in the same
    // order per board: bit-identical values.  With a device-side count every variant whose
    // window the bound n reaches is enqueued and the one that holds the count runs.
    const int64_t TINY = iago_value_tiny_rows(), SMALL = 512;
    const bool host_known = a->n_dev == nullptr;
    const int64_t rows = a->n;
    if (!host_known && iago_value_persistent()) {
        // Device-side count (the value cache: usually 15-20 % of the games have a leaf without a
        // value): ONE launch of the one-board variant on a grid of at most one workgroup per CU,
        // walking the rows with the grid's stride -- one pass up to 256 rows; the rare larger
        // counts take more passes instead of two more (mostly empty) launches per playout.
        P.count_lo = 0;
        P.count_hi = 0x7fffffff;
        const int64_t grid = rows < 256 ? rows : 256;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 1>), dim3((unsigned)grid), dim3(256), lds_alloc_fused(1),
                           (hipStream_t)stream, P);
        return iago_check_launch("iago_value_forward_split");
    }
    if (!host_known || rows <= TINY) {
        P.count_lo = 0;
        P.count_hi = (int32_t)TINY;
        const int64_t r = rows < TINY ? rows : TINY;
        if (TINY > 0)
            hipLaunchKernelGGL((trunk_resident_kernel<true, 1>), dim3((unsigned)r), dim3(256), lds_alloc_fused(1),
                               (hipStream_t)stream, P);
    }
    if (rows > TINY && (!host_known || rows <= SMALL)) {
        P.count_lo = host_known ? 0 : (int32_t)TINY;
        P.count_hi = (int32_t)SMALL;
        const int64_t r = rows < SMALL ? rows : SMALL;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 2>), dim3((unsigned)((r + 1) / 2)), dim3(256),
                           lds_alloc_fused(2), (hipStream_t)stream, P);
    }
    if (rows > SMALL) {
        P.count_lo = host_known ? 0 : (int32_t)SMALL;
        P.count_hi = 0x7fffffff;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(256),
                           lds_alloc_fused(4), (hipStream_t)stream, P);
    }
    return iago_check_launch("iago_value_forward_split");
}

int iago_value_forward_batch(const iago_value_split_args *a, int32_t boards_per_workgroup, int32_t max_workgroups,
                             void *stream)
{
    if (a && a->n == 0)
        return IAGO_OK;
    if (!a || !a->n_dev || max_workgroups < 1 ||
        (boards_per_workgroup != 1 && boards_per_workgroup != 2 && boards_per_workgroup != 4))
        return iago_fail(IAGO_ERR_INVALID, "iago_value_forward_batch: a device-side row count, 1 / 2 / 4 boards per "
                                           "workgroup and max_workgroups >= 1 expected");
    TrunkRParams P;
    if (const int rc = value_params_of(a, P))
        return rc;
    P.count_lo = 0;
    P.count_hi = 0x7fffffff;
    static std::atomic<uint64_t> configured4{0}, configured2{0}, configured1{0};
    const int tb = boards_per_workgroup;
    int64_t grid = (a->n + tb - 1) / tb;
    if (grid > max_workgroups)
        grid = max_workgroups;
    if (tb == 4) {
        if (iago_reserve_lds((const void *)trunk_resident_kernel<true, 4>, lds_alloc_fused(4), configured4,
                             "iago_value_forward_batch: cannot reserve 148 KB of LDS"))
            return IAGO_ERR_HIP;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 4>), dim3((unsigned)grid), dim3(256), lds_alloc_fused(4),
                           (hipStream_t)stream, P);
    } else if (tb == 2) {
        if (iago_reserve_lds((const void *)trunk_resident_kernel<true, 2>, lds_alloc_fused(2), configured2,
                             "iago_value_forward_batch: cannot reserve 75 KB of LDS"))
            return IAGO_ERR_HIP;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 2>), dim3((unsigned)grid), dim3(256), lds_alloc_fused(2),
                           (hipStream_t)stream, P);
    } else {
        if (iago_reserve_lds((const void *)trunk_resident_kernel<true, 1>, lds_alloc_fused(1), configured1,
                             "iago_value_forward_batch: cannot reserve 39 KB of LDS"))
            return IAGO_ERR_HIP;
        hipLaunchKernelGGL((trunk_resident_kernel<true, 1>), dim3((unsigned)grid), dim3(256), lds_alloc_fused(1),
                           (hipStream_t)stream, P);
    }
    return iago_check_launch("iago_value_forward_batch");
}

int iago_value_rollout(const iago_value_split_args *a, const iago_rollout_args *ro, void *stream)
{
    if (!a || !ro)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout: null args");
    if (!a->n_dev || !a->index || a->n < 1)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout: the value net takes its rows from a device-side list "
                                           "(index, n_dev)");
    if (ro->n < 1 || ro->n > 0x7fffffffll || !ro->own || !ro->opp || !ro->z || !ro->table || ((uintptr_t)ro->table & 15u) ||
        ro->log_form || ro->trace || ro->uniforms || ro->throughput_hint != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout: product-form rollout without trace / uniforms expected");
    TrunkRParams P;
    if (const int rc = value_params_of(a, P))
        return rc;
    P.count_lo = 0;
    P.count_hi = 0x7fffffff;
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)value_rollout_kernel, lds_alloc_fused(2), configured,
                         "iago_value_rollout: cannot reserve 80 KB of LDS"))
        return IAGO_ERR_HIP;
    const iago_row::HwParams R = iago_row::hw_params_of(ro);
    const unsigned n_ro = (unsigned)((ro->n + (iago_row::HW_BLOCK / 16) - 1) / (iago_row::HW_BLOCK / 16));
    const unsigned n_val = (unsigned)(a->n < 256 ? a->n : 256);
    hipLaunchKernelGGL(value_rollout_kernel, dim3(n_ro + n_val), dim3(256), lds_alloc_fused(2), (hipStream_t)stream, P, R,
                       n_ro);
    return iago_check_launch("iago_value_rollout");
}

int iago_value_rollout_async(const iago_value_split_args *a, const iago_rollout_args *ro, const iago_mcts_async *y,
                             void *stream)
{
    if (!a || !ro || !y)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout_async: null args");
    if (y->parts < 2 || y->parts > IAGO_ASYNC_MAX_PARTS || !y->wait || !y->done || !y->roll || !y->fq_index ||
        !y->fq_count || !y->step || !y->n_sims || !y->scratch || ((uintptr_t)y->scratch & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout_async: incomplete iago_mcts_async (scratch 16-byte aligned)");
    if (a->n < 1 || !a->own || !a->opp || a->planes)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout_async: the value net takes the boards (own, opp) of "
                                           "the n games");
    if (ro->n != a->n || ro->n > 0x7fffffffll || !ro->own || !ro->opp || !ro->z || !ro->table ||
        ((uintptr_t)ro->table & 15u) || ro->log_form || ro->trace || ro->uniforms || ro->throughput_hint != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_value_rollout_async: product-form rollout of the same n games without "
                                           "trace / uniforms expected");
    iago_value_split_args a2 = *a;
    a2.index = nullptr;
    a2.n_dev = nullptr;
    TrunkRParams P;
    if (const int rc = value_params_of(&a2, P))
        return rc;
    P.count_lo = 0;
    P.count_hi = 0x7fffffff;
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)value_rollout_async_kernel, lds_alloc_fused(2), configured,
                         "iago_value_rollout_async: cannot reserve 80 KB of LDS"))
        return IAGO_ERR_HIP;
    iago_row::HwParams R = iago_row::hw_params_of(ro);
    R.mask = y->roll;
    R.stream_ids = y->done;
    AsyncParams Y;
    Y.parts = y->parts;
    Y.fq_index = y->fq_index;
    Y.fq_count = y->fq_count;
    Y.step = y->step;
    Y.scratch = (char *)y->scratch;
    // pieces of (nearly) equal work in units of a 128 -> 128 layer: block1 0.3, block 2 (64 -> 128) 0.5,
    // blocks 3..8 1 each, head 0.3
    static const int8_t B2[3] = {0, 4, 7}, B3[4] = {0, 3, 5, 7}, B4[5] = {0, 2, 4, 6, 7};
    const int8_t *B = y->parts == 2 ? B2 : y->parts == 3 ? B3 : B4;
    Y.bounds = 0;
    for (int i = 0; i <= y->parts; i++)
        Y.bounds |= (uint64_t)(uint8_t)B[i] << (8 * i);
    const unsigned n_ro = (unsigned)((ro->n + (iago_row::HW_BLOCK / 16) - 1) / (iago_row::HW_BLOCK / 16));
    const unsigned nv = (unsigned)iago_async_nv();
    hipLaunchKernelGGL(value_rollout_async_kernel, dim3(n_ro + (unsigned)y->parts * nv), dim3(256),
                       lds_alloc_fused(2), (hipStream_t)stream, P, R, Y, n_ro, nv);
    return iago_check_launch("iago_value_rollout_async");
}
