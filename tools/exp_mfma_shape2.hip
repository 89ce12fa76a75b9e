// exp_mfma_shape2.hip -- v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 in the access pattern of the
// LDS-resident walks of the nets (csrc/conv_trunk_body.hpp), with CLEAN code for both shapes.
//
// Round 2's tools/exp_mfma_shape.hip had measured the short shape slower (1,291 against 1,470 TFLOP/s); its ISA shows
// why: with float4 accumulators through the builtin hipcc gives every v_mfma_f32_16x16x32_f16 a destination that is
// not its source C and moves the accumulators around them (7 v_accvgpr_* per MFMA).  Here the short shape goes
// through inline asm with the accumulator tied ("+v"): no moves, in place.  Both variants: one wave per SIMD
// (512-register budget), A operands streamed from an L2-resident weight set by 16-byte loads one k-step ahead, B
// operands by ds_read_b128 from a bank-conflict-free LDS image, the Value net's 3 MFMAs per product (hi x hi, hi x
// lo, lo x hi), random f16 data, 468 k16-steps = one board-pair... per "walk", every CU busy.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_mfma_shape2 tools/exp_mfma_shape2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline uint32_t hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// two random f16 in [0.5, 2) with random signs
__device__ inline uint32_t rnd2(uint32_t s)
{
    const uint32_t h = hash(s);
    return ((h & 0x83ff83ffu) | 0x38003800u) + ((h >> 5) & 0x04000400u);
}
__global__ void fill(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = rnd2((uint32_t)i * 2654435761u + 17u);
}

#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

constexpr int STEPS16 = 468;                  // k16-steps of a walk (blocks 2..8 of the nets)
constexpr int W_STRIDE = 256;                 // u32x4 per k16-step and piece: 128 channels x 32 B

// TB boards per walk: N = 64 TB cells
template <int SHAPE, int TB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const u32x4 *wh, const u32x4 *wl,
                                                                                      float *out, int walks)
{
    extern __shared__ __align__(16) char lds[];
    for (int i = threadIdx.x; i < 144 * 1024 / 4; i += 256)
        ((uint32_t *)lds)[i] = rnd2(i * 7919u + blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f;
    if (SHAPE == 32) {
        // the product's loop: lane = (output channel r, k half h); a B tile = 32 cells
        constexpr int NT = 2 * TB;
        const int r = lane & 31, h = lane >> 5;
        const char *base = lds + r * 528 + h * 16;
        const u32x4 *ph = wh + (32 * wv + r) * 2 + h, *pl = wl + (32 * wv + r) * 2 + h;
        float16v am[NT], ac[NT];
        for (int t = 0; t < NT; t++)
            for (int v = 0; v < 16; v++) { am[t][v] = 0.f; ac[t][v] = 0.f; }
        for (int w = 0; w < walks; w++) {
            u32x4 a_hi[2], a_lo[2];
            a_hi[0] = ph[0];
            a_lo[0] = pl[0];
            for (int s0 = 0; s0 < STEPS16; s0 += 2)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int st = s0 + u, nx = st + 1 < STEPS16 ? st + 1 : st;
                    a_hi[u ^ 1] = ph[(size_t)nx * W_STRIDE];
                    a_lo[u ^ 1] = pl[(size_t)nx * W_STRIDE];
                    const half8 ah = __builtin_bit_cast(half8, a_hi[u]), al = __builtin_bit_cast(half8, a_lo[u]);
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const char *p = base + t * 32 * 528 + (st & 7) * 32;
                        const half8 bh = *(const half8 *)p, bl = *(const half8 *)(p + 256);
                        am[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, am[t], 0, 0, 0);
                        ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, ac[t], 0, 0, 0);
                        ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, ac[t], 0, 0, 0);
                    }
                }
        }
        for (int t = 0; t < NT; t++)
            for (int v = 0; v < 16; v++) s += am[t][v] + ac[t][v] * (1.0f / 2048.0f);
    } else {
        // the short shape: lane = (row or column l & 15, k quarter l >> 4 of a k32-step = two 16-channel chunks);
        // a wave's 32 output channels = 2 M tiles, a B tile = 16 cells (rows 544 B apart: 2 x 4 banks per cell)
        constexpr int NN = 4 * TB;
        const int c = lane & 15, kq = lane >> 4;
        const char *base = lds + c * 544 + kq * 16;
        // A: channel 32 wv + 16 m + c, chunk (kq >> 1) of the k32-step, half kq & 1
        const u32x4 *ph = wh + (32 * wv + c) * 2 + (kq & 1) + (kq >> 1) * W_STRIDE;
        const u32x4 *pl = wl + (32 * wv + c) * 2 + (kq & 1) + (kq >> 1) * W_STRIDE;
        float4v am[2][NN], ac[2][NN];
        for (int m = 0; m < 2; m++)
            for (int n = 0; n < NN; n++)
                for (int v = 0; v < 4; v++) { am[m][n][v] = 0.f; ac[m][n][v] = 0.f; }
        for (int w = 0; w < walks; w++) {
            u32x4 a_hi[2][2], a_lo[2][2];
            a_hi[0][0] = ph[0], a_hi[0][1] = ph[32];
            a_lo[0][0] = pl[0], a_lo[0][1] = pl[32];
            for (int s0 = 0; s0 < STEPS16 / 2; s0 += 2)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int st = s0 + u, nx = st + 1 < STEPS16 / 2 ? st + 1 : st;
                    a_hi[u ^ 1][0] = ph[(size_t)nx * 2 * W_STRIDE], a_hi[u ^ 1][1] = ph[(size_t)nx * 2 * W_STRIDE + 32];
                    a_lo[u ^ 1][0] = pl[(size_t)nx * 2 * W_STRIDE], a_lo[u ^ 1][1] = pl[(size_t)nx * 2 * W_STRIDE + 32];
                    half8 bh[NN], bl[NN];
#pragma unroll
                    for (int n = 0; n < NN; n++) {
                        const char *p = base + n * 16 * 544 + (st & 3) * 64;
                        bh[n] = *(const half8 *)p;
                        bl[n] = *(const half8 *)(p + 256);
                    }
                    const half8 ah0 = __builtin_bit_cast(half8, a_hi[u][0]), ah1 = __builtin_bit_cast(half8, a_hi[u][1]);
                    const half8 al0 = __builtin_bit_cast(half8, a_lo[u][0]), al1 = __builtin_bit_cast(half8, a_lo[u][1]);
#pragma unroll
                    for (int n = 0; n < NN; n++) {
                        MFMA16(am[0][n], ah0, bh[n]);
                        MFMA16(am[1][n], ah1, bh[n]);
                    }
#pragma unroll
                    for (int n = 0; n < NN; n++) {
                        MFMA16(ac[0][n], ah0, bl[n]);
                        MFMA16(ac[1][n], ah1, bl[n]);
                    }
#pragma unroll
                    for (int n = 0; n < NN; n++) {
                        MFMA16(ac[0][n], al0, bh[n]);
                        MFMA16(ac[1][n], al1, bh[n]);
                    }
                }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // (the asm MFMAs' results are read below)
        for (int m = 0; m < 2; m++)
            for (int n = 0; n < NN; n++)
                for (int v = 0; v < 4; v++) s += am[m][n][v] + ac[m][n][v] * (1.0f / 2048.0f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int TB> void run(int grid, const u32x4 *wh, const u32x4 *wl, float *out)
{
    const int walks = 200;
    hipFuncSetAttribute((const void *)k<SHAPE, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SHAPE, TB><<<grid, 256, 144 * 1024>>>(wh, wl, out, 20);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<SHAPE, TB><<<grid, 256, 144 * 1024>>>(wh, wl, out, walks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * walks * STEPS16 * (2 * TB) * 3 * 32768.0;
    printf("v_mfma_f32_%s_f16, %d board(s) per walk, grid %3d: %8.3f ms, %6.0f TFLOP/s executed, %6.2f us per walk (%5.2f per board)\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", TB, grid, best, flops / (best * 1e-3) / 1e12, best * 1e3 / walks,
           best * 1e3 / walks / TB);
}

int main()
{
    const size_t n16 = (size_t)(STEPS16 + 2) * W_STRIDE; // u32x4 per piece
    u32x4 *wh, *wl;
    float *out;
    hipMalloc(&wh, n16 * 16);
    hipMalloc(&wl, n16 * 16);
    hipMalloc(&out, 1024 * 256 * 4);
    fill<<<256, 256>>>((uint32_t *)wh, n16 * 4);
    fill<<<256, 256>>>((uint32_t *)wl, n16 * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        run<32, 1>(224, wh, wl, out);
        run<16, 1>(224, wh, wl, out);
        run<32, 2>(224, wh, wl, out);
        run<16, 2>(224, wh, wl, out);
    }
    run<32, 2>(1, wh, wl, out);
    run<16, 2>(1, wh, wl, out);
    return 0;
}
