// exp_mfma_shape.hip -- v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 in the access
// pattern of the LDS-resident Value trunk: per B tile two ds_read_b128 (hi, lo parts, random f16
// data) feed 3 MFMAs (32x32x16: 32 output channels) or 6 (16x16x32: 2 x 16 output channels) --
// the same FLOPs and the same LDS bytes -- with the A operands in registers, one wave per SIMD,
// every CU busy.  Question: does the 16x16x32 shape sustain a higher clock under this load
// (MI355X_MICROARCH.md, DVFS item 7)?
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_mfma_shape tools/exp_mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hash(uint32_t x)
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

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ __align__(16) char lds[];
    for (int i = threadIdx.x; i < 128 * 1024 / 4; i += 256)
        ((uint32_t *)lds)[i] = rnd2(i * 7919u + blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // conflict-free read pattern: 16 B per lane, lanes contiguous
    const char *base = lds + lane * 16 + wv * 1024;
    half8 a_hi[2], a_lo[2];
    for (int i = 0; i < 2; i++) {
        uint32_t w[4], v[4];
        for (int q = 0; q < 4; q++) {
            w[q] = rnd2(lane * 131u + i * 17u + q);
            v[q] = rnd2(lane * 137u + i * 19u + q + 99u);
        }
        a_hi[i] = *(half8 *)w;
        a_lo[i] = *(half8 *)v;
    }
    float s = 0.f;
    if (SHAPE == 32) {
        float16v am[8], ac[8];
        for (int t = 0; t < 8; t++)
            for (int v = 0; v < 16; v++) { am[t][v] = 0.f; ac[t][v] = 0.f; }
        for (int it = 0; it < iters; it++) {
            const char *p = base + (it & 7) * 8192;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const half8 bh = *(const half8 *)(p + t * 4096 % 65536), bl = *(const half8 *)(p + 65536 + t * 4096 % 65536);
                am[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[0], bh, am[t], 0, 0, 0);
                ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[0], bl, ac[t], 0, 0, 0);
                ac[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[0], bh, ac[t], 0, 0, 0);
            }
        }
        for (int t = 0; t < 8; t++)
            for (int v = 0; v < 16; v++) s += am[t][v] + ac[t][v];
    } else {
        float4v am[8][2], ac[8][2];
        for (int t = 0; t < 8; t++)
            for (int r = 0; r < 2; r++)
                for (int v = 0; v < 4; v++) { am[t][r][v] = 0.f; ac[t][r][v] = 0.f; }
        for (int it = 0; it < iters; it++) {
            const char *p = base + (it & 7) * 8192;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const half8 bh = *(const half8 *)(p + t * 4096 % 65536), bl = *(const half8 *)(p + 65536 + t * 4096 % 65536);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    am[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[r], bh, am[t][r], 0, 0, 0);
                    ac[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[r], bl, ac[t][r], 0, 0, 0);
                    ac[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[r], bh, ac[t][r], 0, 0, 0);
                }
            }
        }
        for (int t = 0; t < 8; t++)
            for (int r = 0; r < 2; r++)
                for (int v = 0; v < 4; v++) s += am[t][r][v] + ac[t][r][v];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE> void run(int grid, float *out)
{
    const int iters = 20000;
    hipFuncSetAttribute((const void *)k<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SHAPE><<<grid, 256, 128 * 1024>>>(out, 2000);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<SHAPE><<<grid, 256, 128 * 1024>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * iters * 8 * 3 * 32768.0;
    printf("v_mfma_f32_%s_f16, grid %3d: %.3f ms, %.0f TFLOP/s executed, %.2f us per 1728-MFMA-equivalent layer\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", grid, best, flops / (best * 1e-3) / 1e12,
           best * 1e3 / (iters * 24.0) * 1728.0);
}

int main()
{
    float *out;
    hipMalloc(&out, 1024 * 256 * 4);
    for (int rep = 0; rep < 2; rep++) {
        run<32>(256, out);
        run<16>(256, out);
    }
    run<32>(1, out);
    run<16>(1, out);
    return 0;
}
