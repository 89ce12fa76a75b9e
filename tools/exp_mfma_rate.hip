// exp_mfma_rate.hip -- issue rate of v_mfma_f32_32x32x16_f16 in the access pattern of the
// split-f16 convolution (8 + 8 accumulators of 16 registers, one wave per SIMD):
// bare, and with the 12 ds_read_b128 of a k-step in front of its 24 MFMAs.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_mfma_rate tools/exp_mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ __align__(16) char lds[64 * 1024];
    for (int i = threadIdx.x; i < 64 * 1024 / 16; i += 256)
        ((uint4 *)lds)[i] = make_uint4(i, 1, 2, 3);
    __syncthreads();
    float16v am[4][2], ac[4][2];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++)
            for (int v = 0; v < 16; v++) {
                am[i][j][v] = 0.f;
                ac[i][j][v] = 0.f;
            }
    half8 a_hi[4], a_lo[4], b_hi[2], b_lo[2];
    const int lane = threadIdx.x & 63;
    const char *base = lds + (lane & 31) * 48 + (lane >> 5) * 16;
    for (int i = 0; i < 4; i++) {
        a_hi[i] = *(const half8 *)(base + i * 1536);
        a_lo[i] = *(const half8 *)(base + 8192 + i * 1536);
    }
    for (int j = 0; j < 2; j++) {
        b_hi[j] = *(const half8 *)(base + 16384 + j * 1536);
        b_lo[j] = *(const half8 *)(base + 24576 + j * 1536);
    }
    for (int it = 0; it < iters; it++) {
        if (MODE >= 1) {
            const char *p = base + (it & 7) * 6144;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a_hi[i] = *(const half8 *)(p + i * 1536);
                a_lo[i] = *(const half8 *)(p + 8192 + i * 1536);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                b_hi[j] = *(const half8 *)(p + 16384 + j * 1536);
                b_lo[j] = *(const half8 *)(p + 24576 + j * 1536);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_hi[j], am[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[i], b_lo[j], ac[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo[i], b_hi[j], ac[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++)
            for (int v = 0; v < 16; v++)
                s += am[i][j][v] + ac[i][j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int grid, float *out)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 24.0 * iters;
    printf("%-28s grid %4d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), %.0f TFLOP/s\n", name, grid,
           ms, ms * 1e6 / mfma, ms * 1e6 / mfma * 2.4, grid * 4 * mfma * 32768.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *out;
    hipMalloc(&out, 1024 * 256 * 4);
    run<0>("bare MFMA", 1, out);
    run<0>("bare MFMA", 256, out);
    run<1>("12 ds_read_b128 + 24 MFMA", 1, out);
    run<1>("12 ds_read_b128 + 24 MFMA", 256, out);
    return 0;
}
