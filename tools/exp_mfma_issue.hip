// exp_mfma_issue.hip -- cycles per v_mfma_f32_16x16x32_f16 on ONE SIMD for the instruction patterns of the walks' K loops:
// independent accumulators, the same accumulator again after 1 / 3 / 5 other MFMAs, with one or two ds_read_b128 per six
// MFMAs (bunched or one per gap), with v_add / s_nop fillers.  One workgroup of 4 waves (one per SIMD), s_memtime around
// 2,000 repetitions.   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_mfma_issue tools/exp_mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define M(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define SB __builtin_amdgcn_sched_barrier(0)

template <int PAT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(float *out, unsigned long long *cyc, int reps)
{
    extern __shared__ __align__(16) char lds[];
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 256)
        ((uint32_t *)lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char *p = lds + (lane & 15) * 544 + (lane >> 4) * 16;
    half8 a0 = *(const half8 *)(p), a1 = *(const half8 *)(p + 9000), a2 = *(const half8 *)(p + 18000), a3 = *(const half8 *)(p + 27000);
    half8 b0 = *(const half8 *)(p + 256), b1 = *(const half8 *)(p + 9256);
    half8 B0[4], B1[4];
    for (int i = 0; i < 4; i++) {
        B0[i] = *(const half8 *)(p + 512 * i);
        B1[i] = *(const half8 *)(p + 512 * i + 256);
    }
    unsigned voff = 0;
    float4v c[12];
    for (int i = 0; i < 12; i++)
        for (int v = 0; v < 4; v++)
            c[i][v] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        const char *q = p + (r & 7) * 64;
        if (PAT == 0) { // 12 independent accumulators
#pragma unroll
            for (int i = 0; i < 12; i++) { M(c[i], a0, b0); }
        } else if (PAT == 1) { // same accumulator again after ONE other MFMA (the first 16x16x32 loop)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                M(c[6 * t + 0], a0, b0); M(c[6 * t + 1], a1, b0); M(c[6 * t + 2], a0, b1); M(c[6 * t + 3], a1, b1);
                M(c[6 * t + 2], a2, b0); M(c[6 * t + 3], a3, b0);
            }
        } else if (PAT == 2) { // ... after THREE others (the present loop), no LDS reads
#pragma unroll
            for (int t = 0; t < 2; t++) {
                M(c[6 * t + 2], a0, b1); M(c[6 * t + 3], a1, b1); M(c[6 * t + 0], a0, b0); M(c[6 * t + 1], a1, b0);
                M(c[6 * t + 2], a2, b0); M(c[6 * t + 3], a3, b0);
            }
        } else if (PAT == 3) { // the present loop with its two ds_read_b128 per tile, one per gap
#pragma unroll
            for (int t = 0; t < 2; t++) {
                SB; M(c[6 * t + 2], a0, b1); b0 = *(const half8 *)(q + t * 8704); SB;
                M(c[6 * t + 3], a1, b1); b1 = *(const half8 *)(q + t * 8704 + 256); SB;
                M(c[6 * t + 0], a0, b0); M(c[6 * t + 1], a1, b0); M(c[6 * t + 2], a2, b0); M(c[6 * t + 3], a3, b0); SB;
            }
        } else if (PAT == 4) { // both reads bunched in front of the tile
#pragma unroll
            for (int t = 0; t < 2; t++) {
                SB; b0 = *(const half8 *)(q + t * 8704); b1 = *(const half8 *)(q + t * 8704 + 256); SB;
                M(c[6 * t + 2], a0, b1); M(c[6 * t + 3], a1, b1);
                M(c[6 * t + 0], a0, b0); M(c[6 * t + 1], a1, b0); M(c[6 * t + 2], a2, b0); M(c[6 * t + 3], a3, b0); SB;
            }
        } else if (PAT == 5) { // PAT 2 + an s_nop 0 in front of every dependent MFMA (what hipcc inserts)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                M(c[6 * t + 2], a0, b1); M(c[6 * t + 3], a1, b1); M(c[6 * t + 0], a0, b0); M(c[6 * t + 1], a1, b0);
                asm volatile("s_nop 0"); M(c[6 * t + 2], a2, b0); M(c[6 * t + 3], a3, b0);
            }
        } else if (PAT == 6) { // same accumulator after FIVE others
#pragma unroll
            for (int t = 0; t < 1; t++) {
                M(c[2], a0, b1); M(c[3], a1, b1); M(c[8], a0, b1); M(c[9], a1, b1); M(c[0], a0, b0); M(c[1], a1, b0);
                M(c[2], a2, b0); M(c[3], a3, b0); M(c[8], a2, b0); M(c[9], a3, b0); M(c[6], a0, b0); M(c[7], a1, b0);
            }
        } else if (PAT == 7 || PAT == 8 || PAT == 9) {
            // the present loop, software-pipelined as in the kernel: the reads of a tile go out three tiles ahead of their
            // use (four register sets), one per MFMA gap (7), bunched in front of the tile (8), or (9) one per gap with
            // a v_add_u32 of the address in front of each pair as hipcc emits it
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int cur = t, nxt = (t + 3) & 3;
                const char *qq = q + t * 8704;
                if (PAT == 9)
                    asm volatile("v_add_u32 %0, %1, %0" : "+v"(voff) : "s"(r));
                if (PAT == 8) {
                    SB; B0[nxt] = *(const half8 *)(qq + (voff & 0)); B1[nxt] = *(const half8 *)(qq + 256); SB;
                    M(c[(3 * t + 2) % 12], a0, B1[cur]); M(c[(3 * t + 3) % 12], a1, B1[cur]);
                } else {
                    SB; M(c[(3 * t + 2) % 12], a0, B1[cur]); B0[nxt] = *(const half8 *)(qq + (voff & 0)); SB;
                    M(c[(3 * t + 3) % 12], a1, B1[cur]); B1[nxt] = *(const half8 *)(qq + 256); SB;
                }
                M(c[(3 * t) % 12], a0, B0[cur]); M(c[(3 * t + 1) % 12], a1, B0[cur]);
                M(c[(3 * t + 2) % 12], a2, B0[cur]); M(c[(3 * t + 3) % 12], a3, B0[cur]); SB;
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 12; i++)
        for (int v = 0; v < 4; v++)
            s += c[i][v];
    for (int i = 0; i < 4; i++)
        s += (float)B0[i][0] + (float)B1[i][1];
    out[threadIdx.x] = s + (float)b0[0] + (float)b1[1] + (float)voff;
    if (threadIdx.x == 0)
        cyc[0] = t1 - t0;
}

template <int PAT> void run(const char *what, float *out, unsigned long long *cyc)
{
    const int reps = 2000;
    hipFuncSetAttribute((const void *)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    k<PAT><<<1, 256, 64 * 1024>>>(out, cyc, 100);
    k<PAT><<<1, 256, 64 * 1024>>>(out, cyc, reps);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %6.2f cycles per MFMA\n", what, (double)c / (reps * (PAT >= 7 ? 24.0 : 12.0)));
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 64);
    run<0>("12 independent accumulators", out, cyc);
    run<1>("same accumulator after 1 other MFMA", out, cyc);
    run<2>("same accumulator after 3 other MFMAs", out, cyc);
    run<6>("same accumulator after 5 other MFMAs", out, cyc);
    run<5>("after 3 others + s_nop 0 before the dependent one", out, cyc);
    run<3>("after 3 others + 2 ds_read_b128 per 6 MFMAs, one per gap", out, cyc);
    run<4>("after 3 others + 2 ds_read_b128 per 6 MFMAs, bunched", out, cyc);
    run<7>("pipelined (reads 3 tiles ahead): 2 ds_read_b128 per 6 MFMAs, one per gap", out, cyc);
    run<8>("pipelined: both reads bunched in front of the tile", out, cyc);
    run<9>("pipelined, one per gap, + a v_add_u32 per tile", out, cyc);
    return 0;
}
