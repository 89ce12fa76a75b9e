// exp_l1_share.hip -- can the two waves of a SIMD share a weight stream through the CU's vector L1?  The walks' K loop
// (v_mfma_f32_16x16x32_f16, 3 MFMAs per product, B operands from LDS three tiles ahead, A operands streamed from an
// L2-resident 3.9 MB weight set one k-step ahead) by EIGHT waves: wave-group g = wave >> 2 takes board g of a pair, both
// groups need the SAME weights.  Variants: `shared` -- both groups load the same addresses (the second group's loads
// should hit L1 while the groups stay within a k-step or two of each other); `distinct` -- group 1 streams a copy of its own
// (what a miss in L1 costs: twice the bytes through the CU's L2 port); `four` -- the product's four-wave loop, one pair per
// workgroup, for reference.  224 workgroups, 200 walks each.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_l1_share tools/exp_l1_share.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define SB __builtin_amdgcn_sched_barrier(0)

__device__ __host__ inline uint32_t hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
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
constexpr int STEPS32 = 234, W_STRIDE = 256;

// NW waves; a wave: 2 M tiles x NT tiles of 16 cells
template <int NW, int NT, bool DISTINCT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k(const u32x4 *wh, const u32x4 *wl,
                                                                                                    float *out, int walks, size_t copy)
{
    extern __shared__ __align__(16) char lds[];
    for (int i = threadIdx.x; i < 80 * 1024 / 4; i += 64 * NW)
        ((uint32_t *)lds)[i] = rnd2(i * 7919u + blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3, g = threadIdx.x >> 8;
    const int c = lane & 15, kq = lane >> 4;
    const char *base = lds + c * 544 + kq * 16 + g * 35840;
    const size_t off = (DISTINCT && g) ? copy : 0;
    const u32x4 *ph = wh + off + (32 * wv + c) * 2 + (kq & 1) + (kq >> 1) * W_STRIDE;
    const u32x4 *pl = wl + off + (32 * wv + c) * 2 + (kq & 1) + (kq >> 1) * W_STRIDE;
    float4v am[2][NT], ac[2][NT];
    for (int m = 0; m < 2; m++)
        for (int n = 0; n < NT; n++)
            for (int v = 0; v < 4; v++) { am[m][n][v] = 0.f; ac[m][n][v] = 0.f; }
    for (int w = 0; w < walks; w++) {
        u32x4 a_hi[3][2], a_lo[3][2];
        for (int i = 0; i < 2; i++) {
            a_hi[i][0] = ph[(size_t)i * 2 * W_STRIDE], a_hi[i][1] = ph[(size_t)i * 2 * W_STRIDE + 32];
            a_lo[i][0] = pl[(size_t)i * 2 * W_STRIDE], a_lo[i][1] = pl[(size_t)i * 2 * W_STRIDE + 32];
        }
        half8 bh[4], bl[4];
        for (int i = 0; i < 3; i++) {
            bh[i] = *(const half8 *)(base + (i % NT) * 8704);
            bl[i] = *(const half8 *)(base + (i % NT) * 8704 + 256);
        }
        for (int s0 = 0; s0 < STEPS32; s0 += 3)
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int st = s0 + u, nx = st + 2 < STEPS32 ? st + 2 : STEPS32 - 1;
                const half8 ah0 = __builtin_bit_cast(half8, a_hi[u][0]), ah1 = __builtin_bit_cast(half8, a_hi[u][1]);
                const half8 al0 = __builtin_bit_cast(half8, a_lo[u][0]), al1 = __builtin_bit_cast(half8, a_lo[u][1]);
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const int tile = u * NT + n, cur = tile % 4, nxt = (tile + 3) % 4;
                    const char *p = base + ((tile + 3) % NT) * 8704 + (st & 3) * 64;
                    SB; MFMA16(ac[0][n], ah0, bl[cur]); bh[nxt] = *(const half8 *)p; SB;
                    MFMA16(ac[1][n], ah1, bl[cur]); bl[nxt] = *(const half8 *)(p + 256); SB;
                    MFMA16(am[0][n], ah0, bh[cur]);
                    if (n == 0) a_hi[(u + 2) % 3][0] = ph[(size_t)nx * 2 * W_STRIDE];
                    else if (n == 1) a_hi[(u + 2) % 3][1] = ph[(size_t)nx * 2 * W_STRIDE + 32];
                    else if (n == 2) a_lo[(u + 2) % 3][0] = pl[(size_t)nx * 2 * W_STRIDE];
                    else if (n == 3) a_lo[(u + 2) % 3][1] = pl[(size_t)nx * 2 * W_STRIDE + 32];
                    SB;
                    MFMA16(am[1][n], ah1, bh[cur]); MFMA16(ac[0][n], al0, bh[cur]); MFMA16(ac[1][n], al1, bh[cur]); SB;
                }
            }
        __syncthreads(); // (a layer's barrier: the groups start each layer together; here once per walk)
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    for (int m = 0; m < 2; m++)
        for (int n = 0; n < NT; n++)
            for (int v = 0; v < 4; v++) s += am[m][n][v] + ac[m][n][v] * (1.0f / 2048.0f);
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}

template <int NW, int NT, bool DISTINCT> void run(const char *what, const u32x4 *wh, const u32x4 *wl, float *out, size_t copy)
{
    const int walks = 200, grid = 224;
    hipFuncSetAttribute((const void *)k<NW, NT, DISTINCT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NW, NT, DISTINCT><<<grid, 64 * NW, 80 * 1024>>>(wh, wl, out, 20, copy);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<NW, NT, DISTINCT><<<grid, 64 * NW, 80 * 1024>>>(wh, wl, out, walks, copy);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-64s %7.2f us per pair of boards (K loops only)\n", what, best * 1e3 / walks);
}

int main()
{
    const size_t n16 = (size_t)(2 * STEPS32 + 8) * W_STRIDE; // u32x4 per piece and copy
    u32x4 *wh, *wl;
    float *out;
    hipMalloc(&wh, 2 * n16 * 16);
    hipMalloc(&wl, 2 * n16 * 16);
    hipMalloc(&out, 1024 * 512 * 4);
    fill<<<256, 256>>>((uint32_t *)wh, 2 * n16 * 4);
    fill<<<256, 256>>>((uint32_t *)wl, 2 * n16 * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        run<4, 8, false>("four waves, a pair per workgroup (the product's loop)", wh, wl, out, n16);
        run<8, 4, false>("eight waves, a board per wave group, SHARED weight stream", wh, wl, out, n16);
        run<8, 4, true>("eight waves, a board per wave group, a stream per group", wh, wl, out, n16);
    }
    return 0;
}
