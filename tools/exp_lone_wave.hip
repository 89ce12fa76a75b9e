// exp_lone_wave.hip -- what ONE wave alone on its SIMD pays per instruction (cycles of s_memtime),
// for the instruction kinds of the 16-lanes-per-board rollout kernel: dependent chains, two
// interleaved chains, and independent streams.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_lone_wave tools/exp_lone_wave.hip && tools/_build/exp_lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R32(x) R16(x) R16(x)

// Each test: 32 copies of BODY per iteration, ITER iterations; v0..v7 live registers.
#define TEST(NAME, BODY, NINSTR)                                                                   \
    __global__ void NAME(uint64_t *out, int iters)                                                 \
    {                                                                                              \
        uint32_t a = threadIdx.x * 2654435761u + 12345u, b = threadIdx.x ^ 0x5bd1e995u, c = 77u + threadIdx.x, d = 3u; \
        uint32_t e = a ^ 0x1234567u, f = b + 99u, g = c * 3u, h = 5u;                                   \
        uint32_t sh;                                                                               \
        asm volatile("v_mov_b32 %0, 1" : "=v"(sh));                                                \
        uint32_t lds_addr = (threadIdx.x & 63u) * 16u;                                             \
        __shared__ uint32_t lds[4096];                                                             \
        for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (i * 16) & 0x3ff0;                   \
        __syncthreads();                                                                           \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int it = 0; it < iters; it++) {                                                       \
            asm volatile(R32(BODY)                                                                 \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(lds_addr) \
                         : "v"(sh)                                                                 \
                         : "vcc", "memory", "v100", "v101", "v102", "v103", "s20", "s21");                                                     \
        }                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");                                             \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                           \
        if (a + b + c + d + e + f + g + h + lds_addr == 0x12345u) out[1] = a;                      \
    }

// registers: %0..%7 = a..h, %8 = lds_addr, %9 = sh.  64-bit pairs: [a,b]=%0,%1 are not adjacent registers
// in general, so 64-bit tests use explicit v[N:N+1] via separate kernels below.

TEST(k_and_dep, "v_and_b32 %0, %0, %1\n", 1)
TEST(k_and_ind, "v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7\n", 4)
TEST(k_addf_dep, "v_add_f32 %0, %0, %1\n", 1)
TEST(k_bitop3_dep, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8\n", 1)
TEST(k_alignbit_dep, "v_alignbit_b32 %0, %0, %1, %9\n", 1)
TEST(k_perm_dep, "v_perm_b32 %0, %0, %1, %2\n", 1)
TEST(k_bfe_dep, "v_bfe_i32 %0, %0, %9, 1\n", 1)
TEST(k_bfrev_dep, "v_bfrev_b32 %0, %0\n", 1)
TEST(k_bcnt_dep, "v_bcnt_u32_b32 %0, %0, %1\n", 1)
TEST(k_lshl_dep, "v_lshlrev_b32 %0, %9, %0\n", 1)
TEST(k_cmp_cnd_dep, "v_cmp_ne_u32 vcc, %0, %1\n v_cndmask_b32 %0, %2, %3, vcc\n", 2)
TEST(k_dpp_dep_nop1, "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n", 2)
TEST(k_dpp_2chain_nop0, "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_or_b32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 0\n", 3)
TEST(k_dpp_ind, "v_or_b32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_or_b32_dpp %2, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n", 2)
TEST(k_dpp_valu2_dpp, "v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n", 3)
TEST(k_snop0, "s_nop 0\n", 1)
TEST(k_snop1, "s_nop 1\n", 1)
TEST(k_snop3, "s_nop 3\n", 1)
TEST(k_salu, "s_and_b32 s20, s20, s21\n", 1)
TEST(k_valu_salu, "v_and_b32 %0, %0, %1\n s_and_b32 s20, s20, s21\n", 2)
TEST(k_ldsread_dep, "ds_read_b32 %8, %8\n s_waitcnt lgkmcnt(0)\n", 2)
TEST(k_ldsread128_ind, "ds_read_b128 v[100:103], %8\n", 1)
TEST(k_cmp_branch, "v_cmp_eq_u32 vcc, 0x7fffffff, %0\n s_cbranch_vccnz 1\n s_nop 0\n", 3)
TEST(k_waitcnt0, "s_waitcnt lgkmcnt(0)\n", 1)

// 64-bit and packed tests with fixed register pairs
#define TEST64(NAME, BODY, NINSTR)                                                                 \
    __global__ void NAME(uint64_t *out, int iters)                                                 \
    {                                                                                              \
        uint64_t t0, t1;                                                                           \
        asm volatile("v_mov_b32 v10, 1\n v_mov_b32 v11, 0x3f800000\n v_mov_b32 v12, 0x3f800000\n v_mov_b32 v13, 0x3f800000\n" \
                     "v_mov_b32 v14, 0x3f800000\n v_mov_b32 v15, 0x3f800000\n v_mov_b32 v16, 0x3f800000\n v_mov_b32 v17, 0x3f800000\n" \
                     "v_mov_b32 v18, 0x3f800000\n v_mov_b32 v19, 1\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n" ::: "v10", "v11", "v12", \
                     "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21");                \
        t0 = __builtin_amdgcn_s_memtime();                                                         \
        for (int it = 0; it < iters; it++) {                                                       \
            asm volatile(R32(BODY)::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "vcc"); \
        }                                                                                          \
        t1 = __builtin_amdgcn_s_memtime();                                                         \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                           \
    }
TEST64(k_shl64_dep, "v_lshlrev_b64 v[12:13], v10, v[12:13]\n", 1)
TEST64(k_shl64_ind, "v_lshlrev_b64 v[12:13], v10, v[14:15]\n v_lshlrev_b64 v[16:17], v10, v[18:19]\n", 2)
TEST64(k_pkmul_dep, "v_pk_mul_f32 v[12:13], v[12:13], v[14:15]\n", 1)
TEST64(k_pkmul_2chain, "v_pk_mul_f32 v[12:13], v[12:13], v[14:15]\n v_pk_mul_f32 v[16:17], v[16:17], v[18:19]\n", 2)
TEST64(k_pkmul_ind, "v_pk_mul_f32 v[12:13], v[14:15], v[14:15]\n v_pk_mul_f32 v[16:17], v[18:19], v[18:19]\n", 2)
TEST64(k_pkadd_dep, "v_pk_add_f32 v[12:13], v[12:13], v[14:15]\n", 1)
TEST64(k_mulf_dep, "v_mul_f32 v12, v12, v14\n", 1)
TEST64(k_mulf_2x, "v_mul_f32 v12, v12, v14\n v_mul_f32 v13, v13, v15\n", 2)
TEST64(k_lshladd64_dep, "v_lshl_add_u64 v[12:13], v[12:13], 0, 1\n", 1)
TEST64(k_pkmov, "v_pk_mov_b32 v[12:13], v[14:15], v[16:17] op_sel:[1,0]\n", 1)
TEST64(k_movdpp_add, "v_mov_b32_dpp v13, v12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 v12, v12, v13\n s_nop 0\n", 3)
TEST64(k_adddpp_dep, "v_add_f32_dpp v12, v12, v12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n", 2)

// instruction fetch: 1024 eight-byte instructions in a row (8 KB of straight-line code per loop)
#define R1024(x) R32(R32(x))
__global__ void k_big_bitop3(uint64_t *out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = threadIdx.x ^ 0x5bd1e995u, c = 77u + threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
        asm volatile(R1024("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8\n") : "+v"(a) : "v"(b), "v"(c));
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 0x12345u) out[1] = a;
}
__global__ void k_big_and(uint64_t *out, int iters)
{
    uint32_t a = threadIdx.x * 2654435761u, b = threadIdx.x ^ 0x5bd1e995u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
        asm volatile(R1024("v_and_b32 %0, %0, %1\n") : "+v"(a) : "v"(b));
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 0x12345u) out[1] = a;
}

struct T { const char *name; void (*fn)(uint64_t *, int); int n; };
#define E(NAME, N) {#NAME, NAME, N}

int main()
{
    uint64_t *out;
    hipMalloc(&out, 1024 * 8);
    std::vector<T> tests = {
        E(k_and_dep, 1), E(k_and_ind, 4), E(k_addf_dep, 1), E(k_bitop3_dep, 1), E(k_alignbit_dep, 1), E(k_perm_dep, 1),
        E(k_bfe_dep, 1), E(k_bfrev_dep, 1), E(k_bcnt_dep, 1), E(k_lshl_dep, 1), E(k_cmp_cnd_dep, 2),
        E(k_dpp_dep_nop1, 2), E(k_dpp_2chain_nop0, 3), E(k_dpp_ind, 2), E(k_dpp_valu2_dpp, 3),
        E(k_snop0, 1), E(k_snop1, 1), E(k_snop3, 1), E(k_salu, 1), E(k_valu_salu, 2), E(k_ldsread_dep, 2),
        E(k_ldsread128_ind, 1), E(k_cmp_branch, 3), E(k_waitcnt0, 1),
        E(k_shl64_dep, 1), E(k_shl64_ind, 2), E(k_pkmul_dep, 1), E(k_pkmul_2chain, 2), E(k_pkmul_ind, 2), E(k_pkadd_dep, 1),
        E(k_mulf_dep, 1), E(k_mulf_2x, 2), E(k_lshladd64_dep, 1), E(k_pkmov, 1), E(k_movdpp_add, 3), E(k_adddpp_dep, 2)};
    const int iters = 200;
    for (int grid : {1, 1024}) {
        printf("--- %d workgroup(s) of one wave\n", grid);
        for (auto &t : tests) {
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), 0, 0, out, 10);
            hipLaunchKernelGGL(t.fn, dim3(grid), dim3(64), 0, 0, out, iters);
            hipDeviceSynchronize();
            uint64_t h[1024];
            hipMemcpy(h, out, grid * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < grid; i++) s += (double)h[i];
            s /= grid;
            printf("%-22s %6.2f cycles per group of %d = %5.2f per instruction\n", t.name, s / (iters * 32.0), t.n,
                   s / (iters * 32.0 * t.n));
        }
    }
    for (int grid : {1, 256, 1024}) {
        for (int which = 0; which < 2; which++) {
            auto fn = which ? k_big_and : k_big_bitop3;
            hipLaunchKernelGGL(fn, dim3(grid), dim3(64), 0, 0, out, 2);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(64), 0, 0, out, 20);
            hipDeviceSynchronize();
            uint64_t h[1024];
            hipMemcpy(h, out, grid * 8, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < grid; i++) s += (double)h[i];
            printf("%d waves, 1024 %s in a row: %.2f cycles per instruction\n", grid,
                   which ? "4-byte v_and_b32" : "8-byte v_bitop3_b32", s / grid / (20 * 1024.0));
        }
    }
    return 0;
}
