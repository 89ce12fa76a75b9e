// exp_valu_rates.hip -- issue cost of the VALU instructions the lane-per-board rollout
// leans on, relative to v_and_b32 (whole chip, 8 waves per SIMD, dependent chains).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_valu_rates tools/exp_valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

#define KERNEL(NAME, ASM, ...)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)              \
    {                                                                                  \
        uint32_t a = threadIdx.x, b = blockIdx.x | 1u, c = 5u, d = 3u;                 \
        uint64_t q = ((uint64_t)a << 32) | b, r = 0x12345u + a;                        \
        for (int i = 0; i < iters; i++) {                                              \
            REP16(asm volatile(ASM : __VA_ARGS__);)                                    \
        }                                                                              \
        if (a + b + c + d + (uint32_t)q + (uint32_t)r == 0xdeadbeefu)                  \
            out[0] = a;                                                                \
    }

KERNEL(k_and32, "v_and_b32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_lshr64, "v_lshrrev_b64 %0, %1, %0", "+v"(q) : "v"(c))
KERNEL(k_lshl64, "v_lshlrev_b64 %0, %1, %0", "+v"(q) : "v"(c))
KERNEL(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %1", "+v"(q) : "v"(r))
KERNEL(k_mul24, "v_mul_u32_u24 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_mullo, "v_mul_lo_u32 %0, %1, %0", "+v"(a) : "v"(b))
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_bfrev, "v_bfrev_b32 %0, %0", "+v"(a) : "v"(b))
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80", "+v"(a) : "v"(b), "v"(c))
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(a) : "v"(b))
KERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "+v"(a) : "v"(b))
KERNEL(k_cmp_cnd32, "v_cmp_ne_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmp_cnd64, "v_cmp_ne_u32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "+v"(a) : "v"(b) : "s20", "s21")
KERNEL(k_cnd32_indep, "v_cndmask_b32 %0, %1, %2, vcc", "=v"(a) : "v"(b), "v"(c))
KERNEL(k_cnd32_const, "v_cndmask_b32 %0, 0, %0, vcc", "+v"(a) : "v"(b))
KERNEL(k_cmp_4_cnd, "v_cmp_ne_u32 vcc, %0, %1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmp_16_cnd, "v_cmp_ne_u32 vcc, %0, %1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmp_4_2cnd, "v_cmp_ne_u32 vcc, %0, %1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmps_4_cnd, "v_cmp_ne_u32_e64 s[20:21], %0, %1\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_and_b32 %0, %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "+v"(a) : "v"(b) : "s20", "s21")
KERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_bfi_m1, "v_bfi_b32 %0, %0, %1, -1", "+v"(a) : "v"(c))
KERNEL(k_ashr32, "v_ashrrev_i32 %0, %1, %0", "+v"(a) : "v"(c))
KERNEL(k_ashr32_imm, "v_ashrrev_i32 %0, 31, %0", "+v"(a) : "v"(c))
KERNEL(k_lshl32_imm, "v_lshlrev_b32 %0, 3, %0", "+v"(a) : "v"(c))
KERNEL(k_lshr32_imm, "v_lshrrev_b32 %0, 3, %0", "+v"(a) : "v"(c))
KERNEL(k_lshl64_imm, "v_lshlrev_b64 %0, 7, %0", "+v"(q) : "v"(c))
KERNEL(k_lshr64_imm, "v_lshrrev_b64 %0, 7, %0", "+v"(q) : "v"(c))
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_max32, "v_max_u32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_mov64, "v_mov_b64 %0, %1", "+v"(q) : "v"(r))
KERNEL(k_bitop3_imm, "v_bitop3_b32 %0, %0, %1, s20 bitop3:0x80", "+v"(a) : "v"(b))
KERNEL(k_and_lit, "v_and_b32 %0, 0x7e7e7e7e, %0", "+v"(a) : "v"(b))
KERNEL(k_sub_clamp, "v_sub_u32_e64 %0, %0, %1 clamp", "+v"(a) : "v"(b))
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", "+v"(a) : "v"(b))
KERNEL(k_snop, "s_nop 0", "+v"(a) : "v"(b))
KERNEL(k_add32, "v_add_u32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_sub32, "v_sub_u32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_or32, "v_or_b32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_xor32, "v_xor_b32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_lshl32, "v_lshlrev_b32 %0, %1, %0", "+v"(a) : "v"(c))
KERNEL(k_min32, "v_min_u32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_bfe, "v_bfe_u32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_fma32, "v_fma_f32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_addf32, "v_add_f32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_cmpf32, "v_cmp_le_f32 vcc, %0, %1", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_cmp64s, "v_cmp_ne_u32_e64 s[20:21], %0, %1", "+v"(a) : "v"(b) : "s20", "s21")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, 0, %0, vcc", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_mov, "v_mov_b32 %0, %1", "+v"(a) : "v"(b))
KERNEL(k_not, "v_not_b32 %0, %0", "+v"(a) : "v"(b))
KERNEL(k_and32_imm, "v_and_b32 %0, 0x1c0e07, %0", "+v"(a) : "v"(b))
KERNEL(k_pk_mulf32, "v_pk_mul_f32 %0, %0, %1", "+v"(q) : "v"(r))
KERNEL(k_pk_addf32, "v_pk_add_f32 %0, %0, %1", "+v"(q) : "v"(r))
KERNEL(k_dsread, "ds_read_b32 %0, %1", "+v"(a) : "v"(d))
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0", "+v"(a) : "v"(b))
KERNEL(k_cmp64, "v_cmp_ne_u64 vcc, %0, %1", "+v"(q) : "v"(r) : "vcc")
KERNEL(k_cmp32, "v_cmp_ne_u32 vcc, %0, %1", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %1", "+v"(a) : "v"(b) : "vcc")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_mulf32, "v_mul_f32 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_lshr32, "v_lshrrev_b32 %0, %1, %0", "+v"(a) : "v"(c))
KERNEL(k_pk_add, "v_pk_add_u16 %0, %0, %1", "+v"(a) : "v"(b))
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "+v"(a) : "v"(b))

template <typename K> double run(K kern, uint32_t *out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<2048, 256>>>(out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<2048, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 64);
    const int iters = 4096;
    // 2048 blocks x 4 waves = 8192 waves over 1024 SIMDs = 8 per SIMD (one round)
    const double base = run(k_and32, out, iters);
    const double inst_per_simd = 8.0 * iters * 16;
    printf("v_and_b32: %.3f ms, %.2f cycles/inst/SIMD at 2.4 GHz\n", base,
           base * 1e-3 * 2.4e9 / inst_per_simd);
#define SHOW(k) printf("%-14s %.2fx\n", #k, run(k, out, iters) / base)
    SHOW(k_lshr32);
    SHOW(k_lshr64);
    SHOW(k_lshl64);
    SHOW(k_lshladd64);
    SHOW(k_mul24);
    SHOW(k_mullo);
    SHOW(k_alignbit);
    SHOW(k_bfrev);
    SHOW(k_perm);
    SHOW(k_bitop3);
    SHOW(k_cndmask);
    SHOW(k_cndmask64);
    SHOW(k_cmp_cnd32);
    SHOW(k_cmp_cnd64);
    SHOW(k_cnd32_indep);
    SHOW(k_cnd32_const);
    SHOW(k_cmp_4_cnd);
    SHOW(k_cmp_16_cnd);
    SHOW(k_cmp_4_2cnd);
    SHOW(k_cmps_4_cnd);
    SHOW(k_bfi);
    SHOW(k_bfi_m1);
    SHOW(k_ashr32);
    SHOW(k_ashr32_imm);
    SHOW(k_lshl32_imm);
    SHOW(k_lshr32_imm);
    SHOW(k_lshl64_imm);
    SHOW(k_lshr64_imm);
    SHOW(k_mulhi);
    SHOW(k_max32);
    SHOW(k_mov64);
    SHOW(k_bitop3_imm);
    SHOW(k_and_lit);
    SHOW(k_sub_clamp);
    SHOW(k_cvt_f32_u32);
    SHOW(k_snop);
    SHOW(k_add32);
    SHOW(k_sub32);
    SHOW(k_or32);
    SHOW(k_xor32);
    SHOW(k_lshl32);
    SHOW(k_min32);
    SHOW(k_and_or);
    SHOW(k_add3);
    SHOW(k_lshl_add);
    SHOW(k_bfe);
    SHOW(k_mad24);
    SHOW(k_fma32);
    SHOW(k_addf32);
    SHOW(k_cmpf32);
    SHOW(k_cmp64s);
    SHOW(k_addc);
    SHOW(k_mov);
    SHOW(k_not);
    SHOW(k_and32_imm);
    SHOW(k_pk_mulf32);
    SHOW(k_pk_addf32);
    SHOW(k_dsread);
    SHOW(k_ffbl);
    SHOW(k_cmp64);
    SHOW(k_cmp32);
    SHOW(k_addco);
    SHOW(k_lshl_or);
    SHOW(k_or3);
    SHOW(k_bcnt);
    SHOW(k_mulf32);
    SHOW(k_pk_add);
    SHOW(k_mov_dpp);
    return 0;
}
