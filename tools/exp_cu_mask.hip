// exp_cu_mask.hip -- which CUs a hipExtStreamCreateWithCUMask stream really gets, by mask bit: every workgroup of a
// launch that fills the masked stream (one workgroup per CU: 100 KB of LDS) records its XCC id and its HW_ID, spins
// ~200 us so that all of them are resident together, and the host prints the set of (xcd, se, cu) per mask.
// Also: two launches on complementary masks started back to back -- do they run TOGETHER (both see each other's flag)?
//   hipcc -O3 --offload-arch=gfx950 -o tools/_build/exp_cu_mask tools/exp_cu_mask.hip && tools/_build/exp_cu_mask
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#define CHECK(x)                                                                            \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                           \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

__global__ __launch_bounds__(256) void where_kernel(uint32_t *out, long long spin_ticks, uint32_t *flag_mine, const uint32_t *flag_other,
                                                    uint32_t *saw_other)
{
    extern __shared__ char big[];
    if (threadIdx.x == 0) {
        big[0] = 1;
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID [3:0]
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
        if (flag_mine)
            __hip_atomic_fetch_add(flag_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        uint32_t seen = 0;
        while (wall_clock64() - t0 < spin_ticks) {
            if (flag_other)
                seen |= __hip_atomic_load(flag_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(32);
        }
        if (saw_other && seen)
            __hip_atomic_fetch_add(saw_other, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A workgroup that only keeps time: loops until `ticks` have passed and records the longest gap between two turns of its
// loop (100 MHz ticks) -- a workgroup that is not executed for a while shows up as a gap (round 6: the net launch's last
// workgroups stalled while a second launch ran on the other CUs).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void beat_kernel(long long *gap, long long *turns,
                                                                                                long long ticks)
{
    extern __shared__ char big[];
    if (threadIdx.x == 0) {
        big[0] = 1;
        const long long t0 = wall_clock64();
        long long last = t0, worst = 0, n = 0;
        for (;;) {
            const long long now = wall_clock64();
            worst = now - last > worst ? now - last : worst;
            last = now;
            n++;
            if (now - t0 > ticks)
                break;
            __builtin_amdgcn_s_sleep(16);
        }
        gap[blockIdx.x] = worst;
        turns[blockIdx.x] = n;
    }
}

static int make_stream(const std::vector<int> &bits, int total, hipStream_t *s)
{
    std::vector<uint32_t> words((total + 31) / 32, 0u);
    for (int b : bits)
        words[b / 32] |= 1u << (b % 32);
    CHECK(hipExtStreamCreateWithCUMask(s, (uint32_t)words.size(), words.data()));
    return 0;
}

static void describe(const char *name, const uint32_t *h, int n)
{
    std::map<int, std::set<int>> by_xcd;
    for (int i = 0; i < n; i++) {
        const uint32_t hw = h[2 * i + 1];
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_xcd[(int)h[2 * i]].insert(se * 32 + sh * 16 + cu);
    }
    printf("%s: %d workgroups on", name, n);
    int total = 0;
    for (auto &kv : by_xcd) {
        printf("  xcd%d:%zu", kv.first, kv.second.size());
        total += (int)kv.second.size();
    }
    printf("  = %d distinct CUs; block->xcd of the first 16:", total);
    for (int i = 0; i < 16 && i < n; i++)
        printf(" %u", h[2 * i]);
    printf("\n");
}

int main()
{
    int total = 0;
    CHECK(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d\n", total);
    CHECK(hipFuncSetAttribute((const void *)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    uint32_t *d = nullptr, *flags = nullptr;
    CHECK(hipMalloc(&d, 2 * 1024 * 4));
    CHECK(hipMalloc(&flags, 64));
    std::vector<uint32_t> h(2 * 1024);
    struct Case {
        const char *name;
        std::vector<int> bits;
    };
    std::vector<Case> cases;
    {
        Case c{"bits 0..7", {}};
        for (int i = 0; i < 8; i++) c.bits.push_back(i);
        cases.push_back(c);
    }
    {
        Case c{"bits 0..31", {}};
        for (int i = 0; i < 32; i++) c.bits.push_back(i);
        cases.push_back(c);
    }
    {
        Case c{"every 8th bit (32)", {}};
        for (int i = 0; i < total; i += 8) c.bits.push_back(i);
        cases.push_back(c);
    }
    {
        Case c{"bits 32..255", {}};
        for (int i = 32; i < total; i++) c.bits.push_back(i);
        cases.push_back(c);
    }
    {
        Case c{"bits 0..15", {}};
        for (int i = 0; i < 16; i++) c.bits.push_back(i);
        cases.push_back(c);
    }
    for (auto &c : cases) {
        hipStream_t s;
        if (make_stream(c.bits, total, &s))
            return 1;
        const int n = (int)c.bits.size();
        CHECK(hipMemsetAsync(d, 0xff, 2 * 1024 * 4, s));
        hipLaunchKernelGGL(where_kernel, dim3(n), dim3(256), 100 * 1024, s, d, 20000ll, nullptr, nullptr, nullptr);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(h.data(), d, 2 * n * 4, hipMemcpyDeviceToHost));
        describe(c.name, h.data(), n);
        CHECK(hipStreamDestroy(s));
    }
    // two launches that want to see each other: A on bits 0..31 (64 workgroups of 50 KB: two per CU), B on the rest
    // (224 workgroups, one per CU); each spins 2 ms and counts the workgroups that saw the other launch's flag
    for (int order = 0; order < 2; order++) {
        hipStream_t sa, sb;
        std::vector<int> a, b;
        for (int i = 0; i < total; i++)
            (i < 32 ? a : b).push_back(i);
        if (make_stream(a, total, &sa) || make_stream(b, total, &sb))
            return 1;
        CHECK(hipMemset(flags, 0, 64));
        CHECK(hipDeviceSynchronize());
        uint32_t *da = d, *db = d + 512;
        if (order == 0) {
            hipLaunchKernelGGL(where_kernel, dim3(64), dim3(256), 50 * 1024, sa, da, 200000ll, flags + 0, flags + 1, flags + 2);
            hipLaunchKernelGGL(where_kernel, dim3(total - 32), dim3(256), 100 * 1024, sb, db, 200000ll, flags + 1, flags + 0, flags + 3);
        } else {
            hipLaunchKernelGGL(where_kernel, dim3(total - 32), dim3(256), 100 * 1024, sb, db, 200000ll, flags + 1, flags + 0, flags + 3);
            hipLaunchKernelGGL(where_kernel, dim3(64), dim3(256), 50 * 1024, sa, da, 200000ll, flags + 0, flags + 1, flags + 2);
        }
        CHECK(hipDeviceSynchronize());
        uint32_t f[4];
        CHECK(hipMemcpy(f, flags, 16, hipMemcpyDeviceToHost));
        printf("together (order %d): A started %u of 64, B started %u of %d; A workgroups that saw B: %u, B that saw A: %u\n", order,
               f[0], f[1], total - 32, f[2], f[3]);
        CHECK(hipMemcpy(h.data(), d, 2 * 1024 * 4, hipMemcpyDeviceToHost));
        describe("  A", h.data(), 64);
        describe("  B", h.data() + 1024, total - 32);
        CHECK(hipStreamDestroy(sa));
        CHECK(hipStreamDestroy(sb));
    }
    // Do the last workgroups of a launch on the big mask stall while a second launch runs on the small one?  A: 2 x g
    // workgroups (two per CU, 80 KB of LDS each) on the first g CUs, B: one workgroup (138 KB of LDS, one wave per SIMD) on
    // each of the other CUs; both keep time for 300 ms; B's longest gaps by block index
    long long *gap = nullptr, *turns = nullptr;
    CHECK(hipMalloc(&gap, 2 * 512 * 8));
    CHECK(hipMalloc(&turns, 2 * 512 * 8));
    CHECK(hipFuncSetAttribute((const void *)beat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 138 * 1024));
    for (int g : {16, 24, 32}) {
        for (int rep = 0; rep < 3; rep++) {
            hipStream_t sa, sb;
            std::vector<int> a, b;
            for (int i = 0; i < total; i++)
                (i < g ? a : b).push_back(i);
            if (make_stream(a, total, &sa) || make_stream(b, total, &sb))
                return 1;
            CHECK(hipMemset(gap, 0, 2 * 512 * 8));
            CHECK(hipDeviceSynchronize());
            const int nb = total - g;
            hipLaunchKernelGGL(beat_kernel, dim3(2 * g), dim3(256), 78 * 1024, sa, gap + 512, turns + 512, 30000000ll);
            hipLaunchKernelGGL(beat_kernel, dim3(nb), dim3(256), 138 * 1024, sb, gap, turns, 30000000ll);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> hg(1024), ht(1024);
            CHECK(hipMemcpy(hg.data(), gap, 1024 * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(ht.data(), turns, 1024 * 8, hipMemcpyDeviceToHost));
            int stalled = 0, first = -1;
            long long worst = 0;
            for (int i = 0; i < nb; i++) {
                if (hg[i] > 100000) { // > 1 ms without a turn of the loop
                    stalled++;
                    first = first < 0 ? i : first;
                }
                worst = hg[i] > worst ? hg[i] : worst;
            }
            long long worst_a = 0;
            for (int i = 0; i < 2 * g; i++)
                worst_a = hg[512 + i] > worst_a ? hg[512 + i] : worst_a;
            printf("beat: %d game-side CUs (%d workgroups) + %d workgroups on the rest: B's longest gap %.1f us, %d of B's workgroups "
                   "with a gap > 1 ms (first: block %d); A's longest gap %.1f us\n", g, 2 * g, nb, worst / 100.0, stalled, first,
                   worst_a / 100.0);
            CHECK(hipStreamDestroy(sa));
            CHECK(hipStreamDestroy(sb));
        }
    }
    return 0;
}
