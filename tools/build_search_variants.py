#!/usr/bin/env python3
"""Lab tool: diagnostic variants of csrc/search_kernel.hip (the persistent search), built beside the product library.

    python tools/build_search_variants.py          -> tools/_build/search_log.so, tools/_build/search_phases.so

* search_log.so    -- every request a game sends (value: kind 1, policy: kind 2) is appended to the `trace` buffer
                      as (own, opp, game, kind) instead of the timeline rows: tools/exp_request_log.py counts the
                      positions asked for more than once (what a position table can answer).
* search_phases.so -- game workgroup 0 accumulates 100 MHz clock stamps per phase of its iteration (replies + moves,
                      descent, rollouts, backup, end of iteration) into totals[11..15]: tools/exp_game_phases.py.

The variants are textual patches of the product source (every patch asserts that its anchor is there: a changed
kernel makes this script fail instead of building something else); the other objects are the product's
(iago_amd/_obj, `python -m iago_amd.build` first).  Select a variant with IAGO_HIP_LIB=<path>.
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iago_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")


def patch(s, old, new, count=1):
    assert s.count(old) >= 1, "anchor not found:\n" + old
    return s.replace(old, new, count)


def request_log(s):
    s = patch(s, '''    atomicAdd((unsigned long long *)&S.totals[(uint32_t)g == NOBODY ? 11 : kind], 1ull);
}''', '''    atomicAdd((unsigned long long *)&S.totals[(uint32_t)g == NOBODY ? 11 : kind], 1ull);
    if (S.trace) {
        const unsigned long long row = atomicAdd((unsigned long long *)&S.totals[10], 1ull);
        if ((long long)row < S.trace_rows) {
            S.trace[4 * row + 0] = (int64_t)own;
            S.trace[4 * row + 1] = (int64_t)opp;
            S.trace[4 * row + 2] = g;
            S.trace[4 * row + 3] = 1 + kind;
        }
    }
}''')
    # (the timeline rows and the per-game end rows would overwrite the log)
    s = patch(s, "if (S.trace && blockIdx.x == 0 && tid == 0 && (int64_t)wg_count[0] <", "if (false && S.trace && blockIdx.x == 0 && tid == 0 && (int64_t)wg_count[0] <")
    s = patch(s, "if (S.trace && r == 0u && g < S.trace_rows) {", "if (false && S.trace && r == 0u && g < S.trace_rows) {")
    return s


def phase_stamps(s):
    """Game workgroup 0's iteration by phase (100 MHz stamps of thread 0): replies + moves, descent, the control
    words' loads + the packing of the rollouts, the rollout passes, the backup of the rolled games, the end of the
    iteration (two barriers, pacing); summed in iago_game_phases[0..5], the iterations in [7]."""
    s = patch(s, "namespace {\nusing namespace iago;", "__device__ unsigned long long iago_game_phases[8];\nnamespace {\nusing namespace iago;")
    s = patch(s, """    for (;;) {
        bool busy = false; // this game did something in this iteration""", """    long long ph[6] = {0, 0, 0, 0, 0, 0};
    for (;;) {
        long long c_a = wall_clock64();
        bool busy = false; // this game did something in this iteration""")
    s = patch(s, """            // ---- descent (MCTS.py:105-133): from the root, or on from the leaf whose priors arrived
""", """            { const long long c = wall_clock64(); ph[0] += c - c_a; c_a = c; }
            // ---- descent (MCTS.py:105-133): from the root, or on from the leaf whose priors arrived
""")
    s = patch(s, """        uint32_t c_abort = 0u, c_idle = 0u,""", """        { const long long c = wall_clock64(); ph[1] += c - c_a; c_a = c; }
        uint32_t c_abort = 0u, c_idle = 0u,""")
    s = patch(s, """#pragma unroll 1
            for (int at = 0; at < n_now; at += 16) {""", """            { const long long c = wall_clock64(); ph[2] += c - c_a; c_a = c; }
#pragma unroll 1
            for (int at = 0; at < n_now; at += 16) {""")
    s = patch(s, """        if (mine && rolled) {
            if (state == ST_ROLL) {""", """        { const long long c = wall_clock64(); ph[3] += c - c_a; c_a = c; }
        if (mine && rolled) {
            if (state == ST_ROLL) {""")
    s = patch(s, """        if (tid == 0)
            wg_count[0]++;
        if (mine && r == 0u) {
            const int prog""", """        { const long long c = wall_clock64(); ph[4] += c - c_a; c_a = c; }
        if (tid == 0)
            wg_count[0]++;
        if (mine && r == 0u) {
            const int prog""")
    s = patch(s, """        if (!__syncthreads_or(busy)) {
            if (tid == 0)
                wg_count[1]++;""", """        { const long long c = wall_clock64(); ph[5] += c - c_a; c_a = c; }
        if (!__syncthreads_or(busy)) {
            if (tid == 0)
                wg_count[1]++;""")
    s = patch(s, """    if (tid == 0) {
        atomicAdd((unsigned long long *)&S.totals[2], (unsigned long long)wg_count[0]);""", """    if (tid == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 6; i++)
            atomicAdd(&iago_game_phases[i], (unsigned long long)ph[i]);
        atomicAdd(&iago_game_phases[7], (unsigned long long)wg_count[0]);
    }
    if (tid == 0) {
        atomicAdd((unsigned long long *)&S.totals[2], (unsigned long long)wg_count[0]);""")
    s += """
extern "C" __attribute__((visibility("default"))) int iago_debug_game_phases(unsigned long long *host, int clear)
{
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(iago_game_phases), 8 * 8) != hipSuccess)
        return -1;
    if (clear) {
        static unsigned long long zeros[8];
        if (hipMemcpyToSymbol(HIP_SYMBOL(iago_game_phases), zeros, 8 * 8) != hipSuccess)
            return -1;
    }
    return 0;
}
"""
    return s


STAMP_DEFS = """
// ---- diagnostic build only (tools/build_search_variants.py: search_walkstamps.so)
__device__ unsigned long long iago_walk_stamps[192];
__device__ __forceinline__ void iago_stamp(unsigned long long *st, int i)
{
    if (threadIdx.x == 0)
        st[i] = __builtin_amdgcn_s_memrealtime();
}
"""

WALK_END = """
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime();
            unsigned long long *g = iago_trunk::iago_walk_stamps + KINDBASE;
            WST[23] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 23; i++)
                atomicAdd(&g[i], WST[i + 1] - WST[i]);
            atomicAdd(&g[30], c1 - WST[30]);
            atomicAdd(&g[29], WST[23] - WST[0]);
            atomicAdd(&g[31], 1ull);
        }
"""


def walk_stamps(src):
    """The net workgroups' walks by phase, under load: 100 MHz clock stamps of thread 0 at the entry of a walk, after
    block1, per layer after the K loop / after the barrier that follows it / after the epilogue's barrier, and at the
    end of the head; summed per kind of walk (value pair, value single, policy) in iago_walk_stamps [kind][32]: slots
    0..22 the phases, 29 the whole walks, 30 their shader-clock cycles (s_memtime), 31 the walks counted.  Patched
    COPIES of the two walk headers go beside the variant's source."""
    nop = 'asm volatile("s_nop 15\\n\\ts_nop 7" ::: "memory");'
    sat = "        saturated |= !(vmax <= 65000.0f) || !(vsum.x + vsum.y == vsum.x + vsum.y);\n        __syncthreads();\n    }\n"
    t = open(os.path.join(CSRC, "conv_trunk_body.hpp")).read()
    t = patch(t, "namespace iago_trunk {\n", "namespace iago_trunk {\n" + STAMP_DEFS)
    t = patch(t, "    char *const T = trunk_lds;\n", "    char *const T = trunk_lds;\n    __shared__ unsigned long long wst[32];\n"
              "    iago_stamp(wst, 0);\n    if (threadIdx.x == 0) wst[30] = __builtin_amdgcn_s_memtime();\n")
    t = patch(t, "    __syncthreads();\n\n    // ---- per-lane addresses of the B operand.", "    __syncthreads();\n    iago_stamp(wst, 1);\n\n    // ---- per-lane addresses of the B operand.")
    t = patch(t, "        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T\n",
              "        iago_stamp(wst, 2 + 3 * L);\n        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T\n")
    t = patch(t, "        " + nop + "\n        __syncthreads();\n", "        " + nop + "\n        __syncthreads();\n        iago_stamp(wst, 3 + 3 * L);\n")
    t = patch(t, sat, sat.replace("        __syncthreads();\n    }\n", "        __syncthreads();\n        iago_stamp(wst, 4 + 3 * L);\n    }\n"))
    t = patch(t, "            P.out[W.index ? W.index[row] : row] = v;\n        }\n        return;",
              "            P.out[W.index ? W.index[row] : row] = v;\n        }\n" +
              WALK_END.replace("KINDBASE", "(TB == 2 ? 0 : 32)").replace("WST", "wst") + "        return;")
    # finer stamps (raw, wst[24..]): inside block1 after the staging barrier; inside the head after its MFMAs and after
    # each of its three barriers -> slots 23..27 (head: MFMAs, barrier 1, tap sums + barrier 2, fc10 + barrier 3, the
    # final sum + store) and 28 (block1: entry -> weights staged)
    t = patch(t, "        // the weights of an output channel are wave-uniform: each is fetched once",
              "        iago_stamp(wst, 24);\n        // the weights of an output channel are wave-uniform: each is fetched once")
    t = patch(t, "        __syncthreads();\n        if constexpr (SRCH) {\n#pragma unroll\n            for (int c = 0; c < 16; c++)\n                w10row[c] = ((const float4 *)(W.head_w",
              "        iago_stamp(wst, 25);\n        __syncthreads();\n        iago_stamp(wst, 26);\n        if constexpr (SRCH) {\n#pragma unroll\n            for (int c = 0; c < 16; c++)\n                w10row[c] = ((const float4 *)(W.head_w")
    t = patch(t, "        __syncthreads();\n        if ((tid >> 7) * 2 < TB) {", "        __syncthreads();\n        iago_stamp(wst, 27);\n        if ((tid >> 7) * 2 < TB) {")
    t = patch(t, "        __syncthreads();\n        if (tid < TB && b0 + tid < n_rows) {", "        __syncthreads();\n        iago_stamp(wst, 28);\n        if (tid < TB && b0 + tid < n_rows) {")
    t = patch(t, "            atomicAdd(&g[31], 1ull);\n", "            atomicAdd(&g[31], 1ull);\n"
              "            atomicAdd(&g[23], wst[25] - wst[22]);\n            atomicAdd(&g[24], wst[26] - wst[25]);\n"
              "            atomicAdd(&g[25], wst[27] - wst[26]);\n            atomicAdd(&g[26], wst[28] - wst[27]);\n"
              "            atomicAdd(&g[27], wst[23] - wst[28]);\n            atomicAdd(&g[28], wst[24] - wst[0]);\n")
    open(os.path.join(OUT, "conv_trunk_body_stamped.hpp"), "w").write(t)
    q = open(os.path.join(CSRC, "conv_policy_body.hpp")).read()
    q = patch(q, "    char *const T = policy_lds;\n", "    char *const T = policy_lds;\n    __shared__ unsigned long long pst[32];\n"
              "    iago_trunk::iago_stamp(pst, 0);\n    if (threadIdx.x == 0) pst[30] = __builtin_amdgcn_s_memtime();\n")
    q = patch(q, "    __syncthreads();\n\n    // ---- per-lane addresses of the B operand.", "    __syncthreads();\n    iago_trunk::iago_stamp(pst, 1);\n\n    // ---- per-lane addresses of the B operand.")
    q = patch(q, "        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T.\n",
              "        iago_trunk::iago_stamp(pst, 2 + 3 * L);\n        // ---- epilogue: every wave has read T for the last time; bias, ReLU, split, back into T.\n")
    q = patch(q, nop, nop + "\n        __syncthreads();\n        iago_trunk::iago_stamp(pst, 3 + 3 * L);\n        if (false)")
    q = patch(q, sat, sat.replace("        __syncthreads();\n    }\n", "        __syncthreads();\n        iago_trunk::iago_stamp(pst, 4 + 3 * L);\n    }\n"))
    q = patch(q, "        P.probs[row_id * 64 + lane] = e / sum;\n    }\n}",
              "        P.probs[row_id * 64 + lane] = e / sum;\n    }\n" + WALK_END.replace("KINDBASE", "64").replace("WST", "pst") + "}")
    open(os.path.join(OUT, "conv_policy_body_stamped.hpp"), "w").write(q)
    s = patch(src, '#include "conv_trunk_body.hpp" // (brings rollout_row_body.hpp)', '#include "conv_trunk_body_stamped.hpp"')
    s = patch(s, '#include "conv_policy_body.hpp"', '#include "conv_policy_body_stamped.hpp"')
    s += """
extern "C" __attribute__((visibility("default"))) int iago_debug_walk_stamps(unsigned long long *host, int clear)
{
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(iago_trunk::iago_walk_stamps), 192 * 8) != hipSuccess)
        return -1;
    if (clear) {
        static unsigned long long zeros[192];
        if (hipMemcpyToSymbol(HIP_SYMBOL(iago_trunk::iago_walk_stamps), zeros, 192 * 8) != hipSuccess)
            return -1;
    }
    return 0;
}
"""
    return s


def walk_stamps_no_a(src):
    """TIMING ONLY (wrong numbers): the stamped build with the weight loads of the K loops removed -- the A operands of
    every k-step are those of the first two.  What is left of the K loops' time is MFMA issue + LDS reads."""
    s = walk_stamps(src)
    for name in ("conv_trunk_body_stamped.hpp", "conv_policy_body_stamped.hpp"):
        path = os.path.join(OUT, name)
        t = open(path).read()
        for piece in ("hi", "mid", "lo"):
            w = {"hi": "wh", "mid": "wm", "lo": "wl"}[piece]
            old = "                a_%s[(tap + 2) %% 3][0] = %s[w2], a_%s[(tap + 2) %% 3][1] = %s[w2 + 32];\n" % (piece, w, piece, w)
            if piece == "mid" and "trunk" in name:
                continue
            t = patch(t, old, "                a_%s[(tap + 2) %% 3][0] = a_%s[tap %% 3][1], a_%s[(tap + 2) %% 3][1] = a_%s[tap %% 3][0];\n"
                      % (piece, piece, piece, piece))
        open(path.replace("_stamped", "_stamped_noa"), "w").write(t)
    s = s.replace("conv_trunk_body_stamped.hpp", "conv_trunk_body_stamped_noa.hpp").replace(
        "conv_policy_body_stamped.hpp", "conv_policy_body_stamped_noa.hpp")
    return s


def walk_stamps_no_b(src):
    """TIMING ONLY (wrong numbers): the stamped build with the LDS reads of the K loops removed (every tile multiplies
    the first two tiles' operands)."""
    s = walk_stamps(src)
    for name in ("conv_trunk_body_stamped.hpp", "conv_policy_body_stamped.hpp"):
        path = os.path.join(OUT, name)
        t = open(path).read()
        if "trunk" in name:
            t = patch(t, """                    bh[nxt] = *(const half8 *)p;
                    bl[nxt] = *(const half8 *)(p + 256);
                    __builtin_amdgcn_sched_barrier(0);""", """                    bh[nxt] = bl[cur];
                    bl[nxt] = bh[cur];
                    (void)p;
                    __builtin_amdgcn_sched_barrier(0);""")
        else:
            t = patch(t, """                    bh[nxt] = *(const half8 *)p;
                    bm[nxt] = *(const half8 *)(p + 256);
                    bl[nxt] = *(const half8 *)(p + 512);""", """                    bh[nxt] = bl[cur];
                    bm[nxt] = bh[cur];
                    bl[nxt] = bm[cur];
                    (void)p;""")
        open(path.replace("_stamped", "_stamped_nob"), "w").write(t)
    s = s.replace("conv_trunk_body_stamped.hpp", "conv_trunk_body_stamped_nob.hpp").replace(
        "conv_policy_body_stamped.hpp", "conv_policy_body_stamped_nob.hpp")
    return s


def epilogue_writes_no_conflicts(src):
    """TIMING ONLY (wrong numbers): the walks' epilogues write their f16 pieces to conflict-free addresses (8 bytes per
    lane, the 16 lanes of a ds_write_b64 group 8 bytes apart) instead of the cells' rows, whose 544-byte stride -- chosen
    for the K loops' ds_read_b128 -- puts the 16 cells of a group on 4 banks (4-way).  What SQ_LDS_BANK_CONFLICT and the
    walk lose without those conflicts (VERDICT r05 task 3a)."""
    t = open(os.path.join(CSRC, "conv_trunk_body.hpp")).read()
    t = patch(t, "            char *row = T + wrow[n & 3] + (n >> 2) * BS + (32 * wv + 4 * kq) * 2;",
              "            char *row = T + (n >> 2) * BS + (n & 3) * 8192 + c16 * 8 + kq * 128 + wv * 1024; (void)wrow;")
    t = patch(t, "                *(uint2 *)(row + 32 * m) =", "                *(uint2 *)(row + 512 * m) =")
    t = patch(t, "                *(uint2 *)(row + 32 * m + 256) =", "                *(uint2 *)(row + 512 * m + 4096) =")
    open(os.path.join(OUT, "conv_trunk_body_epiwrite.hpp"), "w").write(t)
    return patch(src, '#include "conv_trunk_body.hpp" // (brings rollout_row_body.hpp)', '#include "conv_trunk_body_epiwrite.hpp"')


def reply_times(s):
    """Post-mortem of a launch that gave up (round 6): per game, in the `trace` buffer, [0] the clock at its last request,
    [1] the clock at which a net workgroup wrote the reply to that request's mailbox (value or priors), [2] the clock at
    which its workgroup left the loop, [3] state | reply tag << 8 (tools/debug_split_abort.py)."""
    s = patch(s, "if (S.trace && blockIdx.x == 0 && tid == 0 && (int64_t)wg_count[0] <", "if (false && S.trace && blockIdx.x == 0 && tid == 0 && (int64_t)wg_count[0] <")
    s = patch(s, "if (S.trace && r == 0u && g < S.trace_rows) {", "if (false && S.trace && r == 0u && g < S.trace_rows) {")
    s = patch(s, """    atomicAdd((unsigned long long *)&S.totals[(uint32_t)g == NOBODY ? 11 : kind], 1ull);
}""", """    atomicAdd((unsigned long long *)&S.totals[(uint32_t)g == NOBODY ? 11 : kind], 1ull);
    if (S.trace && (uint32_t)g != NOBODY && g < S.trace_rows)
        S.trace[4 * g + 0] = wall_clock64();
}""")
    s = patch(s, """                    st(&S.rep_v[(int64_t)(job[6 * tid] & 0x7FFFFFFFu)], ((u64)job[6 * tid + 1] << 32) | bits);""",
              """                    st(&S.rep_v[(int64_t)(job[6 * tid] & 0x7FFFFFFFu)], ((u64)job[6 * tid + 1] << 32) | bits);
                if (S.trace && (job[6 * tid] & 0x7FFFFFFFu) != NOBODY && (int)(job[6 * tid] & 0x7FFFFFFFu) < S.trace_rows)
                    S.trace[4 * (int64_t)(job[6 * tid] & 0x7FFFFFFFu) + 1] = wall_clock64();""")
    s = patch(s, """                st(&S.rep_p[(int64_t)(job[0] & 0x7FFFFFFFu) * 64 + tid], ((u64)job[1] << 32) | __float_as_uint(res_p[tid]));""",
              """                st(&S.rep_p[(int64_t)(job[0] & 0x7FFFFFFFu) * 64 + tid], ((u64)job[1] << 32) | __float_as_uint(res_p[tid]));
            if (S.trace && tid == 0 && (int)(job[0] & 0x7FFFFFFFu) < S.trace_rows)
                S.trace[4 * (int64_t)(job[0] & 0x7FFFFFFFu) + 1] = wall_clock64();""")
    s = patch(s, """        S.cur_node[g] = (int32_t)epoch;
        S.leaf_value[g] = (float)state;""", """        S.cur_node[g] = (int32_t)epoch;
        S.leaf_value[g] = (float)state;
        if (S.trace && g < S.trace_rows) {
            S.trace[4 * g + 2] = wall_clock64();
            S.trace[4 * g + 3] = (int64_t)state | ((int64_t)epoch << 8);
        }""")
    # ... and per net workgroup (rows 4096 + blockIdx.x): [0] the clock at its last loop top, [1] the clock when it last began
    # to wait for a ticket, [2] ring << 32 | ticket it last waited for, [3] stage (1 loop top, 2 waiting in fetch, 3 fetched,
    # 4 left the kernel)
    s = patch(s, """        const long long c0 = wall_clock64();
        if (put_e) {""", """        const long long c0 = wall_clock64();
        if (S.trace && tid == 0 && 4096 + (int)blockIdx.x < S.trace_rows) {
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 0] = c0;
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 3] = 1;
        }
        if (put_e) {""")
    s = patch(s, """        u64 x = 0;
        int status = 0;
        for (uint32_t spins = 0;; spins++) {""", """        u64 x = 0;
        int status = 0;
        if (S.trace && tid == 0 && 4096 + (int)blockIdx.x < S.trace_rows && max_spins == 0u) {
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 1] = wall_clock64();
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 2] = ((int64_t)q << 32) | t;
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 3] = 2;
        }
        for (uint32_t spins = 0;; spins++) {""")
    s = patch(s, """        if (status == 0 && tid < 6)
            job[6 * which + tid] = (uint32_t)x;
        return status;""", """        if (status == 0 && tid < 6)
            job[6 * which + tid] = (uint32_t)x;
        if (S.trace && tid == 0 && 4096 + (int)blockIdx.x < S.trace_rows && max_spins == 0u)
            S.trace[4 * (4096 + (int64_t)blockIdx.x) + 3] = 3 + 10 * status;
        return status;""")
    s = patch(s, """        if (job[28] != 0u) {
            if (tid == 0) {""", """        if (job[28] != 0u) {
            if (S.trace && tid == 0 && 4096 + (int)blockIdx.x < S.trace_rows)
                S.trace[4 * (4096 + (int64_t)blockIdx.x) + 3] += 100;
            if (tid == 0) {""")
    # the ticket of a game's last request
    s = patch(s, """    if (S.trace && (uint32_t)g != NOBODY && g < S.trace_rows)
        S.trace[4 * g + 0] = wall_clock64();""", """    if (S.trace && (uint32_t)g != NOBODY && g < S.trace_rows) {
        S.trace[4 * g + 0] = wall_clock64();
        S.trace[4 * g + 3] = ((int64_t)kind << 32) | t;
    }""")
    return s


def atomic_stores(s):
    """Experiment (round 6): every granule of the protocol written by an atomic exchange instead of an agent-scope store."""
    return patch(s, "__device__ __forceinline__ void st(u64 *p, u64 x) { __hip_atomic_store(p, x, RLX_AGENT); }",
                 "__device__ __forceinline__ void st(u64 *p, u64 x) { (void)__hip_atomic_exchange(p, x, RLX_AGENT); }")


def unroll4(src):
    """... unrolled by four: no back-edge inside a layer (4 chunk pairs; the first layer's 2 run the remainder loop)."""
    for name in ("conv_trunk_body", "conv_policy_body"):
        t = open(os.path.join(CSRC, name + ".hpp")).read()
        t = patch(t, "        for (int cp = 0; cp < n_pairs; cp++) {", "#pragma unroll 4\n        for (int cp = 0; cp < n_pairs; cp++) {")
        open(os.path.join(OUT, name + "_unroll4.hpp"), "w").write(t)
    s = patch(src, '#include "conv_trunk_body.hpp" // (brings rollout_row_body.hpp)', '#include "conv_trunk_body_unroll4.hpp"')
    return patch(s, '#include "conv_policy_body.hpp"', '#include "conv_policy_body_unroll4.hpp"')


def unroll2(src):
    """The chunk-pair loops of the walks' K loops unrolled by two (n_pairs is 2 or 4): half the loop-top waits and address
    updates, twice the code (round 5 measured this once, on a slow box, without an A/B partner)."""
    for name in ("conv_trunk_body", "conv_policy_body"):
        t = open(os.path.join(CSRC, name + ".hpp")).read()
        t = patch(t, "        for (int cp = 0; cp < n_pairs; cp++) {", "#pragma unroll 2\n        for (int cp = 0; cp < n_pairs; cp++) {")
        open(os.path.join(OUT, name + "_unroll2.hpp"), "w").write(t)
    s = patch(src, '#include "conv_trunk_body.hpp" // (brings rollout_row_body.hpp)', '#include "conv_trunk_body_unroll2.hpp"')
    return patch(s, '#include "conv_policy_body.hpp"', '#include "conv_policy_body_unroll2.hpp"')


def rollout_no_conflicts():
    """TIMING ONLY (wrong numbers): rollout_row_kernel.hip with the policy's table reads made bank-conflict-free (every
    lane reads the slot of its own lane number instead of the entry its window selects): the upper bound of what a
    table replicated per lane slot could gain (VERDICT r04 task 7).  -> tools/_build/rollout_noconf.so"""
    body = open(os.path.join(CSRC, "rollout_row_body.hpp")).read()
    body = patch(body, "fac[ky * 2 + pl] = *(const f4 *)(tb + idx + (ky * 2 + pl) * 1024);",
                 "fac[ky * 2 + pl] = *(const f4 *)(tb + ((idx & 0u) | (L.l << 4)) + (ky * 2 + pl) * 1024);")
    open(os.path.join(OUT, "rollout_row_body_noconf.hpp"), "w").write(body)
    src = open(os.path.join(CSRC, "rollout_row_kernel.hip")).read()
    src = patch(src, '#include "rollout_row_body.hpp"', '#include "rollout_row_body_noconf.hpp"')
    return src


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "search_kernel.hip")).read()
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "iago_amd", "_obj", "*.o"))) if "search_kernel" not in o]
    assert objs, "build the product first: python -m iago_amd.build"
    only = sys.argv[1:]
    if "rollout_noconf" in only:
        path = os.path.join(OUT, "rollout_noconf.hip")
        open(path, "w").write(rollout_no_conflicts())
        obj = os.path.join(OUT, "rollout_noconf.o")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", path, "-o", obj])
        others = [o for o in sorted(glob.glob(os.path.join(ROOT, "iago_amd", "_obj", "*.o"))) if "rollout_row_kernel" not in o]
        so = os.path.join(OUT, "rollout_noconf.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + others + [obj])
        print(so)
        return
    for name, fn in (("search_log", request_log), ("search_phases", phase_stamps), ("search_walkstamps", walk_stamps),
                     ("search_walkstamps_noa", walk_stamps_no_a), ("search_walkstamps_nob", walk_stamps_no_b),
                     ("search_epiwrite", epilogue_writes_no_conflicts), ("search_unroll2", unroll2), ("search_unroll4", unroll4),
                     ("search_replytimes", reply_times), ("search_atomicst", atomic_stores)):
        if only and name not in only:
            continue
        if not only and name in ("search_walkstamps_noa", "search_walkstamps_nob", "search_unroll4"):
            continue   # (timing-only builds of round 5's first K loops: their anchors are of that code; kept for the record)
        path = os.path.join(OUT, name + ".hip")
        open(path, "w").write(fn(src))
        obj = os.path.join(OUT, name + ".o")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden",
                               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", path, "-o", obj])
        so = os.path.join(OUT, name + ".so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [obj])
        print(so)


if __name__ == "__main__":
    sys.exit(main())
