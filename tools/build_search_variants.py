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
    s = patch(s, '''    atomicAdd((unsigned long long *)&S.totals[kind], 1ull);
}''', '''    atomicAdd((unsigned long long *)&S.totals[kind], 1ull);
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
    s = patch(s, "if (S.trace && blockIdx.x == 0 && tid == 0 && iters <", "if (false && S.trace && blockIdx.x == 0 && tid == 0 && iters <")
    s = patch(s, "if (S.trace && r == 0u && g < S.trace_rows) {", "if (false && S.trace && r == 0u && g < S.trace_rows) {")
    return s


def phase_stamps(s):
    s = patch(s, '''    for (;;) {
        bool busy = false; // this game did something in this iteration''', '''    long long ph[5] = {0, 0, 0, 0, 0};
    for (;;) {
        long long c_a = wall_clock64();
        bool busy = false; // this game did something in this iteration''')
    s = patch(s, '''            // ---- descent (MCTS.py:105-133): from the root, or on from the leaf whose priors arrived
''', '''            { const long long c = wall_clock64(); ph[0] += c - c_a; c_a = c; }
            // ---- descent (MCTS.py:105-133): from the root, or on from the leaf whose priors arrived
''')
    s = patch(s, '''        const bool rolls = mine && need_z && (state == ST_ROLL || state == ST_ROLL_FRESH);
        {''', '''        const bool rolls = mine && need_z && (state == ST_ROLL || state == ST_ROLL_FRESH);
        { const long long c = wall_clock64(); ph[1] += c - c_a; c_a = c; }
        {''')
    s = patch(s, '''        if (mine) {
            if (state == ST_ROLL) {''', '''        { const long long c = wall_clock64(); ph[2] += c - c_a; c_a = c; }
        if (mine) {
            if (state == ST_ROLL) {''')
    s = patch(s, '''        iters++;
        if (mine && r == 0u) {
            const int prog''', '''        { const long long c = wall_clock64(); ph[3] += c - c_a; c_a = c; }
        iters++;
        if (mine && r == 0u) {
            const int prog''')
    s = patch(s, '''        if (!__syncthreads_or(busy)) {
            idle_iters++;''', '''        { const long long c = wall_clock64(); ph[4] += c - c_a; c_a = c; }
        if (!__syncthreads_or(busy)) {
            idle_iters++;''')
    s = patch(s, '''    if (tid == 0) {
        atomicAdd((unsigned long long *)&S.totals[2], (unsigned long long)iters);''', '''    if (tid == 0 && blockIdx.x == 0)
        for (int i = 0; i < 5; i++)
            S.totals[11 + i] = ph[i];
    if (tid == 0) {
        atomicAdd((unsigned long long *)&S.totals[2], (unsigned long long)iters);''')
    return s


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "search_kernel.hip")).read()
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "iago_amd", "_obj", "*.o"))) if "search_kernel" not in o]
    assert objs, "build the product first: python -m iago_amd.build"
    for name, fn in (("search_log", request_log), ("search_phases", phase_stamps)):
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
