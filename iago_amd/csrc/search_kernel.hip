// search_kernel.hip -- a whole PV-MCTS search (n_sims playouts of every game: MCTS.get_move's loop,
// MCTS.py:139-147), or whole self-play games (the turn loop of game.py:117-142,253-255 around it), as
// ONE persistent launch in which every game runs on its own clock.
//
// Why.  MCTS.playout (MCTS.py:105-133) is sequential inside a game, but the games of a batch are
// independent.  The per-playout engine (mcts_kernels.hip + conv_trunk_kernel.hip: descent, leaf
// evaluation, backup as three launches per playout for all games) makes every game wait for the
// slowest kind of playout -- the 16 % that end on a leaf without a stored value, whose one-board walk
// through the Value net takes 70 us of a 128 us playout --, runs the policy net as batches that hold
// most of the chip while the playouts' kernels want it, and ends every move with a barrier.  Here the
// chip is a pool of workgroups (one per CU) of two kinds:
//   * GAME workgroups (the first ceil(n_games / 32) of the grid: dispatched first, so always resident):
//     each owns 32 games and loops over descent (8 lanes per game: select, expansion, continued
//     descent, exactly descend_kernel's arithmetic) -> rollout of the leaves reached (the
//     16-lanes-per-board body in passes of 16 boards, Philox stream = stream base + turn x n_sims + the
//     game's own playout count) -> backup (mix_backup_path_kernel's arithmetic) and, with whole games,
//     the game's move between two of its searches (most visited child, update_with_move, the stone,
//     the books, the recorded tuple).  A game whose leaf has no stored value -- and whose position the
//     shared position table does not hold -- writes the leaf's position into a request ring and waits
//     for the value (its rollout runs meanwhile); a game whose leaf expands (n_visits >= n_thr,
//     MCTS.py:109) does the same and waits for the priors -- the policy net runs exactly where the
//     reference runs it (no look-ahead: no evaluation is wasted); the workgroup's other games go on.
//   * NET workgroups (the rest of the grid): each takes a ticket of a ring that has an entry waiting
//     (its home ring first: one ring per net, the policy ring at home on 2 of the 8 XCDs), reads the
//     entry, walks the board through the Value net (trunk_item<true, 1>, or <true, 2> for two entries
//     together) or the SLPolicy net (policy_item) -- the kernels' own device functions: bit-identical
//     numbers -- and publishes the result in the game's mailbox (and the position table).
// Scheduling inside a game workgroup (timing only): the games that reach a leaf in an iteration are packed
// into rows of 16 boards for the rollout body (about half of the 32 do: one pass instead of two); a game
// more than `pace_margin` playouts ahead of the batch's mean progress holds while requests queue up (a
// batch ends with its slowest game, and a game is slow when it asks the nets a lot: what the leaders do
// not ask for, the laggards get); a level at which every descending game of the wave has ONE child, a
// pass (the chains the reference grows under a finished game), is followed without scoring.  And idle net
// workgroups are put to use: when a game asks for the priors of a node that is about to expand while net
// workgroups poll and nothing waits in the rings, the node's children are walked through the value net for
// the position table -- requests nobody waits for; the children's first visits find their values there.
// A game's sequence of playouts -- leaves, values, priors, rollouts, backups, expansions, moves -- is
// exactly the reference's; only the interleaving between games changes: trees, moves and results are
// bit-identical to the per-playout engine's (tests/test_search_persistent_gpu.py; the comparisons with
// the oracle's MCTS.py restatement in tests/test_mcts_production_gpu.py run on this kernel).
//
// Inter-workgroup traffic (cdna_hip_programming.md, guideline 16, form R2): every shared word is an
// 8-byte {tag, 32-bit value} granule written by ONE agent-scope atomic store and polled by agent-scope
// atomic loads -- the data is the flag, no fence, no cache invalidation that would cost the net
// workgroups their L2-resident weights.  Ring entry t (ticket t, tag t + 1): 6 granules (kind | game,
// reply tag, the position's four words); replies: one granule (value) or 64 (priors) tagged with the
// request's reply tag.  The tree, the cursors and the paths of a game are touched by its own workgroup
// only.  Every wait is a bounded poll: the launch ends by itself when a clock limit passes (abort word).
#include "mcts_dev.hpp"
#include "conv_trunk_body.hpp" // (brings rollout_row_body.hpp)
#include "conv_policy_body.hpp"

#include <cstdlib>

namespace {
using namespace iago;
using namespace iago_mcts;

typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr uint32_t QCAP = IAGO_SEARCH_QUEUE_ENTRIES; // entries of a request ring (a game has at most ONE request outstanding:
                                                     // >= the 4096 games a launch can hold)
constexpr int CTL_FINISHED = 2, CTL_ABORT = 3;
// two request rings, one per kind of net work: head (tickets handed out) / tail (entries reserved) of ring q
__host__ __device__ constexpr int ctl_head(uint32_t q) { return 8 + 2 * (int)q; }
__host__ __device__ constexpr int ctl_tail(uint32_t q) { return 9 + 2 * (int)q; }
enum { ST_READY = 0, ST_WAIT_PRIOR, ST_PRIOR_READY, ST_ROLL, ST_ROLL_FRESH, ST_WAIT_VALUE, ST_HAVE_VALUE, ST_DONE, ST_TURN, ST_MOVE };
constexpr int CTL_NO_CHILDREN = 4; // a searched root had no children (n_sims below n_thr)
// pacing (below): sum over the games in play of their progress (turn x n_sims + playouts of the turn), games in play
constexpr int CTL_PROGRESS = 5, CTL_PLAYING = 6;
constexpr int CTL_NET_WGS = 7;            // net workgroups of this launch (written by the launch: the grid follows the device)
constexpr int CTL_IDLE = 14;              // net workgroups that found nothing to do at their last look (they poll)
constexpr uint32_t NOBODY = 0x7FFFFFFFu;  // "game" of a request nobody waits for (its value goes to the position table only)
constexpr uint32_t KIND_VALUE = 0u, KIND_POLICY = 1u;
// games of a GAME workgroup: 8 lanes per game in the descent and the backup (16 games = two waves, 32 = all four),
// rollouts in passes of 16 boards (the 16-lanes-per-board body)
constexpr int GAMES_PER_WG = IAGO_SEARCH_GAMES_PER_WORKGROUP; // at most (the launch's own number: SearchParams::games_per_wg)

// dynamic LDS of the kernel: the walks' images (the larger of a value pair's and a policy board's, with their heads), then
// block1's weights of both nets
constexpr int SEARCH_IMG_V = iago_trunk::lds_alloc(2) + iago_trunk::head_lds(2);
constexpr int SEARCH_IMG_P = iago_policy::BS + 1024 + iago_policy::HEAD_FLOATS * 4;
constexpr int SEARCH_IMG_TOP = SEARCH_IMG_V > SEARCH_IMG_P ? SEARCH_IMG_V : SEARCH_IMG_P;

struct SearchParams {
    Tree T;
    const uint64_t *root_own, *root_opp;
    const uint8_t *active;
    float c_puct, lmbda;
    int32_t n_thr, n_sims, n_game_wgs, games_per_wg;
    int32_t *cur_node;
    uint64_t *cur_own, *cur_opp;
    int32_t *path;
    int32_t path_stride;
    int32_t *done;
    uint8_t *roll;
    int8_t *z;
    float *leaf_value;
    int8_t *z_log;
    int32_t *z_log_n;
    int32_t z_log_rows;
    u64 *q_slots;
    uint32_t *ctl;
    u64 *rep_v, *rep_p;
    int64_t *totals; // [0] value evaluations, [1] policy evaluations, [2] game-workgroup iterations, [3] pair walks,
                     // [4] / [5] net workgroups' waiting / walking time (100 MHz ticks), [6] idle game iterations, [7] game time
    int32_t *stats;
    uint64_t *wg_own, *wg_opp;
    long long clock_limit; // wall_clock64 ticks (100 MHz) after which the launch gives up
    // whole self-play games in the launch (iago_selfplay_persistent): max_turns > 0.  A game then walks through
    // its turns on its own: legal moves of the mover, a search of n_sims playouts (or a pass), the most visited
    // move (MCTS.get_move, MCTS.py:147), MCTS.update_with_move (MCTS.py:149-154), the move on the board and the
    // turn bookkeeping of game.py:117-142,253-255 -- what SelfPlayEngine.play drives per turn for all games
    int32_t max_turns;
    uint64_t *game_own, *game_opp;   // [n_games] in: the start positions (own = colour 1, the first mover); out: final
    int32_t *n_turns;                // [n_games] out: turns the game took (even, or max_turns)
    uint64_t *rec_own, *rec_opp;     // optional [max_turns][n_games]: the position before every turn (own = mover)
    uint8_t *rec_valid;              // [max_turns][n_games]: the mover had a move and searched
    int8_t *rec_move;                // [max_turns][n_games]: the move played, -1 = pass / no turn
    int32_t *rec_pi;                 // [max_turns][n_games][64]: the root's visit counts by action
    // optional position table shared by all games and launches: value_func(state) is a pure function of the position, and
    // games that start from one position keep meeting each other's positions (9.5 % of the value requests of 1024 games
    // x 100 playouts repeat a position asked for before: LABNOTES.md).  Direct-mapped, 32-byte entries {seq, own, opp,
    // value bits} under a per-entry sequence lock, every word an agent-scope atomic
    u64 *vtable;
    uint32_t vtable_mask;  // slots - 1 (slots a power of two), 0 = no table
    int64_t *trace;        // optional diagnostic [trace_rows][4]: game workgroup 0 samples (ticks, tail, head, finished) per iteration
    int32_t trace_rows;
    int32_t policy_xcds;   // XCDs (of 8) whose workgroups serve the POLICY ring first
    int32_t pair_backlog;  // entries that must be waiting (beyond the tickets handed out) for a net workgroup to take two
    // pacing: a game that is more than `pace_margin` playouts ahead of the mean progress of the games in play starts no
    // new playout while more than `pace_backlog` requests wait in the rings (< 0: no pacing).  Why: the batch ends with
    // its slowest game, and a game's speed in the net-bound middle of a batch is set by how many requests it makes (end
    // time against requests sent: correlation 0.84-0.90, ends spread over +-20 % at 400 playouts per move; LABNOTES.md);
    // what the leaders do not ask for, the laggards get.  Timing only: a game's own sequence of playouts is untouched
    int32_t pace_margin, pace_backlog;
    // values ahead of their first visit, on idle hands only: when a game asks for the priors of a node that expands while
    // at least `ahead_idle` net workgroups poll and nothing waits in the rings, the node's children (each of them a leaf
    // without a value at ITS first visit: the first in this very playout, the others a few playouts from now) are walked
    // through the value net beside the policy walk and put into the position table (< 0: off; needs the table)
    int32_t ahead_idle;
    int32_t roll_defer; // rollouts: games a full pass of 16 may leave over for the next iteration (0: every game at once)
    int32_t path_lds_cap; // bytes of the launch's dynamic LDS a game workgroup may keep its games' recorded paths in
};

__device__ __forceinline__ u64 ld(const u64 *p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ void st(u64 *p, u64 x) { __hip_atomic_store(p, x, RLX_AGENT); }

// one request: 6 granules of entry `t` (one lane)
__device__ __forceinline__ void send_request(const SearchParams &S, uint32_t kind, int64_t g, uint32_t reply_tag,
                                             uint64_t own, uint64_t opp)
{
    const uint32_t t = __hip_atomic_fetch_add(&S.ctl[ctl_tail(kind)], 1u, RLX_AGENT);
    u64 *e = S.q_slots + ((u64)kind * QCAP + t % QCAP) * 8u;
    const u64 tag = (u64)(t + 1u) << 32;
    st(e + 1, tag | reply_tag);
    st(e + 2, tag | (uint32_t)own);
    st(e + 3, tag | (uint32_t)(own >> 32));
    st(e + 4, tag | (uint32_t)opp);
    st(e + 5, tag | (uint32_t)(opp >> 32));
    st(e + 0, tag | (kind << 31) | (uint32_t)g);
    atomicAdd((unsigned long long *)&S.totals[(uint32_t)g == NOBODY ? 11 : kind], 1ull);
}

__device__ __forceinline__ uint32_t vtable_slot(const SearchParams &S, uint64_t own, uint64_t opp)
{
    uint64_t h = own * 0x9E3779B97F4A7C15ull ^ (opp + 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    h *= 0x94D049BB133111EBull;
    h ^= h >> 32;
    return (uint32_t)h & S.vtable_mask;
}

// value of the position if the table holds it (one lane); the five loads are in flight together, the entry counts only
// if its sequence word is even and did not move
__device__ __forceinline__ bool vtable_get(const SearchParams &S, uint64_t own, uint64_t opp, uint32_t &bits, uint32_t &writer)
{
    const u64 *e = S.vtable + (u64)vtable_slot(S, own, opp) * 4u;
    const u64 s1 = ld(e), o = ld(e + 1), p = ld(e + 2), v = ld(e + 3);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (compiler order only: the words are agent-scope atomics)
    const u64 s2 = ld(e);
    bits = (uint32_t)v;
    writer = (uint32_t)(s1 >> 32);
    return s1 != 0ull && (s1 & 1ull) == 0ull && s1 == s2 && o == own && p == opp && (v >> 32) == (s1 & 0xFFFFFFFFull);
}

// (one lane) the value the net has just computed for the position, in two steps.  vtable_put_begin takes the slot with
// ONE atomic (fetch-or of the sequence word's low bit: odd = being written; a slot somebody else is writing is left
// alone) and stores the position and the value; vtable_put_end -- called by the same lane before its next request, a
// walk later -- publishes the new even sequence word once those stores have landed.  (As one routine -- load, compare-and-
// swap, stores, wait, store -- it was three dependent round trips to L2 behind every value walk: LABNOTES.md, round 5.)
// Sequence word: low half = the sequence number (odd while the entry is being written, never 0 once used), high half =
// the game whose request put the value there (diagnostic: hits by the same game / by another game, totals[12])
__device__ __forceinline__ u64 *vtable_put_begin(const SearchParams &S, uint64_t own, uint64_t opp, uint32_t bits, uint32_t writer,
                                                 u64 &publish)
{
    u64 *e = S.vtable + (u64)vtable_slot(S, own, opp) * 4u;
    const u64 s = __hip_atomic_fetch_or(e, 1ull, RLX_AGENT);
    if (s & 1ull)
        return nullptr;
    uint32_t seq = (uint32_t)s + 2u;
    seq = seq ? seq : 2u;
    st(e + 1, own);
    st(e + 2, opp);
    st(e + 3, ((u64)seq << 32) | bits);   // (the value word carries the sequence number it belongs to)
    publish = ((u64)writer << 32) | seq;
    return e;
}
__device__ __forceinline__ void vtable_put_end(u64 *e, u64 publish)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st(e, publish);
}

template <int CTRL>
__device__ __forceinline__ void most_visited_step(int &n, int &a)
{
    const int on = (int)dpp_u32<CTRL>((uint32_t)n), oa = (int)dpp_u32<CTRL>((uint32_t)a);
    const bool take = on > n || (on == n && oa < a);
    n = take ? on : n;
    a = take ? oa : a;
}

__device__ __forceinline__ bool group8_all(bool x)
{
    return group8_add(x ? 0u : 1u) == 0u;
}

// Node.update_recursive (MCTS.py:51-72) over the recorded path + the leaf mix (MCTS.py:123-125): the
// arithmetic of mix_backup_path_kernel, 8 lanes per game.
__device__ __forceinline__ void backup_game(const SearchParams &S, int64_t g, uint32_t r, int leaf, bool fresh, float vg,
                                            int path_n, const int8_t zl, const int path_at)
{
    const int64_t base = g * (int64_t)S.T.capacity;
    const float lmbda = S.lmbda;
    if (fresh && lmbda < 1.0f && r == 0u)
        S.T.nodes[base + leaf].v = vg; // value_func(leaf), now stored (the value cache)
    const int8_t zg = lmbda > 0.0f ? zl : (int8_t)0; // (the rollout's result, from the workgroup's LDS)
    const float a = (lmbda < 1.0f) ? (float)(1.0 - (double)lmbda) * vg : 0.0f;
    const float b = (lmbda > 0.0f) ? (float)((double)lmbda * (double)zg) : 0.0f;
    const float lv = a + b;
    if (r == 0u) {
        S.leaf_value[g] = lv;
        if (S.z_log && lmbda > 0.0f) {
            const int k = S.z_log_n[g];
            S.z_log_n[g] = k + 1;
            if (k < S.z_log_rows)
                S.z_log[(int64_t)k * S.T.n_games + g] = zg;
        }
    }
    const int len = path_n < S.path_stride ? path_n : S.path_stride;
    const int32_t *const gpath = S.path + g * (int64_t)S.path_stride;
    for (int d = (int)r; d < len; d += 8) {
        // (path_at >= 0: the path is in the workgroup's LDS at that word; an LDS access, not a flat one)
        const int node = path_at >= 0 ? ((const int32_t *)iago_trunk::trunk_lds)[path_at + d] : gpath[d];
        uint2 *nq = (uint2 *)&S.T.nodes[base + node];
        const uint2 old = *nq;
        const int n = (int)old.x + 1;                // MCTS.py:61
        const float q = __uint_as_float(old.y);
        *nq = make_uint2((uint32_t)n, __float_as_uint(q + (lv - q) / (float)n)); // MCTS.py:63
    }
}

// The games of a GAME workgroup (8 lanes per game: 32 games = the four waves).
__device__ __forceinline__ void game_workgroup(const SearchParams &S, const iago_row::HwParams &R, const long long t0)
{
    const Tree &T = S.T;
    const int tid = threadIdx.x;
    const bool mine = tid < 8 * S.games_per_wg;
    const int64_t g = (int64_t)blockIdx.x * S.games_per_wg + (tid >> 3);
    const Lane8 L = make_lane8(threadIdx.x);
    const uint32_t r = L.l8;
    const bool exists = mine && g < T.n_games;
    const int64_t base = exists ? g * (int64_t)T.capacity : 0;
    const bool need_v = S.lmbda < 1.0f, need_z = S.lmbda > 0.0f;
    // the descent's recorded path (Node.update_recursive's ancestors): in the workgroup's dynamic LDS -- a game workgroup
    // walks no net while it has games -- when 32 paths fit there, else in the caller's array.  (From global memory the
    // backup was two dependent round trips to L2: the path entry, then the node.)
    const bool path_lds = (size_t)S.games_per_wg * (size_t)S.path_stride * 4u <= (size_t)S.path_lds_cap;
    const int path_at = path_lds ? (tid >> 3) * S.path_stride : -1; // word of the dynamic LDS where this game's path starts
    int32_t *const gpath = S.path + (exists ? g : 0) * (int64_t)S.path_stride;
    // positions, Philox stream offsets and results between the descent / backup and the rollout passes: LDS (RowHandoff)
    __shared__ uint64_t h_own[GAMES_PER_WG], h_opp[GAMES_PER_WG];
    __shared__ int32_t h_stream[GAMES_PER_WG];
    __shared__ int8_t h_z[GAMES_PER_WG];
    const iago_row::RowHandoff hand = {h_own, h_opp, h_stream, h_z, (int32_t)((int64_t)blockIdx.x * S.games_per_wg)};
    const int gl = tid >> 3; // this game's number within the workgroup

    const bool whole = S.max_turns > 0;
    int state = (exists && S.active[g] != 0 && S.n_sims > 0) ? (whole ? ST_TURN : ST_READY) : ST_DONE;
    uint32_t epoch = 0u; // reply tag of the game's last request (never 0 once used)
    int n_done = 0;
    if (exists && r == 0u) {
        S.done[g] = 0;
        S.roll[g] = 0;
    }
    // whole games: the game's own position (own = side to move), its books (game.py:32,117-142) and its turn
    uint64_t g_own = 0, g_opp = 0;
    int turn = 0, stones = 4;
    bool pass_flg = false, g_over = false;
    const int search_end = whole ? ST_MOVE : ST_DONE; // where a game goes when its search's last playout is backed up
    if (whole && exists) {
        g_own = S.game_own[g];
        g_opp = S.game_opp[g];
        if (state == ST_DONE && r == 0u)
            S.n_turns[g] = 0;
    }
    // cursor of the descent (kept across iterations while the game waits for priors)
    int node = 0, fc = -1, k = 0, nv = 0, path_n = 0, leaf = 0;
    uint32_t vbits = 0;
    uint64_t own = 0, opp = 0;
    bool may_expand = false, leaf_fresh = false;
    float v_reply = 0.0f;
    int st_levels = 0, st_children = 0;
    // (diagnostic counters of the workgroup, kept by thread 0 in LDS: as per-thread 64-bit registers they were 12 VGPRs
    // live across the whole loop -- what the game launch of the role split spilled to scratch memory)
    __shared__ uint32_t wg_count[6]; // [0] iterations, [1] idle iterations, [2..5] iterations with 0 / 1..16 / 17..20 / more games rolled out
    if (tid < 6)
        wg_count[tid] = 0u;
    __shared__ int32_t roll_list[GAMES_PER_WG]; // games whose leaf is rolled out in this iteration, packed
    __shared__ uint32_t roll_wave[BLOCK / 64], roll_wave_old[BLOCK / 64];
    bool deferred = false; // this game's rollout was put off to the next iteration's first pass
    bool table_ready = false;                   // the rollout's factor table is in LDS (from the first pass on)
    // pacing: what this game has added to CTL_PROGRESS / whether CTL_PLAYING counts it; the workgroup's changes of an
    // iteration are collected in LDS and go out as one atomic each; pace[2]: the progress above which a game holds
    __shared__ int32_t pace[4];
    int contrib = 0;
    bool in_play = state != ST_DONE;
    if (tid < 4)
        pace[tid] = tid == 2 ? 0x7fffffff : 0;
    __syncthreads();
    if (mine && in_play && r == 0u)
        atomicAdd(&pace[1], 1);
    int pace_limit = 0x7fffffff;

    for (;;) {
        bool busy = false; // this game did something in this iteration
        if (mine) {
            // ---- replies
            if (state == ST_WAIT_PRIOR) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    ok = ok && (uint32_t)(ld(&S.rep_p[g * 64 + (int)r * 8 + i]) >> 32) == epoch;
                if (group8_all(ok))
                    state = ST_PRIOR_READY;
            } else if (state == ST_WAIT_VALUE) {
                const u64 x = ld(&S.rep_v[g]);
                if ((uint32_t)(x >> 32) == epoch) {
                    v_reply = __uint_as_float((uint32_t)x);
                    state = ST_HAVE_VALUE;
                }
            }
            // (the 8 lanes of a game load the same word in the same instruction, or agree through group8_all:
            // one state per game)
            // ---- backup of the games whose value has arrived (their rollout ran when they descended)
            if (state == ST_HAVE_VALUE) {
                backup_game(S, g, r, leaf, true, v_reply, path_n, h_z[gl], path_at);
                n_done++;
                if (S.trace && r == 0u)
                    atomicAdd((unsigned long long *)&S.totals[9], 1ull); // (diagnostic: playouts over time)
                if (r == 0u)
                    S.done[g] = turn * S.n_sims + n_done;
                state = n_done >= S.n_sims ? search_end : ST_READY;
                busy = true;
            }
            // ---- whole games: the turn's end (the move) and the next turn's start, until the game searches again
            // or is over (a pass leads straight on to the next turn: at most a few rounds)
            if (whole) {
                for (int rep = 0; rep < 6; rep++) {
                    const bool at_move = state == ST_MOVE, at_turn = state == ST_TURN;
                    if (__builtin_amdgcn_ballot_w64(at_move || at_turn) == 0ull)
                        break;
                    busy = busy || at_move || at_turn;
                    const uint64_t lg = group8_legal(to_lane(g_own, L), to_lane(g_opp, L), L);
                    const bool can_move = lg != 0ull && !g_over;
                    if (at_turn && can_move) {
                        // the mover searches: MCTS.get_move(state, color) (game.py:112)
                        n_done = 0;
                        if (r == 0u)
                            S.done[g] = turn * S.n_sims; // (the rollouts' Philox stream: stream base + turn x n_sims + playout)
                        state = ST_READY;
                    }
                    const bool moving = at_move || (at_turn && !can_move);
                    // the root's children are the mover's legal moves in ascending order (Node.expand)
                    const int root = moving ? T.root[g] : 0;
                    const int rfc = moving ? T.nodes[base + root].first_child : -1;
                    int best_n = -1, best_a = 0x7fffffff;
                    int row_n[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int a = (int)(8u * r) + i;
                        int n = 0;
                        if (at_move && rfc >= 0 && ((lg >> a) & 1ull)) {
                            n = T.nodes[base + rfc + __popcll(lg & ((1ull << a) - 1ull))].n_visits;
                            if (n > best_n) { // the first maximum wins (MCTS.py:147)
                                best_n = n;
                                best_a = a;
                            }
                        }
                        row_n[i] = n;
                    }
                    // argmax over the 8 lanes: more visits, then the lower action
                    most_visited_step<DPP_XOR1>(best_n, best_a);
                    most_visited_step<DPP_XOR2>(best_n, best_a);
                    most_visited_step<DPP_HALF_MIRROR>(best_n, best_a);
                    if (moving) {
                        int mv = -1;
                        if (at_move) {
                            mv = best_n >= 0 ? best_a : -2;
                            if (mv == -2 && r == 0u) // max() of an empty children dict (MCTS.py:147): n_sims < n_thr
                                __hip_atomic_store(&S.ctl[CTL_NO_CHILDREN], 1u, RLX_AGENT);
                        }
                        if (S.rec_move) {
                            const int64_t row = (int64_t)turn * T.n_games + g;
                            if (r == 0u) {
                                S.rec_own[row] = g_own;
                                S.rec_opp[row] = g_opp;
                                S.rec_valid[row] = at_move ? 1 : 0;
                                S.rec_move[row] = (int8_t)(at_move ? mv : -1);
                            }
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                S.rec_pi[row * 64 + (int)(8u * r) + i] = row_n[i];
                        }
                        // MCTS.update_with_move (MCTS.py:149-154) for the games not over: the child becomes the root,
                        // or (no such child) a fresh Node(None, 1.0)
                        if (!g_over && r == 0u) {
                            int child = -1;
                            if (rfc >= 0) {
                                if (mv >= 0 && ((lg >> mv) & 1ull))
                                    child = rfc + __popcll(lg & ((1ull << mv) - 1ull));
                                else if (mv == -1 && (int)T.nodes[base + rfc].action == -1)
                                    child = rfc;
                            }
                            if (child >= 0) {
                                T.root[g] = child;
                                T.nodes[base + child].parent = -1;
                            } else {
                                init_node(T, base, -1, -2, 1.0f + 0.1f);
                                T.n_nodes[g] = 1;
                                T.root[g] = 0;
                            }
                        }
                        // the stone, the books, the swap of sides (iago_play_turn; game.py:117-142,253-255)
                        const bool placed = at_move && mv >= 0;
                        const uint64_t f = group8_flips(to_lane(g_own, L), to_lane(g_opp, L), (uint32_t)mv & 63u, L);
                        uint64_t o = g_own, p = g_opp;
                        if (placed) {
                            const uint64_t bit = 1ull << (mv & 63);
                            o = g_own | f | bit;
                            p = g_opp & ~f & ~bit;
                        }
                        const bool was_over = g_over;
                        stones += at_move ? 1 : 0;
                        const bool passing = !at_move && !was_over;
                        if (passing && pass_flg)
                            stones = 64;                       // a pass after a pass ends the game
                        if (!was_over)
                            pass_flg = passing;
                        if (turn % 2 == 1)                     // `while stone_num < 64` once per pair of turns
                            g_over = was_over || stones >= 64;
                        g_own = p;
                        g_opp = o;
                        turn++;
                        if (turn >= S.max_turns || (turn % 2 == 0 && g_over)) {
                            if (r == 0u) {
                                S.n_turns[g] = turn;
                                S.game_own[g] = g_own;         // (colour 1's stones after an even number of turns)
                                S.game_opp[g] = g_opp;
                            }
                            state = ST_DONE;
                            if (S.trace && r == 0u && g < S.trace_rows) { // (diagnostic: the game's end, its requests)
                                int64_t *row = S.trace + 4 * ((int64_t)S.trace_rows - 1 - g);
                                row[0] = wall_clock64() - t0;
                                row[1] = epoch;
                                row[2] = turn;
                            }
                        } else {
                            state = ST_TURN;
                        }
                    }
                }
            }
            // ---- descent (MCTS.py:105-133): from the root, or on from the leaf whose priors arrived
            const bool fresh_start = state == ST_READY && turn * S.n_sims + n_done <= pace_limit;
            bool descending = fresh_start || state == ST_PRIOR_READY;
            bool have_priors = state == ST_PRIOR_READY;
            bool skip_record = state == ST_PRIOR_READY; // the cursor node is on the path already
            bool need_prior = false;
            if (fresh_start) {
                node = T.root[g];
                own = whole ? g_own : S.root_own[g];
                opp = whole ? g_opp : S.root_opp[g];
                const uint4 s0 = ((const uint4 *)&T.nodes[base + node])[0], l0 = ((const uint4 *)&T.nodes[base + node])[1];
                fc = (int)l0.x;
                k = (int)((l0.z >> 8) & 0xFFu);
                nv = (int)s0.x;
                vbits = s0.w;
                path_n = 0;
                may_expand = true;
            }
            const bool went = descending;
            busy = busy || went;
            for (int depth = 0; depth < MAX_DEPTH; depth++) {
                if (descending && !skip_record) {
                    if (r == 0u) {
                        if (path_n < S.path_stride) {
                            if (path_at >= 0)
                                ((int32_t *)iago_trunk::trunk_lds)[path_at + path_n] = node;
                            else
                                gpath[path_n] = node;
                        }
                        else
                            T.overflow[g] = 1; // deeper than the path buffer: reported like a full pool
                    }
                    path_n++;
                }
                skip_record = false;
                const bool expand = descending && may_expand && fc < 0 && nv >= S.n_thr;
                if (__builtin_amdgcn_ballot_w64(expand) != 0ull) {
                    const uint64_t lg = group8_legal(to_lane(own, L), to_lane(opp, L), L);
                    if (expand) {
                        const int kn = lg ? __popcll(lg) : 1;
                        if (lg != 0ull && kn > 1 && !have_priors) {
                            // Node.expand needs policy_func(state) (MCTS.py:118-120): ask for it and wait here
                            need_prior = true;
                            descending = false;
                        } else {
                            may_expand = false;
                            uint32_t fc1 = 0; // first child + 1, 0 = no room
                            if (r == 0u) {
                                const int at = T.n_nodes[g];
                                if (at + kn <= T.capacity) {
                                    T.n_nodes[g] = at + kn;
                                    fc1 = (uint32_t)at + 1u;
                                } else {
                                    T.overflow[g] = 1;
                                }
                            }
                            fc1 = group8_add(fc1);
                            if (fc1 != 0u) {
                                const int nf = (int)fc1 - 1;
                                if (lg == 0ull || kn == 1) {
                                    // pass child / single legal move: Node(node, 1), no net (MCTS.py:112-117)
                                    if (r == 0u)
                                        init_node(T, base + nf, node, lg ? (int)__builtin_ctzll(lg) : -1, 1.0f + 0.1f);
                                } else {
                                    uint32_t row = (uint32_t)(lg >> (8u * r)) & 0xFFu;
                                    int at = nf + __popcll(lg & ((1ull << (8u * r)) - 1ull));
                                    while (row) {
                                        const int a = (int)(8u * r) + __builtin_ctz(row);
                                        row &= row - 1u;
                                        const float p = __uint_as_float((uint32_t)ld(&S.rep_p[g * 64 + a]));
                                        init_node(T, base + at, node, a, p + 0.1f); // MCTS.py:19
                                        at++;
                                    }
                                }
                                if (r == 0u) {
                                    T.nodes[base + node].first_child = nf;
                                    T.nodes[base + node].n_children = (uint8_t)kn;
                                }
                                fc = nf;
                                k = kn;
                            }
                        }
                    }
                    __threadfence_block(); // the new children are read by the other lanes of the group below
                }
                have_priors = false;
                descending = descending && fc >= 0; // leaf reached (MCTS.py:107)
                if (__builtin_amdgcn_ballot_w64(descending) == 0ull)
                    break;
                const int kk = descending ? k : 0;
                st_levels += descending ? 1 : 0;
                st_children += kk;
                // Chains of pass nodes.  At the end of a game neither side has a move, and the reference goes on
                // expanding: a pass child under the pass child, one level deeper every n_thr visits (MCTS.py:109-117) --
                // the last turns of a game descend through 65 such levels per playout on average (400 playouts per
                // move; LABNOTES.md).  A node with ONE child leaves nothing to choose (max over one element, MCTS.py:46):
                // when that is so for every game of the wave that still descends and all those children are passes,
                // the level is the child's record and the swap of sides
                if (__builtin_amdgcn_ballot_w64(descending && k != 1) == 0ull) {
                    const int64_t c = descending ? base + fc : base;
                    const uint4 s0 = ((const uint4 *)&T.nodes[c])[0], l0 = ((const uint4 *)&T.nodes[c])[1];
                    const bool pass_child = (int)(int8_t)(l0.z & 0xFFu) < 0;
                    if (__builtin_amdgcn_ballot_w64(descending && !pass_child) == 0ull) {
                        if (descending) {
                            const uint64_t t = own; // GameFunctions.place_stone(state, -1, c) places nothing; c = 3 - c
                            own = opp;
                            opp = t;
                            node = fc;
                            fc = (int)l0.x;
                            nv = (int)s0.x;
                            k = (int)((l0.z >> 8) & 0xFFu);
                            vbits = s0.w;
                        }
                        continue;
                    }
                }
                const double sq = sqrt((double)nv); // np.sqrt(parent.n_visits), MCTS.py:49
                double best_v = -INFINITY;
                int best_i = 0x7fffffff;
                uint32_t pl[4] = {0u, 0u, 0u, 0u}; // of the best child: first_child, n_visits, action | n_children << 8, v
                for (int j0 = (int)r; j0 < kk; j0 += 16) {
                    const int j1 = j0 + 8;
                    const bool two = j1 < kk;
                    const int64_t c0 = base + fc + j0, c1 = two ? base + fc + j1 : c0;
                    const uint4 s0 = ((const uint4 *)&T.nodes[c0])[0], l0 = ((const uint4 *)&T.nodes[c0])[1];
                    const uint4 s1 = ((const uint4 *)&T.nodes[c1])[0], l1 = ((const uint4 *)&T.nodes[c1])[1];
                    const float p0 = __uint_as_float(s0.z), q0 = __uint_as_float(s0.y);
                    const float p1 = __uint_as_float(s1.z), q1 = __uint_as_float(s1.y);
                    const int n0 = (int)s0.x, n1 = (int)s1.x;
                    {
                        const float cp = S.c_puct * p0;                          // float32, MCTS.py:49
                        const double u = (double)cp * sq / (0.01 + (double)n0);
                        const double v = (double)q0 + u;                         // get_value, MCTS.py:75-76
                        if (v > best_v) { // strict: the first maximum wins (python max, MCTS.py:46)
                            best_v = v;
                            best_i = j0;
                            pl[0] = l0.x, pl[1] = (uint32_t)n0, pl[2] = l0.z & 0xFFFFu, pl[3] = s0.w;
                        }
                    }
                    if (two) {
                        const float cp = S.c_puct * p1;
                        const double u = (double)cp * sq / (0.01 + (double)n1);
                        const double v = (double)q1 + u;
                        if (v > best_v) {
                            best_v = v;
                            best_i = j1;
                            pl[0] = l1.x, pl[1] = (uint32_t)n1, pl[2] = l1.z & 0xFFFFu, pl[3] = s1.w;
                        }
                    }
                }
                argmax_step_payload<DPP_XOR1>(best_v, best_i, pl);
                argmax_step_payload<DPP_XOR2>(best_v, best_i, pl);
                argmax_step_payload<DPP_HALF_MIRROR>(best_v, best_i, pl);
                const int child = fc + best_i;
                const int a = descending ? (int)(int8_t)(pl[2] & 0xFFu) : -1;
                // GameFunctions.place_stone(state, action, c); c = 3 - c  (MCTS.py:131-132)
                const uint64_t f = group8_flips(to_lane(own, L), to_lane(opp, L), (uint32_t)a & 63u, L);
                if (descending) {
                    uint64_t no = own, np_ = opp;
                    if (a >= 0) {
                        const uint64_t bit = 1ull << (a & 63);
                        no = own | f | bit;
                        np_ = opp & ~f & ~bit;
                    }
                    own = np_;
                    opp = no;
                    node = child;
                    fc = (int)pl[0];
                    nv = (int)pl[1];
                    k = (int)(pl[2] >> 8);
                    vbits = pl[3];
                }
            }
            if (went) {
                if (need_prior) {
                    epoch++;
                    if (r == 0u)
                        send_request(S, KIND_POLICY, g, epoch, own, opp);
                    state = ST_WAIT_PRIOR;
                    // The node WILL expand when its priors are back, and its children are leaves without a value at their
                    // first visits -- the first of them in this very playout.  While net workgroups have nothing to do they
                    // walk the children's positions beside the policy walk, for the position table (nobody waits for these)
                    if (pace[3] > 0 && need_v) {
                        uint64_t rest = group8_legal(to_lane(own, L), to_lane(opp, L), L);
                        while (rest) {
                            const uint32_t a2 = (uint32_t)__builtin_ctzll(rest);
                            rest &= rest - 1ull;
                            const uint64_t f2 = group8_flips(to_lane(own, L), to_lane(opp, L), a2, L);
                            const uint64_t bit2 = 1ull << a2;
                            const uint64_t c_own = opp & ~f2 & ~bit2, c_opp = own | f2 | bit2; // the child: the other side moves
                            uint32_t known = 0u, by = 0u;
                            // (the ring holds QCAP entries: at most one per game that waits -- <= QCAP / 2 games when this
                            // is on -- and these, handed out as a budget per workgroup and iteration while the ring was
                            // empty, two iterations' worth of which fit beside the games' own: pace[3])
                            if (r == 0u && !vtable_get(S, c_own, c_opp, known, by) && atomicSub(&pace[3], 1) > 0)
                                send_request(S, KIND_VALUE, (int64_t)NOBODY, (uint32_t)g, c_own, c_opp); // (reply tag: the sender)
                        }
                    }
                } else {
                    if (descending && fc >= 0 && r == 0u)
                        T.overflow[g] = 1; // path longer than MAX_DEPTH: reported like a full pool
                    // the leaf of this playout (MCTS.py:123-127): its rollout runs now, its value is the
                    // stored one or is asked for
                    leaf = node;
                    const float c = __uint_as_float(vbits);
                    leaf_fresh = need_v && c != c;
                    bool ask = leaf_fresh;
                    if (S.vtable_mask) {
                        // has any game of any launch asked for this position before?
                        uint32_t hit = 0u, bits = 0u, by = 0u;
                        if (leaf_fresh && r == 0u && vtable_get(S, own, opp, bits, by)) {
                            hit = 1u;
                            atomicAdd((unsigned long long *)&S.totals[8], 1ull);
                            if (by == (uint32_t)g) // (asked for -- or walked ahead -- by this very game)
                                atomicAdd((unsigned long long *)&S.totals[12], 1ull);
                        }
                        hit = group8_add(hit);
                        bits = group8_add(r == 0u ? bits : 0u);
                        if (hit) {
                            vbits = bits; // (leaf_fresh stays: the backup stores the value in the node)
                            ask = false;
                        }
                    }
                    if (ask)
                        epoch++;
                    if (r == 0u) {
                        S.cur_node[g] = node;
                        S.cur_own[g] = own;
                        S.cur_opp[g] = opp;
                        h_own[gl] = own; // (what the rollout pass reads)
                        h_opp[gl] = opp;
                        h_stream[gl] = turn * S.n_sims + n_done;
                        if (ask)
                            send_request(S, KIND_VALUE, g, epoch, own, opp);
                    }
                    state = ask ? ST_ROLL_FRESH : ST_ROLL;
                }
            }
            if (exists && r == 0u)
                S.roll[g] = (need_z && (state == ST_ROLL || state == ST_ROLL_FRESH)) ? 1 : 0;
        }
        // ---- rollouts of the leaves reached in this iteration (Simulate, mcts_self_play.py:9-134): the games that
        // have one are packed into rows of 16 boards (about half of a workgroup's games reach a leaf in an iteration,
        // the others wait for a net: one pass of the 16-lanes-per-board body instead of two, most of the time).  A
        // board's game does not depend on its row: Philox counters are keyed by the game and its playout count
        // the control words this iteration's end looks at (abort, the rings' depths for the pacing and the values-ahead gate,
        // the games in play and their progress): eight loads in flight together HERE, under the rollouts -- read one after
        // the other by thread 0 between the iteration's last two barriers they were up to six dependent round trips to L2
        // (2 - 4 us of a 34 us iteration, with the whole workgroup waiting).  All of it is timing-only state, one
        // iteration old at most when it is used.
        uint32_t c_abort = 0u, c_idle = 0u, c_t0 = 0u, c_h0 = 0u, c_t1 = 0u, c_h1 = 0u, c_play = 0u, c_prog = 0u;
        if (tid == 0) {
            c_abort = __hip_atomic_load(&S.ctl[CTL_ABORT], RLX_AGENT);
            c_idle = __hip_atomic_load(&S.ctl[CTL_IDLE], RLX_AGENT);
            c_t0 = __hip_atomic_load(&S.ctl[ctl_tail(0)], RLX_AGENT);
            c_h0 = __hip_atomic_load(&S.ctl[ctl_head(0)], RLX_AGENT);
            c_t1 = __hip_atomic_load(&S.ctl[ctl_tail(1)], RLX_AGENT);
            c_h1 = __hip_atomic_load(&S.ctl[ctl_head(1)], RLX_AGENT);
            c_play = __hip_atomic_load(&S.ctl[CTL_PLAYING], RLX_AGENT);
            c_prog = __hip_atomic_load(&S.ctl[CTL_PROGRESS], RLX_AGENT);
        }
        const bool rolls = mine && need_z && (state == ST_ROLL || state == ST_ROLL_FRESH);
        bool rolled = true; // this game's rollout ran in this iteration (or it needs none)
        {
            // Passes of 16 boards.  A pass costs the same whether it plays 16 boards or one, and all games of the workgroup
            // wait for it: when a full pass leaves only a few games over (at most S.roll_defer), they are played in the
            // NEXT iteration's first pass, ahead of that iteration's own (17 .. 20 games rolled out in 16 % of the
            // iterations, more than 20 in 32 %: LABNOTES.md, round 5).  Timing only: a game's rollout is keyed by its
            // own playout count, whenever it runs.
            const uint64_t bal = __builtin_amdgcn_ballot_w64(rolls && r == 0u); // bit 8 j: game j of this wave
            const uint64_t balo = __builtin_amdgcn_ballot_w64(rolls && deferred && r == 0u);
            if ((tid & 63) == 0) {
                roll_wave[tid >> 6] = (uint32_t)(((bal & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
                roll_wave_old[tid >> 6] = (uint32_t)(((balo & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
            }
            if (tid < GAMES_PER_WG)
                roll_list[tid] = -1;
            __syncthreads();
            uint32_t gm = 0u, go = 0u; // bit j: game j of the workgroup has a rollout / one put off in the last iteration
#pragma unroll
            for (int w = 0; w < BLOCK / 64; w++) {
                gm |= roll_wave[w] << (8 * w);
                go |= roll_wave_old[w] << (8 * w);
            }
            const int n_roll = __popc(gm), rem = n_roll & 15;
            const int n_now = (n_roll < 16 || rem > S.roll_defer) ? n_roll : n_roll - rem;
            const uint32_t below = (1u << (tid >> 3)) - 1u;
            // the games put off last time first, then this iteration's own, each in game order
            const int rank = deferred ? __popc(go & below) : __popc(go) + __popc(gm & ~go & below);
            const bool now = rolls && rank < n_now;
            if (now && r == 0u)
                roll_list[rank] = (int32_t)g;
            deferred = rolls && !now;
            rolled = !rolls || now;
            __syncthreads();
            if (tid == 0)
                wg_count[2 + (n_roll == 0 ? 0 : n_roll <= 16 ? 1 : n_roll <= 20 ? 2 : 3)]++; // (diagnostic: totals[10], [13..15])
#pragma unroll 1
            for (int at = 0; at < n_now; at += 16) {
                iago_row::rollout_row_body<false, true, true>(R, 0u, roll_list + at, table_ready, &hand);
                table_ready = true;
                __syncthreads();
            }
        }
        if (mine && rolled) {
            if (state == ST_ROLL) {
                backup_game(S, g, r, leaf, leaf_fresh, __uint_as_float(vbits), path_n, h_z[gl], path_at);
                n_done++;
                if (S.trace && r == 0u)
                    atomicAdd((unsigned long long *)&S.totals[9], 1ull);
                if (r == 0u)
                    S.done[g] = turn * S.n_sims + n_done;
                state = n_done >= S.n_sims ? search_end : ST_READY;
            } else if (state == ST_ROLL_FRESH) {
                state = ST_WAIT_VALUE;
            }
        }
        if (S.trace && blockIdx.x == 0 && tid == 0 && (int64_t)wg_count[0] < S.trace_rows - T.n_games) {
            const int64_t iters = (int64_t)wg_count[0];
            S.trace[4 * iters + 0] = wall_clock64() - t0;
            S.trace[4 * iters + 1] = __hip_atomic_load(&S.ctl[ctl_tail(0)], RLX_AGENT) + __hip_atomic_load(&S.ctl[ctl_tail(1)], RLX_AGENT);
            S.trace[4 * iters + 2] = __hip_atomic_load(&S.ctl[ctl_head(0)], RLX_AGENT) + __hip_atomic_load(&S.ctl[ctl_head(1)], RLX_AGENT);
            S.trace[4 * iters + 3] = (int64_t)__hip_atomic_load(&S.ctl[CTL_FINISHED], RLX_AGENT) |
                                     (__hip_atomic_load(&S.totals[9], RLX_AGENT) << 8);
        }
        if (tid == 0)
            wg_count[0]++;
        if (mine && r == 0u) {
            const int prog = state == ST_DONE ? 0 : turn * S.n_sims + n_done;
            if (prog != contrib)
                atomicAdd(&pace[0], prog - contrib);
            if (in_play && state == ST_DONE)
                atomicAdd(&pace[1], -1);
        }
        contrib = state == ST_DONE ? 0 : turn * S.n_sims + n_done;
        in_play = in_play && state != ST_DONE;
        // (the workgroup's own stores to done / the tree are read by its next iteration: same CU)
        const bool over = wall_clock64() - t0 > S.clock_limit;
        if (over && tid == 0)
            __hip_atomic_store(&S.ctl[CTL_ABORT], 1u, RLX_AGENT);
        const int stop = __syncthreads_or(over || (tid == 0 && c_abort != 0u));
        if (tid == 0) {
            if (pace[0])
                __hip_atomic_fetch_add(&S.ctl[CTL_PROGRESS], (uint32_t)pace[0], RLX_AGENT);
            if (pace[1])
                __hip_atomic_fetch_add(&S.ctl[CTL_PLAYING], (uint32_t)pace[1], RLX_AGENT);
            pace[0] = 0;
            pace[1] = 0;
            int limit = 0x7fffffff;
            pace[3] = 0;
            const int32_t wait0 = (int32_t)(c_t0 - c_h0), wait1 = (int32_t)(c_t1 - c_h1);
            if (S.ahead_idle >= 0 && S.vtable_mask && T.n_games <= (int64_t)(QCAP / 2u) && c_idle >= (uint32_t)S.ahead_idle &&
                wait0 <= 0 && wait1 <= 0)
                // this iteration's share of the ring for requests nobody waits for: the games' own requests (at most one
                // each) and TWO iterations' worth of these (the workgroups look at the rings at different moments: a second
                // burst can be on its way before the first shows in anybody's snapshot) fit the ring together
                pace[3] = (int32_t)((QCAP - (uint32_t)T.n_games) / 2u) / S.n_game_wgs;
            if (S.pace_margin >= 0) {
                if (wait0 + wait1 > S.pace_backlog && c_play != 0u && c_play <= (uint32_t)T.n_games)
                    limit = (int)(c_prog / c_play) + S.pace_margin;
            }
            pace[2] = limit;
        }
        const int all_done = __syncthreads_and(!mine || state == ST_DONE);
        pace_limit = pace[2];
        if (all_done || stop)
            break;
        if (!__syncthreads_or(busy)) {
            if (tid == 0)
                wg_count[1]++;
            __builtin_amdgcn_s_sleep(32); // every game waits for a reply: poll again in ~1 us
        }
    }
    if (exists && r == 0u && S.stats) {
        S.stats[2 * g] += st_levels;
        S.stats[2 * g + 1] += st_children;
    }
    // (a launch that gave up: what every unfinished game was waiting for, for the post-mortem -- the trees are void anyway:
    // cur_node = the reply tag it waits for, leaf_value = its state; tools/debug_split_abort.py)
    if (exists && r == 0u && state != ST_DONE && __hip_atomic_load(&S.ctl[CTL_ABORT], RLX_AGENT) != 0u) {
        S.cur_node[g] = (int32_t)epoch;
        S.leaf_value[g] = (float)state;
    }
    if (tid == 0) {
        atomicAdd((unsigned long long *)&S.totals[2], (unsigned long long)wg_count[0]);
        atomicAdd((unsigned long long *)&S.totals[6], (unsigned long long)wg_count[1]);
        atomicAdd((unsigned long long *)&S.totals[10], (unsigned long long)wg_count[2]);
        atomicAdd((unsigned long long *)&S.totals[13], (unsigned long long)wg_count[3]);
        atomicAdd((unsigned long long *)&S.totals[14], (unsigned long long)wg_count[4]);
        atomicAdd((unsigned long long *)&S.totals[15], (unsigned long long)wg_count[5]);
        atomicAdd((unsigned long long *)&S.totals[7], (unsigned long long)(wall_clock64() - t0));
        __hip_atomic_fetch_add(&S.ctl[CTL_FINISHED], 1u, RLX_AGENT);
    }
}

// A NET workgroup: ticket -> entry -> walk -> reply, until every game workgroup has finished.
// The two kinds of net work have a ring each and a HOME on the chip: the workgroups of `policy_xcds` of the 8
// XCDs (workgroup i of a launch runs on XCD i mod 8) serve the POLICY ring, the others the VALUE ring, so that an
// XCD's 4 MB L2 holds ONE net's weights (3.9 / 5.8 MB) instead of thrashing on both; a workgroup whose home ring is
// empty serves the other one (no CU idles while work waits).
// Two VALUE entries that are in the ring together are walked as a PAIR (trunk_item<true, 2>: the two boards
// share the weight stream, which bounds the one-board walk: 46 instead of 70 us of CU time per board; the same
// products in the same order per board: bit-identical values).
__device__ __forceinline__ void net_workgroup(const SearchParams &S, const iago_trunk::TrunkRParams &VP,
                                              const iago_policy::PolicyParams &PP, const long long t0)
{
    __shared__ __align__(16) uint32_t job[32]; // up to two entries of 6 words (kind | game, reply tag, own lo / hi, opp lo / hi); [28..]: status, count
    const int tid = threadIdx.x;
    const int64_t row0 = 4 * (int64_t)blockIdx.x; // this workgroup's rows of wg_own / wg_opp / out / probs (two in use)
    const uint32_t home = ((int)(blockIdx.x & 7u) >= 8 - S.policy_xcds) ? KIND_POLICY : KIND_VALUE;
    // Tickets are handed out by fetch-and-add whenever the ring shows an entry waiting: several workgroups that saw the
    // same entry all take one, and the later ones wait in fetch() for the ring's next entries.  (Measured and dropped,
    // round 5: head moved by compare-and-swap bounded by the tail, so that nobody is committed to an entry that does not
    // exist -- a retry loop took the 1024-game batch from 0.39 to 3.6 s, one attempt per look to 1.76 s: the losers keep
    // polling and retrying on ONE word from all XCDs, same-address atomics serialise, and every claim queues behind them.)
    // a VALUE ticket taken for a pair whose entry was not there yet: the next round's entry
    uint32_t carry = 0u;
    int n_carry = 0;

    // wave 0: wait for entry t of ring q (at most max_spins polls; 0 = until it comes or the search is over)
    // -> job[6 * which ..]; returns 0 = entry read, 1 = not there yet, 2 = the search is over / given up
    auto fetch = [&](uint32_t q, uint32_t t, uint32_t max_spins, int which) -> int {
        const u64 *e = S.q_slots + ((u64)q * QCAP + t % QCAP) * 8u;
        u64 x = 0;
        int status = 0;
        for (uint32_t spins = 0;; spins++) {
            x = (tid < 6) ? ld(e + tid) : 0ull;
            const bool ok = tid >= 6 || (uint32_t)(x >> 32) == t + 1u;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull)
                break;
            if (max_spins && spins + 1u >= max_spins) {
                status = 1;
                break;
            }
            if ((spins & 15u) == 15u) {
                bool out = false;
                if (tid == 0)
                    out = __hip_atomic_load(&S.ctl[CTL_FINISHED], RLX_AGENT) >= (uint32_t)S.n_game_wgs ||
                          __hip_atomic_load(&S.ctl[CTL_ABORT], RLX_AGENT) != 0u || wall_clock64() - t0 > S.clock_limit;
                if (__builtin_amdgcn_ballot_w64(out) != 0ull) {
                    status = 2;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (status == 0 && tid < 6)
            job[6 * which + tid] = (uint32_t)x;
        return status;
    };
    auto take = [&](uint32_t q) -> uint32_t {
        uint32_t t = 0;
        if (tid == 0)
            t = __hip_atomic_fetch_add(&S.ctl[ctl_head(q)], 1u, RLX_AGENT);
        return __builtin_amdgcn_readfirstlane(t);
    };
    auto backlog = [&](uint32_t q) -> int {
        int d = 0;
        if (tid == 0)
            d = (int32_t)(__hip_atomic_load(&S.ctl[ctl_tail(q)], RLX_AGENT) - __hip_atomic_load(&S.ctl[ctl_head(q)], RLX_AGENT));
        return __builtin_amdgcn_readfirstlane(d);
    };

    // block1's weights and biases of both nets, staged ONCE per workgroup at the top of the dynamic LDS (above the
    // walks' images: conv_trunk_body.hpp, Piece::w1s)
    float *const w1_val = (float *)(iago_trunk::trunk_lds + SEARCH_IMG_TOP), *const w1_pol = w1_val + (64 * 18 + 64);
    __shared__ float res_v[2], res_p[64];
    for (int e = tid; e < 64 * 18 / 4; e += 256) {
        ((float4 *)w1_val)[e] = ((const float4 *)VP.w1)[e];
        ((float4 *)w1_pol)[e] = ((const float4 *)PP.w1)[e];
    }
    if (tid < 16) {
        ((float4 *)(w1_val + 64 * 18))[tid] = ((const float4 *)VP.b1)[tid];
        ((float4 *)(w1_pol + 64 * 18))[tid] = ((const float4 *)PP.b1)[tid];
    }
    // ... and the heads' weights (the value net's 48.5 KB: block9 as MFMA operand, fc10 transposed so that a thread's row is
    // read at 16 B between threads, fc11, block9's bias; the policy net's conv9 and bias10): read once per walk from
    // global memory they are evicted from L2 by the trunks' weight streams in between
    char *const head_v = (char *)(w1_pol + (64 * 18 + 64));
    float *const head_p = (float *)(head_v + iago_trunk::HEAD_W_LDS);
    for (int e = tid; e < 512; e += 256) {
        ((uint4 *)head_v)[e] = VP.w9_hi[e];
        ((uint4 *)(head_v + iago_trunk::HEAD_W_W9LO))[e] = VP.w9_lo[e];
    }
    for (int e = tid; e < 128 * 16; e += 256) // fc10 [128][64]: row j = e / 16, float4 column c = e % 16
        ((float4 *)(head_v + iago_trunk::HEAD_W_W10))[(e & 15) * 128 + (e >> 4)] = ((const float4 *)VP.w10)[e];
    if (tid < 128)
        ((float *)(head_v + iago_trunk::HEAD_W_W11))[tid] = VP.w11[tid];
    if (tid == 0)
        *(float *)(head_v + iago_trunk::HEAD_W_B9) = VP.b9[0];
    if (tid < 128)
        head_p[tid] = PP.w9[tid];
    if (tid < 64)
        head_p[128 + tid] = PP.b10[tid];
    __syncthreads();
    u64 *put_e = nullptr; // (lanes 0 / 1: the position-table entry whose sequence word is still to be published)
    u64 put_word = 0;
    long long t_wait = 0, t_walk = 0, n_pairs = 0;
    for (;;) {
        const long long c0 = wall_clock64();
        if (put_e) {
            vtable_put_end(put_e, put_word);
            put_e = nullptr;
        }
        if (tid < 64) {
            int status = 2, count = 0;
            bool polling = false;
            // an entry waiting in the home ring, else in the other one; nothing anywhere: poll the counters (no
            // ticket is taken for an entry that is not there, so nobody is committed to a ring that stays empty)
            for (;;) {
                uint32_t q = home;
                bool have = n_carry > 0;
                uint32_t t1 = carry;
                if (have) {
                    q = KIND_VALUE;
                    n_carry = 0;
                } else if (backlog(home) > 0) {
                    t1 = take(home);
                    have = true;
                } else if (backlog(home ^ 1u) > 0) {
                    q = home ^ 1u;
                    t1 = take(q);
                    have = true;
                }
                if (have) {
                    // (a ticket below the tail: its producer is writing the entry right now; one beyond it -- two
                    // workgroups saw the same entry -- waits for the next entry of that ring)
                    status = fetch(q, t1, 0u, 0);
                    count = status == 0 ? 1 : 0;
                    if (status == 0 && q == KIND_VALUE && backlog(KIND_VALUE) >= S.pair_backlog) {
                        // (four boards per walk were measured too: the third variant's registers spill in this
                        // kernel and the walks lose more than the shared stream gains: LABNOTES.md, round 4)
                        const uint32_t t2 = take(KIND_VALUE);
                        if (fetch(KIND_VALUE, t2, 8u, 1) == 0) {
                            count = 2;
                        } else {
                            carry = t2; // not there yet: the next round's entry
                            n_carry = 1;
                        }
                    }
                    break;
                }
                bool out = false;
                if (tid == 0)
                    out = __hip_atomic_load(&S.ctl[CTL_FINISHED], RLX_AGENT) >= (uint32_t)S.n_game_wgs ||
                          __hip_atomic_load(&S.ctl[CTL_ABORT], RLX_AGENT) != 0u || wall_clock64() - t0 > S.clock_limit;
                if (__builtin_amdgcn_ballot_w64(out) != 0ull)
                    break; // status 2: every game workgroup is done (no request can come any more), or given up
                if (!polling) {
                    polling = true; // (counted while it polls: values ahead of their visit go to idle hands only)
                    if (tid == 0)
                        __hip_atomic_fetch_add(&S.ctl[CTL_IDLE], 1u, RLX_AGENT);
                }
                __builtin_amdgcn_s_sleep(16);
            }
            if (polling && tid == 0)
                __hip_atomic_fetch_add(&S.ctl[CTL_IDLE], 0xFFFFFFFFu, RLX_AGENT);
            if (tid == 0) {
                job[28] = (uint32_t)status;
                job[29] = (uint32_t)count;
            }
        }
        __syncthreads();
        const long long c1 = wall_clock64();
        t_wait += c1 - c0;
        if (job[28] != 0u) {
            if (tid == 0) {
                atomicAdd((unsigned long long *)&S.totals[3], (unsigned long long)n_pairs);
                atomicAdd((unsigned long long *)&S.totals[4], (unsigned long long)t_wait);
                atomicAdd((unsigned long long *)&S.totals[5], (unsigned long long)t_walk);
            }
            return;
        }
        const uint32_t kind = job[0] >> 31;
        const int count = (int)job[29];
        n_pairs += count - 1;
        // (positions, values and priors go between the request's words and the walks through LDS: a store to global
        // memory read back by this very workgroup was a round trip to L2, ~2 us under load, twice per walk)
        if (kind == KIND_VALUE) {
            iago_trunk::Piece W = iago_trunk::whole_walk(VP);
            W.pos = job;
            W.res = res_v;
            W.w1s = w1_val;
            W.head_w = head_v;
            if (count == 2)
                iago_trunk::trunk_item<true, 2, true>(VP, W, row0, row0 + count);
            else
                iago_trunk::trunk_item<true, 1, true>(VP, W, row0, row0 + count);
            __syncthreads();
            if (tid < count) {
                const uint32_t bits = __float_as_uint(res_v[tid]);
                if ((job[6 * tid] & 0x7FFFFFFFu) != NOBODY)
                    st(&S.rep_v[(int64_t)(job[6 * tid] & 0x7FFFFFFFu)], ((u64)job[6 * tid + 1] << 32) | bits);
                if (S.vtable_mask) {
                    const uint32_t asked = job[6 * tid] & 0x7FFFFFFFu;
                    put_e = vtable_put_begin(S, ((uint64_t)job[6 * tid + 3] << 32) | job[6 * tid + 2],
                                             ((uint64_t)job[6 * tid + 5] << 32) | job[6 * tid + 4], bits,
                                             asked != NOBODY ? asked : job[6 * tid + 1], put_word);
                }
            }
        } else {
            iago_policy::policy_item<true>(PP, row0, job, res_p, w1_pol, head_p);
            __syncthreads();
            if (tid < 64)
                st(&S.rep_p[(int64_t)(job[0] & 0x7FFFFFFFu) * 64 + tid], ((u64)job[1] << 32) | __float_as_uint(res_p[tid]));
        }
        __syncthreads(); // the next item re-stages the LDS image and job[]
        t_walk += wall_clock64() - c1;
    }
}

// ONE grid: the game workgroups first (they are dispatched first, so all of them are resident whatever else
// holds CUs; a net workgroup never waits for another net workgroup, so one that finds no CU free simply starts
// late), then the net workgroups.  A game workgroup whose games are done serves the queue like the others.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void search_kernel(
    SearchParams S, iago_row::HwParams R, iago_trunk::TrunkRParams VP, iago_policy::PolicyParams PP)
{
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) // (what the launch was given: the host sized the grid from the device)
        __hip_atomic_store(&S.ctl[CTL_NET_WGS], (uint32_t)gridDim.x - (uint32_t)S.n_game_wgs, RLX_AGENT);
    if ((int)blockIdx.x < S.n_game_wgs)
        game_workgroup(S, R, t0);
    net_workgroup(S, VP, PP, t0);
}

// The same search as TWO launches that run together, one per role (iago_mcts_search_split: each on a stream of its own
// whose CU mask gives it CUs no other launch of the process gets -- co-residency by construction, not by a guess of how
// many CUs are free).  Each role is compiled for its own register budget: the game workgroups (VALU and latency: lone
// waves) as a kernel of at most 256 registers, TWO workgroups per CU, so that a wave's waits are another wave's issue
// slots; the net workgroups without the game code in their allocation.  Same device functions, same protocol, same
// trees.  What it buys is CUs: in the single launch every game workgroup holds a CU alone (512 registers per lane), so
// beyond 32 game workgroups each one is a net workgroup less -- 2048 games: 64 + 192 workgroups, 17.1 M leaf-evals/s;
// split: 64 game workgroups on 32 CUs + 224 net workgroups, 18.3 M; 4096 games 10.8 -> 15.1 M; at 1024 games (32 + 224
// either way) the two forms measure the same and the single launch stays (LABNOTES.md, round 6).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void search_game_kernel(SearchParams S,
                                                                                                   iago_row::HwParams R)
{
    game_workgroup(S, R, wall_clock64());
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void search_net_kernel(
    SearchParams S, iago_trunk::TrunkRParams VP, iago_policy::PolicyParams PP)
{
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(&S.ctl[CTL_NET_WGS], (uint32_t)gridDim.x, RLX_AGENT);
    net_workgroup(S, VP, PP, t0);
}

} // namespace

namespace {
constexpr int search_lds() { return SEARCH_IMG_TOP + iago_trunk::W1_LDS + iago_policy::W1_LDS + iago_trunk::HEAD_W_LDS + 192 * 4; }

// (iago_mcts_search_streams_create: one launch per masked stream of a kernel that needs more scratch memory per lane than
// the game launch does, so that the runtime sizes the streams' scratch ONCE, before any launch that another launch waits
// for: a stream's first dispatch with a private segment is held until the host has allocated it)
__global__ void search_scratch_warm_kernel(uint32_t *out, int n)
{
    volatile uint32_t a[64];
    for (int i = 0; i < 64; i++)
        a[i] = (uint32_t)(i * n) + threadIdx.x;
    uint32_t sum = 0;
    for (int i = 0; i < 64; i++)
        sum += a[(i * 7 + n) & 63];
    if (n == 0x7fffffff && out)
        out[0] = sum;
}
} // namespace

extern "C" int iago_mcts_search_capacity(int32_t *cus, int32_t *workgroups_per_cu)
{
    if (!cus || !workgroups_per_cu)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_capacity: null pointer");
    static std::atomic<uint64_t> configured{0};
    if (iago_reserve_lds((const void *)search_kernel, search_lds(), configured,
                         "iago_mcts_search_capacity: cannot reserve the nets' LDS image"))
        return IAGO_ERR_HIP;
    // (asked of the runtime once per device: every launch comes through here)
    static std::atomic<int32_t> known[64]; // cus << 8 | workgroups per CU, 0 = not asked yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_capacity: hipGetDevice failed");
    int32_t k = known[dev & 63].load(std::memory_order_acquire);
    if (k == 0) {
        int n_cu = 0, per = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void *)search_kernel, 256, (size_t)search_lds()) != hipSuccess)
            return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_capacity: the device does not answer");
        if (n_cu < 1 || per < 1 || per > 255)
            return iago_fail(IAGO_ERR_CAPACITY, "iago_mcts_search_capacity: the search kernel does not fit a CU of this device");
        k = (int32_t)(n_cu << 8 | per);
        known[dev & 63].store(k, std::memory_order_release);
    }
    *cus = k >> 8;
    *workgroups_per_cu = k & 255;
    return IAGO_OK;
}

// The CU-masked streams of the role split (iago_mcts_search_streams_create): the game launch's, the net launch's, and
// the events that order both after the caller's stream and the caller's stream after both.
struct iago_search_streams {
    int device;
    int32_t cus, game_cus;
    hipStream_t game, net;
    hipEvent_t ready, game_done, net_done;
};

namespace {
int search_launch(const iago_mcts_search_args *a, void *stream, iago_search_streams *sp)
{
    if (!a || !a->tree || !a->value || !a->policy || !a->rollout)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: null args");
    const iago_mcts_tree *tree = a->tree;
    if (tree->n_games < 1 || tree->capacity < 1 || !tree->nodes || ((uintptr_t)tree->nodes & 31u) || !tree->n_nodes ||
        !tree->root || !tree->overflow || !tree->has_v)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: bad tree (the value cache `v` is required)");
    if (tree->n_games > 0x7FFFFFF0ll)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: too many games");
    if (!a->active || !a->cur_node || !a->cur_own || !a->cur_opp || !a->path ||
        a->path_stride < 8 || !a->done || !a->roll || !a->leaf_value || !a->q_slots || !a->ctl || !a->rep_v || !a->rep_p ||
        !a->totals || !a->wg_own || !a->wg_opp || ((uintptr_t)a->q_slots & 63u) || ((uintptr_t)a->rep_p & 7u) ||
        ((uintptr_t)a->rep_v & 7u) || ((uintptr_t)a->ctl & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: null or misaligned state array");
    if (a->n_thr < 1 || a->n_sims < 0 || !(a->lmbda >= 0.0f && a->lmbda <= 1.0f) || a->net_workgroups < 1)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: n_thr >= 1, n_sims >= 0, 0 <= lmbda <= 1, "
                                           "net_workgroups >= 1 expected");
    if (a->max_turns < 0 || (a->max_turns > 0 && (!a->game_own || !a->game_opp || !a->n_turns)) ||
        (a->max_turns > 0 && a->rec_move && (!a->rec_own || !a->rec_opp || !a->rec_valid || !a->rec_pi)))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: whole games (max_turns > 0) need game_own, game_opp, "
                                           "n_turns, and all of rec_own / rec_opp / rec_valid / rec_move / rec_pi or none");
    if (a->max_turns == 0 && (!a->root_own || !a->root_opp))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: root_own / root_opp expected");
    if (a->z_log_rows > 0 && (!a->z_log || !a->z_log_n))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: z_log needs z_log_n");
    const iago_rollout_args *ro = a->rollout;
    if (ro->n != tree->n_games || !ro->z || !ro->table || ((uintptr_t)ro->table & 15u) || ro->log_form || ro->trace ||
        ro->uniforms || ro->throughput_hint != 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: product-form rollout of the n games without "
                                           "trace / uniforms expected");
    const int gpw = a->games_per_workgroup > 0 ? a->games_per_workgroup : GAMES_PER_WG;
    if (gpw != 8 && gpw != 16 && gpw != 24 && gpw != 32)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: games_per_workgroup is 0 (= 32), 8, 16 or 32");
    const int64_t n_game_wgs = (tree->n_games + gpw - 1) / gpw;
    // The grid follows the device: every game workgroup must be resident together with at least one net workgroup (a
    // game waits for replies only net workgroups give), and a net workgroup beyond what fits would only start when
    // another one ends -- at the end of the launch.  Resident workgroups = CUs the launch may count on (max_cus, else
    // the device's) x workgroups of this kernel per CU (its registers and LDS allow one).
    int32_t cus = 0, per_cu = 0;
    if (const int rc = iago_mcts_search_capacity(&cus, &per_cu))
        return rc;
    if (a->max_cus < 0)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: max_cus < 0");
    int64_t resident = (int64_t)(a->max_cus > 0 && a->max_cus < cus ? a->max_cus : cus) * per_cu;
    // the games' recorded paths in the launch's dynamic LDS when they fit there (else in the caller's array)
    int path_lds_cap = SEARCH_IMG_TOP, game_lds = 0;
    if (sp) {
        // Role split: the game launch has sp->game_cus CUs of its own and the net launch all the others.  A game
        // workgroup keeps its paths in LDS when TWO workgroups with them fit a CU (else in the caller's array)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev != sp->device || sp->cus != cus)
            return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_split: the streams belong to another device");
        if (a->max_cus != 0)
            return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_split: max_cus must be 0 (the split owns the device's CUs)");
        const size_t want = (size_t)gpw * (size_t)a->path_stride * 4u;
        // (asked of the runtime once per device and LDS size: every launch comes through here)
        static std::atomic<int32_t> static_lds{-1};
        int32_t fixed = static_lds.load(std::memory_order_acquire);
        if (fixed < 0) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, (const void *)search_game_kernel) != hipSuccess)
                return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_split: hipFuncGetAttributes failed");
            fixed = (int32_t)fa.sharedSizeBytes;
            static_lds.store(fixed, std::memory_order_release);
        }
        game_lds = (want + (size_t)fixed + 256u) * 2u <= (size_t)160 * 1024u ? (int)want : 0;
        path_lds_cap = game_lds;
        static std::atomic<uint64_t> configured_game{0};
        if (game_lds && iago_reserve_lds((const void *)search_game_kernel, 96 * 1024, configured_game,
                                         "iago_mcts_search_split: cannot reserve the game workgroups' LDS"))
            return IAGO_ERR_HIP;
        static std::atomic<uint64_t> occ_known[64]; // per device: LDS bytes << 8 | workgroups per CU (+ 1 << 63: valid)
        int per_game = 0;
        const uint64_t seen = occ_known[dev & 63].load(std::memory_order_acquire);
        if ((seen >> 63) && ((seen >> 8) & 0xFFFFFFull) == (uint64_t)game_lds) {
            per_game = (int)(seen & 0xFF);
        } else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_game, (const void *)search_game_kernel, 256, (size_t)game_lds) != hipSuccess)
                return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_split: the device does not answer");
            occ_known[dev & 63].store((1ull << 63) | ((uint64_t)game_lds << 8) | (uint64_t)(per_game & 0xFF), std::memory_order_release);
        }
        if (n_game_wgs > (int64_t)sp->game_cus * per_game)
            return iago_fail(IAGO_ERR_CAPACITY, "iago_mcts_search_split: the game workgroups do not fit the game launch's CUs "
                                                "(more game CUs, fewer games per launch, or the single launch)");
        // The net launch takes at most 7/8 of the device's CUs (224 of 256), whatever the game launch leaves.  Measured, not
        // understood: with 232 .. 240 net workgroups beside a game launch on 16 / 24 CUs, about one batch in 60 froze --
        // the LAST workgroups of the net launch (block index >= 224: exactly those, in every post-mortem) each took a
        // ticket, stopped executing in the same 100 us, and went on the moment the game launch had ended (at its clock
        // limit); with at most 224, 960 batches in a row on the same settings: none (LABNOTES.md, round 6)
        const int64_t net_cus = cus - (sp->game_cus > cus / 8 ? sp->game_cus : cus / 8);
        resident = n_game_wgs + net_cus * per_cu;
    }
    if (n_game_wgs + 1 > resident)
        return iago_fail(IAGO_ERR_CAPACITY, "iago_mcts_search_persistent: the game workgroups and one net workgroup do not "
                                            "fit the device together (fewer games per launch, or the per-playout launches)");
    const int64_t net_wgs = a->net_workgroups < resident - n_game_wgs ? a->net_workgroups : resident - n_game_wgs;
    const int64_t grid = n_game_wgs + net_wgs;
    if (a->value->n < 4 * grid || a->policy->n < 4 * grid || a->value->planes || a->value->index || a->value->n_dev ||
        a->policy->index || a->policy->n_dev || !a->value->own || a->value->own != a->wg_own ||
        a->value->opp != a->wg_opp || a->policy->own != a->wg_own || a->policy->opp != a->wg_opp)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: the nets read their rows from wg_own / wg_opp "
                                           "(four rows per workgroup of the grid: n >= 4 x (game + net workgroups)), no gather list, no "
                                           "device count");
    iago_trunk::TrunkRParams VP;
    if (const int rc = iago_trunk::value_params_of(a->value, VP))
        return rc;
    VP.count_lo = 0;
    VP.count_hi = 0x7fffffff;
    iago_policy::PolicyParams PP;
    if (const int rc = iago_policy::policy_params_of(a->policy, PP))
        return rc;
    SearchParams S;
    S.T = *tree;
    S.root_own = a->root_own;
    S.root_opp = a->root_opp;
    S.active = a->active;
    S.c_puct = a->c_puct;
    S.lmbda = a->lmbda;
    S.n_thr = a->n_thr;
    S.n_sims = a->n_sims;
    S.n_game_wgs = (int32_t)n_game_wgs;
    S.games_per_wg = gpw;
    S.cur_node = a->cur_node;
    S.cur_own = a->cur_own;
    S.cur_opp = a->cur_opp;
    S.path = a->path;
    S.path_stride = a->path_stride;
    S.done = a->done;
    S.roll = a->roll;
    S.z = ro->z;
    S.leaf_value = a->leaf_value;
    S.z_log = a->z_log_rows > 0 ? a->z_log : nullptr;
    S.z_log_n = a->z_log_n;
    S.z_log_rows = a->z_log_rows;
    S.q_slots = (u64 *)a->q_slots;
    S.ctl = a->ctl;
    S.rep_v = (u64 *)a->rep_v;
    S.rep_p = (u64 *)a->rep_p;
    S.totals = a->totals;
    S.stats = a->stats;
    S.wg_own = a->wg_own;
    S.wg_opp = a->wg_opp;
    S.clock_limit = (long long)(a->time_limit_ms > 0 ? a->time_limit_ms : 2000) * 100000ll; // wall_clock64: 100 MHz
    // (tuning knob: a pair shares the weight stream -- 46 instead of 70 us of CU time per board -- but takes 92 us:
    // worth it only while entries queue up)
    static const int pair_backlog = [] {
        const char *e = getenv("IAGO_PERSISTENT_PAIR");
        const int v = e ? atoi(e) : 1;
        return v < 1 ? 0x7fffffff : v;
    }();
    S.pair_backlog = pair_backlog;
    // (tuning knob: 27 % of the nets' CU time is the policy net's)
    static const int policy_xcds = [] {
        const char *e = getenv("IAGO_PERSISTENT_POLICY_XCDS");
        const int v = e ? atoi(e) : 2;
        return v < 0 ? 0 : (v > 7 ? 7 : v);
    }();
    S.policy_xcds = policy_xcds;
    // (tuning knob of the pacing: requests that must be waiting for a leader to hold)
    const char *pace_env = getenv("IAGO_PERSISTENT_PACE_BACKLOG"); // (read per launch: the tests vary it)
    const int pace_backlog = pace_env ? atoi(pace_env) : 128;
    const int pace_margin = a->pace_margin == 0 ? 16 : (a->pace_margin < 0 ? -1 : a->pace_margin);
    S.pace_margin = pace_margin;
    S.pace_backlog = pace_backlog;
    // (tuning knob: net workgroups that must poll for values to be walked ahead of their visit; -1 = never)
    const char *ahead_env = getenv("IAGO_PERSISTENT_AHEAD"); // (read per launch: the tests vary it)
    S.ahead_idle = ahead_env ? atoi(ahead_env) : 4;
    // (tuning knob: games that a full pass of 16 rollouts may leave over for the next iteration)
    const char *defer_env = getenv("IAGO_PERSISTENT_ROLL_DEFER"); // (read per launch: the tests vary it)
    S.roll_defer = defer_env ? atoi(defer_env) : 10;
    if (S.roll_defer < 0 || S.roll_defer > 15)
        S.roll_defer = S.roll_defer < 0 ? 0 : 15;
    S.path_lds_cap = path_lds_cap;
    S.max_turns = a->max_turns;
    S.game_own = a->game_own;
    S.game_opp = a->game_opp;
    S.n_turns = a->n_turns;
    S.rec_own = a->rec_own;
    S.rec_opp = a->rec_opp;
    S.rec_valid = a->rec_valid;
    S.rec_move = a->rec_move;
    S.rec_pi = a->rec_pi;
    S.vtable = nullptr;
    S.vtable_mask = 0u;
    if (a->vtable_slots > 0) {
        if (!a->vtable || ((uintptr_t)a->vtable & 31u) || (a->vtable_slots & (a->vtable_slots - 1)) != 0)
            return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_persistent: vtable must be 32-byte aligned, vtable_slots a "
                                               "power of two");
        S.vtable = (u64 *)a->vtable;
        S.vtable_mask = (uint32_t)(a->vtable_slots - 1);
    }
    S.trace = a->trace_rows > 0 ? a->trace : nullptr;
    S.trace_rows = a->trace_rows;
    iago_row::HwParams R = iago_row::hw_params_of(ro);
    R.own = a->cur_own;
    R.opp = a->cur_opp;
    R.mask = a->roll;
    R.stream_ids = a->done;
    constexpr int lds = search_lds(); // (reserved for the kernel by iago_mcts_search_capacity above)
    // every polled word starts from zero: the control block, the request ring and the reply mailboxes
    if (hipMemsetAsync(a->ctl, 0, 64, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(a->q_slots, 0, (size_t)2 * QCAP * 64, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(a->rep_v, 0, (size_t)tree->n_games * 8, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync(a->rep_p, 0, (size_t)tree->n_games * 512, (hipStream_t)stream) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_persistent: hipMemsetAsync failed");
    if (!sp) {
        hipLaunchKernelGGL(search_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, S, R, VP, PP);
        return iago_check_launch("iago_mcts_search_persistent");
    }
    // both launches after everything queued on the caller's stream so far (the zeroing above included), the caller's
    // stream after both.  The game launch first: the net workgroups leave when the game workgroups have finished
    static std::atomic<uint64_t> configured_net{0};
    if (iago_reserve_lds((const void *)search_net_kernel, lds, configured_net,
                         "iago_mcts_search_split: cannot reserve the nets' LDS image"))
        return IAGO_ERR_HIP;
    if (hipEventRecord(sp->ready, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(sp->game, sp->ready, 0) != hipSuccess ||
        hipStreamWaitEvent(sp->net, sp->ready, 0) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_split: cannot order the launches after the stream");
    hipLaunchKernelGGL(search_game_kernel, dim3((unsigned)n_game_wgs), dim3(256), game_lds, sp->game, S, R);
    int rc = iago_check_launch("iago_mcts_search_split (game launch)");
    if (rc == IAGO_OK) {
        hipLaunchKernelGGL(search_net_kernel, dim3((unsigned)net_wgs), dim3(256), lds, sp->net, S, VP, PP);
        rc = iago_check_launch("iago_mcts_search_split (net launch)");
    }
    // (also after a failed launch: whatever did start is waited for by the caller's stream)
    if (hipEventRecord(sp->game_done, sp->game) != hipSuccess || hipEventRecord(sp->net_done, sp->net) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)stream, sp->game_done, 0) != hipSuccess ||
        hipStreamWaitEvent((hipStream_t)stream, sp->net_done, 0) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_split: cannot order the stream after the launches");
    return rc;
}
} // namespace

extern "C" int iago_mcts_search_persistent(const iago_mcts_search_args *a, void *stream)
{
    return search_launch(a, stream, nullptr);
}

extern "C" int iago_mcts_search_split(const iago_mcts_search_args *a, iago_search_streams *streams, void *stream)
{
    if (!streams)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_split: null streams");
    return search_launch(a, stream, streams);
}

extern "C" int iago_mcts_search_streams_create(int32_t game_cus, iago_search_streams **out)
{
    if (!out)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_streams_create: null pointer");
    *out = nullptr;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_streams_create: the device does not answer");
    // (a multiple of 8 from every side: mask bit i is a CU of XCD i mod 8 -- tools/exp_cu_mask.hip --, so both launches
    // get the same number of CUs on every XCD and workgroup i of a launch still runs on XCD i mod 8)
    if (game_cus < 8 || game_cus % 8 != 0 || cus % 8 != 0 || game_cus > cus / 2)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_search_streams_create: game_cus is a multiple of 8, at most half the "
                                           "device's CUs");
    uint32_t gm[32] = {0}, nm[32] = {0};
    if (cus > 1024)
        return iago_fail(IAGO_ERR_CAPACITY, "iago_mcts_search_streams_create: more CUs than the mask holds");
    for (int i = 0; i < cus; i++)
        (i < game_cus ? gm : nm)[i / 32] |= 1u << (i % 32);
    iago_search_streams *sp = new iago_search_streams();
    sp->device = dev;
    sp->cus = cus;
    sp->game_cus = game_cus;
    const uint32_t words = (uint32_t)((cus + 31) / 32);
    bool ok = hipExtStreamCreateWithCUMask(&sp->game, words, gm) == hipSuccess;
    ok = ok && hipExtStreamCreateWithCUMask(&sp->net, words, nm) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sp->ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sp->game_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sp->net_done, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(search_scratch_warm_kernel, dim3(1), dim3(64), 0, sp->game, (uint32_t *)nullptr, 1);
        hipLaunchKernelGGL(search_scratch_warm_kernel, dim3(1), dim3(64), 0, sp->net, (uint32_t *)nullptr, 1);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(sp->game) == hipSuccess &&
             hipStreamSynchronize(sp->net) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        iago_mcts_search_streams_destroy(sp);
        return iago_fail(IAGO_ERR_HIP, "iago_mcts_search_streams_create: no CU-masked streams on this device (use the single "
                                       "launch, iago_mcts_search_persistent)");
    }
    *out = sp;
    return IAGO_OK;
}

extern "C" int iago_mcts_search_streams_destroy(iago_search_streams *sp)
{
    if (!sp)
        return IAGO_OK;
    if (sp->game)
        (void)hipStreamDestroy(sp->game);
    if (sp->net)
        (void)hipStreamDestroy(sp->net);
    if (sp->ready)
        (void)hipEventDestroy(sp->ready);
    if (sp->game_done)
        (void)hipEventDestroy(sp->game_done);
    if (sp->net_done)
        (void)hipEventDestroy(sp->net_done);
    delete sp;
    return IAGO_OK;
}
