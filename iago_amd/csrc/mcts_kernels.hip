// mcts_kernels.hip -- PV-MCTS tree arithmetic for thousands of lockstep games.
//
// Replaces the reference's recursive dict tree (MCTS.py:10-154); contracts
// and citations per entry point are in include/iago_hip.h.  The tree lives in
// HBM as struct-of-arrays pools (one segment per game); the host loop launches
//   select -> [policy net] -> expand -> select(continue) -> [value net, rollout]
//   -> leaf_values -> backup
// once per simulation for all games at once.
//
// Mapping: 8 lanes per game (othello_dev.hpp).  In select, the children of a
// node are scored 8 at a time (lane j takes children j, j+8, ...; their
// (n, Q, P) are contiguous, so a group reads 3 coalesced 32-byte runs), the
// argmax is a 3-step DPP butterfly on (float64 score, child index), and the
// chosen move is applied with the direction-per-lane flip primitive.
// Scores are float64 exactly as the reference computes them under numpy >= 2
// (float32 P and Q, float64 sqrt / divide / add); no FMA contraction can occur
// in these expressions (no multiply feeds an add).
#include "mcts_dev.hpp"

using namespace iago;
using namespace iago_mcts;

namespace {

__global__ __launch_bounds__(BLOCK) void reset_kernel(Tree T, const uint8_t *__restrict__ mask)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= T.n_games || (mask && !mask[g]))
        return;
    init_node(T, g * T.capacity, -1, -2, 1.0f + 0.1f); // Node(None, 1.0), MCTS.py:81
    T.n_nodes[g] = 1;
    T.root[g] = 0;
    T.overflow[g] = 0;
}

__global__ __launch_bounds__(BLOCK) void select_kernel(
    Tree T, const uint64_t *__restrict__ root_own, const uint64_t *__restrict__ root_opp,
    const uint8_t *__restrict__ active, float c_puct, int n_thr, int from_root,
    int32_t *__restrict__ cur_node, uint64_t *__restrict__ cur_own, uint64_t *__restrict__ cur_opp,
    uint8_t *__restrict__ needs_expand, uint64_t *__restrict__ legal_out,
    int32_t *__restrict__ stats)
{
    int st_levels = 0, st_children = 0;

    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t g = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const bool live = g < T.n_games && active[g] != 0;
    const int64_t base = live ? g * (int64_t)T.capacity : 0;

    int node = 0;
    uint64_t own = 0, opp = 0;
    if (live) {
        node = from_root ? T.root[g] : cur_node[g];
        own = from_root ? root_own[g] : cur_own[g];
        opp = from_root ? root_opp[g] : cur_opp[g];
    }
    bool descending = live;
    for (int depth = 0; depth < MAX_DEPTH; depth++) {
        const int fc = descending ? T.nodes[base + node].first_child : -1;
        descending = descending && fc >= 0; // leaf reached (MCTS.py:107)
        if (__builtin_amdgcn_ballot_w64(descending) == 0ull)
            break;
        const int k = descending ? (int)T.nodes[base + node].n_children : 0;
        const int pn = descending ? T.nodes[base + node].n_visits : 0;
        st_levels += descending ? 1 : 0;
        st_children += k;
        const double sq = sqrt((double)pn); // np.sqrt(parent.n_visits), MCTS.py:49
        double best_v = -INFINITY;
        int best_i = 0x7fffffff;
        for (int j = (int)L.l8; j < k; j += 8) {
            const int64_t c = base + fc + j;
            const float cp = c_puct * T.nodes[c].p;                        // float32, MCTS.py:49
            const double u = (double)cp * sq / (0.01 + (double)T.nodes[c].n_visits);
            const double v = (double)T.nodes[c].q + u;                     // get_value, MCTS.py:75-76
            if (v > best_v) { // strict: the first maximum wins (python max, MCTS.py:46)
                best_v = v;
                best_i = j;
            }
        }
        argmax_step<DPP_XOR1>(best_v, best_i);
        argmax_step<DPP_XOR2>(best_v, best_i);
        argmax_step<DPP_HALF_MIRROR>(best_v, best_i);
        const int child = fc + best_i;
        const int a = descending ? (int)T.nodes[base + child].action : -1;
        // GameFunctions.place_stone(state, action, c); c = 3 - c  (MCTS.py:131-132)
        const uint64_t f =
            group8_flips(to_lane(own, L), to_lane(opp, L), (uint32_t)a & 63u, L);
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
        }
    }
    // leaf: expansion test (MCTS.py:109) and its legal moves (MCTS.py:111)
    const uint64_t legal = group8_legal(to_lane(own, L), to_lane(opp, L), L);
    if (live && L.l8 == 0 && descending && T.nodes[base + node].first_child >= 0)
        T.overflow[g] = 1; // path longer than MAX_DEPTH: reported like a full pool
    if (live && L.l8 == 0) {
        cur_node[g] = node;
        cur_own[g] = own;
        cur_opp[g] = opp;
        const bool ne = T.nodes[base + node].first_child < 0 && T.nodes[base + node].n_visits >= n_thr;
        needs_expand[g] = ne ? 1 : 0;
        legal_out[g] = legal;
        if (stats) {
            stats[2 * g] += st_levels;
            stats[2 * g + 1] += st_children;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void expand_kernel(Tree T, const int32_t *__restrict__ games,
                                                       int64_t n_expand,
                                                       const int32_t *__restrict__ cur_node,
                                                       const uint64_t *__restrict__ legal,
                                                       const float *__restrict__ probs,
                                                       const int32_t *__restrict__ n_dev)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t i = gtid >> 3;
    const uint32_t r = threadIdx.x & 7u;
    const bool live = i < n_expand && (!n_dev || i < (int64_t)*n_dev);
    const int64_t g = live ? games[i] : 0;
    const int64_t base = g * (int64_t)T.capacity;
    const int node = live ? cur_node[g] : 0;
    const uint64_t lg = live ? legal[g] : 0ull;
    const int k = lg ? __popcll(lg) : 1;
    // lane 0 allocates k nodes; every lane of the group learns the start
    uint32_t fc1 = 0; // first child + 1, 0 = no room / already expanded
    if (live && r == 0u && T.nodes[base + node].first_child < 0) {
        const int at = T.n_nodes[g];
        if (at + k <= T.capacity) {
            T.n_nodes[g] = at + k;
            fc1 = (uint32_t)at + 1u;
        } else {
            T.overflow[g] = 1;
        }
    }
    fc1 = group8_add(fc1);
    if (!live || fc1 == 0u)
        return;
    const int fc = (int)fc1 - 1;
    if (lg == 0ull || k == 1) {
        // pass child / single legal move: Node(node, 1), no net (MCTS.py:112-117)
        if (r == 0u) {
            const int a = lg ? (int)__builtin_ctzll(lg) : -1;
            init_node(T, base + fc, node, a, 1.0f + 0.1f);
        }
    } else {
        // Node.expand (MCTS.py:27-37): lane r creates the children of board row r
        uint32_t row = (uint32_t)(lg >> (8u * r)) & 0xFFu;
        int at = fc + __popcll(lg & ((1ull << (8u * r)) - 1ull));
        while (row) {
            const int a = (int)(8u * r) + __builtin_ctz(row);
            row &= row - 1u;
            init_node(T, base + at, node, a, probs[i * 64 + a] + 0.1f); // MCTS.py:19
            at++;
        }
    }
    if (r == 0u) {
        T.nodes[base + node].first_child = fc;
        T.nodes[base + node].n_children = (uint8_t)k;
    }
}

__global__ __launch_bounds__(BLOCK) void leaf_values_kernel(const float *__restrict__ v,
                                                            const int8_t *__restrict__ z,
                                                            float lmbda, float *__restrict__ out,
                                                            int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n)
        return;
    // (1-lmbda)*v + lmbda*z with numpy>=2 scalar promotion (MCTS.py:123-125):
    // the python-float factors are rounded to float32, products and sum in float32
    const float a = (lmbda < 1.0f) ? (float)(1.0 - (double)lmbda) * v[i] : 0.0f;
    const float b = (lmbda > 0.0f) ? (float)((double)lmbda * (double)z[i]) : 0.0f;
    out[i] = a + b;
}

__global__ __launch_bounds__(BLOCK) void backup_kernel(Tree T, const uint8_t *__restrict__ active,
                                                       const int32_t *__restrict__ cur_node,
                                                       const float *__restrict__ leaf_value)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= T.n_games || !active[g])
        return;
    const int64_t base = g * (int64_t)T.capacity;
    const float lv = leaf_value[g];
    int node = cur_node[g];
    for (int depth = 0; node >= 0 && depth <= MAX_DEPTH; depth++) {
        const int n = T.nodes[base + node].n_visits + 1; // MCTS.py:61
        const float q = T.nodes[base + node].q;
        T.nodes[base + node].n_visits = n;
        T.nodes[base + node].q = q + (lv - q) / (float)n; // MCTS.py:63
        node = T.nodes[base + node].parent;               // MCTS.py:71-72, same value, no sign flip
    }
}

// leaf mix (leaf_values_kernel) + backup (backup_kernel) in one launch; every game's mixed
// value is also stored to `leaf_value`, thread 0 bumps the optional playout counter.
__global__ __launch_bounds__(BLOCK) void mix_backup_kernel(Tree T, const uint8_t *__restrict__ active,
                                                           const int32_t *__restrict__ cur_node,
                                                           const float *__restrict__ v,
                                                           const int8_t *__restrict__ z, float lmbda,
                                                           float *__restrict__ leaf_value,
                                                           uint32_t *__restrict__ counter)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g == 0 && counter)
        *counter += 1u;
    if (g >= T.n_games)
        return;
    // value_func(leaf) (MCTS.py:97-103,124) is a pure function of the leaf's position: with a
    // value cache (T.v) the net ran only for the leaves it has not seen (iago_mcts_fresh_leaves),
    // whose values are in v[g] and are stored now; the others take the stored value
    const int64_t base = g * (int64_t)T.capacity;
    float vg = (lmbda < 1.0f) ? v[g] : 0.0f;
    if (T.has_v && lmbda < 1.0f && active[g]) {
        float *slot = &T.nodes[base + cur_node[g]].v;
        const float cached = *slot;
        if (cached != cached)
            *slot = vg;
        else
            vg = cached;
    }
    // (1-lmbda)*v + lmbda*z exactly as leaf_values_kernel (MCTS.py:123-125)
    const float a = (lmbda < 1.0f) ? (float)(1.0 - (double)lmbda) * vg : 0.0f;
    const float b = (lmbda > 0.0f) ? (float)((double)lmbda * (double)z[g]) : 0.0f;
    const float lv = a + b;
    leaf_value[g] = lv;
    if (!active[g])
        return;
    int node = cur_node[g];
    for (int depth = 0; node >= 0 && depth <= MAX_DEPTH; depth++) {
        const int n = T.nodes[base + node].n_visits + 1; // MCTS.py:61
        const float q = T.nodes[base + node].q;
        T.nodes[base + node].n_visits = n;
        T.nodes[base + node].q = q + (lv - q) / (float)n; // MCTS.py:63
        node = T.nodes[base + node].parent;
    }
}

__global__ __launch_bounds__(BLOCK) void best_move_kernel(Tree T, const uint8_t *__restrict__ active,
                                                          int8_t *__restrict__ move,
                                                          int32_t *__restrict__ visits)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= T.n_games || (active && !active[g]))
        return;
    const int64_t base = g * (int64_t)T.capacity;
    const int root = T.root[g];
    const int fc = T.nodes[base + root].first_child;
    const int k = fc >= 0 ? (int)T.nodes[base + root].n_children : 0;
    int best = -2, best_n = -1;
    if (visits)
        for (int a = 0; a < 64; a++)
            visits[g * 64 + a] = 0;
    for (int j = 0; j < k; j++) {
        const int n = T.nodes[base + fc + j].n_visits;
        const int a = (int)T.nodes[base + fc + j].action;
        if (n > best_n) { // first maximum wins (MCTS.py:147)
            best_n = n;
            best = a;
        }
        if (visits && a >= 0)
            visits[g * 64 + a] = n;
    }
    move[g] = (int8_t)best;
}

__global__ __launch_bounds__(BLOCK) void advance_root_kernel(Tree T, const uint8_t *__restrict__ mask,
                                                             const int8_t *__restrict__ move)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= T.n_games || (mask && !mask[g]))
        return;
    const int64_t base = g * (int64_t)T.capacity;
    const int root = T.root[g];
    const int fc = T.nodes[base + root].first_child;
    const int k = fc >= 0 ? (int)T.nodes[base + root].n_children : 0;
    const int a = (int)move[g];
    for (int j = 0; j < k; j++) {
        if ((int)T.nodes[base + fc + j].action == a) { // last_move in self.root.children (MCTS.py:150)
            T.root[g] = fc + j;
            T.nodes[base + fc + j].parent = -1; // self.root.parent = None (MCTS.py:152)
            return;
        }
    }
    init_node(T, base, -1, -2, 1.0f + 0.1f); // self.root = Node(None, 1.0) (MCTS.py:154)
    T.n_nodes[g] = 1;
    T.root[g] = 0;
}

// ---- policy look-ahead (iago_mcts_mix_backup_lookahead / _store_priors / _expand_cached).
// The reference evaluates the policy net at the visit that expands a leaf (MCTS.py:109-121), a
// handful of games per lockstep playout: nine latency-bound launches on the critical path of
// EVERY playout.  policy_func(state) is a pure function of the leaf's position, so it can be
// evaluated earlier: a leaf is queued when its visit count reaches n_thr - K, the queue is
// flushed through the net every K playouts (one larger batch), the priors wait in a per-game
// cache, and the expansion -- at the same visit as in the reference, with the same values --
// reads them from there.  An unexpanded leaf carries its cache tag in first_child:
// -1 = not queued, -2 - seq = priors of sequence number seq (slot seq % slots of its game).
struct Lookahead {
    int32_t trigger;            // visit count at which a leaf is queued (n_thr - K)
    int32_t slots;              // cache slots per game
    int32_t *next_seq;          // [n_games] sequence numbers handed out so far
    int32_t *cache_seq;         // [n_games][slots] sequence number stored in a slot, -1 = empty
    float *cache;               // [n_games][slots][64] priors (raw softmax entries)
    int32_t *q_count;           // entries in the queue
    int32_t q_capacity;
    uint64_t *q_own, *q_opp;    // the queued leaves' positions (own = side to move)
    int32_t *q_game, *q_seq;
    int32_t *error;             // raised when a cache slot was recycled before its leaf expanded,
                                // a leaf reaches n_thr without priors, or the queue is full
    int32_t *clear_word;        // optional: set to 0 by the backup (the fresh-leaf count of the next descent)
    int32_t *path, *path_len;   // optional: the nodes of a game's last descent, root first ([n_games][path_stride])
    int32_t path_stride;
    int8_t *z_log;              // optional [z_log_rows][n_games]: the rollout result the backup mixed in, one
    int32_t *z_log_n;           // row per playout of a game (z_log_n [n_games] = rows written so far)
    int32_t z_log_rows;
    // game-asynchronous steps (iago_mcts_async, include/iago_hip.h); y_wait == nullptr: lockstep playouts
    int32_t y_parts;
    int32_t *y_wait, *y_done;
    uint8_t *y_roll;
    int64_t *y_fq_index;
    int32_t *y_fq_count;
    uint32_t *y_step;
    const int32_t *y_n_sims;
    // value look-ahead (iago_mcts_value_ahead); va_x_count == nullptr: off
    int32_t *va_x_count;
    int32_t va_x_capacity;
    int32_t *va_x_game, *va_x_node;
    uint64_t *va_x_own, *va_x_opp;
};

// Diagnostic record of the parity tests (tests/test_mcts_production_gpu.py): the z every playout of
// game g backed up, in playout order -- what the oracle's rollout_fn replays.
__device__ __forceinline__ void log_z(const Lookahead &A, int64_t g, int64_t n_games, int8_t zg)
{
    const int k = A.z_log_n[g];
    A.z_log_n[g] = k + 1;
    if (k < A.z_log_rows)
        A.z_log[(int64_t)k * n_games + g] = zg;
}

__global__ __launch_bounds__(BLOCK) void mix_backup_lookahead_kernel(
    Tree T, const uint8_t *__restrict__ active, const int32_t *__restrict__ cur_node,
    const uint64_t *__restrict__ cur_own, const uint64_t *__restrict__ cur_opp, const float *__restrict__ v,
    const int8_t *__restrict__ z, float lmbda, float *__restrict__ leaf_value, uint32_t *__restrict__ counter,
    Lookahead A)
{
    const int64_t g = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g == 0 && counter)
        *counter += 1u;
    if (g == 0 && A.clear_word)
        *A.clear_word = 0;
    if (g >= T.n_games)
        return;
    // value_func(leaf) (MCTS.py:97-103,124) is a pure function of the leaf's position: with a
    // value cache (T.v) the net ran only for the leaves it has not seen (iago_mcts_fresh_leaves),
    // whose values are in v[g] and are stored now; the others take the stored value
    const int64_t base = g * (int64_t)T.capacity;
    float vg = (lmbda < 1.0f) ? v[g] : 0.0f;
    if (T.has_v && lmbda < 1.0f && active[g]) {
        float *slot = &T.nodes[base + cur_node[g]].v;
        const float cached = *slot;
        if (cached != cached)
            *slot = vg;
        else
            vg = cached;
    }
    // (1-lmbda)*v + lmbda*z exactly as leaf_values_kernel (MCTS.py:123-125)
    const float a = (lmbda < 1.0f) ? (float)(1.0 - (double)lmbda) * vg : 0.0f;
    const float b = (lmbda > 0.0f) ? (float)((double)lmbda * (double)z[g]) : 0.0f;
    const float lv = a + b;
    leaf_value[g] = lv;
    if (!active[g])
        return;
    if (A.z_log && lmbda > 0.0f)
        log_z(A, g, T.n_games, z[g]);
    const int leaf = cur_node[g];
    int node = leaf;
    for (int depth = 0; node >= 0 && depth <= MAX_DEPTH; depth++) {
        const int n = T.nodes[base + node].n_visits + 1; // MCTS.py:61
        const float q = T.nodes[base + node].q;
        T.nodes[base + node].n_visits = n;
        T.nodes[base + node].q = q + (lv - q) / (float)n; // MCTS.py:63
        node = T.nodes[base + node].parent;
    }
    // the leaf of this playout crosses the trigger exactly once (it gains one visit per playout
    // that ends on it): queue its position for the next flush
    if (T.nodes[base + leaf].n_visits == A.trigger && T.nodes[base + leaf].first_child == -1) {
        const int pos = atomicAdd(A.q_count, 1);
        if (pos < A.q_capacity) {
            const int seq = A.next_seq[g];
            A.next_seq[g] = seq + 1;
            T.nodes[base + leaf].first_child = -2 - seq;
            A.q_own[pos] = cur_own[g];
            A.q_opp[pos] = cur_opp[g];
            A.q_game[pos] = (int32_t)g;
            A.q_seq[pos] = seq;
        } else {
            *A.error = 1;
        }
    }
}

// The same backup with the path the descent recorded (iago_mcts_descend with A.path): the 8 lanes
// of a game update its nodes side by side -- no climb through `parent`, whose loads depend on
// each other (one round trip per level).  Same arithmetic per node: same trees.
__global__ __launch_bounds__(BLOCK) void mix_backup_path_kernel(
    Tree T, const uint8_t *__restrict__ active, const int32_t *__restrict__ cur_node,
    const uint64_t *__restrict__ cur_own, const uint64_t *__restrict__ cur_opp, const float *__restrict__ v,
    const int8_t *__restrict__ z, float lmbda, float *__restrict__ leaf_value, uint32_t *__restrict__ counter,
    Lookahead A)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t g = gtid >> 3;
    const uint32_t r = threadIdx.x & 7u;
    if (gtid == 0 && counter)
        *counter += 1u;
    if (gtid == 0 && A.clear_word && !A.y_wait)
        *A.clear_word = 0;
    if (A.y_wait && gtid == 0) {
        // the step is over: the queue row the last piece of the value net has just drained is the
        // one the next step's descent appends to
        const uint32_t s = *A.y_step;
        A.y_fq_count[(s + 1u) % (uint32_t)A.y_parts] = 0;
        *A.y_step = s + 1u;
    }
    if (g >= T.n_games)
        return;
    const int64_t base = g * (int64_t)T.capacity;
    bool act = active[g] != 0;
    if (A.y_wait) {
        // a game completes a playout in this step when its leaf had a stored value (it descended
        // in this step: roll) or when the value it has been waiting for has arrived (wait == 1)
        const int w = A.y_wait[g];
        if (w > 1) {
            if (r == 0u)
                A.y_wait[g] = w - 1;
            act = false;
        } else if (w == 1) {
            if (r == 0u)
                A.y_wait[g] = 0;
        } else {
            act = act && A.y_roll[g] != 0;
        }
        if (act && r == 0u)
            A.y_done[g] += 1;
    }
    const int leaf = cur_node[g];
    float vg = (lmbda < 1.0f) ? v[g] : 0.0f;
    if (T.has_v && lmbda < 1.0f && act) {
        // (every lane reads the slot, lane 0 fills it: the loads of a wave come before its stores)
        float *slot = &T.nodes[base + leaf].v;
        const float cached = *slot;
        if (cached != cached) {
            if (r == 0u)
                *slot = vg;
        } else {
            vg = cached;
        }
    }
    // (1-lmbda)*v + lmbda*z exactly as leaf_values_kernel (MCTS.py:123-125)
    const float a = (lmbda < 1.0f) ? (float)(1.0 - (double)lmbda) * vg : 0.0f;
    const float b = (lmbda > 0.0f) ? (float)((double)lmbda * (double)z[g]) : 0.0f;
    const float lv = a + b;
    if (r == 0u && (act || !A.y_wait))
        leaf_value[g] = lv;
    if (!act)
        return;
    if (A.z_log && lmbda > 0.0f && r == 0u)
        log_z(A, g, T.n_games, z[g]);
    // lane 0 decides the queueing from the leaf's count BEFORE this playout (+ 1 = after it)
    int leaf_n = 0, leaf_tag = 0;
    if (r == 0u) {
        leaf_n = T.nodes[base + leaf].n_visits;
        leaf_tag = T.nodes[base + leaf].first_child;
    }
    const int len = A.path_len[g];
    const int32_t *path = A.path + g * (int64_t)A.path_stride;
    for (int d = (int)r; d < len; d += 8) {
        const int node = path[d];
        uint2 *nq = (uint2 *)&T.nodes[base + node]; // (n_visits, Q): one 8-byte load, one 8-byte store
        const uint2 old = *nq;
        const int n = (int)old.x + 1;                // MCTS.py:61
        const float q = __uint_as_float(old.y);
        *nq = make_uint2((uint32_t)n, __float_as_uint(q + (lv - q) / (float)n)); // MCTS.py:63
    }
    if (r == 0u && leaf_n + 1 == A.trigger && leaf_tag == -1) {
        const int pos = atomicAdd(A.q_count, 1);
        if (pos < A.q_capacity) {
            const int seq = A.next_seq[g];
            A.next_seq[g] = seq + 1;
            T.nodes[base + leaf].first_child = -2 - seq;
            A.q_own[pos] = cur_own[g];
            A.q_opp[pos] = cur_opp[g];
            A.q_game[pos] = (int32_t)g;
            A.q_seq[pos] = seq;
        } else {
            *A.error = 1;
        }
    }
}

// probs [>= *q_count][64]: the net's outputs for the queued positions, row i for queue entry i
__global__ __launch_bounds__(BLOCK) void store_priors_kernel(Lookahead A, const float *__restrict__ probs,
                                                             int64_t *__restrict__ total)
{
    const int count = min(*A.q_count, A.q_capacity);
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (t == 0 && total)
        *total += count;
    for (int64_t e = t; e < (int64_t)count * 16; e += (int64_t)gridDim.x * BLOCK) {
        const int i = (int)(e >> 4), piece = (int)(e & 15);
        const int64_t g = A.q_game[i];
        const int seq = A.q_seq[i], slot = seq % A.slots;
        ((float4 *)(A.cache + (g * A.slots + slot) * 64))[piece] = ((const float4 *)(probs + (int64_t)i * 64))[piece];
        if (piece == 0)
            A.cache_seq[g * A.slots + slot] = seq;
    }
}

// Node.expand (MCTS.py:27-37, 109-121) for every active game whose cursor sits on a leaf with
// n_visits >= n_thr, the priors taken from the cache; expanded[g] = 1 for those games (the
// mask of the descent that continues into the new children), 0 for the others.
__global__ __launch_bounds__(BLOCK) void expand_cached_kernel(Tree T, const uint8_t *__restrict__ active,
                                                              const uint8_t *__restrict__ needs_expand,
                                                              const int32_t *__restrict__ cur_node,
                                                              const uint64_t *__restrict__ legal, Lookahead A,
                                                              uint8_t *__restrict__ expanded)
{
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t g = gtid >> 3;
    const uint32_t r = threadIdx.x & 7u;
    const bool live = g < T.n_games && active[g] != 0 && needs_expand[g] != 0;
    if (g < T.n_games && r == 0u)
        expanded[g] = live ? 1 : 0;
    const int64_t base = live ? g * (int64_t)T.capacity : 0;
    const int node = live ? cur_node[g] : 0;
    const uint64_t lg = live ? legal[g] : 0ull;
    const int k = lg ? __popcll(lg) : 1;
    const int tag = live ? T.nodes[base + node].first_child : 0;
    uint32_t fc1 = 0; // first child + 1, 0 = no room / already expanded
    if (live && r == 0u && tag < 0) {
        const int at = T.n_nodes[g];
        if (at + k <= T.capacity) {
            T.n_nodes[g] = at + k;
            fc1 = (uint32_t)at + 1u;
        } else {
            T.overflow[g] = 1;
        }
    }
    fc1 = group8_add(fc1);
    if (!live || fc1 == 0u)
        return;
    const int fc = (int)fc1 - 1;
    if (lg == 0ull || k == 1) {
        // pass child / single legal move: Node(node, 1), no net (MCTS.py:112-117)
        if (r == 0u) {
            const int a = lg ? (int)__builtin_ctzll(lg) : -1;
            init_node(T, base + fc, node, a, 1.0f + 0.1f);
        }
    } else {
        const int seq = -2 - tag;
        const int slot = tag <= -2 ? seq % A.slots : 0;
        const float *probs = A.cache + (g * A.slots + slot) * 64;
        if (tag > -2 || A.cache_seq[g * A.slots + slot] != seq) {
            if (r == 0u)
                *A.error = 2; // no priors for this leaf (not queued, or its slot was recycled)
        }
        uint32_t row = (uint32_t)(lg >> (8u * r)) & 0xFFu;
        int at = fc + __popcll(lg & ((1ull << (8u * r)) - 1ull));
        while (row) {
            const int a = (int)(8u * r) + __builtin_ctz(row);
            row &= row - 1u;
            init_node(T, base + at, node, a, probs[a] + 0.1f); // MCTS.py:19
            at++;
        }
    }
    if (r == 0u) {
        T.nodes[base + node].first_child = fc;
        T.nodes[base + node].n_children = (uint8_t)k;
    }
}

// ---- the whole descent of a playout in ONE launch (iago_mcts_descend): select from the root
// (MCTS.py:129-133), and where the leaf it reaches has n_visits >= n_thr (MCTS.py:109) the
// expansion from the prior cache and the continued descent into the new children (MCTS.py:
// 110-121), then the list of the leaves that have no cached value yet -- select_kernel,
// expand_cached_kernel, select_kernel again and fresh_leaves_kernel with the same arithmetic.
// A game's 8 lanes expand together; the fresh list is appended with an atomic counter (its
// order varies from run to run: the value net's output for a board does not depend on its row).
__global__ __launch_bounds__(BLOCK) void descend_kernel(
    Tree T, const uint64_t *__restrict__ root_own, const uint64_t *__restrict__ root_opp,
    const uint8_t *__restrict__ active, float c_puct, int n_thr, int32_t *__restrict__ cur_node,
    uint64_t *__restrict__ cur_own, uint64_t *__restrict__ cur_opp, uint64_t *__restrict__ legal_out,
    int32_t *__restrict__ stats, Lookahead A, int64_t *fresh_index, int32_t *fresh_count,
    int64_t *__restrict__ fresh_total)
{
    int st_levels = 0, st_children = 0;
    const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t g = gtid >> 3;
    const Lane8 L = make_lane8(threadIdx.x);
    const uint32_t r = L.l8;
    bool live = g < T.n_games && active[g] != 0;
    if (A.y_wait) {
        // a game-asynchronous step: the games that wait for a value and the games that have
        // completed the search's playouts sit this step out
        live = live && A.y_wait[g] == 0 && A.y_done[g] < *A.y_n_sims;
        if (g < T.n_games && r == 0u)
            A.y_roll[g] = live ? 1 : 0;
        const uint32_t row = *A.y_step % (uint32_t)A.y_parts;
        fresh_index = A.y_fq_index + (int64_t)row * T.n_games;
        fresh_count = A.y_fq_count + row;
    }
    const int64_t base = live ? g * (int64_t)T.capacity : 0;

    // The descent is a chain of dependent loads (a level cannot start before the previous one
    // has chosen its child): ONE round trip per level.  A lane that scores child c also loads
    // c's action and c's own header (first_child, n_children, n_visits -- the statistics are
    // loaded anyway -- and the stored value); the argmax butterfly carries them along, so the
    // winner's move and the next level's header arrive with the winner's index.
    int node = 0, fc = -1, k = 0, nv = 0;
    uint32_t vbits = 0; // T.nodes[node].v (the value cache), NaN = not evaluated
    uint64_t own = 0, opp = 0;
    if (live) {
        node = T.root[g];
        own = root_own[g];
        opp = root_opp[g];
        const uint4 s0 = ((const uint4 *)&T.nodes[base + node])[0], l0 = ((const uint4 *)&T.nodes[base + node])[1];
        fc = (int)l0.x;
        k = (int)((l0.z >> 8) & 0xFFu);
        nv = (int)s0.x;
        vbits = s0.w;
    }
    int32_t *const path = (A.path && live) ? A.path + g * (int64_t)A.path_stride : nullptr;
    int path_n = 0;
    bool descending = live, may_expand = live;
    for (int depth = 0; depth < MAX_DEPTH; depth++) {
        if (path && descending && r == 0u) {
            if (depth < A.path_stride)
                path[depth] = node;
            else
                T.overflow[g] = 1; // deeper than the path buffer: reported like a full pool
        }
        path_n += descending ? 1 : 0;
        // a leaf with n_visits >= n_thr expands here and the descent goes on into its children
        // (once per playout: the new children have no visits)
        const bool expand = descending && may_expand && fc < 0 && nv >= n_thr;
        if (__builtin_amdgcn_ballot_w64(expand) != 0ull) {
            const uint64_t lg = group8_legal(to_lane(own, L), to_lane(opp, L), L);
            if (expand) {
                may_expand = false;
                const int kn = lg ? __popcll(lg) : 1;
                const int tag = fc;
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
                        const int seq = -2 - tag;
                        const int slot = tag <= -2 ? seq % A.slots : 0;
                        const float *probs = A.cache + (g * A.slots + slot) * 64;
                        if ((tag > -2 || A.cache_seq[g * A.slots + slot] != seq) && r == 0u)
                            *A.error = 2; // no priors for this leaf (not queued, or its slot was recycled)
                        uint32_t row = (uint32_t)(lg >> (8u * r)) & 0xFFu;
                        int at = nf + __popcll(lg & ((1ull << (8u * r)) - 1ull));
                        while (row) {
                            const int a = (int)(8u * r) + __builtin_ctz(row);
                            row &= row - 1u;
                            init_node(T, base + at, node, a, probs[a] + 0.1f); // MCTS.py:19
                            at++;
                        }
                    }
                    if (r == 0u) {
                        T.nodes[base + node].first_child = nf;
                        T.nodes[base + node].n_children = (uint8_t)kn;
                        if (A.va_x_count) {
                            // value look-ahead: the new children will be first-visited during this
                            // node's next visits (a full queue drops the hint)
                            const int pos = atomicAdd(A.va_x_count, 1);
                            if (pos < A.va_x_capacity) {
                                A.va_x_game[pos] = (int32_t)g;
                                A.va_x_node[pos] = node;
                                A.va_x_own[pos] = own;
                                A.va_x_opp[pos] = opp;
                            }
                        }
                    }
                    fc = nf;
                    k = kn;
                }
            }
            __threadfence_block(); // the new children are read by the other lanes of the group below
        }
        descending = descending && fc >= 0; // leaf reached (MCTS.py:107)
        if (__builtin_amdgcn_ballot_w64(descending) == 0ull)
            break;
        const int kk = descending ? k : 0;
        st_levels += descending ? 1 : 0;
        st_children += kk;
        const double sq = sqrt((double)nv); // np.sqrt(parent.n_visits), MCTS.py:49
        double best_v = -INFINITY;
        int best_i = 0x7fffffff;
        uint32_t pl[4] = {0u, 0u, 0u, 0u}; // of the best child: first_child, n_visits, action | n_children << 8, v
        // two children per lane and pass (16 per group): their loads are all in flight together
        for (int j0 = (int)r; j0 < kk; j0 += 16) {
            const int j1 = j0 + 8;
            const bool two = j1 < kk;
            const int64_t c0 = base + fc + j0, c1 = two ? base + fc + j1 : c0;
            // a child's record as two 16-byte loads: (n_visits, Q, P, v) | (first_child, parent, action |
            // n_children << 8, -) -- one 32-byte sector per child
            const uint4 s0 = ((const uint4 *)&T.nodes[c0])[0], l0 = ((const uint4 *)&T.nodes[c0])[1];
            const uint4 s1 = ((const uint4 *)&T.nodes[c1])[0], l1 = ((const uint4 *)&T.nodes[c1])[1];
            const float p0 = __uint_as_float(s0.z), q0 = __uint_as_float(s0.y);
            const float p1 = __uint_as_float(s1.z), q1 = __uint_as_float(s1.y);
            const int n0 = (int)s0.x, n1 = (int)s1.x;
            const int f0 = (int)l0.x, f1 = (int)l1.x;
            const uint32_t a0 = l0.z & 0xFFFFu, a1 = l1.z & 0xFFFFu;
            const uint32_t w0 = s0.w, w1 = s1.w;
            {
                const float cp = c_puct * p0;                          // float32, MCTS.py:49
                const double u = (double)cp * sq / (0.01 + (double)n0);
                const double v = (double)q0 + u;                       // get_value, MCTS.py:75-76
                if (v > best_v) { // strict: the first maximum wins (python max, MCTS.py:46)
                    best_v = v;
                    best_i = j0;
                    pl[0] = (uint32_t)f0, pl[1] = (uint32_t)n0, pl[2] = a0, pl[3] = w0;
                }
            }
            if (two) {
                const float cp = c_puct * p1;
                const double u = (double)cp * sq / (0.01 + (double)n1);
                const double v = (double)q1 + u;
                if (v > best_v) {
                    best_v = v;
                    best_i = j1;
                    pl[0] = (uint32_t)f1, pl[1] = (uint32_t)n1, pl[2] = a1, pl[3] = w1;
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
    const uint64_t legal = group8_legal(to_lane(own, L), to_lane(opp, L), L);
    if (live && r == 0 && descending && fc >= 0)
        T.overflow[g] = 1; // path longer than MAX_DEPTH: reported like a full pool
    if (live && r == 0) {
        cur_node[g] = node;
        cur_own[g] = own;
        cur_opp[g] = opp;
        legal_out[g] = legal;
        if (A.path)
            A.path_len[g] = path_n < A.path_stride ? path_n : A.path_stride;
        if (stats) {
            stats[2 * g] += st_levels;
            stats[2 * g + 1] += st_children;
        }
        if (fresh_index) {
            const float c = __uint_as_float(vbits);
            if (c != c) {
                // (at most one entry per game and playout; a count the previous backup did not clear
                // -- a playout aborted between descent and backup -- must not run past the list)
                const int pos = atomicAdd(fresh_count, 1);
                if (pos < T.n_games)
                    fresh_index[pos] = g;
                else
                    T.overflow[g] = 1;
                if (A.y_wait)
                    A.y_wait[g] = A.y_parts; // the value arrives with the last piece, parts - 1 steps on
                if (fresh_total)
                    atomicAdd((unsigned long long *)fresh_total, 1ull);
            }
        }
    }
}

// ---- pool compaction (iago_mcts_compact).  Subtree reuse (advance_root) keeps the chosen
// child's subtree and abandons its siblings' nodes in the pool; the reference's garbage
// collector frees them (MCTS.py:149-152 drops the last reference).  Here a game's live subtree
// is re-laid in breadth-first order from index 0: children stay contiguous and in their order,
// so every later select / expand / backup behaves exactly as on the uncompacted pool.
//   plan:   one thread per game walks the subtree with `order` as its queue (order[new] = old)
//           and writes the new parent / first_child links into the scratch pool;
//   gather: the statistics of the live nodes into the scratch pool, in the new order;
//   commit: scratch -> pool, n_nodes = live nodes, root = 0.
__global__ __launch_bounds__(64) void compact_plan_kernel(Tree T, Tree S, const uint8_t *__restrict__ mask,
                                                          int32_t *__restrict__ order)
{
    const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= T.n_games)
        return;
    if (mask && !mask[g]) {
        S.n_nodes[g] = -1; // not compacted
        return;
    }
    const int64_t base = g * (int64_t)T.capacity;
    int32_t *q = order + base;
    q[0] = T.root[g];
    S.nodes[base].parent = -1;
    int head = 0, tail = 1;
    while (head < tail) {
        const int o = q[head];
        const int fc = T.nodes[base + o].first_child;
        const int k = fc >= 0 ? (int)T.nodes[base + o].n_children : 0;
        S.nodes[base + head].first_child = k ? tail : fc; // (fc < 0: an unexpanded leaf keeps its prior-cache tag)
        for (int j = 0; j < k; j++) {
            q[tail + j] = fc + j;
            S.nodes[base + tail + j].parent = head;
        }
        tail += k;
        head++;
    }
    S.n_nodes[g] = tail;
}

__global__ __launch_bounds__(BLOCK) void compact_gather_kernel(Tree T, Tree S, const int32_t *__restrict__ order)
{
    const int64_t g = blockIdx.x;
    const int count = S.n_nodes[g];
    const int64_t base = g * (int64_t)T.capacity;
    for (int i = threadIdx.x; i < count; i += BLOCK) {
        const int64_t o = base + order[base + i];
        S.nodes[base + i].n_children = T.nodes[o].n_children;
        S.nodes[base + i].action = T.nodes[o].action;
        S.nodes[base + i].n_visits = T.nodes[o].n_visits;
        S.nodes[base + i].q = T.nodes[o].q;
        S.nodes[base + i].p = T.nodes[o].p;
        if (T.has_v && S.has_v)
            S.nodes[base + i].v = T.nodes[o].v;
    }
}

__global__ __launch_bounds__(BLOCK) void compact_commit_kernel(Tree T, Tree S)
{
    const int64_t g = blockIdx.x;
    const int count = S.n_nodes[g];
    const int64_t base = g * (int64_t)T.capacity;
    for (int i = threadIdx.x; i < count; i += BLOCK) {
        T.nodes[base + i].parent = S.nodes[base + i].parent;
        T.nodes[base + i].first_child = S.nodes[base + i].first_child;
        T.nodes[base + i].n_children = S.nodes[base + i].n_children;
        T.nodes[base + i].action = S.nodes[base + i].action;
        T.nodes[base + i].n_visits = S.nodes[base + i].n_visits;
        T.nodes[base + i].q = S.nodes[base + i].q;
        T.nodes[base + i].p = S.nodes[base + i].p;
        if (T.has_v && S.has_v)
            T.nodes[base + i].v = S.nodes[base + i].v;
    }
    if (threadIdx.x == 0 && count >= 0) {
        T.n_nodes[g] = count;
        T.root[g] = 0;
    }
}

// ---- value look-ahead (iago_mcts_value_ahead_rows / _store).  8 lanes per queued node: its
// legal moves in ascending order are its children in pool order (Node.expand, MCTS.py:27-37);
// every child without a stored value becomes a row of the value net's next batch.
struct ValueAhead {
    int32_t x_capacity, row_capacity;
    int32_t *x_count;
    const int32_t *x_game, *x_node;
    const uint64_t *x_own, *x_opp;
    int32_t *row_count;
    uint64_t *row_own, *row_opp;
    int64_t *row_node;
    float *row_v;
    int64_t *total;
};

__global__ __launch_bounds__(BLOCK) void value_ahead_rows_kernel(Tree T, ValueAhead V)
{
    const int count = min(*V.x_count, V.x_capacity);
    const Lane8 L = make_lane8(threadIdx.x);
    for (int64_t e = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 3; e < (((int64_t)count + 31) & ~31ll);
         e += ((int64_t)gridDim.x * BLOCK) >> 3) {
        // (the loop bound is rounded up to a whole workgroup's 32 entries: the 8 lanes of a group
        // and the groups of a wave run the cross-lane primitives together)
        const bool live = e < count;
        const int64_t g = live ? V.x_game[e] : 0;
        const int node = live ? V.x_node[e] : 0;
        const uint64_t own = live ? V.x_own[e] : 0ull, opp = live ? V.x_opp[e] : 0ull;
        const int64_t base = g * (int64_t)T.capacity;
        const int fc = live ? T.nodes[base + node].first_child : -1;
        const int k = fc >= 0 ? (int)T.nodes[base + node].n_children : 0;
        uint64_t lg = group8_legal(to_lane(own, L), to_lane(opp, L), L);
        // (a node of a finished position has one pass child: the same stones, the other side to move)
        const bool pass = lg == 0ull;
        int j = 0;
        // wave-uniform trip count: the longest list of the wave's groups
        while (__builtin_amdgcn_ballot_w64(live && j < k) != 0ull) {
            const bool on = live && j < k;
            const int a = pass ? -1 : (lg ? (int)__builtin_ctzll(lg) : 0);
            const uint64_t f = group8_flips(to_lane(own, L), to_lane(opp, L), (uint32_t)a & 63u, L);
            if (on && L.l8 == 0u) {
                const int64_t c = base + fc + j;
                const float cv = T.nodes[c].v;
                if (cv != cv) {
                    uint64_t no = own, np_ = opp;
                    if (a >= 0) {
                        const uint64_t bit = 1ull << (a & 63);
                        no = own | f | bit;
                        np_ = opp & ~f & ~bit;
                    }
                    const int pos = atomicAdd(V.row_count, 1);
                    if (pos < V.row_capacity) {
                        V.row_own[pos] = np_; // the child's mover is the other side (MCTS.py:131-132)
                        V.row_opp[pos] = no;
                        V.row_node[pos] = c;
                    }
                }
            }
            lg &= lg - 1ull;
            j++;
        }
    }
}

// (separate launch: the count is final only when every workgroup of the rows kernel is done)
__global__ void value_ahead_clamp_kernel(ValueAhead V)
{
    const int n = min(*V.row_count, V.row_capacity);
    *V.row_count = n;
    if (V.total)
        *V.total += n;
}

__global__ __launch_bounds__(BLOCK) void value_ahead_store_kernel(Tree T, ValueAhead V)
{
    const int count = min(*V.row_count, V.row_capacity);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (int64_t)gridDim.x * BLOCK) {
        float *slot = &T.nodes[V.row_node[i]].v;
        const float cur = *slot;
        if (cur != cur) // (a visit in the meantime has stored the same number)
            *slot = V.row_v[i];
    }
}

inline unsigned grid_for(int64_t threads) { return (unsigned)((threads + BLOCK - 1) / BLOCK); }

int check_tree(const Tree *t, const char *who)
{
    if (!t)
        return iago_fail(IAGO_ERR_INVALID, who);
    if (t->n_games < 0 || t->capacity < 1 || !t->nodes || ((uintptr_t)t->nodes & 31u) || !t->n_nodes || !t->root ||
        !t->overflow)
        return iago_fail(IAGO_ERR_INVALID, who);
    return IAGO_OK;
}

// The games whose cursor sits on a leaf due for expansion, in ascending order: one block
// scans the flags in chunks of 1024 (ballot + prefix over the 16 waves).
__global__ __launch_bounds__(1024) void pending_kernel(const uint8_t *__restrict__ needs_expand,
                                                       const uint8_t *__restrict__ active, int n,
                                                       uint8_t *__restrict__ pending,
                                                       int64_t *__restrict__ index,
                                                       int32_t *__restrict__ games, int32_t *__restrict__ count,
                                                       int64_t *__restrict__ total)
{
    __shared__ int wave_sum[16];
    __shared__ int base;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0)
        base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < n; g0 += 1024) {
        const int g = g0 + tid;
        const bool p = g < n && needs_expand[g] != 0 && active[g] != 0;
        if (g < n)
            pending[g] = p ? 1 : 0;
        const unsigned long long m = __ballot(p);
        if (lane == 0)
            wave_sum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++)
            off += wave_sum[w];
        if (p) {
            const int k = off + __popcll(m & ((1ull << lane) - 1ull));
            index[k] = g;
            games[k] = g;
        }
        __syncthreads();
        if (tid == 0) {
            int t = base;
            for (int w = 0; w < 16; w++)
                t += wave_sum[w];
            base = t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        *count = base;
        if (total)
            *total += base;
    }
}

// The active games whose leaf has no cached value yet (T.v is NaN), in ascending order, and
// their number: the rows of the value net's next launch.
__global__ __launch_bounds__(1024) void fresh_leaves_kernel(Tree T, const uint8_t *__restrict__ active,
                                                            const int32_t *__restrict__ cur_node,
                                                            int64_t *__restrict__ index, int32_t *__restrict__ count,
                                                            int64_t *__restrict__ total)
{
    __shared__ int wave_sum[16];
    __shared__ int base;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = (int)T.n_games;
    if (tid == 0)
        base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < n; g0 += 1024) {
        const int g = g0 + tid;
        bool p = false;
        if (g < n && active[g] != 0) {
            const float c = T.nodes[(int64_t)g * T.capacity + cur_node[g]].v;
            p = c != c;
        }
        const unsigned long long m = __ballot(p);
        if (lane == 0)
            wave_sum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++)
            off += wave_sum[w];
        if (p)
            index[off + __popcll(m & ((1ull << lane) - 1ull))] = g;
        __syncthreads();
        if (tid == 0) {
            int t = base;
            for (int w = 0; w < 16; w++)
                t += wave_sum[w];
            base = t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        *count = base;
        if (total)
            *total += base;
    }
}

} // namespace

extern "C" {

int iago_mcts_reset(const iago_mcts_tree *tree, const uint8_t *mask, void *stream)
{
    if (check_tree(tree, "iago_mcts_reset: bad tree"))
        return IAGO_ERR_INVALID;
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(reset_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, mask);
    return iago_check_launch("iago_mcts_reset");
}

int iago_mcts_select(const iago_mcts_tree *tree, const uint64_t *root_own, const uint64_t *root_opp,
                     const uint8_t *active, float c_puct, int32_t n_thr, int from_root,
                     int32_t *cur_node, uint64_t *cur_own, uint64_t *cur_opp, uint8_t *needs_expand,
                     uint64_t *legal, int32_t *stats, void *stream)
{
    if (check_tree(tree, "iago_mcts_select: bad tree"))
        return IAGO_ERR_INVALID;
    if (!active || !cur_node || !cur_own || !cur_opp || !needs_expand || !legal ||
        (from_root && (!root_own || !root_opp)))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_select: null pointer");
    if (n_thr < 1)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_select: n_thr must be >= 1 (n_thr = 0 never "
                                           "terminates in the reference either)");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(select_kernel, dim3(grid_for(tree->n_games * 8)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, root_own, root_opp, active, c_puct, n_thr,
                       from_root, cur_node, cur_own, cur_opp, needs_expand, legal, stats);
    return iago_check_launch("iago_mcts_select");
}

int iago_mcts_expand(const iago_mcts_tree *tree, const int32_t *games, int64_t n_expand,
                     const int32_t *cur_node, const uint64_t *legal, const float *probs,
                     const int32_t *n_dev, void *stream)
{
    if (check_tree(tree, "iago_mcts_expand: bad tree"))
        return IAGO_ERR_INVALID;
    if (n_expand < 0 || (n_expand > 0 && (!games || !cur_node || !legal || !probs)))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_expand: null pointer or negative count");
    if (n_expand == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(expand_kernel, dim3(grid_for(n_expand * 8)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, games, n_expand, cur_node, legal, probs, n_dev);
    return iago_check_launch("iago_mcts_expand");
}

int iago_mcts_pending(const uint8_t *needs_expand, const uint8_t *active, int64_t n, uint8_t *pending,
                      int64_t *index, int32_t *games, int32_t *count, int64_t *total, void *stream)
{
    if (n < 0 || (n > 0 && (!needs_expand || !active || !pending || !index || !games || !count)))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_pending: null pointer or negative n");
    if (n > (1 << 24))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_pending: more than 2^24 games");
    hipLaunchKernelGGL(pending_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, needs_expand, active,
                       (int)n, pending, index, games, count, total);
    return iago_check_launch("iago_mcts_pending");
}

int iago_leaf_values(const float *v, const int8_t *z, float lmbda, float *leaf_value, int64_t n,
                     void *stream)
{
    if (n < 0 || (n > 0 && (!leaf_value || (lmbda < 1.0f && !v) || (lmbda > 0.0f && !z))))
        return iago_fail(IAGO_ERR_INVALID, "iago_leaf_values: null pointer or negative n");
    if (n == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(leaf_values_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream,
                       v, z, lmbda, leaf_value, n);
    return iago_check_launch("iago_leaf_values");
}

int iago_mcts_backup(const iago_mcts_tree *tree, const uint8_t *active, const int32_t *cur_node,
                     const float *leaf_value, void *stream)
{
    if (check_tree(tree, "iago_mcts_backup: bad tree"))
        return IAGO_ERR_INVALID;
    if (!active || !cur_node || !leaf_value)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_backup: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(backup_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, active, cur_node, leaf_value);
    return iago_check_launch("iago_mcts_backup");
}

int iago_mcts_mix_backup(const iago_mcts_tree *tree, const uint8_t *active, const int32_t *cur_node,
                         const float *v, const int8_t *z, float lmbda, float *leaf_value,
                         uint32_t *counter, void *stream)
{
    if (check_tree(tree, "iago_mcts_mix_backup: bad tree"))
        return IAGO_ERR_INVALID;
    if (!active || !cur_node || !leaf_value || (lmbda < 1.0f && !v) || (lmbda > 0.0f && !z))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_mix_backup: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(mix_backup_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, active, cur_node, v, z, lmbda, leaf_value, counter);
    return iago_check_launch("iago_mcts_mix_backup");
}

int iago_mcts_best_move(const iago_mcts_tree *tree, const uint8_t *active, int8_t *move,
                        int32_t *visits, void *stream)
{
    if (check_tree(tree, "iago_mcts_best_move: bad tree"))
        return IAGO_ERR_INVALID;
    if (!move)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_best_move: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(best_move_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, active, move, visits);
    return iago_check_launch("iago_mcts_best_move");
}

int iago_mcts_advance_root(const iago_mcts_tree *tree, const uint8_t *mask, const int8_t *move,
                           void *stream)
{
    if (check_tree(tree, "iago_mcts_advance_root: bad tree"))
        return IAGO_ERR_INVALID;
    if (!move)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_advance_root: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(advance_root_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                       (hipStream_t)stream, *tree, mask, move);
    return iago_check_launch("iago_mcts_advance_root");
}

int iago_mcts_compact(const iago_mcts_tree *tree, const iago_mcts_tree *scratch, int32_t *order,
                      const uint8_t *mask, void *stream)
{
    if (check_tree(tree, "iago_mcts_compact: bad tree") || check_tree(scratch, "iago_mcts_compact: bad scratch tree"))
        return IAGO_ERR_INVALID;
    if (!order || scratch->n_games != tree->n_games || scratch->capacity != tree->capacity ||
        scratch->nodes == tree->nodes)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_compact: scratch must be a second pool of the same shape");
    if (tree->n_games == 0)
        return IAGO_OK;
    const unsigned games = (unsigned)tree->n_games;
    hipLaunchKernelGGL(compact_plan_kernel, dim3((games + 63) / 64), dim3(64), 0, (hipStream_t)stream, *tree,
                       *scratch, mask, order);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(games), dim3(BLOCK), 0, (hipStream_t)stream, *tree, *scratch,
                       order);
    hipLaunchKernelGGL(compact_commit_kernel, dim3(games), dim3(BLOCK), 0, (hipStream_t)stream, *tree, *scratch);
    return iago_check_launch("iago_mcts_compact");
}

} // extern "C"

namespace {
int lookahead_of(const iago_mcts_lookahead *a, Lookahead &A, const char *who)
{
    if (!a || a->trigger < 1 || a->slots < 1 || a->q_capacity < 1 || !a->next_seq || !a->cache_seq || !a->cache ||
        !a->q_count || !a->q_own || !a->q_opp || !a->q_game || !a->q_seq || !a->error)
        return iago_fail(IAGO_ERR_INVALID, who);
    if ((uintptr_t)a->cache & 15u)
        return iago_fail(IAGO_ERR_INVALID, who);
    A.trigger = a->trigger;
    A.slots = a->slots;
    A.next_seq = a->next_seq;
    A.cache_seq = a->cache_seq;
    A.cache = a->cache;
    A.q_count = a->q_count;
    A.q_capacity = a->q_capacity;
    A.q_own = a->q_own;
    A.q_opp = a->q_opp;
    A.q_game = a->q_game;
    A.q_seq = a->q_seq;
    A.error = a->error;
    A.clear_word = a->clear_word;
    A.path = a->path_stride > 0 ? a->path : nullptr;
    A.path_len = a->path_len;
    A.path_stride = a->path_stride;
    if (A.path && (!A.path_len || a->path_stride < 8))
        return iago_fail(IAGO_ERR_INVALID, who);
    A.z_log = a->z_log_rows > 0 ? a->z_log : nullptr;
    A.z_log_n = a->z_log_n;
    A.z_log_rows = a->z_log_rows;
    if (A.z_log && !A.z_log_n)
        return iago_fail(IAGO_ERR_INVALID, who);
    A.va_x_count = nullptr;
    A.va_x_capacity = 0;
    A.va_x_game = A.va_x_node = nullptr;
    A.va_x_own = A.va_x_opp = nullptr;
    if (const iago_mcts_value_ahead *v = a->value_ahead) {
        if (v->x_capacity < 1 || !v->x_count || !v->x_game || !v->x_node || !v->x_own || !v->x_opp)
            return iago_fail(IAGO_ERR_INVALID, who);
        A.va_x_count = v->x_count;
        A.va_x_capacity = v->x_capacity;
        A.va_x_game = v->x_game;
        A.va_x_node = v->x_node;
        A.va_x_own = v->x_own;
        A.va_x_opp = v->x_opp;
    }
    A.y_wait = nullptr;
    A.y_parts = 0;
    if (const iago_mcts_async *y = a->async) {
        if (y->parts < 2 || y->parts > IAGO_ASYNC_MAX_PARTS || !y->wait || !y->done || !y->roll || !y->fq_index ||
            !y->fq_count || !y->step || !y->n_sims || !A.path)
            return iago_fail(IAGO_ERR_INVALID, who);
        A.y_parts = y->parts;
        A.y_wait = y->wait;
        A.y_done = y->done;
        A.y_roll = y->roll;
        A.y_fq_index = y->fq_index;
        A.y_fq_count = y->fq_count;
        A.y_step = y->step;
        A.y_n_sims = y->n_sims;
    }
    return IAGO_OK;
}
} // namespace

extern "C" {

int iago_mcts_mix_backup_lookahead(const iago_mcts_tree *tree, const uint8_t *active, const int32_t *cur_node,
                                   const uint64_t *cur_own, const uint64_t *cur_opp, const float *v,
                                   const int8_t *z, float lmbda, float *leaf_value, uint32_t *counter,
                                   const iago_mcts_lookahead *la, void *stream)
{
    if (check_tree(tree, "iago_mcts_mix_backup_lookahead: bad tree"))
        return IAGO_ERR_INVALID;
    Lookahead A;
    if (lookahead_of(la, A, "iago_mcts_mix_backup_lookahead: bad look-ahead state"))
        return IAGO_ERR_INVALID;
    if (!active || !cur_node || !cur_own || !cur_opp || !leaf_value || (lmbda < 1.0f && !v) || (lmbda > 0.0f && !z))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_mix_backup_lookahead: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    if (A.path) // the descent recorded the path: 8 lanes per game update its nodes side by side
        hipLaunchKernelGGL(mix_backup_path_kernel, dim3(grid_for(tree->n_games * 8)), dim3(BLOCK), 0,
                           (hipStream_t)stream, *tree, active, cur_node, cur_own, cur_opp, v, z, lmbda, leaf_value,
                           counter, A);
    else
        hipLaunchKernelGGL(mix_backup_lookahead_kernel, dim3(grid_for(tree->n_games)), dim3(BLOCK), 0,
                           (hipStream_t)stream, *tree, active, cur_node, cur_own, cur_opp, v, z, lmbda, leaf_value,
                           counter, A);
    return iago_check_launch("iago_mcts_mix_backup_lookahead");
}

int iago_mcts_store_priors(const iago_mcts_lookahead *la, const float *probs, int64_t *total, void *stream)
{
    Lookahead A;
    if (lookahead_of(la, A, "iago_mcts_store_priors: bad look-ahead state"))
        return IAGO_ERR_INVALID;
    if (!probs || ((uintptr_t)probs & 15u))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_store_priors: probs must be non-null and 16-byte aligned");
    hipLaunchKernelGGL(store_priors_kernel, dim3(64), dim3(BLOCK), 0, (hipStream_t)stream, A, probs, total);
    return iago_check_launch("iago_mcts_store_priors");
}

int iago_mcts_expand_cached(const iago_mcts_tree *tree, const uint8_t *active, const uint8_t *needs_expand,
                            const int32_t *cur_node, const uint64_t *legal, const iago_mcts_lookahead *la,
                            uint8_t *expanded, void *stream)
{
    if (check_tree(tree, "iago_mcts_expand_cached: bad tree"))
        return IAGO_ERR_INVALID;
    Lookahead A;
    if (lookahead_of(la, A, "iago_mcts_expand_cached: bad look-ahead state"))
        return IAGO_ERR_INVALID;
    if (!active || !needs_expand || !cur_node || !legal || !expanded)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_expand_cached: null pointer");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(expand_cached_kernel, dim3(grid_for(tree->n_games * 8)), dim3(BLOCK), 0, (hipStream_t)stream,
                       *tree, active, needs_expand, cur_node, legal, A, expanded);
    return iago_check_launch("iago_mcts_expand_cached");
}

int iago_mcts_fresh_leaves(const iago_mcts_tree *tree, const uint8_t *active, const int32_t *cur_node,
                           int64_t *index, int32_t *count, int64_t *total, void *stream)
{
    if (check_tree(tree, "iago_mcts_fresh_leaves: bad tree"))
        return IAGO_ERR_INVALID;
    if (!tree->has_v || !active || !cur_node || !index || !count)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_fresh_leaves: null pointer (the tree needs its value cache `v`)");
    if (tree->n_games > 0x7fffffffll)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_fresh_leaves: too many games");
    hipLaunchKernelGGL(fresh_leaves_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *tree, active, cur_node,
                       index, count, total);
    return iago_check_launch("iago_mcts_fresh_leaves");
}

int iago_mcts_descend(const iago_mcts_tree *tree, const uint64_t *root_own, const uint64_t *root_opp,
                      const uint8_t *active, float c_puct, int32_t n_thr, int32_t *cur_node, uint64_t *cur_own,
                      uint64_t *cur_opp, uint64_t *legal, int32_t *stats, const iago_mcts_lookahead *la,
                      int64_t *fresh_index, int32_t *fresh_count, int64_t *fresh_total, void *stream)
{
    if (check_tree(tree, "iago_mcts_descend: bad tree"))
        return IAGO_ERR_INVALID;
    Lookahead A;
    if (lookahead_of(la, A, "iago_mcts_descend: bad look-ahead state"))
        return IAGO_ERR_INVALID;
    if (!root_own || !root_opp || !active || !cur_node || !cur_own || !cur_opp || !legal)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_descend: null pointer");
    if (A.y_wait && !tree->has_v)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_descend: game-asynchronous steps need the tree's value cache");
    if (fresh_index && (!fresh_count || !tree->has_v))
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_descend: the fresh-leaf list needs its count word and the "
                                           "tree's value cache");
    if (n_thr < 1)
        return iago_fail(IAGO_ERR_INVALID, "iago_mcts_descend: n_thr must be >= 1");
    if (tree->n_games == 0)
        return IAGO_OK;
    hipLaunchKernelGGL(descend_kernel, dim3(grid_for(tree->n_games * 8)), dim3(BLOCK), 0, (hipStream_t)stream, *tree,
                       root_own, root_opp, active, c_puct, n_thr, cur_node, cur_own, cur_opp, legal, stats, A,
                       fresh_index, fresh_count, fresh_total);
    return iago_check_launch("iago_mcts_descend");
}

} // extern "C"

namespace {
int value_ahead_of(const iago_mcts_tree *tree, const iago_mcts_value_ahead *v, ValueAhead &V, const char *who)
{
    if (!v || v->x_capacity < 1 || v->row_capacity < 1 || !v->x_count || !v->x_game || !v->x_node || !v->x_own ||
        !v->x_opp || !v->row_count || !v->row_own || !v->row_opp || !v->row_node || !v->row_v || !tree->has_v)
        return iago_fail(IAGO_ERR_INVALID, who);
    V.x_capacity = v->x_capacity;
    V.row_capacity = v->row_capacity;
    V.x_count = v->x_count;
    V.x_game = v->x_game;
    V.x_node = v->x_node;
    V.x_own = v->x_own;
    V.x_opp = v->x_opp;
    V.row_count = v->row_count;
    V.row_own = v->row_own;
    V.row_opp = v->row_opp;
    V.row_node = v->row_node;
    V.row_v = v->row_v;
    V.total = v->total;
    return IAGO_OK;
}
} // namespace

extern "C" {

int iago_mcts_value_ahead_rows(const iago_mcts_tree *tree, const iago_mcts_value_ahead *va, void *stream)
{
    if (check_tree(tree, "iago_mcts_value_ahead_rows: bad tree"))
        return IAGO_ERR_INVALID;
    ValueAhead V;
    if (value_ahead_of(tree, va, V, "iago_mcts_value_ahead_rows: incomplete iago_mcts_value_ahead (the tree needs "
                                    "its value cache `v`)"))
        return IAGO_ERR_INVALID;
    // (a fixed small grid: the queue holds a few dozen nodes per playout)
    hipLaunchKernelGGL(value_ahead_rows_kernel, dim3(32), dim3(BLOCK), 0, (hipStream_t)stream, *tree, V);
    hipLaunchKernelGGL(value_ahead_clamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, V);
    return iago_check_launch("iago_mcts_value_ahead_rows");
}

int iago_mcts_value_ahead_store(const iago_mcts_tree *tree, const iago_mcts_value_ahead *va, void *stream)
{
    if (check_tree(tree, "iago_mcts_value_ahead_store: bad tree"))
        return IAGO_ERR_INVALID;
    ValueAhead V;
    if (value_ahead_of(tree, va, V, "iago_mcts_value_ahead_store: incomplete iago_mcts_value_ahead (the tree needs "
                                    "its value cache `v`)"))
        return IAGO_ERR_INVALID;
    hipLaunchKernelGGL(value_ahead_store_kernel, dim3(16), dim3(BLOCK), 0, (hipStream_t)stream, *tree, V);
    return iago_check_launch("iago_mcts_value_ahead_store");
}

} // extern "C"
