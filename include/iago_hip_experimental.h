/*
 * iago_hip_experimental.h -- entry points of the library OUTSIDE the drop-in boundary INTEGRATION.md describes:
 *   (1) two schedules of the PER-PLAYOUT search engine -- game-asynchronous steps (round 3) and the value look-ahead (round
 *       4) --, built, proven tree-identical to the reference's order of evaluation and MEASURED SLOWER than the engines that
 *       serve the path (the persistent search wherever the split-f16 Value net and the three-piece
 *       SLPolicy apply and the batch fits one launch, the lockstep per-playout launches of iago_hip.h otherwise).  They stay
 *       so that the measurements of LABNOTES.md can be repeated (engine.BatchedMCTS(async_steps=True) / (value_ahead=True));
 *       nothing on the product's default paths calls them;
 *   (2) the per-phase forms of a playout (rounds 1 - 2), superseded by the one-launch descent / backup of iago_hip.h and by
 *       the persistent search: what engine.BatchedMCTS drives for arbitrary callables as nets.
 * Conventions as in iago_hip.h.
 */
#ifndef IAGO_HIP_EXPERIMENTAL_H
#define IAGO_HIP_EXPERIMENTAL_H

#include "iago_hip.h"
#include "iago_hip_layers.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Game-asynchronous search steps.  MCTS.playout (MCTS.py:105-133) is sequential INSIDE a game --
 * playout i + 1 selects on the statistics playout i backed up -- but the games of a batch are
 * independent, and only ~1 playout in 6 ends on a leaf whose value_func (MCTS.py:97-103) has not
 * been computed yet.  In lockstep every playout of every game waits for those evaluations (one
 * board's walk through the Value net on one CU: ~70 us of a ~125 us playout).  With this state a
 * search advances in STEPS: per step every game that is not waiting descends
 * (iago_mcts_descend); a game whose leaf has a stored value completes its playout in the same step
 * (rollout, iago_mcts_mix_backup_lookahead); a game whose leaf is fresh is queued for the value net
 * and WAITS `parts` steps while the net walks its board in `parts` pieces, one per step's
 * iago_value_rollout_async launch (each piece a few layers, the board's activations parked in
 * `scratch` in between), beside the rollouts of the games that go on; its playout completes in the
 * step its value arrives.  A game's sequence of playouts -- leaves, values, rollouts (Philox stream
 * id = stream base + the game's own playout count), backups, expansions -- is exactly the lockstep
 * one; only the interleaving between games changes: trees are bit-identical.
 * All arrays caller-owned device memory: wait / done [n_games] int32 (zeroed before a search),
 * roll [n_games] uint8, fq_index [parts][n_games], fq_count [parts] (zeroed before a search), step
 * (one word, any value), n_sims (one word: playouts per game of this search), scratch
 * [parts][n_games][IAGO_VALUE_IMAGE_BYTES].
 */
#define IAGO_VALUE_IMAGE_BYTES 34816   /* a board's activations between two pieces: 64 cells x 544 B */
#define IAGO_ASYNC_MAX_PARTS 4
typedef struct iago_mcts_async {
    int32_t parts;           /* pieces of the value net = steps a fresh leaf waits: 2..IAGO_ASYNC_MAX_PARTS */
    int32_t reserved;
    int32_t *wait;           /* steps until the game's value arrives (0 = not waiting) */
    int32_t *done;           /* playouts the game has completed in this search */
    uint8_t *roll;           /* the game descended in this step (its leaf is rolled out) */
    int64_t *fq_index;       /* the games queued at step s in row (s mod parts) */
    int32_t *fq_count;
    uint32_t *step;          /* step counter (iago_mcts_mix_backup_lookahead increments it) */
    const int32_t *n_sims;
    void *scratch;
} iago_mcts_async;

/*
 * The leaf evaluation of one game-asynchronous step in ONE launch: the rollouts
 * (Simulate, mcts_self_play.py:9-134) of the games that descended in this step (async->roll; game
 * g draws from Philox stream rollout->stream_id (+ *stream_id_dev) + async->done[g]) and, for the
 * leaves queued 0 .. parts-1 steps ago, piece 0 .. parts-1 of the Value net (iago_value_forward_split's
 * arithmetic, bit-identical values; piece parts-1 writes value->out[game]).  value->index / n_dev
 * are ignored (the queues of `async` take their place); value->n = n_games.
 */
IAGO_API int iago_value_rollout_async(const iago_value_split_args *value, const iago_rollout_args *rollout,
                                      const iago_mcts_async *async, void *stream);

/*
 * Value look-ahead.  value_func(state) (MCTS.py:97-103) is a pure function of the position, and
 * under Node.select's score (MCTS.py:44-49,75-76: u = c_puct*P*sqrt(N)/(0.01+n) with P >= 0.1) an
 * unvisited child outscores every visited one, so the children of a node that has just expanded
 * are first-visited one after the other during the node's next visits -- each of those visits ends
 * on a leaf without a stored value (the value cache's NaN), i.e. on a one-board walk of the value
 * net on the playouts' critical path.  With this state iago_mcts_descend records every node it
 * expands (x_* queue: game, node, the node's position), iago_mcts_value_ahead_rows turns the
 * queued nodes into one row per child that has no value yet (the child's position, own = side to
 * move there, and the global index of its record), the caller runs the value net on the rows as
 * ONE batch off the critical path (iago_value_forward_batch -> row_v) and
 * iago_mcts_value_ahead_store writes the results into the children's `v` -- exactly the number the
 * first visit would have computed (the net's output for a board does not depend on its batch), so
 * the trees are bit-identical; a child that is visited before its value has landed is evaluated
 * in place as before.  Everything here is a hint: a full queue drops entries, nothing is reported.
 * Arrays caller-owned: x_* [x_capacity], row_* [row_capacity], the count words zeroed by the caller
 * (x_count after iago_mcts_value_ahead_rows has consumed the queue, row_count before it runs).
 * The queues must be drained (rows + net + store) before iago_mcts_compact / iago_mcts_reset /
 * a change of the value net's weights: rows address nodes by pool index.
 */
typedef struct iago_mcts_value_ahead {
    int32_t x_capacity, row_capacity;
    int32_t *x_count;
    int32_t *x_game, *x_node;     /* game and local id of an expanded node */
    uint64_t *x_own, *x_opp;      /* its position, own = side to move (the children's mover is the other side) */
    int32_t *row_count;
    uint64_t *row_own, *row_opp;  /* a child's position, own = side to move at the child */
    int64_t *row_node;            /* game * capacity + local id of the child */
    float *row_v;                 /* the value net's output for the row */
    int64_t *total;               /* optional: rows produced so far (accumulated by iago_mcts_value_ahead_rows) */
} iago_mcts_value_ahead;
IAGO_API int iago_mcts_value_ahead_rows(const iago_mcts_tree *tree, const iago_mcts_value_ahead *va, void *stream);
IAGO_API int iago_mcts_value_ahead_store(const iago_mcts_tree *tree, const iago_mcts_value_ahead *va, void *stream);

/*
 * The same net on a batch whose row count is known only on the device (args->n_dev required,
 * args->n = the buffers' capacity), for work OFF the playouts' critical path (the value
 * look-ahead): `boards_per_workgroup` (1, 2 or 4) boards share a workgroup's weight stream and at
 * most `max_workgroups` workgroups walk the rows with the grid's stride, so that the launch leaves
 * the rest of the chip to the kernels of the playouts beside it.  Same products in the same order
 * per board as iago_value_forward_split: bit-identical values.
 */
IAGO_API int iago_value_forward_batch(const iago_value_split_args *args, int32_t boards_per_workgroup,
                                      int32_t max_workgroups, void *stream);

/*
 * ---- The per-phase forms of a playout (rounds 1 - 2): one launch per phase of MCTS.playout (MCTS.py:105-133).  Superseded
 * twice -- by iago_mcts_descend + iago_mcts_mix_backup_lookahead (one-launch descent and backup, iago_hip.h) and by the
 * persistent search --; they remain what engine.BatchedMCTS drives for ARBITRARY callables as nets (one host
 * synchronisation per playout: the stand-in nets of the parity tests) and behind the IAGO_FUSED_* = 0 knobs.
 */
/*
 * Descend from the cursor to a leaf with Node.select (MCTS.py:39-49,75-76):
 * child score = Q + c_puct*P*sqrt(parent.n)/(0.01+n), argmax, first wins;
 * apply the chosen move to the cursor board (GameFunctions.place_stone,
 * game.py:180-207; -1 = pass) and switch sides (MCTS.py:130-133).
 * This is the non-leaf branch of MCTS.playout (MCTS.py:129-133).
 *   from_root != 0: the cursor starts at the game's root with board
 *     (root_own, root_opp) (own = side to move at the root);
 *   from_root == 0: it continues from (cur_node, cur_own, cur_opp) -- used after
 *     an expansion, mirroring the recursion of MCTS.py:121.
 * Games with active[g] == 0 are skipped.  On return, per active game:
 *   cur_node/cur_own/cur_opp: the leaf and its position (own = side to move),
 *   needs_expand: 1 iff leaf.n_visits >= n_thr (MCTS.py:109),
 *   legal: the leaf's legal-move mask (game.py:210-235) when needs_expand.
 * stats (optional, int32 [n_games][2]): ACCUMULATES per game the levels
 * descended and the children scored by this call -- the harness turns them
 * into the algorithmic bytes of the tree arrays.
 */
IAGO_API int iago_mcts_select(const iago_mcts_tree *tree, const uint64_t *root_own,
                              const uint64_t *root_opp, const uint8_t *active, float c_puct,
                              int32_t n_thr, int from_root, int32_t *cur_node, uint64_t *cur_own,
                              uint64_t *cur_opp, uint8_t *needs_expand, uint64_t *legal,
                              int32_t *stats, void *stream);

/*
 * Expand the leaves listed in `games` (int32 game ids, n_expand of them):
 * MCTS.playout's expansion branch (MCTS.py:110-120) + Node.expand
 * (MCTS.py:27-37).  0 legal moves: one pass child (-1) with prior 1; exactly
 * one: that child with prior 1 (no net); otherwise one child per legal move,
 * ascending, with prior probs[i][a] (raw softmax entry, not renormalised,
 * MCTS.py:96-98).  probs: float32 [n_expand][64], row i belongs to games[i]
 * (rows of single-move / pass leaves are ignored and may be garbage).
 * A game whose pool is full gets overflow[g] = 1 and is left unexpanded.
 * n_dev: optional device-side count (see iago_encode_planes_indexed): min(n_expand, *n_dev)
 * leaves are expanded.
 */
IAGO_API int iago_mcts_expand(const iago_mcts_tree *tree, const int32_t *games, int64_t n_expand,
                              const int32_t *cur_node, const uint64_t *legal, const float *probs,
                              const int32_t *n_dev, void *stream);

/*
 * The games a playout has to expand before it can go on (MCTS.py:109: the leaf reached
 * n_thr visits): pending[g] = needs_expand[g] && active[g] (0/1), their ids in ascending
 * order as index[] (int64) and games[] (int32, what iago_mcts_expand takes), *count =
 * how many.  index / games must hold n entries.  One small launch in place of a mask,
 * a stream compaction and a type conversion.  total (optional): a device int64 that
 * accumulates the counts (the number of policy evaluations of a search, read once at its end).
 */
IAGO_API int iago_mcts_pending(const uint8_t *needs_expand, const uint8_t *active, int64_t n,
                               uint8_t *pending, int64_t *index, int32_t *games, int32_t *count,
                               int64_t *total, void *stream);

/*
 * Node.update_recursive (MCTS.py:51-72) from cur_node up to the root of every
 * active game: n += 1; Q += (leaf_value - Q)/n; the SAME value at every
 * level (the reference does not flip the sign).
 */
IAGO_API int iago_mcts_backup(const iago_mcts_tree *tree, const uint8_t *active,
                              const int32_t *cur_node, const float *leaf_value, void *stream);

/*
 * iago_leaf_values + iago_mcts_backup in one launch (MCTS.py:123-127): leaf_value[g] =
 * (1-lmbda)*v[g] + lmbda*z[g] for every game, backed up along the path of the active
 * ones.  counter: optional device word incremented by one (the playout number that
 * iago_rollout_args.stream_id_dev reads when the playouts replay from a hipGraph).
 */
IAGO_API int iago_mcts_mix_backup(const iago_mcts_tree *tree, const uint8_t *active,
                                  const int32_t *cur_node, const float *v, const int8_t *z, float lmbda,
                                  float *leaf_value, uint32_t *counter, void *stream);

/* iago_mcts_expand for every active game with needs_expand, priors from the look-ahead's cache (iago_mcts_lookahead,
 * iago_hip.h); expanded[g] = 1 for those games, else 0 */
IAGO_API int iago_mcts_expand_cached(const iago_mcts_tree *tree, const uint8_t *active, const uint8_t *needs_expand,
                                     const int32_t *cur_node, const uint64_t *legal, const iago_mcts_lookahead *la,
                                     uint8_t *expanded, void *stream);

/* the active games whose leaf (cur_node) has no stored value yet: index[0 .. *count), ascending; *total += *count when
 * given (the value cache of iago_hip.h; iago_mcts_descend lists them in the same launch as the descent) */
IAGO_API int iago_mcts_fresh_leaves(const iago_mcts_tree *tree, const uint8_t *active, const int32_t *cur_node,
                                    int64_t *index, int32_t *count, int64_t *total, void *stream);



#ifdef __cplusplus
}
#endif
#endif /* IAGO_HIP_EXPERIMENTAL_H */
