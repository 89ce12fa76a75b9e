/*
 * iago_hip_experimental.h -- entry points of two schedules of the PER-PLAYOUT search engine that were built, proven
 * tree-identical to the reference's order of evaluation and MEASURED SLOWER than the engines that serve the path
 * (the persistent search, iago_mcts_search_persistent, wherever the split-f16 Value net and the
 * three-piece SLPolicy apply and the batch fits one launch; the lockstep per-playout launches of include/iago_hip.h
 * otherwise): game-asynchronous steps (round 3) and the value look-ahead (round 4).  They stay in the library so that the
 * measurements of LABNOTES.md can be repeated (engine.BatchedMCTS(async_steps=True) / (value_ahead=True)); nothing on the
 * product's default paths calls them, and they are not part of the drop-in boundary INTEGRATION.md describes.
 * Conventions as in iago_hip.h.
 */
#ifndef IAGO_HIP_EXPERIMENTAL_H
#define IAGO_HIP_EXPERIMENTAL_H

#include "iago_hip.h"

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

#ifdef __cplusplus
}
#endif
#endif /* IAGO_HIP_EXPERIMENTAL_H */
