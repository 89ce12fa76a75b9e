/*
 * iago_hip.h -- C ABI of the MI355X (gfx950) Othello self-play hot path: the drop-in boundary for the hot path of
 * shionhonda/IaGo (board rules, plane encoding, leaf rollout, the nets, PV-MCTS, the self-play drivers, the REINFORCE
 * update).  The reference has no FFI layer: the path is in-process Python method calls on ONE (8,8) float32 board.  Each
 * entry point below is the batched restatement of one of those methods and cites the reference interface it replaces
 * (file:line in the reference repository); INTEGRATION.md shows the ctypes stubs a maintainer of the reference would add.
 * Sibling headers of the same library: iago_hip_layers.h (single layers of the nets, format conversions),
 * iago_hip_experimental.h (superseded and measured-slower forms, outside the boundary).
 *
 * Conventions
 *   - Boards are bitboards: one 64-bit word per colour, bit a = row*8 + col (the reference's action index,
 *     game.py:184).  `own` holds the stones of the side to move, `opp` the other side's.
 *   - Every pointer is a CALLER-OWNED DEVICE pointer (hipMalloc / a torch tensor's data_ptr()) unless the parameter is
 *     documented as host memory.  `stream` is a hipStream_t passed as void*; NULL = the default stream.  Calls only
 *     enqueue work: no host synchronisation, no allocation, no internal threads; re-entrant.
 *   - Return value: IAGO_OK (0) or a negative iago_status; nothing is thrown across the ABI.  iago_last_error() returns a
 *     thread-local message.  There is NO CPU fallback: without a HIP device every launch fails with IAGO_ERR_HIP.
 */
#ifndef IAGO_HIP_H
#define IAGO_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define IAGO_API __attribute__((visibility("default")))
#else
#define IAGO_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum iago_status {
    IAGO_OK = 0,
    IAGO_ERR_INVALID = -1, /* bad argument (null pointer, negative size, bad option) */
    IAGO_ERR_HIP = -2,     /* a HIP runtime call failed; see iago_last_error() */
    IAGO_ERR_CAPACITY = -3 /* a caller-provided pool/buffer is too small */
} iago_status;

#define IAGO_PASS (-1)          /* pass action, game.py:181 */
#define IAGO_TRACE_PASS 0xFF    /* pass marker in uint8 action traces */
#define IAGO_MAX_TURNS 128      /* upper bound on turns of one game (<= 124) */
#define IAGO_ROLLOUT_TABLE_FLOATS (3 * 2 * 256 * 8 + 64 + 4 + 2 * 512)

IAGO_API int iago_abi_version(void);
IAGO_API const char *iago_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host); never fails. */
IAGO_API int iago_device_count(void);

/* ------------------------------------------------------------------ rules */

/*
 * legal[b] = bit mask of the legal moves of `own` on board b.  Replaces GameFunctions.legal_actions(state, color)
 * (game.py:210-235), GameEnv.valid_pos(color) (rl_env.py:114-138), Simulate.legal_actions (mcts_self_play.py:64-89),
 * rl_self_play.Game.legal_actions (src/rl_self_play.py:63-88).  The reference's ascending action list is the ascending
 * set-bit order of the mask.
 */
IAGO_API int iago_legal_moves(const uint64_t *own, const uint64_t *opp, uint64_t *legal, int64_t n, void *stream);

/*
 * In place: own[b] |= bit(action[b]) | flips, opp[b] &= ~(flips | bit), where flips are the opponent runs bracketed from
 * action[b] in the 8 directions.  action -1 (IAGO_PASS) leaves the board unchanged.  Like the reference, NO legality
 * check: an illegal or occupied target is overwritten and whatever it brackets is flipped.  The side to move is NOT
 * switched (the caller swaps own/opp).  Replaces GameFunctions.place_stone(state, action, color) (game.py:180-207),
 * GameEnv.place_stone (rl_env.py:88-112), mcts_self_play.py:36-62, src/rl_self_play.py:36-61.   action: int8[n].
 */
IAGO_API int iago_apply_moves(uint64_t *own, uint64_t *opp, const int8_t *action, int64_t n, void *stream);

/*
 * One turn of the reference's game loops for n lockstep games -- Game.turn inside Game.__call__
 * (src/rl_self_play.py:130-145,27-31; game.py:117-142,253-255; mcts_self_play.py:124-134,25-29;
 * GameEnv.step's bookkeeping rl_env.py:41-74) -- in one launch:
 *   game b's mover places action[b] where active[b] != 0 (iago_apply_moves; otherwise it has no
 *   move and passes); stone_num += 1 per stone placed; a pass directly after a pass sets
 *   stone_num = 64; pass_flg = "this turn was a pass" (games already done keep theirs);
 *   with close_pair != 0 (the second turn of a `while stone_num < 64` pair) done |= stone_num >= 64;
 *   the sides swap IN MEMORY (own = the next mover's stones on return);
 *   legal[b] = the next mover's legal moves (0 for a game that is done), active_next[b] =
 *   legal[b] != 0 (the `active` of the next call).
 * stone_num int32, pass_flg / done / active / active_next uint8, action int8 (-1 = pass); active and
 * active_next may not alias.
 */
IAGO_API int iago_play_turn(uint64_t *own, uint64_t *opp, const int8_t *action, const uint8_t *active,
                            int32_t *stone_num, uint8_t *pass_flg, uint8_t *done, int close_pair,
                            uint64_t *legal, uint8_t *active_next, int64_t n, void *stream);

/*
 * planes: float32 (n,2,8,8) NCHW; channel 0 = opp (the opponent of the side to move), channel 1 = own.  Replaces
 * GameFunctions.make_state_var(state, color) (game.py:168-174; copies mcts_self_play.py:91-97,
 * src/rl_self_play.py:102-108).  The un-swapped observation of GameEnv (rl_env.py:36-38,70-72: [state==1, state==2]) is
 * the same call with own = player-2 stones and opp = player-1 stones.
 */
IAGO_API int iago_encode_planes(const uint64_t *own, const uint64_t *opp, float *planes, int64_t n, void *stream);
/* The same for a gather list: row b of `planes` encodes board index[b] (int64; each in
 * [0, number of boards)): the planes of the few games a playout expands (MCTS.py:109-113).
 *
 * Device-side counts.  Several entry points of the expansion path take `n_dev`, an optional
 * device word (NULL = absent): the call then processes the first min(n, *n_dev) items, n
 * being the capacity of the buffers (it sizes the grid).  The number of leaves a playout
 * expands is only known on the device (iago_mcts_pending writes it); with n_dev the whole
 * playout -- select, policy net, expand, value net, rollout, backup -- is a fixed sequence
 * of launches with no host synchronisation, i.e. one hipGraph replay. */
IAGO_API int iago_encode_planes_indexed(const uint64_t *own, const uint64_t *opp, const int64_t *index,
                                        float *planes, int64_t n, const int32_t *n_dev, void *stream);

/* z[b] = sign(popcount(own) - popcount(opp)) as int8.  Replaces Simulate.judge(color) (mcts_self_play.py:113-121),
 * GameEnv.judge (rl_env.py:141-149), rl_self_play.Game.judge (src/rl_self_play.py:91-100). */
IAGO_API int iago_judge(const uint64_t *own, const uint64_t *opp, int8_t *z, int64_t n, void *stream);

/*
 * Masked sampling of one move per board from a policy's probabilities: Simulate.get_action /
 * rl_self_play.Game.get_action (mcts_self_play.py:100-106, src/rl_self_play.py:111-122): p = prob(float32) *
 * valid(float64 0/1), normalised by its float64 sum, then numpy.random.choice = inverse CDF (cumsum / last, first index
 * with cdf > u).  Arithmetic is float64 in cell order, bit-identical to oracle/othello_oracle.c (orc_masked_probs +
 * orc_choice_cdf).  probs: float32 [n][64]; legal: masks from iago_legal_moves; action: int8 [n], -1 where legal == 0.
 * u per board: uniforms[b] (float64, optional) or the Philox draw (id_base + b, step>>2, stream_id, 0)[step&3] >> 8
 * scaled to [0,1).
 */
IAGO_API int iago_sample_moves(const float *probs, const uint64_t *legal, const double *uniforms,
                               uint64_t seed, uint32_t id_base, uint32_t step, uint32_t stream_id,
                               int8_t *action, int64_t n, void *stream);

/*
 * The 8-fold dihedral augmentation of a (position, action) data set, in the
 * reference's order (load.py:56-74): variant 0 = identity, 1..3 = successive
 * np.rot90 (counter-clockwise, (y,x) -> (7-x, y), load.rotate load.py:12-16),
 * 4 = transpose of variant 3 (load.transpose load.py:18-22), 5..7 = three more
 * rotations.  own/opp/action: [n]; outputs [8][n]; action -1 stays -1.
 */
IAGO_API int iago_augment8(const uint64_t *own, const uint64_t *opp, const int8_t *action,
                           uint64_t *own_out, uint64_t *opp_out, int8_t *action_out, int64_t n,
                           void *stream);

/* ---------------------------------------------------------------- rollout */

/*
 * Host helper: expand RolloutPolicy parameters (network.py:49-64; conv1/W
 * (1,2,3,3) as 18 floats, channel 0 = opponent plane, channel 1 = side to
 * move; bias2/b as 64 floats) into the table blob the rollout kernel stages in
 * LDS.  w18, b64, blob are HOST pointers; blob has IAGO_ROLLOUT_TABLE_FLOATS
 * floats and is then copied to the device by the caller.  w18 == NULL builds
 * the uniform policy (every legal move equally likely).
 * Layout: E[3 ky][2 plane][2 half][256 row byte][4] row-pattern contributions
 * to 8 adjacent outputs; bias[64]; mode[4]; CT[2 plane][512] per-cell contributions
 * indexed by the 3x3 neighbourhood pattern (used by the lane-per-board kernel).  mode[0] == 1: PRODUCT form (E and bias hold exp() of the
 * contributions, each factor shifted by its own maximum so that all lie in (0,1]), chosen
 * when the logit range is < 60 so that no partial product leaves float32's
 * range; mode[0] == 0: LOG form (raw sums; the kernel does max / exp2).  Pass
 * log_form = (mode[0] == 0) to iago_rollout.
 */
IAGO_API int iago_rollout_build_table(const float *w18, const float *b64, float *blob);

typedef struct iago_rollout_args {
    const uint64_t *own;     /* [n] side to move at the leaf */
    const uint64_t *opp;     /* [n] */
    int64_t n;
    const float *table;      /* device copy of the blob from iago_rollout_build_table */
    const float *uniforms;   /* optional device [IAGO_MAX_TURNS][n] float32 in [0,1); NULL = Philox */
    uint64_t seed;           /* Philox4x32-10 key */
    uint32_t id_base;        /* rollout b draws from counter (id_base + b, turn/4, stream, 0) */
    uint32_t stream_id;
    const uint32_t *stream_id_dev; /* optional device word added to stream_id when the kernel
                                starts: lets a captured hipGraph replay the launch with a
                                new Philox stream (the caller bumps the word on the stream) */
    int8_t *z;               /* [n] result from the leaf side-to-move's view */
    uint64_t *final_own;     /* optional [n]: final stones of the leaf side to move */
    uint64_t *final_opp;     /* optional [n] */
    uint8_t *n_turns;        /* optional [n]: turns played (passes included) */
    uint8_t *trace;          /* optional [IAGO_MAX_TURNS][n]: action per turn, 0xFF = pass */
    int log_form;            /* 0: blob is in PRODUCT form, 1: LOG form (see above) */
    int throughput_hint;     /* kernel choice (all three play the same games up to float32
                                rounding at CDF boundaries).  0: automatic -- n >= 32768: the
                                lane-per-board kernel (least work per board, needs ~10^5
                                boards in flight); below: half a wave per board (lowest latency
                                of one launch; product form).  1: lane per board (the caller
                                keeps the chip full with other launches).  2: 8 lanes per
                                board (the round-1 kernel; also serves the log form) */
} iago_rollout_args;

/*
 * Play every board to the end with the rollout policy and report the result.
 * Replaces Simulate(state)(color) (mcts_self_play.py:9-29,100-134) -- rules,
 * RolloutPolicy forward (network.py:59-64), masked sampling
 * (mcts_self_play.py:100-106: prob*valid renormalised, numpy choice =
 * inverse CDF, first index with cdf > u), pass / double-pass / full-board
 * termination in the reference's paired-turn loop, judge.
 */
IAGO_API int iago_rollout(const iago_rollout_args *args, void *stream);


/* ------------------------------------------------------------------- nets */

/*
 * The nets of network.py as whole-net launches.  Arithmetic, operand formats ("split f16": a float32 operand as two or
 * three f16 pieces, float32 accumulation; weights [cin/16][3][3][cout][16]) and the `overflow` word (set to 1 when an
 * activation left the f16 range or was NaN before the clamp: the call's results are saturated, not the reference's) are
 * described with the layer-level entry points in iago_hip_layers.h (iago_conv3x3_split).
 */
/*
 * The WHOLE SLPolicy net (SLPolicy.__call__, network.py:15-47: 8 x [conv3x3 + bias + ReLU], conv
 * 1x1 128 -> 1, per-cell bias, softmax) in ONE launch on the f16 matrix units with float32-exact
 * products: every float32 operand as three f16 pieces (33 bits), six MFMAs per product sum into
 * three float32 accumulators -- a float32 convolution with its own summation order at 2.7x the
 * float32 MFMA rate (the policy's distribution must hold 1e-5 on a near one-hot net: the
 * two-piece split of the Value net is not enough).  Activations stay in LDS from block1 to the
 * head, one board per workgroup.  For the batches a search evaluates the policy on.
 *   own / opp: the boards (own = side to move; iago_encode_planes fused in), row b = board
 *   index[b] when a gather list is given; n_dev: optional device-side row count.
 *   w_hi / w_mid / w_lo[k]: block 2+k as three f16 tensors [cin/16][3][3][128][16]
 *   (hi = f16(w), mid = f16((w - hi) 2^11), lo = f16(((w - hi) 2^11 - mid) 2^11)); w1 [64][2][3][3],
 *   b1 [64], bias[k] [128], w9 [128], b10 [64] (w1, b1, w9 and every weight piece 16-byte aligned); probs [n][64].
 *   overflow: see iago_conv3x3_split.
 */
typedef struct iago_policy_split3_args {
    const uint64_t *own, *opp;
    const int64_t *index;
    const int32_t *n_dev;
    int64_t n;
    const float *w1, *b1;
    const void *w_hi[7], *w_mid[7], *w_lo[7];
    const float *bias[7];
    const float *w9, *b10;
    float *probs;
    uint32_t *overflow;
    int32_t parts;          /* 0 / 1: one launch; 2..7: the 7 blocks as that many launches of 7 / parts blocks */
    int32_t scratch_rows;   /* parts > 1: rows the scratch holds; 0 = n.  A batch of more rows runs as chunks of
                               scratch_rows rows (each chunk its `parts` launches), so the scratch stays bounded
                               whatever n is */
    void *scratch;          /* parts > 1: [scratch_rows or n][51,200] bytes, a board's activations between the
                               launches */
} iago_policy_split3_args;
IAGO_API int iago_policy_forward_split3(const iago_policy_split3_args *args, void *stream);
/*
 * The WHOLE Value net (Value.__call__(x, train=False), network.py:66-96, as MCTS.playout calls
 * it, MCTS.py:123-124) in ONE launch: iago_value_stem(_boards) + the 7-layer
 * iago_conv3x3_split_trunk + iago_value_head with the activations resident in LDS from block1
 * to fc11 -- no intermediate tensor is written.  Same arithmetic as the three calls (block1 in
 * float32, blocks 2..8 in split-f16 MFMAs, bit-identical to them; block9 as a 9-row MFMA
 * product in the same split arithmetic, fc10 / fc11 in float32).
 *   planes: float32 [n][2][8][8], or NULL: the boards themselves in own / opp (own = side to
 *   move; iago_encode_planes fused in).  w_hi/w_lo/bias[k]: block 2+k as in iago_conv3x3_split
 *   (w_hi[0] has cin = 64).  w9_hi/w9_lo: block9's weight as MFMA operand
 *   [8 chunks of 16 input channels][32 rows][16] f16, row r < 9 = kernel tap r (ky*3+kx),
 *   rows 9..31 zero, split like every weight (hi = f16(w), lo = f16((w - hi) * 2^11)).
 *   b9 [1], w10 [128][64], w11 [1][128], out [n].  overflow: see iago_conv3x3_split.
 * Up to 512 rows a workgroup takes two boards instead of four (twice the workgroups, half the
 * latency of each; the same products in the same order per board: bit-identical values); with
 * a device-side count both variants are enqueued and the one the count selects runs.
 */
typedef struct iago_value_split_args {
    const float *planes;
    const uint64_t *own, *opp;
    int64_t n;
    const float *w1, *b1;            /* 16-byte aligned */
    const void *w_hi[7], *w_lo[7];
    const float *bias[7];
    const void *w9_hi, *w9_lo;
    const float *b9, *w10, *w11;
    float *out;
    uint32_t *overflow;
    const int64_t *index;  /* optional gather list (boards only): row b evaluates board index[b] of own / opp
                              and writes out[index[b]] -- the leaves of a playout that need the net */
    const int32_t *n_dev;  /* optional device-side row count: only the first min(n, *n_dev) rows */
} iago_value_split_args;
IAGO_API int iago_value_forward_split(const iago_value_split_args *args, void *stream);

/*
 * The leaf evaluation of a playout (MCTS.py:123-125) in ONE launch: iago_rollout of all leaves
 * (product form, no trace / recorded uniforms) and iago_value_forward_split on the leaves of a
 * device-side list (args->index, args->n_dev: the leaves without a stored value,
 * iago_mcts_fresh_leaves / iago_mcts_descend) as two kinds of workgroups of one grid.  Same
 * results as the two calls.
 */
IAGO_API int iago_value_rollout(const iago_value_split_args *args, const iago_rollout_args *rollout,
                                void *stream);

/* ------------------------------------------------------------------- MCTS */

/*
 * Device-resident search trees of `n_games` lockstep games.  Game g owns nodes
 * [g*capacity, (g+1)*capacity) of ONE array of 32-byte records; node ids stored in the records are
 * LOCAL to the game (0 .. capacity-1).  Replaces the reference's dict-of-Node tree
 * (MCTS.py:10-76): parent / children{action: Node} / n_visits / Q / P.  The children of a node are
 * stored contiguously in ascending action order (the reference's dict insertion order,
 * MCTS.py:34-36), so "first maximum wins" (MCTS.py:46,147) is "lowest child index wins".
 * dtypes follow the reference under numpy >= 2: Q, P float32; scores float64.
 * Layout (round 3; struct-of-arrays before): what scoring a child reads -- n_visits, Q, P and the
 * stored value -- is the first 16 bytes of its record, the links the descent carries along
 * (first_child, action, n_children) the second 16: one 32-byte sector per child instead of
 * one sector in each of seven arrays.
 */
typedef struct iago_mcts_node {
    int32_t n_visits;      /* MCTS.py:14 */
    float q;               /* MCTS.py:15,63 */
    float p;               /* prior + 0.1 (MCTS.py:19) */
    float v;               /* value_func(node) once evaluated, NaN before (the value cache, see
                              iago_mcts_fresh_leaves); maintained only when the tree's has_v is set */
    int32_t first_child;   /* local id of child 0, -1 = leaf (MCTS.py:24-25), <= -2: look-ahead tag */
    int32_t parent;        /* local id, -1 = root (MCTS.py:12,21) */
    int8_t action;         /* move leading to the node; -1 = pass child (MCTS.py:114) */
    uint8_t n_children;
    uint16_t reserved0;
    int32_t reserved1;
} iago_mcts_node;

typedef struct iago_mcts_tree {
    int64_t n_games;
    int32_t capacity;      /* nodes per game */
    int32_t has_v;         /* 1: the value cache (node.v) is in use */
    iago_mcts_node *nodes; /* [n_games*capacity], 32-byte aligned */
    int32_t *n_nodes;      /* [n_games] nodes allocated so far */
    int32_t *root;         /* [n_games] local id of the current root */
    int32_t *overflow;     /* [n_games] set to 1 when an expansion did not fit */
} iago_mcts_tree;

/*
 * (Re)initialise every game flagged in `mask` (NULL = all) to a single fresh
 * root Node(None, 1.0) (MCTS.py:81,154): n=0, Q=0, P=1.1.
 */
IAGO_API int iago_mcts_reset(const iago_mcts_tree *tree, const uint8_t *mask, void *stream);

/*
 * leaf_value = (1-lmbda)*v + lmbda*z in the reference's float32 arithmetic
 * (MCTS.py:123-125); v may be NULL when lmbda >= 1, z when lmbda <= 0.
 */
IAGO_API int iago_leaf_values(const float *v, const int8_t *z, float lmbda, float *leaf_value,
                              int64_t n, void *stream);

/*
 * MCTS.get_move's final choice (MCTS.py:147): the most visited child of the
 * root, first wins.  move[g] = action (int8, -1 = pass) or -2 if the root has
 * no children; visits (optional, int32 [n_games][64]) receives the root
 * children's visit counts by action (pass is not recorded), zero elsewhere --
 * the build's pi target, which the reference does not have.
 */
IAGO_API int iago_mcts_best_move(const iago_mcts_tree *tree, const uint8_t *active, int8_t *move,
                                 int32_t *visits, void *stream);

/*
 * MCTS.update_with_move (MCTS.py:149-154) for every game with mask[g] != 0:
 * the child reached by move[g] becomes the root (its parent link is cut), or,
 * if the root has no such child, the game's pool is reset to a fresh root.
 */
IAGO_API int iago_mcts_advance_root(const iago_mcts_tree *tree, const uint8_t *mask,
                                    const int8_t *move, void *stream);

/*
 * Garbage collection of the node pools.  iago_mcts_advance_root keeps the chosen child's
 * subtree and leaves its siblings' nodes behind in the pool (the reference drops its last
 * reference to them, MCTS.py:149-152, and Python frees them).  This call re-lays the live
 * subtree of every game with mask[g] != 0 (NULL = all) in breadth-first order from index 0:
 * children stay contiguous and in their order, statistics are copied unchanged, root = 0,
 * n_nodes = the number of live nodes -- a search behaves exactly as on the uncompacted pool.
 * scratch: a second pool of the same shape (temporary copy target), order: int32
 * [n_games*capacity] (temporary).  Call between searches (no cursor may be in flight).
 */
IAGO_API int iago_mcts_compact(const iago_mcts_tree *tree, const iago_mcts_tree *scratch, int32_t *order,
                               const uint8_t *mask, void *stream);

/*
 * Policy look-ahead.  The reference evaluates the policy net at the visit that expands a leaf
 * (MCTS.policy_func inside MCTS.playout, MCTS.py:93-96,109-121) -- for lockstep games a few
 * positions per playout, nine latency-bound launches on every playout's critical path.
 * policy_func(state) depends on nothing but the leaf's position, so the engine evaluates it
 * EARLIER: a leaf is queued when its visit count reaches `trigger` = n_thr - K, the queue is
 * flushed through the net every K playouts as one batch, the priors wait in a per-game cache
 * and the expansion -- at the same visit and with the same values as in the reference --
 * takes them from there.  Trees are bit-identical to those of iago_mcts_expand.
 *   iago_mcts_mix_backup_lookahead: iago_mcts_mix_backup + queueing of the playout's leaf
 *     (its position from cur_own / cur_opp) when its visit count reaches `trigger`;
 *   iago_mcts_store_priors: rows 0 .. *q_count-1 of probs (the net's outputs for q_own / q_opp
 *     in queue order) into the cache; *total += *q_count when given.  The caller zeroes
 *     *q_count afterwards;
 *   iago_mcts_expand_cached: iago_mcts_expand for every active game with needs_expand, priors
 *     from the cache; expanded[g] = 1 for those games, else 0.
 * An unexpanded leaf carries its cache tag in first_child (-1 not queued, -2 - seq queued).
 * *error is raised (1: queue full, 2: a leaf reached n_thr without valid priors -- its slot
 * was recycled (`slots` too small) or it never crossed the trigger).  All arrays caller-owned:
 * next_seq [n_games] (zeroed with the trees), cache_seq [n_games][slots] (-1), cache
 * [n_games][slots][64] float32, the queue arrays [q_capacity].
 */


/* (optional extensions of the per-playout engine, measured slower and kept for reproduction only: their state structs and
   entry points are in include/iago_hip_experimental.h) */
struct iago_mcts_async;
struct iago_mcts_value_ahead;
typedef struct iago_mcts_lookahead {
    int32_t trigger, slots;
    int32_t *next_seq;
    int32_t *cache_seq;
    float *cache;
    int32_t *q_count;
    int32_t q_capacity;
    int32_t path_stride;   /* entries per game of `path` (>= 8), 0 = no path */
    uint64_t *q_own, *q_opp;
    int32_t *q_game, *q_seq;
    int32_t *error;
    int32_t *clear_word;   /* optional device word iago_mcts_mix_backup_lookahead sets to 0 (the fresh-leaf
                              count that the next iago_mcts_descend appends to) */
    int32_t *path;         /* optional [n_games][path_stride]: iago_mcts_descend records the nodes it visits, root
                              first (Node.update_recursive's ancestors, MCTS.py:66-72, in reverse), and their
                              number in path_len [n_games]; iago_mcts_mix_backup_lookahead then updates them side
                              by side instead of climbing through `parent`.  A descent deeper than path_stride
                              sets the game's `overflow` */
    int32_t *path_len;
    int8_t *z_log;         /* optional diagnostic record [z_log_rows][n_games] (z_log_rows = 0: none):
                              iago_mcts_mix_backup_lookahead stores the rollout result z it mixed into the leaf
                              value (MCTS.py:124-125) of game g's k-th playout since z_log_n[g] was zeroed in row
                              k, and counts in z_log_n [n_games] -- what a replay of the search through the
                              reference's MCTS.playout needs from Simulate (tests/test_mcts_production_gpu.py) */
    int32_t *z_log_n;
    int32_t z_log_rows;
    int32_t reserved;
    const struct iago_mcts_async *async; /* optional: iago_mcts_descend and iago_mcts_mix_backup_lookahead run one
                                     game-asynchronous STEP instead of one lockstep playout (`active` is then
                                     the search's mask; `counter` and `clear_word` are not used) */
    const struct iago_mcts_value_ahead *value_ahead; /* optional: iago_mcts_descend queues every node it expands
                                     for the value look-ahead (below) */
} iago_mcts_lookahead;
IAGO_API int iago_mcts_mix_backup_lookahead(const iago_mcts_tree *tree, const uint8_t *active,
                                            const int32_t *cur_node, const uint64_t *cur_own,
                                            const uint64_t *cur_opp, const float *v, const int8_t *z, float lmbda,
                                            float *leaf_value, uint32_t *counter,
                                            const iago_mcts_lookahead *la, void *stream);
IAGO_API int iago_mcts_store_priors(const iago_mcts_lookahead *la, const float *probs, int64_t *total,
                                    void *stream);

/*
 * Value cache.  MCTS.playout evaluates value_func(state) at every visit of a leaf
 * (MCTS.py:97-103,123-124) -- n_thr times for a leaf that goes on to expand, with the same
 * result every time: the net is a pure function of the position (the rollout, MCTS.py:125, is
 * not, and runs every time).  With `v` in the tree the value of a node is computed at its first
 * visit only: iago_mcts_fresh_leaves lists the active games whose leaf (cur_node) has no value
 * yet (index[0 .. *count), ascending; *total += *count when given), the value net runs on
 * those boards (iago_value_forward_split with index / n_dev: values to v_out[game]), and
 * iago_mcts_mix_backup(_lookahead) stores the new values into the tree and takes the stored
 * one for every other leaf.  ~85 % of a search's value evaluations go away; the trees are
 * bit-identical (the net's output for a board does not depend on its batch).  iago_mcts_reset,
 * expansion and iago_mcts_advance_root mark new nodes as not evaluated; after a change of the
 * value net's weights the caller fills `v` with NaN.
 */
/*
 * The whole descent of a playout in one launch: iago_mcts_select from the root, for the games
 * whose leaf has n_visits >= n_thr iago_mcts_expand_cached and the continued iago_mcts_select
 * into the new children (MCTS.py:105-121,129-133), and -- when fresh_index is given -- the list
 * of iago_mcts_fresh_leaves, appended through *fresh_count (which the caller, or the previous
 * iago_mcts_mix_backup_lookahead through `clear_word`, has set to 0); the list's order is not
 * defined.  Same arithmetic as the separate calls: same trees.
 */
IAGO_API int iago_mcts_descend(const iago_mcts_tree *tree, const uint64_t *root_own, const uint64_t *root_opp,
                               const uint8_t *active, float c_puct, int32_t n_thr, int32_t *cur_node,
                               uint64_t *cur_own, uint64_t *cur_opp, uint64_t *legal, int32_t *stats,
                               const iago_mcts_lookahead *la, int64_t *fresh_index, int32_t *fresh_count,
                               int64_t *fresh_total, void *stream);


/*
 * A whole search -- n_sims playouts of every active game, the loop of MCTS.get_move (MCTS.py:139-147)
 * around MCTS.playout (MCTS.py:105-133) -- as ONE persistent launch in which every game runs on its
 * own clock (csrc/search_kernel.hip).  The first ceil(n_games / IAGO_SEARCH_GAMES_PER_WORKGROUP) workgroups each own that many games and
 * loop over iago_mcts_descend's descent, iago_rollout's leaf rollout (Philox stream = rollout->stream_id
 * (+ *stream_id_dev) + the game's own playout count) and iago_mcts_mix_backup's backup for them; a game
 * whose leaf has no stored value (iago_mcts_fresh_leaves) or expands (n_visits >= n_thr, MCTS.py:109:
 * MCTS.policy_func exactly where the reference calls it, no look-ahead) queues the position and waits
 * for the reply while the workgroup's other games go on; the `net_workgroups` other workgroups serve the
 * queue with iago_value_forward_split's / iago_policy_forward_split3's one-board walks (bit-identical
 * numbers).  A game's sequence of leaves, values, priors, rollouts and backups is the reference's: the
 * trees are bit-identical to those of the per-playout launches.
 *   value / policy: the nets' weights as for iago_value_forward_split / iago_policy_forward_split3 with
 *     own = wg_own, opp = wg_opp (four rows per workgroup of the grid: a net workgroup walks two boards
 *     through the value net together when two are queued), n >= 4 x (game workgroups + net_workgroups),
 *     out / probs sized for n rows, no planes / index / n_dev, parts = 1.
 *   rollout: table, seed, id_base, stream_id(_dev) and z [n_games] of iago_rollout_args (product form);
 *     its own / opp are ignored (the leaves' positions cur_own / cur_opp take their place).
 *   State, caller-owned device memory, no initialisation needed: cur_node / cur_own / cur_opp / done
 *     [n_games], roll [n_games] uint8, path [n_games][path_stride], leaf_value [n_games], q_slots
 *     [2][IAGO_SEARCH_QUEUE_ENTRIES][8] uint64 (64-byte aligned: one request ring per net), ctl [16] uint32 (16-byte aligned; after the
 *     launch ctl[3] != 0: the launch gave up after time_limit_ms), rep_v [n_games] uint64, rep_p
 *     [n_games][64] uint64, wg_own / wg_opp [4 x grid].  totals [16] int64 ACCUMULATES value evaluations, policy
 *     evaluations, game-workgroup iterations, pair walks, the net workgroups' waiting and walking time (100 MHz
 *     ticks, summed over the workgroups), idle game-workgroup iterations and the game workgroups' run time; stats as for iago_mcts_select; z_log as in
 *     iago_mcts_lookahead.
 * At most one workgroup per CU is resident; the grid is sized from the device (iago_mcts_search_capacity below,
 * max_cus), so net_workgroups is an upper bound.  Nothing else should occupy the device while the launch runs: a game
 * workgroup that finds no CU free for its net workgroups gives up after time_limit_ms.
 */
#define IAGO_SEARCH_QUEUE_ENTRIES 4096
#define IAGO_SEARCH_GAMES_PER_WORKGROUP 32   /* games a game workgroup owns (at most, and by default) */
typedef struct iago_mcts_search_args {
    const iago_mcts_tree *tree;
    const uint64_t *root_own, *root_opp;
    const uint8_t *active;
    float c_puct, lmbda;
    int32_t n_thr, n_sims;
    int32_t net_workgroups, time_limit_ms;
    const iago_value_split_args *value;
    const iago_policy_split3_args *policy;
    const iago_rollout_args *rollout;
    int32_t *cur_node;
    uint64_t *cur_own, *cur_opp;
    int32_t *path;
    int32_t path_stride, z_log_rows;
    int32_t *done;
    uint8_t *roll;
    float *leaf_value;
    int8_t *z_log;
    int32_t *z_log_n;
    void *q_slots;
    uint32_t *ctl;
    void *rep_v, *rep_p;
    int64_t *totals;
    int32_t *stats;
    uint64_t *wg_own, *wg_opp;
    /* Whole self-play games in the launch: max_turns > 0 (0: one search from root_own / root_opp).  Every game then
       also walks through its TURNS on its own clock -- the mover's legal moves (game.py:210-235); a search of n_sims
       playouts and the most visited move (MCTS.get_move, MCTS.py:139-147), or a pass; MCTS.update_with_move
       (MCTS.py:149-154); the stone and the books of game.py:117-142,253-255 (stone_num, pass_flg, `while
       stone_num < 64` once per pair of turns: iago_play_turn's arithmetic) -- until it is over or max_turns turns
       are played: what engine.SelfPlayEngine.play drives turn by turn for all games in lockstep.  Playout p of a
       game's turn t draws from Philox stream rollout->stream_id + t * n_sims + p.  The trees must be reset and hold
       a whole game (no compaction in the launch).
         game_own / game_opp [n_games]: in: the start positions (own = the first mover); out: the final positions
           as they stand after n_turns swaps of sides;  n_turns [n_games]: out;
         rec_* (optional, all or none) [max_turns][n_games] (rec_pi: [max_turns][n_games][64] int32): per turn the
           position before it (own = mover), whether the mover searched, the move (-1: pass / no turn), the root's
           visit counts by action; rows of turns a game did not play are not written.
       ctl [16]: ctl[3] != 0: gave up after time_limit_ms; ctl[4] != 0: a searched root had no children
       (n_sims < n_thr: the reference's max() of an empty dict, MCTS.py:147). */
    int32_t max_turns;
    int32_t games_per_workgroup; /* games a game workgroup owns: 0 (= IAGO_SEARCH_GAMES_PER_WORKGROUP), 8, 16 or 32; the first
                                    ceil(n_games / that) workgroups of the grid are the game workgroups */
    uint64_t *game_own, *game_opp;
    int32_t *n_turns;
    uint64_t *rec_own, *rec_opp;
    uint8_t *rec_valid;
    int8_t *rec_move;
    int32_t *rec_pi;
    void *vtable;          /* optional position table [vtable_slots][4] uint64 (32-byte aligned, vtable_slots a power of two, 0 =
                              none), zeroed by the caller ONCE and whenever the value net's weights change, kept across
                              launches: MCTS.value_func (MCTS.py:97-103) is a pure function of the position, so a position any
                              game has asked for before is answered from the table instead of the queue (same number, same
                              trees).  An entry's sequence word carries the game whose request put the value there in its
                              high half.  totals[8] counts the hits, totals[12] those of them on a value the asking game
                              itself had put there (or had walked ahead); totals [16] */
    int64_t vtable_slots;
    int64_t *trace;       /* optional diagnostic [trace_rows][4].  Rows 0 .. trace_rows - n_games - 1: game workgroup 0 records
                             (100 MHz ticks since its start, requests queued so far, tickets handed out so far, game
                             workgroups finished | playouts of all games so far << 8) once per iteration of its loop; with
                             max_turns > 0 row trace_rows - 1 - g: (ticks at the end of game g, requests it sent, its turns) */
    int32_t trace_rows;
    int32_t pace_margin;  /* pacing of the leading games: a game more than pace_margin playouts ahead of the mean progress of
                             the games in play starts no new playout while requests queue up (the batch ends with its slowest
                             game; what the leaders do not ask of the nets, the laggards get).  Timing only -- a game's own
                             sequence of playouts, hence every tree and move, is unchanged.  0 = default (16), < 0 = off */
    int32_t max_cus;      /* CUs the launch may count on when fewer than the device's are free for it (a CU-masked stream, a
                             device shared with another job); 0 = all CUs of the device */
    int32_t reserved;
} iago_mcts_search_args;
IAGO_API int iago_mcts_search_persistent(const iago_mcts_search_args *args, void *stream);
/*
 * What the persistent launch can count on: the CUs of the current device and the workgroups of the search kernel that
 * fit one CU (its registers and LDS allow one).  iago_mcts_search_persistent sizes its grid from these: all game
 * workgroups + min(net_workgroups, CUs x workgroups per CU - game workgroups) net workgroups -- every workgroup of the
 * launch is resident from the start -- and returns IAGO_ERR_CAPACITY when the game workgroups and ONE net workgroup do
 * not fit (the launch could never finish: games wait for replies only net workgroups give).  ctl[7] after a launch: the
 * net workgroups it ran with.
 */
IAGO_API int iago_mcts_search_capacity(int32_t *cus, int32_t *workgroups_per_cu);
/*
 * The same search as TWO launches that run together, one per role: the game workgroups (a kernel of its own register
 * budget: two per CU) on a stream masked to `game_cus` CUs, the net workgroups on a stream masked to all the other CUs
 * of the device -- co-resident by construction.  Same arguments, protocol, trees, moves and records (MCTS.py:105-154,
 * game.py:117-142); max_cus must be 0.  For batches whose games need more than 32 workgroups: in the single launch
 * every game workgroup holds a CU alone (2048 games: 17.1 -> 18.3 M leaf-evals/s, 4096 games: 10.8 -> 15.1 M).
 *   iago_mcts_search_streams_create: the two streams and their events for the CURRENT device, once per process and
 *     device (game_cus: a multiple of 8, at most half the device's CUs; IAGO_ERR_HIP where the runtime gives no
 *     CU-masked streams: use iago_mcts_search_persistent there).  _destroy releases them.
 *   iago_mcts_search_split: both launches start after everything queued on `stream` so far, `stream` continues after
 *     both; IAGO_ERR_CAPACITY when the game workgroups do not fit game_cus CUs.  The net launch takes one workgroup
 *     per CU that is not the games', at most 7/8 of the device's CUs.  No host synchronisation.  One call at a
 *     time per `streams` object (its events order the launches); destroy it only when the device has finished with it.
 */
typedef struct iago_search_streams iago_search_streams;
IAGO_API int iago_mcts_search_streams_create(int32_t game_cus, iago_search_streams **out);
IAGO_API int iago_mcts_search_streams_destroy(iago_search_streams *streams);
IAGO_API int iago_mcts_search_split(const iago_mcts_search_args *args, iago_search_streams *streams, void *stream);

/*
 * Whole policy-vs-policy games -- src/rl_self_play.py:8-149, Game(model1, model2)() for n games -- in ONE launch: a
 * workgroup plays a game from its first turn to its last (csrc/selfplay_policy_kernel.hip).  Per turn: the mover's
 * SLPolicy on the board (iago_policy_forward_split3's one-board walk: the same probabilities), the masked draw of
 * src/rl_self_play.py:111-127 (iago_sample_moves with uniforms = NULL: turn t of game b draws the Philox uniform of
 * (seed, id_base + b, step t, stream 0)), the stone and the books of iago_play_turn (src/rl_self_play.py:27-31,130-145).
 * The games are those of the launch-per-turn loop over iago_policy_forward_split3 / iago_sample_moves / iago_play_turn,
 * record for record.
 *   model1 (colour 1, moves first) / model2: weights as for iago_policy_forward_split3; both read their rows from the
 *     SAME own / opp arrays of >= n rows, which the launch writes, and write their distributions to their probs
 *     (>= n rows each); no index, no n_dev; parts is ignored.  Their overflow words are raised as usual.
 *   own / opp [n]: in: the start positions (own = colour 1: the standard start, colour 2 possibly with a handicap
 *     stone, src/train_rl.py:43-46); out: the final positions (own = colour 1).
 *   rec_own / rec_opp / rec_act [max_turns / 2][n]: the position before each of colour 1's turns (own = colour 1)
 *     and its move (-1 = no move: a pass, or the game was over) -- rows of turns after a game's end hold its final
 *     board, as the lockstep loop records them.  n_turns [n]: the even turn at which `while stone_num < 64`
 *     (src/rl_self_play.py:28) ended the game, max_turns if it never did.
 *   bad_probs [1]: raised (never cleared) when a draw met NaN / inf / zero-mass probabilities (numpy.random.choice
 *     raises there, src/rl_self_play.py:122); the game goes on from cell 0 like iago_play_turn after action 64.
 *   max_turns: even, <= IAGO_MAX_TURNS.
 */
typedef struct iago_selfplay_policy_args {
    const iago_policy_split3_args *model1, *model2;
    uint64_t *own, *opp;
    int64_t n;
    uint64_t seed;
    uint32_t id_base;
    int32_t max_turns;
    uint64_t *rec_own, *rec_opp;
    int8_t *rec_act;
    int32_t *n_turns;
    uint32_t *bad_probs;
} iago_selfplay_policy_args;
IAGO_API int iago_selfplay_policy(const iago_selfplay_policy_args *args, void *stream);

/*
 * ---- The gradients of the REINFORCE update of SLPolicy (src/train_rl.py:55-66: pred = model(x), loss =
 * mean(softmax_cross_entropy(pred, y) * r), loss.backward()) in split-f16 arithmetic on the matrix units
 * (csrc/policy_grad_kernels.hip), replacing the float32 convolutions of a tensor library in the update.  Its layer-level
 * pieces (iago_conv3x3_wgrad_split, iago_conv3x3_bwd_data_split, iago_split_scaled) are in iago_hip_layers.h.
 *
 * iago_policy_reinforce_grad: the whole of src/train_rl.py:61-65 -- pred = model1(x), loss = mean(softmax_cross_entropy(
 *   pred, y) * r), model1.cleargrads(), loss.backward() -- for n recorded rows: forward with every block's output kept
 *   (iago_value_stem_boards on block1's weights, 7 x iago_conv3x3_split), the head and the loss forward and backward
 *   (logits = conv9 + bias10, p = softmax, c = logsumexp(p) - p[a] as the reference computes it, dlogits through both
 *   softmaxes), then per block 8 .. 2 iago_split_scaled (+ the bias gradient), iago_conv3x3_wgrad_split,
 *   iago_conv3x3_bwd_data_split, and block 1's gradients from the float32 gradient at its pre-activations.  ~48
 *   launches on `stream`, no host synchronisation; deterministic (fixed summation orders).
 *   own / opp [n]: the recorded positions, own = the mover (the planes of game.py:168-174 are built in the kernel);
 *   action [n] int32 (0..63); reward [n] float32 (z; rows added as padding carry 0); n_mean: the row count the mean
 *   divides by.  w1 [64][2][3][3], b1 [64]; blocks 2..8 (index 0..6): w_hi / w_lo as for iago_conv3x3_split, wt_hi /
 *   wt_lo the transposed form of iago_conv3x3_bwd_data_split, bias [128]; w9 [128], b10 [64].
 *   g_*: the gradients, float32 in the parameters' own layouts ([co][ci][3][3]); loss: device float; probs: optional
 *   [n][64], the model's output.  workspace: iago_policy_grad_workspace_bytes(n) bytes, 256-byte aligned (308 KB per
 *   row + 247 MB: every block keeps its partial sums for the one reduction at the end).  overflow: see
 *   iago_conv3x3_split (the forward's activations: bit 0); bit 1 is raised when an action lies outside 0 .. 63 (the
 *   reference's F.softmax_cross_entropy raises there): that call's loss and gradients must not be used.
 */
typedef struct iago_policy_grad_args {
    const uint64_t *own, *opp;
    const int32_t *action;
    const float *reward;
    int64_t n, n_mean;
    const float *w1, *b1;
    const void *w_hi[7], *w_lo[7], *wt_hi[7], *wt_lo[7];
    const float *bias[7];
    const float *w9, *b10;
    float *g_w1, *g_b1;
    float *g_w[7], *g_b[7];
    float *g_w9, *g_b10;
    float *loss;
    float *probs;
    void *workspace;
    int64_t workspace_bytes;
    uint32_t *overflow;
} iago_policy_grad_args;
IAGO_API int64_t iago_policy_grad_workspace_bytes(int64_t n);
IAGO_API int iago_policy_reinforce_grad(const iago_policy_grad_args *args, void *stream);

/*
 * chainer.optimizers.Adam with the optimizer_hooks.WeightDecay hook as src/train_rl.py:24-26,66 call them, over all
 * parameters in one launch (Chainer's documented rule, float32, every operation rounded on its own):
 *   g = grad + weight_decay * w;  m += (1 - beta1)(g - m);  v += (1 - beta2)(g g - v);  w -= alpha_t m / (sqrt(v) + eps)
 * with alpha_t = alpha sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller.  m / v are updated in place; w too
 * where step[k] is NULL, otherwise the step alpha_t m / (sqrt(v) + eps) is written there and the caller subtracts it
 * (a tensor library then sees its parameter change).
 */
#define IAGO_ADAM_MAX_TENSORS 24
typedef struct iago_adam_args {
    float *p[IAGO_ADAM_MAX_TENSORS];
    const float *g[IAGO_ADAM_MAX_TENSORS];
    float *m[IAGO_ADAM_MAX_TENSORS], *v[IAGO_ADAM_MAX_TENSORS], *step[IAGO_ADAM_MAX_TENSORS];
    int64_t count[IAGO_ADAM_MAX_TENSORS];
    int32_t n_tensors;
    float alpha_t, one_minus_beta1, one_minus_beta2, eps, weight_decay;
} iago_adam_args;
IAGO_API int iago_adam_chainer(const iago_adam_args *args, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IAGO_HIP_H */
