/*
 * iago_hip.h -- C ABI of the MI355X (gfx950) Othello self-play hot path.
 *
 * This is the drop-in boundary for the hot path of shionhonda/IaGo (board
 * rules, plane encoding, leaf rollout, PV-MCTS tree arithmetic).  The
 * reference has no FFI layer: the path is in-process Python method calls on
 * ONE (8,8) float32 board.  Each entry point below is the batched restatement
 * of one of those methods and cites the reference interface it replaces
 * (file:line in the reference repository).  INTEGRATION.md shows the ctypes
 * stubs a maintainer of the reference would add.
 *
 * Conventions
 *   - Boards are bitboards: one 64-bit word per colour, bit a = row*8 + col
 *     (the reference's action index, game.py:184).  `own` holds the stones of
 *     the side to move, `opp` the other side's.
 *   - Every pointer is a CALLER-OWNED DEVICE pointer (hipMalloc / a torch
 *     tensor's data_ptr()) unless the parameter is documented as host memory.
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.
 *     Calls only enqueue work: no host synchronisation, no allocation, no
 *     internal threads; re-entrant.
 *   - Return value: IAGO_OK (0) or a negative iago_status; nothing is thrown
 *     across the ABI.  iago_last_error() returns a thread-local message.
 *   - There is NO CPU fallback: without a HIP device every launch fails with
 *     IAGO_ERR_HIP.
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
#define IAGO_ROLLOUT_TABLE_FLOATS (3 * 2 * 256 * 8)

IAGO_API int iago_abi_version(void);
IAGO_API const char *iago_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host); never fails. */
IAGO_API int iago_device_count(void);

/* ------------------------------------------------------------------ rules */

/*
 * legal[b] = bit mask of the legal moves of `own` on board b.
 * Replaces GameFunctions.legal_actions(state, color) (game.py:210-235),
 * GameEnv.valid_pos(color) (rl_env.py:114-138), Simulate.legal_actions
 * (mcts_self_play.py:64-89), rl_self_play.Game.legal_actions
 * (src/rl_self_play.py:63-88).  The reference's ascending action list is the
 * ascending set-bit order of the mask.
 */
IAGO_API int iago_legal_moves(const uint64_t *own, const uint64_t *opp, uint64_t *legal, int64_t n,
                     void *stream);

/*
 * In place: own[b] |= bit(action[b]) | flips, opp[b] &= ~flips, where flips are
 * the opponent runs bracketed from action[b] in the 8 directions.  action -1
 * (IAGO_PASS) leaves the board unchanged.  Like the reference, NO legality
 * check: an illegal or occupied target is overwritten and whatever it brackets
 * is flipped.  The side to move is NOT switched (the caller swaps own/opp).
 * Replaces GameFunctions.place_stone(state, action, color) (game.py:180-207),
 * GameEnv.place_stone (rl_env.py:88-112), mcts_self_play.py:36-62,
 * src/rl_self_play.py:36-61.   action: int8[n].
 */
IAGO_API int iago_apply_moves(uint64_t *own, uint64_t *opp, const int8_t *action, int64_t n, void *stream);

/*
 * planes: float32 (n,2,8,8) NCHW; channel 0 = opp (the opponent of the side
 * to move), channel 1 = own.  Replaces GameFunctions.make_state_var(state,
 * color) (game.py:168-174; copies mcts_self_play.py:91-97,
 * src/rl_self_play.py:102-108).  The un-swapped observation of GameEnv
 * (rl_env.py:36-38,70-72: [state==1, state==2]) is the same call with
 * own = player-2 stones and opp = player-1 stones.
 */
IAGO_API int iago_encode_planes(const uint64_t *own, const uint64_t *opp, float *planes, int64_t n,
                       void *stream);

/*
 * z[b] = sign(popcount(own) - popcount(opp)) as int8.  Replaces
 * Simulate.judge(color) (mcts_self_play.py:113-121), GameEnv.judge
 * (rl_env.py:141-149), rl_self_play.Game.judge (src/rl_self_play.py:91-100).
 */
IAGO_API int iago_judge(const uint64_t *own, const uint64_t *opp, int8_t *z, int64_t n, void *stream);

/* ---------------------------------------------------------------- rollout */

/*
 * Host helper: expand RolloutPolicy parameters (network.py:49-64; conv1/W
 * (1,2,3,3) as 18 floats, channel 0 = opponent plane, channel 1 = side to
 * move) into the row-lookup tables the rollout kernel stages in LDS.
 * w18, table are HOST pointers; table has IAGO_ROLLOUT_TABLE_FLOATS floats
 * and is then copied to the device by the caller.
 */
IAGO_API int iago_rollout_build_table(const float *w18, float *table);

typedef struct iago_rollout_args {
    const uint64_t *own;     /* [n] side to move at the leaf */
    const uint64_t *opp;     /* [n] */
    int64_t n;
    const float *table;      /* device, from iago_rollout_build_table */
    const float *bias;       /* device, 64 floats (bias2/b) */
    const float *uniforms;   /* optional device [IAGO_MAX_TURNS][n] float32 in [0,1); NULL = Philox */
    uint64_t seed;           /* Philox4x32-10 key */
    uint32_t id_base;        /* rollout b draws from counter (id_base + b, turn/4, stream, 0) */
    uint32_t stream_id;
    int8_t *z;               /* [n] result from the leaf side-to-move's view */
    uint64_t *final_own;     /* optional [n]: final stones of the leaf side to move */
    uint64_t *final_opp;     /* optional [n] */
    uint8_t *n_turns;        /* optional [n]: turns played (passes included) */
    uint8_t *trace;          /* optional [IAGO_MAX_TURNS][n]: action per turn, 0xFF = pass */
    int uniform_policy;      /* 1: ignore table/bias, every legal move equally likely */
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

#ifdef __cplusplus
}
#endif
#endif /* IAGO_HIP_H */
