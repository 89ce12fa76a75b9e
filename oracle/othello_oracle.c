/*
 * oracle/othello_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, array-and-loop restatement of the reference's (shionhonda/IaGo)
 * Othello rules, plane encoding, judge, masked sampling and leaf rollout.
 * It deliberately keeps the reference's data model -- an 8x8 board of float
 * cells holding 0 (empty) / 1 / 2 and the reference's loop order -- and uses
 * no bitboards, so it is an independent check of the HIP bitboard kernels.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (iago_amd/) never links, imports or calls
 * it; the product fails loudly if its HIP library is missing.
 *
 * Parity pin: every function here is checked against golden vectors recorded
 * from the *real* reference code (imported from /root/reference under stub
 * chainer/numba modules by tests/golden/make_golden.py); see
 * tests/test_oracle_golden.py.
 *
 * Citations are file:line in the reference repository.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Search directions, in the reference's fixed order.
 * rl_env.py:95-96,122-123  game.py:188-189,218-219  mcts_self_play.py:42-43,71-72 */
static const int DYS[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
static const int DXS[8] = {-1, 0, 1, -1, 1, -1, 0, 1};

/* game.py:164-165  rl_env.py:82-83  mcts_self_play.py:138-139 */
static int is_outside(int y, int x) { return y < 0 || y > 7 || x < 0 || x > 7; }

/*
 * legal_actions -- game.py:210-235 == rl_env.py:114-138 == mcts_self_play.py:64-89
 * == src/rl_self_play.py:63-88.  Row-major scan of empty cells; for each of the
 * 8 directions: neighbour must be the opponent (state+color==3), walk while
 * opponent, stay in bounds, landing cell must equal color; first success
 * appends a = i*8+j and breaks.  Returns the count, actions ascending.
 */
ORC_API int orc_legal_actions(const float *state, int color, int *actions)
{
    int n = 0;
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) {
            if (state[i * 8 + j] != 0.0f)
                continue;
            for (int d = 0; d < 8; d++) {
                int dy = DYS[d], dx = DXS[d];
                if (is_outside(i + dy, j + dx))
                    continue;
                if (state[(i + dy) * 8 + (j + dx)] + (float)color != 3.0f)
                    continue;
                int ry = i + dy, rx = j + dx, out = 0;
                while (state[ry * 8 + rx] + (float)color == 3.0f) {
                    ry += dy;
                    rx += dx;
                    out = is_outside(ry, rx);
                    if (out)
                        break;
                }
                if (out)
                    continue;
                if (state[ry * 8 + rx] == (float)color) {
                    actions[n++] = i * 8 + j;
                    break;
                }
            }
        }
    }
    return n;
}

/*
 * place_stone -- game.py:180-207 (action==-1 returns the state unchanged,
 * game.py:181-182) == rl_env.py:88-112 (1-origin position, flips via 3-state)
 * == mcts_self_play.py:36-62 == src/rl_self_play.py:36-61.
 * NO legality check: the target cell is overwritten and whatever is bracketed
 * from it gets flipped, exactly as the reference does.
 */
ORC_API void orc_place_stone(float *state, int action, int color)
{
    if (action == -1)
        return;
    int py = action / 8, px = action % 8;
    state[py * 8 + px] = (float)color;
    for (int d = 0; d < 8; d++) {
        int dy = DYS[d], dx = DXS[d];
        if (is_outside(py + dy, px + dx))
            continue;
        if (state[(py + dy) * 8 + (px + dx)] + (float)color != 3.0f)
            continue;
        int ry = py + dy, rx = px + dx, out = 0;
        while (state[ry * 8 + rx] + (float)color == 3.0f) {
            ry += dy;
            rx += dx;
            out = is_outside(ry, rx);
            if (out)
                break;
        }
        if (out)
            continue;
        if (state[ry * 8 + rx] == (float)color) {
            ry -= dy;
            rx -= dx;
            while (state[ry * 8 + rx] + (float)color == 3.0f) {
                state[ry * 8 + rx] = (float)color;
                ry -= dy;
                rx -= dx;
            }
        }
    }
}

/*
 * make_state_var -- game.py:168-174 (copies: mcts_self_play.py:91-97,
 * src/rl_self_play.py:102-108).  color==1 swaps 1<->2 through
 * s*(3-s)*(3-s)/2, then planes [s==1, s==2].  Net effect: channel 0 = the
 * opponent of the side to move, channel 1 = the side to move.
 * out: 2*64 floats, NCHW of a (1,2,8,8) tensor.
 */
ORC_API void orc_make_state_var(const float *state, int color, float *out)
{
    for (int k = 0; k < 64; k++) {
        float s = state[k];
        if (color == 1)
            s = s * (3.0f - s) * (3.0f - s) / 2.0f;
        out[k] = (s == 1.0f) ? 1.0f : 0.0f;
        out[64 + k] = (s == 2.0f) ? 1.0f : 0.0f;
    }
}

/* rl_env.py:27-39,70-72 observation: planes [state==1, state==2], no swap. */
ORC_API void orc_env_obs(const float *state, float *out)
{
    for (int k = 0; k < 64; k++) {
        out[k] = (state[k] == 1.0f) ? 1.0f : 0.0f;
        out[64 + k] = (state[k] == 2.0f) ? 1.0f : 0.0f;
    }
}

/* judge -- mcts_self_play.py:113-121 (from `color`'s view); rl_env.py:141-149
 * and src/rl_self_play.py:91-100 are the color==1 case. */
ORC_API int orc_judge(const float *state, int color)
{
    int me = 0, op = 0;
    for (int k = 0; k < 64; k++) {
        if (state[k] == (float)color)
            me++;
        if (state[k] == (float)(3 - color))
            op++;
    }
    return me > op ? 1 : (me < op ? -1 : 0);
}

/* ------------------------------------------------------------------ */
/* Sampling                                                            */
/* ------------------------------------------------------------------ */

/*
 * numpy.random.choice(64, p=p) semantics (third-party numpy, mtrand.pyx
 * `choice`: cdf = p.cumsum(); cdf /= cdf[-1]; idx = cdf.searchsorted(u,
 * side='right')), given the uniform u instead of the MT19937 draw.
 * Call sites: mcts_self_play.py:106, src/rl_self_play.py:122, game.py:103.
 */
ORC_API int orc_choice_cdf(const double *p, int n, double u)
{
    double cdf[64];
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        acc += p[i];
        cdf[i] = acc;
    }
    for (int i = 0; i < n; i++)
        cdf[i] /= acc;
    int idx = 0; /* searchsorted side='right': first idx with cdf[idx] > u */
    while (idx < n && cdf[idx] <= u)
        idx++;
    return idx;
}

/*
 * RolloutPolicy forward -- network.py:49-64: conv 3x3, 2->1 channels, pad 1,
 * no bias (cross-correlation, W (1,2,3,3)); reshape (1,64); + bias2 (64);
 * softmax(axis=1).  float32 arithmetic like Chainer's.
 * x: (2,8,8) planes from make_state_var.  prob: 64 floats.  logits optional.
 */
ORC_API void orc_rollout_policy(const float *x, const float *w18, const float *b64,
                                float *prob, float *logits_out)
{
    float h[64];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            float acc = 0.0f;
            for (int c = 0; c < 2; c++)
                for (int ky = 0; ky < 3; ky++)
                    for (int kx = 0; kx < 3; kx++) {
                        int y = i + ky - 1, xx = j + kx - 1;
                        if (is_outside(y, xx))
                            continue;
                        acc += w18[c * 9 + ky * 3 + kx] * x[c * 64 + y * 8 + xx];
                    }
            h[i * 8 + j] = acc + b64[i * 8 + j];
        }
    float m = h[0];
    for (int k = 1; k < 64; k++)
        if (h[k] > m)
            m = h[k];
    float s = 0.0f;
    for (int k = 0; k < 64; k++) {
        prob[k] = expf(h[k] - m);
        s += prob[k];
    }
    for (int k = 0; k < 64; k++)
        prob[k] /= s;
    if (logits_out)
        memcpy(logits_out, h, sizeof h);
}

/*
 * Masked distribution of Simulate.get_action -- mcts_self_play.py:100-106:
 * prob (f32 softmax) * valid (f64 0/1), normalised by the f64 sum.
 */
ORC_API void orc_masked_probs(const float *prob, const int *actions, int n_actions, double *p)
{
    double valid[64];
    for (int k = 0; k < 64; k++)
        valid[k] = 0.0;
    for (int k = 0; k < n_actions; k++)
        valid[actions[k]] = 1.0;
    double s = 0.0;
    for (int k = 0; k < 64; k++) {
        p[k] = (double)prob[k] * valid[k];
        s += p[k];
    }
    for (int k = 0; k < 64; k++)
        p[k] /= s;
}

/* ------------------------------------------------------------------ */
/* Philox4x32-10 counter RNG (Salmon et al., SC'11) -- the build's      */
/* replacement for numpy's MT19937 stream (RNG     */
/* parity is by replay).  key = (seed_lo, seed_hi), counter = (game,    */
/* turn>>2, stream, 0); uniform = (word[turn&3] >> 8) * 2^-24 in [0,1). */
/* ------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

ORC_API void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                        uint32_t *out4)
{
    uint32_t c[4] = {c0, c1, c2, c3};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    memcpy(out4, c, sizeof c);
}

ORC_API float orc_uniform(uint64_t seed, uint32_t game, uint32_t step, uint32_t stream)
{
    /* one Philox block serves 4 consecutive turns: word step&3 of counter step>>2 */
    uint32_t c[4] = {game, step >> 2, stream, 0};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c[step & 3] >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ */
/* Leaf rollout                                                        */
/* ------------------------------------------------------------------ */

/*
 * Simulate(state)(color) -- mcts_self_play.py:9-29,124-134.
 *   stone_num = 64 - #empty; pass_flg = False
 *   while stone_num < 64: turn(color); turn(3-color)
 *   return judge(color)
 * turn(c): legal_actions(c); if any: get_action -> place_stone; pass_flg =
 * False; stone_num += 1; else: if pass_flg: stone_num = 64; pass_flg = True.
 *
 * The uniform for the t-th call of turn() (t counts every turn, passes
 * included, from 0) is uniforms[t] if uniforms != NULL, otherwise
 * orc_uniform(seed, game_id, t, 0).  A turn without a legal move consumes no
 * uniform but still advances t.  trace (optional, >=130 bytes): the action of
 * every turn, 0xFF for a pass, and *n_turns their count.
 * The resample-on-illegal recursion (mcts_self_play.py:107-109) cannot fire:
 * masked-out cells carry probability exactly 0.
 */
ORC_API int orc_simulate(float *state, int color, const float *w18, const float *b64,
                         const float *uniforms, uint64_t seed, uint32_t game_id,
                         uint8_t *trace, int *n_turns)
{
    int empties = 0;
    for (int k = 0; k < 64; k++)
        if (state[k] == 0.0f)
            empties++;
    int stone_num = 64 - empties;
    int pass_flg = 0;
    int t = 0;
    while (stone_num < 64) {
        for (int half = 0; half < 2; half++) {
            int c = half == 0 ? color : 3 - color;
            int actions[64];
            int n = orc_legal_actions(state, c, actions);
            if (n > 0) {
                float x[128], prob[64];
                double p[64];
                orc_make_state_var(state, c, x);
                orc_rollout_policy(x, w18, b64, prob, 0);
                orc_masked_probs(prob, actions, n, p);
                double u = uniforms ? (double)uniforms[t]
                                    : (double)orc_uniform(seed, game_id, (uint32_t)t, 0);
                int a = orc_choice_cdf(p, 64, u);
                orc_place_stone(state, a, c);
                pass_flg = 0;
                stone_num += 1;
                if (trace)
                    trace[t] = (uint8_t)a;
            } else {
                if (pass_flg)
                    stone_num = 64;
                pass_flg = 1;
                if (trace)
                    trace[t] = 0xFF;
            }
            t++;
        }
    }
    if (n_turns)
        *n_turns = t;
    return orc_judge(state, color);
}

/*
 * Uniform-random playout on the same turn structure (the "board logic only"
 * workload of SURVEY.md section 6): action = actions[floor(u * n)].
 * Used for traces and as a CPU baseline of the pure rules path.
 */
ORC_API int orc_random_playout_stream(float *state, int color, uint64_t seed, uint32_t game_id, uint32_t stream,
                                      uint8_t *trace, int *n_turns);
ORC_API int orc_random_playout(float *state, int color, uint64_t seed, uint32_t game_id,
                               uint8_t *trace, int *n_turns)
{
    return orc_random_playout_stream(state, color, seed, game_id, 0, trace, n_turns);
}

/* The same game drawn from Philox stream `stream` of (seed, game_id): a search rolls the leaf of its
 * i-th playout out on stream i (the build's keying; the reference draws from numpy's global state). */
ORC_API int orc_random_playout_stream(float *state, int color, uint64_t seed, uint32_t game_id, uint32_t stream,
                                      uint8_t *trace, int *n_turns)
{
    int empties = 0;
    for (int k = 0; k < 64; k++)
        if (state[k] == 0.0f)
            empties++;
    int stone_num = 64 - empties, pass_flg = 0, t = 0;
    while (stone_num < 64) {
        for (int half = 0; half < 2; half++) {
            int c = half == 0 ? color : 3 - color;
            int actions[64];
            int n = orc_legal_actions(state, c, actions);
            if (n > 0) {
                float u = orc_uniform(seed, game_id, (uint32_t)t, stream);
                int k = (int)(u * (float)n);
                if (k >= n)
                    k = n - 1;
                orc_place_stone(state, actions[k], c);
                pass_flg = 0;
                stone_num += 1;
                if (trace)
                    trace[t] = (uint8_t)actions[k];
            } else {
                if (pass_flg)
                    stone_num = 64;
                pass_flg = 1;
                if (trace)
                    trace[t] = 0xFF;
            }
            t++;
        }
    }
    if (n_turns)
        *n_turns = t;
    return orc_judge(state, color);
}

/* Batched driver for the CPU baseline: n rollouts from the given start
 * position; returns the sum of turns played (board-steps). */
ORC_API long orc_simulate_batch(const float *state0, int color, const float *w18,
                                const float *b64, uint64_t seed, uint32_t first_game,
                                int n, int8_t *z_out)
{
    long steps = 0;
    for (int g = 0; g < n; g++) {
        float s[64];
        int nt = 0;
        memcpy(s, state0, sizeof s);
        int z = orc_simulate(s, color, w18, b64, 0, seed, first_game + (uint32_t)g, 0, &nt);
        if (z_out)
            z_out[g] = (int8_t)z;
        steps += nt;
    }
    return steps;
}

/* ------------------------------------------------------------------ */
/* MCTS node arithmetic -- MCTS.py:10-76                                */
/* ------------------------------------------------------------------ */

/*
 * Node.U -- MCTS.py:48-49: c_puct * P * sqrt(parent.n_visits) / (0.01 + n_visits)
 * with P = prob + 0.1 (MCTS.py:19).  dtypes as the reference produces them
 * under numpy >= 2 (the version the golden vectors were recorded with): P is
 * float32 (float32 prior + weak python float), c_puct*P stays float32,
 * np.sqrt(int) is float64, so U is float64; Q is a float32 running mean
 * (MCTS.py:63) and get_value = Q + u (MCTS.py:75-76) is float64.
 */
ORC_API float orc_node_P(float prob) { return prob + 0.1f; }

ORC_API double orc_node_U(float c_puct, float P, int parent_n, int n)
{
    float cp = c_puct * P;
    return (double)cp * sqrt((double)parent_n) / (0.01 + (double)n);
}

ORC_API float orc_node_update_Q(float Q, float leaf_value, int n_after)
{
    return Q + (leaf_value - Q) / (float)n_after;
}

/* leaf mix -- MCTS.py:123-125: (1-lmbda)*v + lmbda*z; v float32, z int. */
ORC_API float orc_leaf_value(float lmbda, float v, int z)
{
    float a = (float)(1.0 - (double)lmbda) * v;
    double b = (double)lmbda * (double)z;
    return a + (float)b;
}
