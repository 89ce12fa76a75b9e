"""Policy-vs-policy self-play games of the REINFORCE trainer
(src/rl_self_play.py:8-149): `Game(model1, model2)()` with the reference's
return value, plus `play_batch`, the lockstep batched form the GPU engine is
built for.  model1 is the learner (colour 1, moves first); only colour-1 plies
are recorded, as perspective-swapped boards with their actions
(src/rl_self_play.py:134-138)."""
import numpy as np
import torch

from . import boards, engine, ops


# turn from which play_batch reads the end-of-batch flag back every second turn (before it the flag is logged on the
# device); 0 = from the first turn on, the reference's loop to the letter (the tests compare the two)
SYNC_FROM = 56


def _move_probs(model, own, opp):
    """model(make_state_var(...)) for every board.  An SLPolicy module is evaluated by its one-launch
    three-piece kernel straight from the boards: one board per workgroup, so a board's distribution
    does not depend on how many boards are in the batch -- the games a rank plays do not depend on how
    the games were sharded (the planes-fed float32 kernels pick their tiling by batch size and agree
    only to ~1e-6).  Any other callable sees the planes, as in the reference."""
    with torch.no_grad():
        fb = getattr(model, "forward_boards_split3", None)
        if fb is not None and getattr(model, "split3", False) and not getattr(model, "training", False):
            return fb(own, opp)
        return model(ops.encode_planes(own, opp))


# the games of play_batch as ONE launch (iago_selfplay_policy: a workgroup per game, the walk's workgroup goes on to the
# draw, the stone and the next walk) wherever both models are SLPolicy modules on their three-piece kernel and the draws
# are the Philox ones; False: the launch-per-turn loop everywhere (the tests compare the two)
ONE_LAUNCH = True


def _play_one_launch(model1, model2, own, opp, seed, game_id_base):
    """iago_selfplay_policy on the start positions own / opp (consumed: the final boards on return)."""
    import ctypes as C
    from . import _lib
    B, dev = own.numel(), own.device
    T = ops.IAGO_MAX_TURNS
    rows_own, rows_opp = torch.empty_like(own), torch.empty_like(opp)
    probs = torch.empty((2, B, 64), dtype=torch.float32, device=dev)
    a1, keep1 = model1.search_args(rows_own, rows_opp, probs[0])
    a2, keep2 = model2.search_args(rows_own, rows_opp, probs[1])
    rec_own = torch.empty((T // 2, B), dtype=torch.int64, device=dev)
    rec_opp = torch.empty((T // 2, B), dtype=torch.int64, device=dev)
    rec_act = torch.empty((T // 2, B), dtype=torch.int8, device=dev)
    n_turns = torch.empty(B, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    a = _lib.SelfplayPolicyArgs()
    a.model1, a.model2 = C.addressof(a1), C.addressof(a2)
    a.own, a.opp, a.n = own.data_ptr(), opp.data_ptr(), B
    a.seed, a.id_base, a.max_turns = int(seed) & 0xFFFFFFFFFFFFFFFF, int(game_id_base) & 0xFFFFFFFF, T
    a.rec_own, a.rec_opp, a.rec_act = rec_own.data_ptr(), rec_opp.data_ptr(), rec_act.data_ptr()
    a.n_turns, a.bad_probs = n_turns.data_ptr(), bad.data_ptr()
    ops.check(_lib.lib().iago_selfplay_policy(C.byref(a), ops._stream()), "iago_selfplay_policy")
    del keep1, keep2
    host = torch.cat([n_turns.max().reshape(1), bad]).tolist()     # (the batch's one read-back)
    return rec_own, rec_opp, rec_act, int(host[0]), bool(host[1])


def play_batch(model1, model2, n_games, handicap=None, seed=0, game_id_base=0, uniforms=None,
               device="cuda"):
    """n_games lockstep games.  handicap: optional (n_games,) int64 bit masks of
    extra colour-2 stones (src/train_rl.py:43-46).  uniforms: optional iterator
    of float64 (n_games,) tensors, one per turn that has a mover (parity tests).

    Returns dict: own/opp (T1,B) int64 recorded learner positions (own = colour
    1 = the mover), action (T1,B) int8 (-1 where that game did not move),
    z (B,) int8 from colour 1's view, final_p1/final_p2, n_turns.

    Shard-independence of the games (a rank's games do not depend on how many other games share
    its batch) holds for SLPolicy modules in eval mode with `split3` on (one board per workgroup);
    that kernel clamps activations at 65000, so both models' saturation flags are read once at the
    end of the batch and a saturated net raises here (the reference's float32 range is unbounded:
    set `model.split3 = False` for such weights) instead of playing on from clamped priors."""
    B = n_games
    own = torch.full((B,), engine.START_OWN, dtype=torch.int64, device=device)
    opp = torch.full((B,), engine.START_OPP, dtype=torch.int64, device=device)
    if handicap is not None:
        opp = opp | handicap
    if (ONE_LAUNCH and uniforms is None and B > 0
            and all(getattr(m, "search_args", None) is not None and getattr(m, "forward_boards_split3", None) is not None
                    and getattr(m, "split3", False) and not getattr(m, "training", False) for m in (model1, model2))):
        rec_own, rec_opp, rec_act, t, bad = _play_one_launch(model1, model2, own, opp, seed, game_id_base)
        for model in (model1, model2):
            model.check_saturation()
        if bad:
            raise ValueError("probabilities contain NaN")      # numpy.random.choice, src/rl_self_play.py:122
        return dict(own=rec_own[: t // 2], opp=rec_opp[: t // 2], action=rec_act[: t // 2],
                    z=ops.judge(own, opp), final_p1=own, final_p2=opp, n_turns=t)
    stone_num = torch.full((B,), 4, dtype=torch.int32, device=device)  # src/rl_self_play.py:20
    pass_flg = torch.zeros(B, dtype=torch.uint8, device=device)
    done = torch.zeros(B, dtype=torch.uint8, device=device)
    T = ops.IAGO_MAX_TURNS
    rec_own = torch.empty((T // 2, B), dtype=torch.int64, device=device)
    rec_opp = torch.empty((T // 2, B), dtype=torch.int64, device=device)
    rec_act = torch.empty((T // 2, B), dtype=torch.int8, device=device)
    a_max = torch.full((B,), -1, dtype=torch.int8, device=device)   # 64 = iago_sample_moves met NaN probabilities
    legal = ops.legal_moves(own, opp)          # of the mover; 0 for a finished game from turn 1 on
    active = (legal != 0).to(torch.uint8)
    legal_next, active_next = torch.empty_like(legal), torch.empty_like(active)
    # `while stone_num < 64` is tested once per pair of turns (src/rl_self_play.py:28-30).  Reading that flag back
    # costs the loop a host round trip, and no lockstep batch is over before turn SYNC_FROM in practice: up to there
    # the flag goes into a device-side log (turns played past the true end change nothing: a finished game has no
    # legal move, so no action, no stone, no record), and the log gives the turn the batch really ended at
    sync_from = SYNC_FROM if uniforms is None else 0
    over_log = torch.zeros(T // 2 + 1, dtype=torch.uint8, device=device)
    t = 0
    while t < T:
        color = 1 if t % 2 == 0 else 2
        probs = _move_probs(model1 if color == 1 else model2, own, opp)
        u = next(uniforms) if (uniforms is not None and bool(active.any().item())) else None
        a = ops.sample_moves(probs.to(torch.float32).contiguous(), legal, uniforms=u,
                             seed=seed, id_base=game_id_base, step=t)
        torch.maximum(a_max, a, out=a_max)
        if color == 1:
            torch._foreach_copy_([rec_own[t // 2], rec_opp[t // 2], rec_act[t // 2]], [own, opp, a])
        # the move, stone_num / pass_flg / done (`while stone_num < 64` per pair of turns,
        # src/rl_self_play.py:28-30,130-145), the swap of sides and the next mover's moves
        ops.play_turn(own, opp, a, active, stone_num, pass_flg, done, t % 2 == 1, legal_next, active_next)
        legal, legal_next = legal_next, legal
        active, active_next = active_next, active
        t += 1
        if t % 2 == 0:
            if t >= sync_from:
                if bool(done.all().item()):
                    break
            else:
                torch.amin(done, dim=0, out=over_log[t // 2])    # 1 iff every game is over
    if sync_from:
        first = torch.nonzero(over_log[: sync_from // 2]).reshape(-1)[:1].tolist()   # (one read for the whole batch)
        if first:
            t = 2 * first[0]      # the batch was over there already: the later turns were idle (an even number of swaps)
    nan_seen = a_max > 63
    rec_own, rec_opp, rec_act = rec_own[: (t + 1) // 2], rec_opp[: (t + 1) // 2], rec_act[: (t + 1) // 2]
    p1, p2 = (own, opp) if t % 2 == 0 else (opp, own)
    for model in (model1, model2):
        if getattr(model, "check_saturation", None) is not None:
            model.check_saturation()   # raises and clears the flag (one readback per model and batch)
    if bool(nan_seen.any().item()):
        # iago_sample_moves returns 64 when no cell's CDF exceeds u: NaN probabilities.
        # numpy.random.choice raises here in the reference (src/rl_self_play.py:122)
        raise ValueError("probabilities contain NaN")
    return dict(own=rec_own, opp=rec_opp, action=rec_act,
                z=ops.judge(p1, p2), final_p1=p1, final_p2=p2, n_turns=t)


class Game(object):
    """src/rl_self_play.py:8-31: one game; `game.state` may be edited before the
    call (the handicap stone of src/train_rl.py:43-46)."""

    def __init__(self, model1, model2, seed=0, uniforms=None):
        self.state = boards.initial_state()
        self.states, self.actions = [], []
        self.stone_num, self.pass_flg = 4, False
        self.model1, self.model2 = model1, model2
        self.seed, self.uniforms = seed, uniforms

    def __call__(self):
        p1, p2 = boards.state_to_bits(self.state)
        extra = p2 & ~engine.START_OPP
        if p1 != engine.START_OWN or (p2 & engine.START_OPP) != engine.START_OPP:
            raise ValueError("Game starts from the standard position (+ optional colour-2 stones)")
        hc = torch.from_numpy(np.array([extra], dtype=np.uint64).view(np.int64)).cuda()
        us = None
        if self.uniforms is not None:
            it = iter(self.uniforms)
            us = (torch.tensor([next(it)], dtype=torch.float64, device="cuda")
                  for _ in iter(int, 1))
        r = play_batch(self.model1, self.model2, 1, handicap=hc, seed=self.seed, uniforms=us)
        own, opp, act = (ops.tensor_to_bits(r["own"])[:, 0], ops.tensor_to_bits(r["opp"])[:, 0],
                         r["action"].cpu().numpy()[:, 0])
        for o, p, a in zip(own, opp, act):
            if a >= 0:
                # the recorded board is colour-swapped: the learner's stones are 2s
                self.states.append(boards.bits_to_state(p, o))
                self.actions.append(int(a))
        boards.bits_to_state(ops.tensor_to_bits(r["final_p1"])[0],
                             ops.tensor_to_bits(r["final_p2"])[0], out=self.state)
        self.stone_num = 64
        return self.states, self.actions, int(r["z"].item())
