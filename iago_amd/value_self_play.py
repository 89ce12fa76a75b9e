"""Value-network training data by self-play (value_self_play.py:11-59 driven by
gen_value_data.py:6-19), batched on the HIP board kernels.

One game of the reference: the SL policy plays both colours until `stop_num` stones
are on the board (stop_num ~ U{4..63}, gen_value_data.py:14), the position is RECORDED
from the side to move's point of view, that side plays one uniformly random legal move
(no legal move: the game is dropped with result -1, value_self_play.py:46-48), then
the RL policy plays both colours to the end; the sample is (recorded position, result
for the side that was to move there).  Quirks kept: the nets' outputs go through an
extra softmax (value_self_play.py:143,169-171), the draw is over all 64 cells and an
illegal draw falls back to a uniformly random legal move (value_self_play.py:145-148),
and the stop test runs after every single turn, not per pair of turns.

`generate` plays n_games in lockstep (every game is at the same ply, so all boards of a
step share the colour to move): rules, plane encoding and both selection rules run
through the C ABI (iago_legal_moves, iago_encode_planes, iago_sample_moves with an
all-ones mask = the unmasked draw, iago_sample_moves with equal probabilities = the
uniform legal move, iago_apply_moves, iago_judge).  `SelfPlay` is the reference-shaped
B = 1 view.  The text file gen_value_data.py appends to is not reproduced (load.py's
parser is out of scope); `generate` returns the samples as device tensors.
"""
import numpy as np
import torch

from . import boards, engine, ops, rl_self_play

MAX_TURNS = 2 * ops.IAGO_MAX_TURNS  # every turn either places a stone or is one of <= 2 passes in a row


def _softmax_like_reference(out):
    """value_self_play.py:169-171: exp(x) / sum(exp(x)) in the net's float32, no max shift."""
    ex = torch.exp(out.to(torch.float32))
    return (ex / ex.sum(dim=1, keepdim=True)).contiguous()


def generate(model_sl, model_rl, n_games, stop_num=None, seed=0, game_id_base=0, draws=None,
             device="cuda"):
    """n_games lockstep games.  stop_num: (n_games,) ints in [4, 64] (None: drawn uniformly
    from {4..63} like gen_value_data.py:14, from a generator seeded with `seed`).
    draws: replay mode for ONE game (parity tests): an iterator of the uniforms consumed, in
    order, by numpy.random.choice and random.choice in the reference run; otherwise the
    draws are Philox words keyed by (seed, game_id_base + game, ply).

    Returns dict: own/opp (B,) int64 recorded positions (own = the side to move there),
    z (B,) int8 result from that side's view (-1 also for dropped games), dropped (B,) bool,
    color (B,) int8 colour to move at the recorded position, final_p1/final_p2, n_turns."""
    B = n_games
    if draws is not None and B != 1:
        raise ValueError("replay mode (draws) plays one game")
    if stop_num is None:
        g = torch.Generator().manual_seed(int(seed))
        stop_num = torch.randint(4, 64, (B,), generator=g)
    stop = torch.as_tensor(stop_num, dtype=torch.int32).reshape(B).to(device)
    own = torch.full((B,), engine.START_OWN, dtype=torch.int64, device=device)
    opp = torch.full((B,), engine.START_OPP, dtype=torch.int64, device=device)
    stone_num = torch.full((B,), 4, dtype=torch.int32, device=device)
    pass_flg = torch.zeros(B, dtype=torch.bool, device=device)
    finished = torch.zeros(B, dtype=torch.bool, device=device)
    recorded = torch.zeros(B, dtype=torch.bool, device=device)
    dropped = torch.zeros(B, dtype=torch.bool, device=device)
    rec_own, rec_opp = torch.zeros_like(own), torch.zeros_like(opp)
    t_rec = torch.zeros(B, dtype=torch.int32, device=device)
    z = torch.zeros(B, dtype=torch.int8, device=device)
    every_cell = torch.full((B,), -1, dtype=torch.int64, device=device)
    equal = torch.full((B, 64), 1.0 / 64, dtype=torch.float32, device=device)
    it = iter(draws) if draws is not None else None
    nan_seen = torch.zeros((), dtype=torch.bool, device=device)

    def u_next():
        return torch.tensor([next(it)], dtype=torch.float64, device=device)

    t = 0
    while t < MAX_TURNS:
        legal = ops.legal_moves(own, opp)
        has = legal != 0
        live = ~finished
        phase_a = live & (stone_num < stop)                  # value_self_play.py:34
        rec_now = live & ~phase_a & ~recorded                # :38-50
        phase_c = live & recorded & (stone_num < 64)         # :55
        movers = (phase_a | phase_c) & has
        action = torch.full((B,), -1, dtype=torch.int8, device=device)
        if bool(movers.any().item()):
            out = torch.zeros((B, 64), dtype=torch.float32, device=device)
            # a model that has a mover evaluates the WHOLE batch and its movers' rows are kept: the batch a net sees
            # has one size from the first turn to the last (a gathered sub-batch changes size every turn, and every new
            # size sends MIOpen through its solver search: 58 s for 1024 games, measured), and an SLPolicy module goes
            # through its one-board-per-workgroup kernel, whose rows do not depend on the batch around them
            for model, mask in ((model_sl, phase_a & has), (model_rl, phase_c & has)):
                if bool(mask.any().item()):
                    pm = rl_self_play._move_probs(model, own, opp).reshape(B, 64).to(torch.float32)
                    out = torch.where(mask.reshape(B, 1), pm, out)
            p = _softmax_like_reference(out)
            # np.random.choice(64, p=softmax(...)): unmasked = every cell "legal"
            a1 = ops.sample_moves(p, every_cell, uniforms=u_next() if it is not None else None,
                                  seed=seed, id_base=game_id_base, step=t, stream_id=0)
            # iago_sample_moves returns 64 for a row whose probabilities hold NaN / inf (the
            # reference's un-shifted softmax overflows for saturated nets and np.random.choice
            # then raises, value_self_play.py:143): remembered here, raised after the loop
            nan_seen = nan_seen | (movers & (a1 > 63)).any()
            ok = ((legal >> a1.to(torch.int64).clamp(0, 63)) & 1).bool() & (a1 >= 0) & (a1 < 64)
            need_fallback = movers & ~ok
            a2 = a1
            if it is None or bool(need_fallback.any().item()):
                # random.choice(positions): the floor(u * n)-th legal cell
                a2 = ops.sample_moves(equal, legal, uniforms=u_next() if it is not None else None,
                                      seed=seed, id_base=game_id_base, step=t, stream_id=1)
            action = torch.where(movers, torch.where(ok, a1, a2), action)
        if bool(rec_now.any().item()):
            rec_own = torch.where(rec_now, own, rec_own)
            rec_opp = torch.where(rec_now, opp, rec_opp)
            t_rec = torch.where(rec_now, torch.full_like(t_rec, t), t_rec)
            recorded = recorded | rec_now
            stuck = rec_now & ~has                            # :46-48: return state, -1
            dropped = dropped | stuck
            z = torch.where(stuck, torch.full_like(z, -1), z)
            finished = finished | stuck
            rnd = rec_now & has
            if bool(rnd.any().item()):
                ar = ops.sample_moves(equal, legal, uniforms=u_next() if it is not None else None,
                                      seed=seed, id_base=game_id_base, step=t, stream_id=2)
                action = torch.where(rnd, ar, action)
        ops.apply_moves(own, opp, action)
        placed = action >= 0
        passing = (phase_a | phase_c) & ~has
        stone_num = stone_num + placed.to(torch.int32)
        stone_num = torch.where(passing & pass_flg, torch.full_like(stone_num, 64), stone_num)
        pass_flg = torch.where(passing, torch.ones_like(pass_flg),
                               torch.where(placed, torch.zeros_like(pass_flg), pass_flg))
        own, opp = opp, own
        t += 1
        # `while stone_num < 64` of the third phase fails: judge from the recorded side's view
        over = recorded & ~finished & (stone_num >= 64)
        if bool(over.any().item()):
            zz = ops.judge(own, opp)                          # from the side to move NOW
            flip = ((t - t_rec) % 2 == 1)                     # the recorded side is `opp` now
            zz = torch.where(flip, -zz, zz)
            z = torch.where(over, zz, z)
            finished = finished | over
        if bool(finished.all().item()):
            break
    if bool(nan_seen.item()):
        raise ValueError("probabilities contain NaN")   # numpy.random.choice's message
    if not bool(finished.all().item()):
        raise RuntimeError("value_self_play.generate: a game did not finish in %d turns" % MAX_TURNS)
    if it is not None and next(it, None) is not None:
        raise ValueError("replay: the game consumed fewer draws than recorded")
    p1, p2 = (own, opp) if t % 2 == 0 else (opp, own)
    color = torch.where(t_rec % 2 == 0, torch.ones_like(z), torch.full_like(z, 2))
    return dict(own=rec_own, opp=rec_opp, z=z, dropped=dropped, color=color, final_p1=p1,
                final_p2=p2, n_turns=t, stop_num=stop)


class SelfPlay(object):
    """value_self_play.SelfPlay(stop_num)() -> (state, result) (value_self_play.py:11-59).
    The nets are passed in (the reference loads ./models/sl_model.npz and rl_model.npz,
    value_self_play.py:25-28).  `state`: (8,8) float32 board of the recorded position with
    the side to move as 2s and its opponent as 1s (value_self_play.py:38-43)."""

    def __init__(self, stop_num, model0, model1, seed=0, draws=None):
        self.stop_num, self.model0, self.model1 = stop_num, model0, model1
        self.seed, self.draws = seed, draws
        self.state = boards.initial_state()
        self.stone_num, self.pass_flg = 4, False

    def __call__(self):
        r = generate(self.model0, self.model1, 1, stop_num=[self.stop_num], seed=self.seed,
                     draws=self.draws)
        own, opp = ops.tensor_to_bits(r["own"])[0], ops.tensor_to_bits(r["opp"])[0]
        boards.bits_to_state(ops.tensor_to_bits(r["final_p1"])[0], ops.tensor_to_bits(r["final_p2"])[0],
                             out=self.state)
        return boards.bits_to_state(opp, own), int(r["z"].item())


def generate_dataset(model_sl, model_rl, size, batch=4096, seed=0):
    """gen_value_data.main (gen_value_data.py:6-19) without the text file: `size` games in
    lockstep batches; returns (own, opp, z) numpy arrays of the kept samples."""
    owns, opps, zs = [], [], []
    done = 0
    while done < size:
        n = min(batch, size - done)
        r = generate(model_sl, model_rl, n, seed=seed + done, game_id_base=done)
        owns.append(ops.tensor_to_bits(r["own"]))
        opps.append(ops.tensor_to_bits(r["opp"]))
        zs.append(r["z"].cpu().numpy())
        done += n
    return np.concatenate(owns), np.concatenate(opps), np.concatenate(zs)
