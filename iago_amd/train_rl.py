"""REINFORCE self-play training (src/train_rl.py:13-88) on the GPU engine.

One "set" = 2N = 64 games of the learner (colour 1) against an opponent drawn
from the pool of earlier checkpoints; odd games give colour 2 an extra stone on
one of (2,4),(3,5),(4,2),(5,3) (src/train_rl.py:41-47).  The update reproduces
the reference's loss exactly, including its quirk: `pred` is already a softmax
output and `F.softmax_cross_entropy(pred, y)` applies log-softmax to it again
(src/train_rl.py:61-64), loss = mean(CE_i * z_i).  Optimiser: Chainer's Adam
with the WeightDecay(5e-4) hook (src/train_rl.py:24-26); Chainer is not in the
reference tree (unpinned, readme.md:13), so `ChainerAdam` restates its published
update rule -- PARITY UNPINNED, see DESIGN.md.

Multi-GPU (BASELINE configs[4]): the games of a set shard over the ranks and the
recorded (state, action, z) tuples are all-gathered (iago_amd.dist); every rank
then computes the update from the same batch and rank 0's parameters are
broadcast (one 3.8 MB collective per set), so the replicas cannot drift even
where a backward kernel is not bit-reproducible across GPUs.  Opponent and
handicap draws come from ONE seed (rank 0's), checkpoints are written atomically
by rank 0 and followed by a barrier before anyone lists the pool again.
"""
import collections
import glob
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import dist as idist
from . import _lib, engine, network, ops, rl_self_play


POOL_CACHE = 64   # opponent snapshots kept as ready modules (3.8 MB of parameters + 5.8 MB of weight pieces each)
# the update's loss and gradients through iago_policy_reinforce_grad (split-f16 kernels on the matrix units) instead of
# autograd over the tensor library's float32 convolutions (2 ms against 9 ms at 1,900 rows); "0": autograd
NATIVE_GRAD = os.environ.get("IAGO_NATIVE_GRAD", "1") != "0"


class ChainerAdam(object):
    """chainer.optimizers.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8) +
    optimizer_hooks.WeightDecay(rate) as called at src/train_rl.py:24-26,66.

    Chainer's rule (v4+, from its documentation): the hook adds rate*w to every
    gradient (biases included) before the update; then per parameter
        m += (1-beta1)(g - m);  v += (1-beta2)(g*g - v)
        w -= alpha_t * m / (sqrt(v) + eps),
        alpha_t = alpha * sqrt(1 - beta2^t) / (1 - beta1^t).
    State layout of `state_dict_npz` follows serializers.save_npz(optimizer):
    't', 'epoch', '<param path>/t', '<param path>/m', '<param path>/v'.
    """

    def __init__(self, model, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=5e-4):
        self.model = model
        self.alpha, self.beta1, self.beta2, self.eps, self.wd = alpha, beta1, beta2, eps, weight_decay
        self.t = 0
        self.state = {n: (torch.zeros_like(p), torch.zeros_like(p))
                      for n, p in model.named_parameters()}

    @torch.no_grad()
    def update(self):
        self.t += 1
        a_t = self.alpha * math.sqrt(1.0 - self.beta2 ** self.t) / (1.0 - self.beta1 ** self.t)
        ps, gs, ms, vs = [], [], [], []
        for n, p in self.model.named_parameters():
            if p.grad is None:
                continue
            ps.append(p)
            gs.append(p.grad)
            ms.append(self.state[n][0])
            vs.append(self.state[n][1])
        if not ps:
            return
        if (NATIVE_GRAD and len(ps) <= 24
                and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ps + gs)):
            # the same rule, the same roundings: ONE launch (iago_adam_chainer) for g, m, v and the step, then the
            # subtraction as a tensor operation (it bumps p._version, the key of every cached weight layout) -- instead of
            # 14 multi-tensor launches
            steps = self.__dict__.setdefault("_steps", {})
            ss = [steps.setdefault(id(p), torch.empty_like(p)) for p in ps]
            ops.adam_chainer([p.detach() for p in ps], gs, ms, vs, a_t, self.beta1, self.beta2, self.eps, self.wd, steps=ss)
            torch._foreach_sub_(ps, ss)
            return
        # the rule above, operation by operation (each product and sum rounded on its own, as numpy rounds them), over
        # all parameters at once: 14 multi-tensor launches instead of 12 per parameter
        g = torch._foreach_add(gs, torch._foreach_mul(ps, self.wd))          # g = p.grad + wd * p
        d = torch._foreach_sub(g, ms)
        torch._foreach_mul_(d, 1.0 - self.beta1)
        torch._foreach_add_(ms, d)                                            # m += (g - m) (1 - beta1)
        gg = torch._foreach_mul(g, g)
        torch._foreach_sub_(gg, vs)
        torch._foreach_mul_(gg, 1.0 - self.beta2)
        torch._foreach_add_(vs, gg)                                           # v += (g g - v) (1 - beta2)
        den = torch._foreach_sqrt(vs)
        torch._foreach_add_(den, self.eps)
        num = torch._foreach_mul(ms, a_t)
        torch._foreach_div_(num, den)
        torch._foreach_sub_(ps, num)                                          # p -= a_t m / (sqrt(v) + eps)

    def state_dict_npz(self):
        out = {"t": np.asarray(self.t, np.int64), "epoch": np.asarray(0, np.int64)}
        keys = {id(p): k for k, p in self.model._npz_map().items()}
        for n, p in self.model.named_parameters():
            k = keys[id(p)]
            m, v = self.state[n]
            out[k + "/t"] = np.asarray(self.t, np.int64)
            out[k + "/m"] = m.detach().cpu().numpy()
            out[k + "/v"] = v.detach().cpu().numpy()
        return out

    def save_npz(self, path):
        _atomic_savez(path, self.state_dict_npz())

    def load_npz(self, path):
        src = np.load(path)
        self.t = int(src["t"])
        keys = {id(p): k for k, p in self.model._npz_map().items()}
        for n, p in self.model.named_parameters():
            k = keys[id(p)]
            m, v = self.state[n]
            m.copy_(torch.from_numpy(src[k + "/m"]))
            v.copy_(torch.from_numpy(src[k + "/v"]))
        return self


def _atomic_savez(path, arrays):
    """np.savez to a temporary name in the same directory + rename: a reader never sees
    a partially written checkpoint."""
    final = path if path.endswith(".npz") else path + ".npz"
    tmp = "%s.part%d" % (final, os.getpid())  # not *.npz: invisible to the pool listing
    with open(tmp, "wb") as f:
        np.savez(f, **arrays)
    os.replace(tmp, final)


def _canonical(gathered):
    """Gathered tuple rows (rank 0's first, then rank 1's, ...) in the order of their `key`
    column: independent of how the games were sharded.  The key column is dropped."""
    order = torch.argsort(gathered["key"], stable=True)
    return {k: v[order] for k, v in gathered.items() if k != "key"}


def reinforce_loss(model, own, opp, actions, rewards, pad_to=None):
    """src/train_rl.py:55-64.  own/opp: recorded learner positions (own = the
    learner = mover); the reference rebuilds planes [x==1, x==2] from the
    colour-swapped boards, which is exactly encode_planes(own, opp).

    pad_to: round the batch up to a multiple of it with zero-reward rows (they
    add exactly 0 to the loss and its gradient; the mean still divides by the
    true count) so that MIOpen sees a handful of batch shapes instead of a new
    one -- and a new kernel search -- every set."""
    n = own.numel()
    if pad_to and n % pad_to:
        extra = pad_to - n % pad_to
        own = torch.cat([own, own[:1].expand(extra)])
        opp = torch.cat([opp, opp[:1].expand(extra)])
        actions = torch.cat([actions, actions[:1].expand(extra)])
        rewards = torch.cat([rewards, torch.zeros(extra, dtype=rewards.dtype,
                                                  device=rewards.device)])
    x = ops.encode_planes(own.contiguous(), opp.contiguous())
    pred = model(x)                                                  # softmax probabilities
    c = F.cross_entropy(pred, actions.to(torch.int64), reduction="none")  # log-softmax AGAIN
    return torch.sum(c * rewards.to(torch.float32)) / n              # F.mean(c * r)


class ReinforceTrainer(object):
    """The loop of src/train_rl.py:28-81 with the reference's constants."""

    def __init__(self, model1, pool_dir=None, N=32, seed=0, alpha=1e-3, device="cuda"):
        self.model1 = model1.to(device)
        self.opt = ChainerAdam(self.model1, alpha=alpha)
        # one seed for every rank: the opponent / handicap draws and the Philox key of the
        # games (keyed by GLOBAL game id) must not depend on the rank
        seed = idist.broadcast_object(seed)
        self.N, self.seed, self.device = N, seed, device
        self.pool_dir = pool_dir
        self.rs = np.random.RandomState(seed)
        self.models, self.cnt, self.set_index = 1, 0, 0
        self.log = []
        self.gather_seconds = 0.0   # time spent in the all-gathers of the sets' tuples so far (N > 1)
        self.sync_replicas()

    @classmethod
    def from_checkpoint(cls, pool_dir, k, models=None, **kw):
        """Resume as the reference's main() does (src/train_rl.py:22-30): the learner is
        pool_dir/model<k>.npz, the optimizer state pool_dir/optimizers/<k>.npz (Adam's t and
        moments), `models` the number of the next snapshot (--models; default k + 1), cnt = 0."""
        model = network.SLPolicy().load_npz(os.path.join(pool_dir, "model%d.npz" % k))
        tr = cls(model, pool_dir=pool_dir, **kw)
        tr.opt.load_npz(os.path.join(pool_dir, "optimizers", "%d.npz" % k))
        tr.models = k + 1 if models is None else models
        tr.sync_replicas()
        return tr

    def sync_replicas(self):
        """Every rank takes rank 0's parameters and Adam moments."""
        # (the parameters themselves, not their .data: broadcast_tensors copies under no_grad,
        # which bumps p._version -- the key of every cached weight layout and captured graph)
        ts = list(self.model1.parameters())
        for n, _ in self.model1.named_parameters():
            ts.extend(self.opt.state[n])
        idist.broadcast_tensors(ts)

    def pick_opponent(self):
        """np.random.choice(glob('../models/RL/*.npz')) (src/train_rl.py:33-37)."""
        paths = sorted(glob.glob(os.path.join(self.pool_dir, "*.npz"))) if self.pool_dir else []
        if paths:
            # a snapshot never changes once it is in the pool (written under a temporary name, then renamed): its
            # module -- parameters on the device, the search-path weight pieces derived from them -- is kept, the most
            # recently used POOL_CACHE of them (reading the file and splitting the weights again was 3.7 ms of a 22 ms set)
            path = paths[self.rs.randint(len(paths))]
            key = (path, os.path.getmtime(path))
            cache = self.__dict__.setdefault("_pool_cache", collections.OrderedDict())
            m2 = cache.pop(key, None)
            if m2 is None:
                m2 = network.SLPolicy().to(self.device).eval()
                m2.load_npz(path)
                while len(cache) >= POOL_CACHE:
                    cache.popitem(last=False)
            cache[key] = m2
            return m2
        # no pool yet: self-play against the current weights -- the learner itself plays both colours (its weight pieces
        # are split once; a copy would be re-split every set)
        return self.model1

    def play_set(self, model2):
        """2N games; odd games carry the handicap stone (src/train_rl.py:41-47).
        Games shard over the ranks; returns the gathered tuples of the whole set."""
        n_total = 2 * self.N
        lo, hi = idist.shard_range(n_total)
        cells = np.array(engine.HANDICAP_CELLS)[self.rs.randint(4, size=n_total)]
        hc = np.where(np.arange(n_total) % 2 == 1, np.uint64(1) << cells.astype(np.uint64),
                      np.uint64(0)).astype(np.uint64)
        self.model1.eval()
        r = rl_self_play.play_batch(self.model1, model2, hi - lo,
                                    handicap=ops.bits_to_tensor(hc[lo:hi], self.device),
                                    seed=self.seed, game_id_base=self.set_index * n_total + lo)
        T1, B = r["action"].shape
        # the rows in which the learner moved, in (turn, game) order -- ONE compaction (a device-to-host count) for all
        # columns
        at = torch.nonzero(r["action"].reshape(-1) >= 0).reshape(-1)
        cols = dict(own=r["own"].reshape(-1)[at], opp=r["opp"].reshape(-1)[at], action=r["action"].reshape(-1)[at],
                    z=r["z"][at % B])
        if idist.world_size() == 1:
            return cols, int((r["z"] == 1).sum().item())
        # row key = (turn, global game): sorted by it the gathered batch is the one a single rank
        # playing all 2N games records (turn-major), whatever the number of ranks -- the update then
        # sums the same rows in the same order
        cols["key"] = (at // B) * n_total + (lo + at % B)
        t0 = time.perf_counter()
        tup = _canonical(idist.gather_tuples(cols))
        wins = idist.gather_tuples(dict(win=(r["z"] == 1).to(torch.int8)))["win"]
        n_win = int(wins.sum().item())                   # (a read-back: the gathers have completed)
        self.gather_seconds += time.perf_counter() - t0  # (bench.py: `gather_ms` of the weak-scaling set)
        return tup, n_win

    def _update(self, own, opp, actions, rewards):
        """src/train_rl.py:55-66 on a gathered batch: loss, backward, Adam step; every replica
        then takes rank 0's parameters (one collective)."""
        # (the split-f16 kernels clamp at the f16 range like the three-piece forward: a model set to `split3 = False`
        # -- the documented remedy when activations leave that range -- takes float32 autograd here as well)
        if (NATIVE_GRAD and isinstance(self.model1, network.SLPolicy) and getattr(self.model1, "split3", False)
                and own.is_cuda):
            loss = self.model1.reinforce_grads(own, opp, actions, rewards)
            # the forward's saturation word and the loss in ONE read-back, BEFORE Adam is applied and broadcast: the
            # gradients of a clamped net (or of a row with an action outside 0 .. 63) must not reach the parameters
            flag = self.model1._overflow_flag(own.device)
            loss_v, bad = torch.stack([loss.to(torch.float64), flag[0].to(torch.float64)]).tolist()
            if bad:
                flag.zero_()
                raise _lib.IagoError(
                    "REINFORCE update: %s; no update was applied.  Set `model1.split3 = False` (float32 forward and "
                    "autograd update) or IAGO_NATIVE_GRAD=0 (float32 autograd update only)"
                    % ("an action lies outside 0 .. 63" if int(bad) & 2 else
                       "an activation of the update's forward left the f16 range of the split kernels (|a| > 65000) "
                       "or is NaN"))
            self.opt.update()
            idist.broadcast_tensors(list(self.model1.parameters()))  # replicas stay identical
            return torch.tensor(loss_v, dtype=torch.float32)
        self.model1.train()
        for p in self.model1.parameters():
            p.grad = None
        # batch shapes MIOpen gets to see: multiples of a quarter of the batch's power-of-two bucket, at least 512 -- a
        # set's ~1,920 rows always run as 2,048, a 64-game PV-MCTS round's ~3,840 as 4,096 until it falls below 3,073 (a
        # new shape costs a 15 s solver search on a fresh box: seen once inside a 1000-round run with multiples of 512)
        n = int(own.numel())
        granule = max(512, (1 << max(n - 1, 1).bit_length()) // 4)
        loss = reinforce_loss(self.model1, own, opp, actions, rewards, pad_to=granule)
        loss.backward()
        self.opt.update()
        idist.broadcast_tensors(list(self.model1.parameters()))  # replicas stay identical
        return loss

    def step_from_tuples(self, tup, colour=None):
        """One REINFORCE update from the tuples of a PV-MCTS self-play round (BASELINE
        configs[4]: "self-play feeding train_rl.py REINFORCE update on gathered (s, pi, z)"):
        tup = engine.SelfPlayResult.tuples() of THIS rank's games -- own / opp (the searched
        position, own = the mover), move (the move played = argmax of pi), z (the game's result
        from the mover's view), colour (1 / 2).  The rows of all ranks are all-gathered
        (idist.gather_tuples: the reference's list concatenation, src/train_rl.py:48-51) and go
        through the same loss as a policy-vs-policy set (src/train_rl.py:55-66: x = the mover's
        planes, y = the action, r = the result).  colour = 1 keeps the learner's plies only, as
        the reference records them (src/rl_self_play.py:134-138); None = both colours (in PV-MCTS
        self-play both sides are the learner).  The win rate / snapshot gating of step() belongs
        to games against a pool opponent and is not touched.  Returns dict(loss, n_tuples)."""
        keep = slice(None)
        if colour is not None:
            keep = tup["colour"] == colour
        key = tup["turn"].to(torch.int64) * (1 << 32) + tup["game"].to(torch.int64)
        if idist.world_size() > 1:
            # the canonical order needs globally unique game ids: every rank must have built its
            # engine with its own game_id_base (idist.shard_range); with the default 0 everywhere the
            # ranks also play IDENTICAL games, whose duplicate rows would silently enter the update
            gm = tup["game"].to(torch.int64)
            span = torch.stack([gm.min(), gm.max()]) if gm.numel() else torch.tensor([1, 0], device=key.device)
            spans = idist.gather_tuples(dict(lo=span[:1], hi=span[1:]))
            lo, hi = spans["lo"].tolist(), spans["hi"].tolist()
            live = sorted((a, b) for a, b in zip(lo, hi) if a <= b)
            if any(live[i][1] >= live[i + 1][0] for i in range(len(live) - 1)):
                raise ValueError("step_from_tuples: the ranks' game id ranges overlap (%s): build every rank's "
                                 "BatchedMCTS with game_id_base = iago_amd.dist.shard_range(n_games)[0]" % live)
        g = _canonical(idist.gather_tuples(dict(own=tup["own"][keep], opp=tup["opp"][keep],
                                                action=tup["move"][keep], z=tup["z"][keep], key=key[keep])))
        if g["z"].numel() == 0:
            raise ValueError("step_from_tuples: no tuples")
        loss = self._update(g["own"], g["opp"], g["action"], g["z"])
        out = dict(loss=float(loss.item()), n_tuples=int(g["z"].numel()))
        self.model1.check_saturation()
        self.log.append(out)
        return out

    def step(self):
        """One set + one update; returns dict(rate, loss, saved)."""
        model2 = self.pick_opponent()
        tup, result = self.play_set(model2)
        loss = self._update(tup["own"], tup["opp"], tup["action"], tup["z"])
        rate = result / (2 * self.N)
        saved = False
        if rate > 0.5:                                             # src/train_rl.py:71-72
            self.cnt += 1
        if self.cnt > 4 * math.sqrt(self.models) and rate > 0.6:  # src/train_rl.py:73-79
            if self.pool_dir and idist.rank() == 0:
                _atomic_savez(os.path.join(self.pool_dir, "model%d.npz" % self.models),
                              self.model1.npz_dict())
                os.makedirs(os.path.join(self.pool_dir, "optimizers"), exist_ok=True)
                self.opt.save_npz(os.path.join(self.pool_dir, "optimizers", "%d.npz" % self.models))
            if self.pool_dir:
                idist.barrier()  # nobody lists the pool before the new checkpoint is complete
            self.models += 1
            self.cnt = 0
            saved = True
        self.set_index += 1
        out = dict(rate=rate, loss=float(loss.item()), saved=saved, n_tuples=int(tup["z"].numel()),
                   stop=rate < 0.2)                                # src/train_rl.py:80-81
        self.log.append(out)
        return out
