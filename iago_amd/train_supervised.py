"""Supervised trainers of the reference (train_policy.py:10-88, train_value.py:9-70)
on device-resident bitboard data sets: minibatches of 4096, Chainer's Adam with
the WeightDecay(5e-4) hook (train_rl.ChainerAdam), per-epoch test loss.

The reference's policy loss applies F.softmax_cross_entropy to the model's
OUTPUT, which is already a softmax (train_policy.py:59-60, network.py:46) -- the
same double softmax as in REINFORCE; it is reproduced here.  Data sets are
(own, opp, label) tensors with own = the side to move; the reference's
[x==1, x==2] planes of "white(2) to play" boards (load.py:41-47) are
encode_planes(own = the 2-stones, opp = the 1-stones).  Data preparation
(load.py) and the 8-fold augmentation are ops.augment8.
"""
import torch
import torch.nn.functional as F

from . import ops
from .train_rl import ChainerAdam

MINIBATCH = 4096  # train_policy.py:43, train_value.py:33


def policy_loss(model, own, opp, actions):
    pred = model(ops.encode_planes(own, opp))                 # probabilities
    return F.cross_entropy(pred, actions.to(torch.int64)), pred  # softmax_cross_entropy(pred, y)


def value_loss(model, own, opp, results):
    pred = model(ops.encode_planes(own, opp))
    return F.mse_loss(pred, results.to(torch.float32)), pred   # mean_squared_error


class SupervisedTrainer(object):
    """kind = 'policy' (SLPolicy / RolloutPolicy, labels = actions) or 'value'
    (Value, labels = results)."""

    def __init__(self, model, kind, seed=0, device="cuda"):
        if kind not in ("policy", "value"):
            raise ValueError("kind must be 'policy' or 'value'")
        self.model, self.kind = model.to(device), kind
        self.opt = ChainerAdam(self.model)                     # optimizers.Adam() + WeightDecay(5e-4)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.loss_fn = policy_loss if kind == "policy" else value_loss

    def epoch(self, own, opp, labels):
        """One shuffled sweep (train_policy.py:46-62); returns the mean minibatch loss."""
        n = own.numel()
        perm = torch.randperm(n, device=own.device, generator=self.gen)
        self.model.train()
        total, count = 0.0, 0
        for lo in range(0, n, MINIBATCH):
            idx = perm[lo:lo + MINIBATCH]
            for p in self.model.parameters():
                p.grad = None
            loss, _ = self.loss_fn(self.model, own[idx], opp[idx], labels[idx])
            loss.backward()
            self.opt.update()
            total += float(loss.item())
            count += 1
        return total / max(count, 1)

    @torch.no_grad()
    def evaluate(self, own, opp, labels):
        """Test loss (and accuracy for policies), train_policy.py:63-68."""
        self.model.eval()
        n = own.numel()
        if n <= MINIBATCH:
            loss, pred = self.loss_fn(self.model, own, opp, labels)
            out = {"loss": float(loss.item())}
            if self.kind == "policy":
                out["accuracy"] = float((pred.argmax(dim=1) == labels.to(torch.int64)).float().mean())
            return out
        # a test set beyond one minibatch goes through in minibatches (the reference hands the whole set to the net at
        # once: activations of 128 x 64 floats per sample and layer, and a convolution shape MIOpen has not seen --
        # 22 s of solver search for 20,000 samples, measured); the means are the sample-weighted means of the pieces
        total = torch.zeros((), dtype=torch.float64, device=own.device)
        hits = torch.zeros((), dtype=torch.float64, device=own.device)
        for lo in range(0, n, MINIBATCH):
            sl = slice(lo, min(lo + MINIBATCH, n))
            loss, pred = self.loss_fn(self.model, own[sl], opp[sl], labels[sl])
            total += loss.to(torch.float64) * (sl.stop - sl.start)
            if self.kind == "policy":
                hits += (pred.argmax(dim=1) == labels[sl].to(torch.int64)).sum()
        out = {"loss": float((total / n).item())}
        if self.kind == "policy":
            out["accuracy"] = float((hits / n).item())
        return out
