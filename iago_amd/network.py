"""SLPolicy / RolloutPolicy / Value as PyTorch(-ROCm) modules.

Same architecture, call signature and parameter names as the reference's
network.py:5-96 (Chainer): `model(x)` with x (B,2,8,8) float32 planes
(channel 0 = opponent of the side to move, channel 1 = side to move) returns
(B,64) softmax probabilities for the policies and (B,) values for Value.
`load_npz` / `save_npz` read and write the reference's Chainer npz layout
(models/*.npz: `block1/conv/W`, `conv9/W`, `bias10/b`, `fc10/W`, ...), so the
shipped checkpoints drive these modules unchanged.

On the GPU the 3x3 convolutions run through MIOpen / hipBLASLt (MFMA); the
rollout policy additionally lives fused inside the HIP rollout kernel
(csrc/rollout_kernel.hip).
"""
import math

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class Block(nn.Module):
    """network.py:5-13: Convolution2D(None, out, 3, pad=1) + ReLU."""

    def __init__(self, in_channels, out_channels, ksize=3, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, ksize, padding=pad)

    def forward(self, x):
        if (x.is_cuda and not torch.is_grad_enabled() and x.shape[-1] == 8 and x.shape[-2] == 8
                and x.dtype == torch.float32 and self.conv.bias is not None
                and not torch.is_autocast_enabled()):
            # inference on the GPU: MIOpen convolution without bias + one fused
            # bias/ReLU pass (HIP) instead of PyTorch's two elementwise kernels
            from . import ops
            y = F.conv2d(x, self.conv.weight, None, padding=self.conv.padding)
            return ops.bias_relu_(y.contiguous(), self.conv.bias)
        return F.relu(self.conv(x))


def _lecun_normal_(w):
    # Chainer's default initialiser: LeCunNormal, std = sqrt(1 / fan_in); biases zero
    # (Chainer source is not in the reference tree: from its documentation, unpinned)
    fan_in = w[0].numel()
    with torch.no_grad():
        w.normal_(0.0, 1.0 / math.sqrt(fan_in))


class _NpzMixin(object):
    """Chainer `serializers.save_npz/load_npz` interop (MCTS.py:83-85,
    mcts_self_play.py:19, src/train_rl.py:23,76)."""

    def _npz_map(self):
        out = {}
        for name, p in self.named_parameters():
            parts = name.split(".")
            leaf = {"weight": "W", "bias": "b", "b": "b"}[parts[-1]]
            out["/".join(parts[:-1] + [leaf])] = p
        return out

    def load_npz(self, path_or_dict, prefix=""):
        src = np.load(path_or_dict) if isinstance(path_or_dict, str) else path_or_dict
        keys = set(src.files) if hasattr(src, "files") else set(src.keys())
        m = self._npz_map()
        missing = [k for k in m if prefix + k not in keys]
        if missing:
            raise KeyError("npz is missing %s (has %s)" % (missing, sorted(keys)[:6]))
        with torch.no_grad():
            for k, p in m.items():
                a = np.asarray(src[prefix + k], dtype=np.float32)
                if tuple(a.shape) != tuple(p.shape):
                    raise ValueError("%s: shape %s, expected %s" % (k, a.shape, tuple(p.shape)))
                p.copy_(torch.from_numpy(a))
        return self

    def npz_dict(self):
        return {k: p.detach().cpu().numpy() for k, p in self._npz_map().items()}

    def save_npz(self, path):
        np.savez(path, **self.npz_dict())

    def reset_parameters_chainer(self, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("weight"):
                _lecun_normal_(p)
            else:
                nn.init.zeros_(p)
        return self


class Bias(nn.Module):
    """L.Bias(shape=(64,)): adds a learned vector along axis 1 (network.py:32,57)."""

    def __init__(self, n):
        super().__init__()
        self.b = nn.Parameter(torch.zeros(n))

    def forward(self, x):
        return x + self.b


class SLPolicy(nn.Module, _NpzMixin):
    """network.py:15-47."""

    def __init__(self):
        super().__init__()
        self.block1 = Block(2, 64)
        for k in range(2, 9):
            setattr(self, "block%d" % k, Block(64 if k == 2 else 128, 128))
        self.conv9 = nn.Conv2d(128, 1, 1, bias=False)
        self.bias10 = Bias(64)
        self.reset_parameters_chainer()

    def logits(self, x):
        h = x
        for k in range(1, 9):
            h = getattr(self, "block%d" % k)(h)
        h = self.conv9(h).reshape(-1, 64)
        return self.bias10(h)

    # Small inference batches on the GPU (the expansions of a playout: a few dozen boards)
    # run through the float32 matrix-unit kernels of csrc/conv_kernels.hip, which spread
    # one board over four workgroups; larger batches through MIOpen.
    F32_MAX_BATCH = 192

    def _use_f32_kernels(self, x):
        return (x.is_cuda and not self.training and not torch.is_grad_enabled()
                and x.dtype == torch.float32 and 0 < x.shape[0] <= self.F32_MAX_BATCH
                and not torch.is_autocast_enabled())

    def forward(self, x):
        if self._use_f32_kernels(x):
            from . import ops
            h = _f32_trunk(self, x)
            return ops.policy_head(h, self.conv9.weight.reshape(128), self.bias10.b)
        return F.softmax(self.logits(x), dim=1)

    def forward_counted(self, x, n_dev):
        """forward(x[:k]) with k = min(len(x), *n_dev) known only on the device (int32 CUDA
        word): the float32 matrix-unit kernels on a fixed grid, rows past k untouched.  What
        the search engine calls on the leaves a playout expands (MCTS.py:109-121), so that
        a whole playout is one hipGraph replay.  Inference only."""
        if not (x.is_cuda and x.dtype == torch.float32 and not self.training):
            raise ValueError("forward_counted: float32 CUDA planes, eval mode")
        from . import ops
        with torch.no_grad():
            h = _f32_trunk(self, x, n_dev)
            return ops.policy_head(h, self.conv9.weight.reshape(128), self.bias10.b, n_dev)

    # The whole net in one launch with three f16 pieces per float32 operand
    # (csrc/conv_policy_kernel.hip) instead of the per-layer float32 kernels.
    split3 = os.environ.get("IAGO_POLICY_SPLIT3", "1") != "0"
    # launches per forward: two half-nets leave the CUs to the playouts' kernels sooner than one
    # launch that holds them for the whole net (tuning knob: DESIGN.md)
    split3_parts = int(os.environ.get("IAGO_POLICY_PARTS", "2"))

    def _split3_layers(self):
        from . import ops
        ws = [getattr(self, "block%d" % k).conv for k in range(2, 9)]
        key = tuple((c.weight._version, c.weight.data_ptr(), c.bias._version) for c in ws)
        hit = self.__dict__.get("_split3_cache")
        if hit is None or hit[0] != key:
            pieces = ops.split_weights3_many([c.weight for c in ws])
            layers = [p3 + (c.bias.detach().float().contiguous(),) for p3, c in zip(pieces, ws)]
            hit = (key, layers, None)
            self.__dict__["_split3_cache"] = hit
        return hit[1]

    def _bwd_layers(self):
        """Blocks 2..8 in the backward-data form of iago_policy_reinforce_grad (transposed, flipped, two f16 pieces),
        rebuilt when the weights change."""
        from . import ops
        ws = [getattr(self, "block%d" % k).conv.weight for k in range(2, 9)]
        key = tuple((w._version, w.data_ptr()) for w in ws)
        hit = self.__dict__.get("_bwd_cache")
        if hit is None or hit[0] != key:
            hit = (key, ops.split_weights_transposed_many(ws))
            self.__dict__["_bwd_cache"] = hit
        return hit[1]

    GRAD_CHUNK_ROWS = int(os.environ.get("IAGO_GRAD_CHUNK_ROWS", "4096"))   # 1.25 GB of scratch per chunk

    def reinforce_grads(self, own, opp, action, reward, n_mean=None, probs=None):
        """src/train_rl.py:61-65 on the matrix units in split-f16 arithmetic (iago_policy_reinforce_grad):
        cleargrads + loss.backward() for loss = mean(softmax_cross_entropy(self(x), action) * reward); every
        parameter's .grad is overwritten.  own / opp: the recorded positions (own = the mover); action in 0 .. 63 (an
        action outside raises bit 1 of the module's overflow word: ReinforceTrainer reads it before Adam).  Returns the loss (0-dim device tensor).  The first two f16 pieces of the search
        path's three-piece weights are the forward's.  The kernels' scratch (308 KB per row + 247 MB, kept between
        calls) is freed by ops.release_grad_workspace()."""
        from . import ops
        convs = [getattr(self, "block%d" % k).conv for k in range(2, 9)]
        params = [self.block1.conv.weight, self.block1.conv.bias, self.conv9.weight, self.bias10.b]
        for c in convs:
            params += [c.weight, c.bias]
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("reinforce_grads: float32 CUDA parameters expected")
            if p.grad is None:
                p.grad = torch.empty_like(p)
        layers = [(hi, mid, bias) for hi, mid, lo, bias in self._split3_layers()]
        grads = dict(w1=self.block1.conv.weight.grad, b1=self.block1.conv.bias.grad,
                     w=[c.weight.grad for c in convs], b=[c.bias.grad for c in convs],
                     w9=self.conv9.weight.grad, b10=self.bias10.b.grad)
        n = own.numel()
        n_mean = n if n_mean is None else n_mean
        own, opp = own.contiguous(), opp.contiguous()
        action, reward = action.to(torch.int32).contiguous(), reward.to(torch.float32).contiguous()
        layers_t = self._bwd_layers()

        def rows(lo, hi):
            return ops.policy_reinforce_grad(own[lo:hi], opp[lo:hi], action[lo:hi], reward[lo:hi], n_mean,
                                             self.block1.conv.weight.detach(), self.block1.conv.bias.detach(), layers,
                                             layers_t, self.conv9.weight.detach(), self.bias10.b.detach(), grads,
                                             probs=None if probs is None else probs[lo:hi],
                                             overflow=self._overflow_flag(own.device))
        # The kernels' scratch is 308 KB per row: a PV-MCTS round of 1024 games (~61 k rows, both colours) would take
        # 18 GB at once.  Larger batches run in chunks of GRAD_CHUNK_ROWS rows, every chunk dividing by the same
        # n_mean, the chunks' gradients added in chunk order (deterministic); a batch within one chunk is one call.
        chunk = self.GRAD_CHUNK_ROWS
        if n <= chunk:
            return rows(0, n)
        total = [torch.zeros_like(p) for p in params]
        loss = None
        for lo in range(0, n, chunk):
            part = rows(lo, min(n, lo + chunk))
            loss = part if loss is None else loss + part
            for t, p in zip(total, params):
                t += p.grad
        for t, p in zip(total, params):
            p.grad.copy_(t)
        return loss

    def _split3_template(self):
        """iago_policy_split3_args with this module's weights, rebuilt when they change."""
        from . import ops
        layers = self._split3_layers()
        small = (self.block1.conv.weight, self.block1.conv.bias, self.conv9.weight, self.bias10.b)
        key2 = tuple((t._version, t.data_ptr()) for t in small)
        hit = self.__dict__["_split3_cache"]
        if hit[2] is None or hit[2][0] != key2:
            tmpl = ops.policy_split3_prepare(self.block1.conv.weight, self.block1.conv.bias, layers,
                                             self.conv9.weight, self.bias10.b)
            hit = (hit[0], hit[1], (key2, tmpl))
            self.__dict__["_split3_cache"] = hit
        return hit[2][1]

    def _overflow_flag(self, device):
        """Device word the three-piece kernel raises when an activation leaves the f16 range
        (|a| > 65000: the 'hi' piece cannot hold it) or is NaN."""
        f = self.__dict__.get("_ovf")
        if f is None or f.device != device:
            f = torch.zeros(1, dtype=torch.int32, device=device)
            self.__dict__["_ovf"] = f
        return f

    def check_saturation(self):
        """Raise if any three-piece forward since the last check saturated (one host sync; the
        search engine calls it once per search).  `split3 = False` evaluates in float32, with
        the reference's unbounded range."""
        f = self.__dict__.get("_ovf")
        if f is not None and int(f.item()) != 0:
            f.zero_()
            from ._lib import IagoError
            raise IagoError("SLPolicy: an activation left the f16 range of the three-piece split "
                            "(|a| > 65000) or is NaN; the priors of this search are saturated.  Set "
                            "`policy.split3 = False` to evaluate in float32")

    def forward_boards_split3(self, own, opp, index=None, n=None, n_dev=None, overflow=None):
        """The move distributions of boards (own = side to move; rows index[0..n) when a gather
        list is given) through the one-launch kernel.  Inference only.  overflow: the device
        word to raise on saturation (default: the module's, see check_saturation)."""
        if self.training or not own.is_cuda:
            raise ValueError("forward_boards_split3: CUDA boards, eval mode")
        from . import ops
        if overflow is None:
            overflow = self._overflow_flag(own.device)
        rows = own.numel() if n is None else n
        scratch = self._split3_scratch(own.device, rows) if self.split3_parts > 1 else None
        with torch.no_grad():
            return ops.policy_forward_split3_prepared(self._split3_template(), own, opp, n=n, index=index,
                                                      n_dev=n_dev, overflow=overflow, parts=self.split3_parts,
                                                      scratch=scratch)

    # rows of a multi-launch forward's scratch buffer: 4096 x 51,200 B = 205 MB per stream that
    # calls the net, whatever the batch (a longer batch runs as chunks of this many rows)
    SPLIT3_SCRATCH_ROWS = 4096

    def release_scratch(self):
        """Drop this module's references to the scratch buffers of its multi-launch forwards.
        engine.BatchedMCTS._capture calls it before every (re-)capture; an engine keeps the buffers
        its own captured graphs address alive itself (BatchedMCTS._scratch_refs)."""
        self.__dict__.pop("_split3_scratch_pool", None)

    def _split3_scratch(self, device, rows):
        """The buffer a multi-launch forward parks the boards' activations in: one per stream
        (calls on different streams must not share it), min(rows, SPLIT3_SCRATCH_ROWS) rows,
        kept until release_scratch() -- a captured hipGraph of the search engine holds its
        address."""
        from . import ops
        rows = min(rows, self.SPLIT3_SCRATCH_ROWS)
        key = (str(device), torch.cuda.current_stream(device).cuda_stream)
        pool = self.__dict__.setdefault("_split3_scratch_pool", {})
        buf = pool.get(key)
        if buf is None or buf.shape[0] < rows:
            # (a larger request replaces the buffer; the old one stays referenced so that graphs
            # captured with it keep valid memory)
            if buf is not None:
                pool.setdefault("retired", []).append(buf)
            buf = pool[key] = torch.empty((rows, ops.POLICY_SCRATCH_ROW_BYTES), dtype=torch.uint8, device=device)
        return buf

    def search_args(self, own, opp, probs):
        """iago_policy_split3_args for the persistent search (iago_mcts_search_persistent): this
        module's three-piece weights, rows read from own / opp, distributions written to probs
        ((rows, 64) float32).  Returns (args, tensors to keep alive)."""
        if self.training or not own.is_cuda:
            raise ValueError("search_args: CUDA, eval mode")
        from . import _lib
        a = _lib.PolicySplit3Args.from_buffer_copy(self._split3_template())
        a.own, a.opp, a.n, a.probs = own.data_ptr(), opp.data_ptr(), own.numel(), probs.data_ptr()
        a.parts = 1
        a.overflow = self._overflow_flag(own.device).data_ptr()
        return a, (self.__dict__["_split3_cache"], own, opp, probs)

    def forward_counted_boards(self, own, opp, index, n, n_dev):
        """forward_counted on make_state_var of boards index[0..n) (own = side to move) without
        materialising the planes."""
        if self.training or not own.is_cuda:
            raise ValueError("forward_counted_boards: CUDA boards, eval mode")
        from . import ops
        if self.split3:
            return self.forward_boards_split3(own, opp, index, n, n_dev)
        with torch.no_grad():
            h = _f32_trunk(self, None, n_dev, boards=(own, opp, index, n))
            return ops.policy_head(h, self.conv9.weight.reshape(128), self.bias10.b, n_dev)


def _f32_weights(module, k):
    """Cached [4][9][cin][32] layout of block k's weight (ops.f32_weights)."""
    from . import ops
    w = getattr(module, "block%d" % k).conv.weight
    cache = module.__dict__.setdefault("_f32_cache", {})
    key = (w._version, w.data_ptr(), str(w.device))
    hit = cache.get(k)
    if hit is None or hit[0] != key:
        hit = (key, ops.f32_weights(w))
        cache[k] = hit
    return hit[1]


def _f32_trunk(module, x, n_dev=None, boards=None):
    """blocks 1..8 of SLPolicy / Value in float32 on the matrix units (small batches).
    boards = (own, opp, index, n): block1 straight from the bitboards instead of planes x."""
    from . import ops
    if boards is not None:
        own, opp, index, n = boards
        h = ops.stem_f32_boards(own, opp, index, module.block1.conv.weight, module.block1.conv.bias, n, n_dev)
    else:
        h = ops.stem_f32(x.contiguous(), module.block1.conv.weight, module.block1.conv.bias, n_dev)
    for k in range(2, 9):
        h = ops.conv3x3_f32(h, _f32_weights(module, k), getattr(module, "block%d" % k).conv.bias, n_dev)
    return h


class RolloutPolicy(nn.Module, _NpzMixin):
    """network.py:49-64."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(2, 1, 3, padding=1, bias=False)
        self.bias2 = Bias(64)
        self.reset_parameters_chainer()

    def logits(self, x):
        return self.bias2(self.conv1(x).reshape(-1, 64))

    def forward(self, x):
        return F.softmax(self.logits(x), dim=1)

    def kernel_weights(self):
        """(w18, b64) numpy arrays for ops.RolloutWeights."""
        return (self.conv1.weight.detach().cpu().numpy().reshape(18),
                self.bias2.b.detach().cpu().numpy().reshape(64))


class Value(nn.Module, _NpzMixin):
    """network.py:66-96: no tanh, no activation after fc10, dropout 0.4 in
    training mode only (MCTS.py:86 runs it with train=False)."""

    def __init__(self):
        super().__init__()
        self.block1 = Block(2, 64)
        for k in range(2, 9):
            setattr(self, "block%d" % k, Block(64 if k == 2 else 128, 128))
        self.block9 = Block(128, 1)
        self.fc10 = nn.Linear(64, 128, bias=False)
        self.fc11 = nn.Linear(128, 1, bias=False)
        self.reset_parameters_chainer()

    # Inference on the GPU runs through csrc/conv_kernels.hip: blocks 2..8 (>98 % of the
    # FLOPs) as split-f16 MFMA convolutions (22-bit products, float32 accumulation),
    # block1 and block9 + fc10 + fc11 as fused float32 kernels; the whole forward stays
    # within 1e-6 of the float32 one.  Set to False for MIOpen's
    # float32 convolutions everywhere.
    split_f16 = True
    SPLIT_MIN_BATCH = 192   # below: float32 matrix-unit kernels (126 us at <= 64 boards, 225 us at 128)

    def _split_weights(self, k):
        from . import ops
        w = getattr(self, "block%d" % k).conv.weight
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (w._version, w.data_ptr(), str(w.device))
        hit = cache.get(k)
        if hit is None or hit[0] != key:
            hit = (key,) + ops.split_weights(w)
            cache[k] = hit
        return hit[1], hit[2]

    def _overflow_flag(self, device):
        """Device word the split-f16 kernels raise when an activation leaves the f16 range
        (|a| > 65000) or is NaN: the forward is then saturated, not the reference's."""
        f = self.__dict__.get("_ovf")
        if f is None or f.device != device:
            f = torch.zeros(1, dtype=torch.int32, device=device)
            self.__dict__["_ovf"] = f
        return f

    def check_saturation(self):
        """Raise if any split-f16 forward since the last check saturated (one host sync; the
        search engine calls it once per search).  The float32 path (`split_f16 = False`) has
        the reference's unbounded range."""
        f = self.__dict__.get("_ovf")
        if f is not None and int(f.item()) != 0:
            f.zero_()
            from ._lib import IagoError
            raise IagoError("Value net: an activation left the split-f16 range (|a| > 65000) or is "
                            "NaN; the results of this search are saturated.  Set "
                            "`value.split_f16 = False` to evaluate in float32")

    def _use_split(self, x):
        return (self.split_f16 and x.is_cuda and not self.training and not torch.is_grad_enabled()
                and x.dtype == torch.float32 and x.shape[0] >= self.SPLIT_MIN_BATCH
                and not torch.is_autocast_enabled())

    def _split_trunk_head(self, a):
        """Blocks 2..8 + head on SplitActs: the three-launch form (the fused forward below is
        what inference runs; this is its cross-check and the path of callers that hold
        split activations)."""
        from . import ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        a = ops.conv3x3_split_trunk(a, layers, overflow=self._overflow_flag(a.hi.device))  # blocks 2..8
        return ops.value_head(a, self.block9.conv.weight, self.block9.conv.bias,
                              self.fc10.weight, self.fc11.weight)

    def _head_weights(self):
        from . import ops
        w = self.block9.conv.weight
        key = (w._version, w.data_ptr(), str(w.device))
        hit = self.__dict__.get("_head_cache")
        if hit is None or hit[0] != key:
            hit = (key, ops.split_head_weights(w))
            self.__dict__["_head_cache"] = hit
        return hit[1]

    fused = True   # one launch for the whole net (iago_value_forward_split); False: stem, trunk, head

    def _forward_split(self, x, device):
        """x: planes (n,2,8,8) or (own, opp).  The whole net in one launch."""
        from . import ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        return ops.value_forward_split(x, self.block1.conv.weight, self.block1.conv.bias, layers,
                                       self._head_weights(), self.block9.conv.weight, self.block9.conv.bias,
                                       self.fc10.weight, self.fc11.weight,
                                       overflow=self._overflow_flag(device))

    def forward_boards_counted(self, own, opp, index, n_dev, out, rollout=None):
        """The net on boards index[0 .. *n_dev) only (int64 gather list and int32 count on the
        device: the leaves of a playout that have no cached value), values to out[index[i]];
        one enqueue whatever the count (iago_value_forward_split picks its variant on the
        device).  Split-f16 path for every count."""
        if not (self.split_f16 and own.is_cuda and not self.training and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled()):
            raise ValueError("forward_boards_counted: CUDA boards, eval mode, split_f16")
        from . import ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        return ops.value_forward_split((own, opp), self.block1.conv.weight, self.block1.conv.bias, layers,
                                       self._head_weights(), self.block9.conv.weight, self.block9.conv.bias,
                                       self.fc10.weight, self.fc11.weight,
                                       overflow=self._overflow_flag(own.device), index=index, n_dev=n_dev, out=out,
                                       rollout=rollout)

    def search_args(self, own, opp, out):
        """iago_value_split_args for the persistent search (iago_mcts_search_persistent): this module's
        split-f16 weights, rows read from own / opp, values written to out (one slot per workgroup of
        the launch).  Returns (args, tensors to keep alive while the args are in use)."""
        if not (self.split_f16 and own.is_cuda and not self.training):
            raise ValueError("search_args: CUDA, eval mode, split_f16")
        import ctypes as C
        from . import _lib, ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        head = self._head_weights()
        a = _lib.ValueSplitArgs()
        a.own, a.opp, a.n, a.out = own.data_ptr(), opp.data_ptr(), own.numel(), out.data_ptr()
        ops.value_split_weights(a, self.block1.conv.weight, self.block1.conv.bias, layers, head,
                                self.block9.conv.bias, self.fc10.weight, self.fc11.weight)
        a.overflow = self._overflow_flag(own.device).data_ptr()
        return a, (layers, head, own, opp, out)

    def forward_boards_batch(self, own, opp, n_dev, out, boards_per_workgroup=2, max_workgroups=192):
        """The net on boards 0 .. *n_dev (device-side count) of own / opp, values to out[i], as a batch
        OFF the playouts' critical path (the search's value look-ahead): boards_per_workgroup boards
        share a workgroup's weight stream, at most max_workgroups workgroups walk the rows
        (iago_value_forward_batch).  Bit-identical per board to every other split-f16 forward."""
        if not (self.split_f16 and own.is_cuda and not self.training and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled()):
            raise ValueError("forward_boards_batch: CUDA boards, eval mode, split_f16")
        from . import ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        return ops.value_forward_split((own, opp), self.block1.conv.weight, self.block1.conv.bias, layers,
                                       self._head_weights(), self.block9.conv.weight, self.block9.conv.bias,
                                       self.fc10.weight, self.fc11.weight,
                                       overflow=self._overflow_flag(own.device), n_dev=n_dev, out=out,
                                       batch=(boards_per_workgroup, max_workgroups))

    def forward_boards_async(self, own, opp, out, rollout, async_ref):
        """One game-asynchronous search step's leaf evaluation (iago_value_rollout_async): the
        rollouts of the games that descended in this step + one piece of this net for every queue
        of fresh leaves; values of the leaves whose last piece ran land in out[game]."""
        if not (self.split_f16 and own.is_cuda and not self.training and not torch.is_grad_enabled()
                and not torch.is_autocast_enabled()):
            raise ValueError("forward_boards_async: CUDA boards, eval mode, split_f16")
        from . import ops
        layers = [self._split_weights(k) + (getattr(self, "block%d" % k).conv.bias,) for k in range(2, 9)]
        return ops.value_forward_split((own, opp), self.block1.conv.weight, self.block1.conv.bias, layers,
                                       self._head_weights(), self.block9.conv.weight, self.block9.conv.bias,
                                       self.fc10.weight, self.fc11.weight,
                                       overflow=self._overflow_flag(own.device), out=out, rollout=rollout,
                                       async_ref=async_ref)

    def forward_boards(self, own, opp):
        """forward(make_state_var(...)) for int64 bitboards (own = side to move) without the
        planes tensor, when the split-f16 path applies; None otherwise."""
        if not (self.split_f16 and own.is_cuda and not self.training and not torch.is_grad_enabled()
                and own.numel() >= self.SPLIT_MIN_BATCH and not torch.is_autocast_enabled()):
            return None
        if self.fused:
            return self._forward_split((own, opp), own.device)
        from . import ops
        return self._split_trunk_head(ops.value_stem_boards(own, opp, self.block1.conv.weight,
                                                            self.block1.conv.bias,
                                                            overflow=self._overflow_flag(own.device)))

    def forward(self, x):
        if self._use_split(x) and self.fused:
            return self._forward_split(x.contiguous(), x.device)
        elif self._use_split(x):
            from . import ops
            a = ops.value_stem(x.contiguous(), self.block1.conv.weight, self.block1.conv.bias,
                               overflow=self._overflow_flag(x.device))
            return self._split_trunk_head(a)
        elif (self.split_f16 and x.is_cuda and not self.training and not torch.is_grad_enabled()
              and x.dtype == torch.float32 and x.shape[0] > 0 and not torch.is_autocast_enabled()):
            # small batches (single-game serving): float32 matrix-unit kernels
            from . import ops
            return ops.value_head(ops.split_nchw(_f32_trunk(self, x), overflow=self._overflow_flag(x.device)),
                                  self.block9.conv.weight,
                                  self.block9.conv.bias, self.fc10.weight, self.fc11.weight)
        else:
            h = x
            for k in range(1, 10):
                h = getattr(self, "block%d" % k)(h)
        h = self.fc10(h.reshape(-1, 64))
        h = F.dropout(h, 0.4, training=self.training)
        return self.fc11(h).reshape(-1)
