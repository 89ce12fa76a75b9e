"""Tensor-level wrappers over the C ABI (include/iago_hip.h).

Boards are int64 CUDA tensors holding the 64-bit bitboards (bit a = row*8+col;
`own` = side to move).  Every function launches on torch's current stream and
requires CUDA tensors: there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import IAGO_MAX_TURNS, IAGO_ROLLOUT_TABLE_FLOATS, RolloutArgs, check


try:   # (torch.cuda.current_stream() costs ~8 us of Python per call: 40 % of a small launch's host time)
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:  # a torch without these internals
    _raw_stream = _cur_device = None


def _stream():
    """The current HIP stream of the current device as a void* for the C ABI."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.IagoError("%s must be a CUDA tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return C.c_void_p(t.data_ptr())


def bits_to_tensor(values, device="cuda"):
    """Python ints / numpy uint64 -> int64 tensor with the same bit patterns."""
    a = np.asarray(values, dtype=np.uint64).reshape(-1)
    return torch.from_numpy(a.view(np.int64).copy()).to(device)


def tensor_to_bits(t):
    """int64 tensor -> numpy uint64 array."""
    return t.detach().cpu().numpy().view(np.uint64)


def legal_moves(own, opp):
    """Bit mask of legal moves per board (game.py:210-235, rl_env.py:114-138)."""
    n = own.numel()
    out = torch.empty_like(own)
    check(_lib.lib().iago_legal_moves(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                      _dev(out, torch.int64, "legal"), n, _stream()),
          "iago_legal_moves")
    return out


def apply_moves(own, opp, action):
    """In place place_stone (game.py:180-207); action int8, -1 = pass."""
    n = own.numel()
    if action.numel() != n or opp.numel() != n:
        raise ValueError("own/opp/action sizes differ")
    check(_lib.lib().iago_apply_moves(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                      _dev(action, torch.int8, "action"), n, _stream()),
          "iago_apply_moves")
    return own, opp


def play_turn(own, opp, action, active, stone_num, pass_flg, done, close_pair, legal, active_next):
    """One turn of n lockstep games in place (iago_play_turn): the move, stone_num / pass_flg /
    done, the swap of sides (own = the next mover afterwards), the next mover's legal moves and
    the next turn's `active`.  uint8 flags, int32 stone_num, int8 action."""
    n = own.numel()
    for name, t in (("opp", opp), ("action", action), ("active", active), ("stone_num", stone_num),
                    ("pass_flg", pass_flg), ("done", done), ("legal", legal), ("active_next", active_next)):
        if t.numel() != n:
            raise ValueError("play_turn: %s has %d entries, expected %d" % (name, t.numel(), n))
    if active.data_ptr() == active_next.data_ptr():
        raise ValueError("play_turn: active and active_next may not alias")
    check(_lib.lib().iago_play_turn(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                    _dev(action, torch.int8, "action"), _dev(active, torch.uint8, "active"),
                                    _dev(stone_num, torch.int32, "stone_num"), _dev(pass_flg, torch.uint8, "pass_flg"),
                                    _dev(done, torch.uint8, "done"), 1 if close_pair else 0,
                                    _dev(legal, torch.int64, "legal"), _dev(active_next, torch.uint8, "active_next"),
                                    n, _stream()), "iago_play_turn")


def encode_planes(own, opp, out=None):
    """(n,2,8,8) float32 planes, channel 0 = opp, channel 1 = own (game.py:168-174)."""
    n = own.numel()
    if out is None:
        out = torch.empty((n, 2, 8, 8), dtype=torch.float32, device=own.device)
    elif out.numel() != n * 128:
        raise ValueError("planes buffer has the wrong size")
    check(_lib.lib().iago_encode_planes(_dev(own, torch.int64, "own"),
                                        _dev(opp, torch.int64, "opp"),
                                        _dev(out, torch.float32, "planes"), n, _stream()),
          "iago_encode_planes")
    return out


def _count(n_dev):
    """Optional int32 CUDA word holding a device-side item count (iago_hip.h, `n_dev`)."""
    return _dev(n_dev, torch.int32, "n_dev") if n_dev is not None else None


def encode_planes_indexed(own, opp, index, out, n_dev=None):
    """Planes of boards index[0..k) (int64 tensor) into out[:k] (game.py:168-174); with
    n_dev only the first min(k, *n_dev) rows."""
    k = index.numel()
    if out.numel() < k * 128:
        raise ValueError("planes buffer is too small")
    check(_lib.lib().iago_encode_planes_indexed(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                                _dev(index, torch.int64, "index"),
                                                _dev(out, torch.float32, "planes"), k, _count(n_dev),
                                                _stream()),
          "iago_encode_planes_indexed")
    return out


def judge(own, opp):
    """sign(#own - #opp) as int8 (mcts_self_play.py:113-121)."""
    n = own.numel()
    out = torch.empty(n, dtype=torch.int8, device=own.device)
    check(_lib.lib().iago_judge(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                _dev(out, torch.int8, "z"), n, _stream()), "iago_judge")
    return out


def augment8(own, opp, action):
    """8-fold dihedral augmentation in the reference's order (load.py:56-74):
    returns (8, n) own, opp (int64) and action (int8)."""
    n = own.numel()
    oo = torch.empty((8, n), dtype=torch.int64, device=own.device)
    po = torch.empty((8, n), dtype=torch.int64, device=own.device)
    ao = torch.empty((8, n), dtype=torch.int8, device=own.device)
    check(_lib.lib().iago_augment8(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                   _dev(action, torch.int8, "action"), _dev(oo, torch.int64, "o"),
                                   _dev(po, torch.int64, "p"), _dev(ao, torch.int8, "a"), n,
                                   _stream()), "iago_augment8")
    return oo, po, ao


def bias_relu_(x, bias):
    """In place max(x + bias[c], 0) on a (n, C, 8, 8) float32 tensor: the epilogue
    of network.Block (network.py:9-13) as one pass."""
    if x.dim() != 4 or x.shape[2] != 8 or x.shape[3] != 8:
        raise ValueError("x must be (n, C, 8, 8)")
    check(_lib.lib().iago_bias_relu(_dev(x, torch.float32, "x"), _dev(bias, torch.float32, "bias"),
                                    x.shape[0], x.shape[1], _stream()), "iago_bias_relu")
    return x


# ---- split-f16 convolution stack (csrc/conv_kernels.hip) --------------------------

class SplitActs(object):
    """Activations in split channel blocks: f16 tensors hi, lo of shape
    (n, C/16, 64, 16); value = hi + lo * 2**-11 (include/iago_hip.h)."""

    __slots__ = ("hi", "lo", "channels")

    def __init__(self, hi, lo, channels):
        self.hi, self.lo, self.channels = hi, lo, channels

    @property
    def n(self):
        return self.hi.shape[0]


def split_weights(weight):
    """(cout, cin, 3, 3) float32 conv weight -> (w_hi, w_lo) f16 tensors
    [cin/16][3][3][cout][16] as iago_conv3x3_split expects them."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cin % 16 or cout != 128:
        raise ValueError("split_weights: need a (128, 16k, 3, 3) weight")
    w = weight.detach().to(torch.float32).permute(2, 3, 0, 1)          # ky, kx, co, ci
    w = w.reshape(3, 3, cout, cin // 16, 16).permute(3, 0, 1, 2, 4).contiguous()
    hi = w.to(torch.float16)
    lo = ((w - hi.to(torch.float32)) * 2048.0).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


def _flag(overflow):
    """Optional int32 CUDA word the split-f16 kernels raise on saturation (iago_hip.h)."""
    return _dev(overflow, torch.int32, "overflow") if overflow is not None else None


def split_nchw(x, overflow=None):
    """(n, C, 8, 8) float32 -> SplitActs."""
    n, c = x.shape[0], x.shape[1]
    if x.dim() != 4 or x.shape[2] != 8 or x.shape[3] != 8 or c % 16:
        raise ValueError("x must be (n, 16k, 8, 8)")
    hi = torch.empty((n, c // 16, 64, 16), dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().iago_split_nchw(_dev(x, torch.float32, "x"), _dev(hi, torch.float16, "hi"),
                                     _dev(lo, torch.float16, "lo"), n, c, _flag(overflow), _stream()),
          "iago_split_nchw")
    return SplitActs(hi, lo, c)


def merge_nchw(a):
    """SplitActs -> (n, C, 8, 8) float32."""
    y = torch.empty((a.n, a.channels, 8, 8), dtype=torch.float32, device=a.hi.device)
    check(_lib.lib().iago_merge_nchw(_dev(a.hi, torch.float16, "hi"), _dev(a.lo, torch.float16, "lo"),
                                     _dev(y, torch.float32, "y"), a.n, a.channels, _stream()),
          "iago_merge_nchw")
    return y


def conv3x3_split_trunk(a, layers, overflow=None):
    """Consecutive conv3x3_split layers in one launch.  layers: list of (w_hi, w_lo, bias);
    the first reads SplitActs `a`, each next one its predecessor's output."""
    n = a.n
    descs = (_lib.ConvSplitLayer * len(layers))()
    keep = []
    cur, cin = a, a.channels
    for k, (w_hi, w_lo, bias) in enumerate(layers):
        if w_hi.shape != (cin // 16, 3, 3, 128, 16):
            raise ValueError("layer %d: weight blocks %s do not match %d input channels"
                             % (k, tuple(w_hi.shape), cin))
        hi = torch.empty((n, 8, 64, 16), dtype=torch.float16, device=a.hi.device)
        lo = torch.empty_like(hi)
        d = descs[k]
        d.x_hi, d.x_lo = cur.hi.data_ptr(), cur.lo.data_ptr()
        d.w_hi, d.w_lo = _dev(w_hi, torch.float16, "w_hi").value, _dev(w_lo, torch.float16, "w_lo").value
        d.bias = _dev(bias, torch.float32, "bias").value
        d.y_hi, d.y_lo = hi.data_ptr(), lo.data_ptr()
        d.cin = cin
        cur, cin = SplitActs(hi, lo, 128), 128
        keep.append(cur)
    check(_lib.lib().iago_conv3x3_split_trunk(descs, len(layers), n, _flag(overflow), _stream()),
          "iago_conv3x3_split_trunk")
    return cur


def value_stem(planes, w1, b1, overflow=None):
    """relu(conv3x3(planes, w1) + b1), 2 -> 64 channels (Value.block1, network.py:68-70):
    float32 planes (n, 2, 8, 8) -> SplitActs with 64 channels."""
    n = planes.shape[0]
    if tuple(planes.shape[1:]) != (2, 8, 8) or tuple(w1.shape) != (64, 2, 3, 3):
        raise ValueError("value_stem: planes (n,2,8,8) and w1 (64,2,3,3) expected")
    hi = torch.empty((n, 4, 64, 16), dtype=torch.float16, device=planes.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().iago_value_stem(_dev(planes, torch.float32, "planes"), _dev(w1, torch.float32, "w1"),
                                     _dev(b1, torch.float32, "b1"), _dev(hi, torch.float16, "y_hi"),
                                     _dev(lo, torch.float16, "y_lo"), n, _flag(overflow), _stream()),
          "iago_value_stem")
    return SplitActs(hi, lo, 64)


def value_stem_boards(own, opp, w1, b1, overflow=None):
    """value_stem on the boards themselves (own = side to move): plane encoding fused in."""
    n = own.numel()
    hi = torch.empty((n, 4, 64, 16), dtype=torch.float16, device=own.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().iago_value_stem_boards(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                            _dev(w1, torch.float32, "w1"), _dev(b1, torch.float32, "b1"),
                                            _dev(hi, torch.float16, "y_hi"), _dev(lo, torch.float16, "y_lo"),
                                            n, _flag(overflow), _stream()), "iago_value_stem_boards")
    return SplitActs(hi, lo, 64)


def value_head(a, w9, b9, w10, w11):
    """Value.block9 + fc10 + fc11 with train=False (network.py:78-96) on SplitActs with
    128 channels -> (n,) float32."""
    if a.channels != 128 or tuple(w9.shape) != (1, 128, 3, 3) or tuple(w10.shape) != (128, 64) \
            or tuple(w11.shape) != (1, 128):
        raise ValueError("value_head: unexpected shapes")
    out = torch.empty((a.n,), dtype=torch.float32, device=a.hi.device)
    check(_lib.lib().iago_value_head(_dev(a.hi, torch.float16, "x_hi"), _dev(a.lo, torch.float16, "x_lo"),
                                     _dev(w9, torch.float32, "w9"), _dev(b9, torch.float32, "b9"),
                                     _dev(w10, torch.float32, "w10"), _dev(w11, torch.float32, "w11"),
                                     _dev(out, torch.float32, "out"), a.n, _stream()), "iago_value_head")
    return out


def split_weights3(weight):
    """(128, cin, 3, 3) float32 conv weight -> (w_hi, w_mid, w_lo) f16 tensors
    [cin/16][3][3][128][16] with w == hi + mid 2^-11 + lo 2^-22 exactly
    (iago_policy_forward_split3)."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cin % 16 or cout != 128:
        raise ValueError("split_weights3: need a (128, 16k, 3, 3) weight")
    w = weight.detach().to(torch.float32).permute(2, 3, 0, 1)          # ky, kx, co, ci
    w = w.reshape(3, 3, cout, cin // 16, 16).permute(3, 0, 1, 2, 4).contiguous()
    hi = w.to(torch.float16)
    r1 = (w - hi.to(torch.float32)) * 2048.0
    mid = r1.to(torch.float16)
    lo = ((r1 - mid.to(torch.float32)) * 2048.0).to(torch.float16)
    return hi.contiguous(), mid.contiguous(), lo.contiguous()


def split_weights3_many(weights):
    """split_weights3 of several conv weights with the elementwise work of equally shaped ones
    done on one stacked tensor (a training loop re-splits all 7 blocks after every update)."""
    out = [None] * len(weights)
    groups = {}
    for i, w in enumerate(weights):
        groups.setdefault(tuple(w.shape), []).append(i)
    for shape, idx in groups.items():
        cout, cin, kh, kw = shape
        if (kh, kw) != (3, 3) or cin % 16 or cout != 128:
            raise ValueError("split_weights3: need (128, 16k, 3, 3) weights")
        w = torch.stack([weights[i].detach().to(torch.float32) for i in idx])          # L, co, ci, ky, kx
        w = w.permute(0, 3, 4, 1, 2).reshape(len(idx), 3, 3, cout, cin // 16, 16).permute(0, 4, 1, 2, 3, 5).contiguous()
        hi = w.to(torch.float16)
        r1 = (w - hi.to(torch.float32)) * 2048.0
        mid = r1.to(torch.float16)
        lo = ((r1 - mid.to(torch.float32)) * 2048.0).to(torch.float16)
        for j, i in enumerate(idx):
            out[i] = (hi[j], mid[j], lo[j])
    return out


POLICY_SCRATCH_ROW_BYTES = 51200   # a board's LDS image in iago_policy_forward_split3 (64 rows of 800 B)


def policy_split3_prepare(w1, b1, layers, w9, b10):
    """The weight half of iago_policy_split3_args, checked once: a template that
    policy_forward_split3_prepared copies per call.  The caller keeps the tensors alive."""
    a = _lib.PolicySplit3Args()
    if len(layers) != 7 or tuple(w1.shape) != (64, 2, 3, 3) or w9.numel() != 128 or b10.numel() != 64:
        raise ValueError("policy_forward_split3: unexpected shapes")
    a.w1, a.b1 = _dev(w1, torch.float32, "w1").value, _dev(b1, torch.float32, "b1").value
    for k, (w_hi, w_mid, w_lo, bias) in enumerate(layers):
        if w_hi.shape != ((4 if k == 0 else 8), 3, 3, 128, 16):
            raise ValueError("policy_forward_split3: layer %d: weight blocks %s" % (k, tuple(w_hi.shape)))
        a.w_hi[k] = _dev(w_hi, torch.float16, "w_hi").value
        a.w_mid[k] = _dev(w_mid, torch.float16, "w_mid").value
        a.w_lo[k] = _dev(w_lo, torch.float16, "w_lo").value
        a.bias[k] = _dev(bias, torch.float32, "bias").value
    a.w9 = _dev(w9.reshape(128), torch.float32, "w9").value
    a.b10 = _dev(b10, torch.float32, "b10").value
    return a


def policy_forward_split3_prepared(template, own, opp, n=None, index=None, n_dev=None, overflow=None, parts=1,
                                   scratch=None):
    """policy_forward_split3 with the weights of policy_split3_prepare."""
    a = _lib.PolicySplit3Args.from_buffer_copy(template)
    n = own.numel() if n is None else n
    if index is not None:
        n = min(n, index.numel())
        a.index = _dev(index, torch.int64, "index").value
    if n_dev is not None:
        a.n_dev = _dev(n_dev, torch.int32, "n_dev").value
    a.own, a.opp = _dev(own, torch.int64, "own").value, _dev(opp, torch.int64, "opp").value
    a.n = n
    probs = torch.empty((n, 64), dtype=torch.float32, device=own.device)
    a.probs = probs.data_ptr()
    if parts > 1:
        _set_scratch(a, parts, scratch, n)
    f = _flag(overflow)
    a.overflow = f.value if f is not None else None
    check(_lib.lib().iago_policy_forward_split3(C.byref(a), _stream()), "iago_policy_forward_split3")
    return probs


def _set_scratch(a, parts, scratch, n):
    """parts > 1: scratch = uint8 [rows][POLICY_SCRATCH_ROW_BYTES]; a batch of more than `rows`
    rows runs as chunks of `rows` (iago_policy_split3_args.scratch_rows)."""
    rows = 0 if scratch is None else scratch.numel() // POLICY_SCRATCH_ROW_BYTES
    if rows < 1:
        raise ValueError("policy_forward_split3: parts > 1 needs a scratch buffer of (rows, %d) bytes"
                         % POLICY_SCRATCH_ROW_BYTES)
    a.parts, a.scratch = parts, _dev(scratch, torch.uint8, "scratch").value
    a.scratch_rows = min(rows, n, 0x7FFFFFFF)


def policy_forward_split3(own, opp, w1, b1, layers, w9, b10, n=None, index=None, n_dev=None, overflow=None,
                          parts=1, scratch=None):
    """The whole SLPolicy net in one launch (iago_policy_forward_split3): boards (own = side to
    move) -> (n, 64) probabilities.  layers: the 7 (w_hi, w_mid, w_lo, bias) of blocks 2..8
    (split_weights3); index / n_dev: optional device-side gather list and row count."""
    a = _lib.PolicySplit3Args()
    n = own.numel() if n is None else n
    if index is not None:
        n = min(n, index.numel())
        a.index = _dev(index, torch.int64, "index").value
    if n_dev is not None:
        a.n_dev = _dev(n_dev, torch.int32, "n_dev").value
    if len(layers) != 7 or tuple(w1.shape) != (64, 2, 3, 3) or w9.numel() != 128 or b10.numel() != 64:
        raise ValueError("policy_forward_split3: unexpected shapes")
    a.own, a.opp = _dev(own, torch.int64, "own").value, _dev(opp, torch.int64, "opp").value
    a.n = n
    a.w1, a.b1 = _dev(w1, torch.float32, "w1").value, _dev(b1, torch.float32, "b1").value
    for k, (w_hi, w_mid, w_lo, bias) in enumerate(layers):
        if w_hi.shape != ((4 if k == 0 else 8), 3, 3, 128, 16):
            raise ValueError("policy_forward_split3: layer %d: weight blocks %s" % (k, tuple(w_hi.shape)))
        a.w_hi[k] = _dev(w_hi, torch.float16, "w_hi").value
        a.w_mid[k] = _dev(w_mid, torch.float16, "w_mid").value
        a.w_lo[k] = _dev(w_lo, torch.float16, "w_lo").value
        a.bias[k] = _dev(bias, torch.float32, "bias").value
    a.w9 = _dev(w9.reshape(128), torch.float32, "w9").value
    a.b10 = _dev(b10, torch.float32, "b10").value
    probs = torch.empty((n, 64), dtype=torch.float32, device=own.device)
    a.probs = probs.data_ptr()
    if parts > 1:
        # parts launches (per chunk of the scratch's rows); scratch not shared with a call on
        # another stream (the launches of a call, and the calls of a stream, run in order)
        _set_scratch(a, parts, scratch, n)
    f = _flag(overflow)
    a.overflow = f.value if f is not None else None
    check(_lib.lib().iago_policy_forward_split3(C.byref(a), _stream()), "iago_policy_forward_split3")
    return probs


def split_head_weights(w9):
    """Value.block9's (1, 128, 3, 3) weight as the MFMA operand of iago_value_forward_split:
    (hi, lo) f16 [8 chunks][32 rows][16], row r < 9 = kernel tap r, the other rows zero."""
    if tuple(w9.shape) != (1, 128, 3, 3):
        raise ValueError("split_head_weights: (1, 128, 3, 3) expected")
    w = w9.detach().to(torch.float32).reshape(8, 16, 9).permute(0, 2, 1)   # chunk, tap, channel
    a = torch.zeros((8, 32, 16), dtype=torch.float32, device=w9.device)
    a[:, :9, :] = w
    hi = a.to(torch.float16)
    lo = ((a - hi.to(torch.float32)) * 2048.0).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


def value_split_weights(a, w1, b1, layers, head, b9, w10, w11):
    """The weight half of an iago_value_split_args, checked (value_forward_split, the persistent search)."""
    if len(layers) != 7 or tuple(w1.shape) != (64, 2, 3, 3) or tuple(w10.shape) != (128, 64) \
            or tuple(w11.shape) != (1, 128):
        raise ValueError("value_forward_split: unexpected shapes")
    a.w1, a.b1 = _dev(w1, torch.float32, "w1").value, _dev(b1, torch.float32, "b1").value
    for k, (w_hi, w_lo, bias) in enumerate(layers):
        if w_hi.shape != ((4 if k == 0 else 8), 3, 3, 128, 16):
            raise ValueError("value_forward_split: layer %d: weight blocks %s" % (k, tuple(w_hi.shape)))
        a.w_hi[k] = _dev(w_hi, torch.float16, "w_hi").value
        a.w_lo[k] = _dev(w_lo, torch.float16, "w_lo").value
        a.bias[k] = _dev(bias, torch.float32, "bias").value
    a.w9_hi, a.w9_lo = _dev(head[0], torch.float16, "w9_hi").value, _dev(head[1], torch.float16, "w9_lo").value
    a.b9 = _dev(b9, torch.float32, "b9").value
    a.w10, a.w11 = _dev(w10, torch.float32, "w10").value, _dev(w11, torch.float32, "w11").value


def value_forward_split(x, w1, b1, layers, head, w9, b9, w10, w11, overflow=None, index=None, n_dev=None,
                        out=None, rollout=None, async_ref=None, batch=None):
    """The whole Value net in one launch (iago_value_forward_split).  x: float32 planes
    (n, 2, 8, 8) or a pair (own, opp) of int64 bitboards (own = side to move); layers: the 7
    (w_hi, w_lo, bias) of blocks 2..8 (split_weights); head: split_head_weights(w9).
    index / n_dev (boards only): evaluate boards index[0 .. min(n, *n_dev)) and write their
    values to out[index[i]] (out: (n_boards,) float32, the other entries untouched).
    rollout: a PreparedRollout (rollout_prepare) to play in the SAME launch (iago_value_rollout:
    the leaf evaluation of a playout, value net on the listed leaves + rollout of all).
    batch = (boards per workgroup, max workgroups): the device-counted batch form for work off the
    playouts' critical path (iago_value_forward_batch; n_dev required)."""
    a = _lib.ValueSplitArgs()
    if isinstance(x, tuple):
        own, opp = x
        n, dev = own.numel(), own.device
        a.own, a.opp = _dev(own, torch.int64, "own").value, _dev(opp, torch.int64, "opp").value
    else:
        if tuple(x.shape[1:]) != (2, 8, 8):
            raise ValueError("value_forward_split: planes (n, 2, 8, 8) expected")
        n, dev = x.shape[0], x.device
        a.planes = _dev(x, torch.float32, "planes").value
    a.n = n
    value_split_weights(a, w1, b1, layers, head, b9, w10, w11)
    if index is not None:
        if not isinstance(x, tuple) or out is None:
            raise ValueError("value_forward_split: a gather list needs (own, opp) boards and an `out` buffer")
        n = min(index.numel(), n)
        a.n = n
        a.index = _dev(index, torch.int64, "index").value
    if n_dev is not None:
        a.n_dev = _dev(n_dev, torch.int32, "n_dev").value
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=dev)
    a.out = _dev(out, torch.float32, "out").value
    f = _flag(overflow)
    a.overflow = f.value if f is not None else None
    if async_ref is not None:
        # one game-asynchronous step's leaf evaluation: the rollouts of the games that descended and
        # one piece of the value net for every queue of fresh leaves (iago_value_rollout_async)
        check(_lib.lib().iago_value_rollout_async(C.byref(a), rollout.ref, async_ref, _stream()),
              "iago_value_rollout_async")
    elif rollout is not None:
        check(_lib.lib().iago_value_rollout(C.byref(a), rollout.ref, _stream()), "iago_value_rollout")
    elif batch is not None:
        check(_lib.lib().iago_value_forward_batch(C.byref(a), int(batch[0]), int(batch[1]), _stream()),
              "iago_value_forward_batch")
    else:
        check(_lib.lib().iago_value_forward_split(C.byref(a), _stream()), "iago_value_forward_split")
    return out


def conv3x3_split(a, w_hi, w_lo, bias, overflow=None):
    """relu(conv3x3(a, w) + bias) on SplitActs: Block.__call__ (network.py:9-13) on
    the MFMA units in split-f16 arithmetic.  w_hi, w_lo from split_weights."""
    cin = a.channels
    if w_hi.shape != (cin // 16, 3, 3, 128, 16):
        raise ValueError("weight blocks %s do not match %d input channels" % (tuple(w_hi.shape), cin))
    hi = torch.empty((a.n, 8, 64, 16), dtype=torch.float16, device=a.hi.device)
    lo = torch.empty_like(hi)
    check(_lib.lib().iago_conv3x3_split(_dev(a.hi, torch.float16, "x_hi"), _dev(a.lo, torch.float16, "x_lo"),
                                        _dev(w_hi, torch.float16, "w_hi"), _dev(w_lo, torch.float16, "w_lo"),
                                        _dev(bias, torch.float32, "bias"),
                                        _dev(hi, torch.float16, "y_hi"), _dev(lo, torch.float16, "y_lo"),
                                        a.n, cin, 128, _flag(overflow), _stream()), "iago_conv3x3_split")
    return SplitActs(hi, lo, 128)


# ---- gradients of the REINFORCE update in split-f16 arithmetic (csrc/policy_grad_kernels.hip) ----

WGRAD_GROUPS = 32   # groups of boards whose partial weight gradients are summed in order (a multiple of 8)


def conv3x3_wgrad_split(dy, x, scale_exp=None, groups=WGRAD_GROUPS, part=None):
    """Weight gradient of one 3x3 block: dy, x SplitActs (dy: 128 channels, times 2**scale_exp), returns
    dW (128, cin, 3, 3) float32 (include/iago_hip.h)."""
    cin = x.channels
    if dy.channels != 128 or dy.n != x.n:
        raise ValueError("conv3x3_wgrad_split: dy must be (n, 128) channels on the rows of x")
    if part is None:
        part = torch.empty((groups, 9, 128, cin), dtype=torch.float32, device=x.hi.device)
    dw = torch.empty((128, cin, 3, 3), dtype=torch.float32, device=x.hi.device)
    check(_lib.lib().iago_conv3x3_wgrad_split(_dev(dy.hi, torch.float16, "dy_hi"), _dev(dy.lo, torch.float16, "dy_lo"),
                                              _dev(x.hi, torch.float16, "x_hi"), _dev(x.lo, torch.float16, "x_lo"),
                                              x.n, cin, _dev(part, torch.float32, "part"), groups,
                                              _dev(scale_exp, torch.int32, "scale_exp") if scale_exp is not None else None,
                                              _dev(dw, torch.float32, "dw"), _stream()), "iago_conv3x3_wgrad_split")
    return dw


def split_weights_transposed(weight):
    """The backward-data form of a (128, cin, 3, 3) block weight: split_weights of Wt[ci][co][ky][kx] =
    W[co][ci][2-ky][2-kx], padded to 128 rows."""
    cout, cin = weight.shape[0], weight.shape[1]
    wt = weight.detach().to(torch.float32).flip(2, 3).permute(1, 0, 2, 3)
    if cin < 128:
        wt = torch.cat([wt, torch.zeros((128 - cin, cout, 3, 3), dtype=torch.float32, device=wt.device)])
    return split_weights(wt.contiguous())


def split_weights_transposed_many(weights):
    """split_weights_transposed of several block weights, the elementwise work of equally shaped ones on one stacked
    tensor (the update re-splits all 7 blocks after every step)."""
    out = [None] * len(weights)
    groups = {}
    for i, w in enumerate(weights):
        groups.setdefault(tuple(w.shape), []).append(i)
    for shape, idx in groups.items():
        cout, cin, kh, kw = shape
        if (kh, kw) != (3, 3) or cout != 128 or cin not in (64, 128):
            raise ValueError("split_weights_transposed: need (128, 64|128, 3, 3) weights")
        w = torch.stack([weights[i].detach().to(torch.float32) for i in idx])            # L, co, ci, ky, kx
        wt = w.flip(3, 4).permute(0, 3, 4, 2, 1)                                          # L, ky', kx', ci, co
        if cin < 128:
            wt = torch.cat([wt, torch.zeros((len(idx), 3, 3, 128 - cin, cout), dtype=torch.float32, device=w.device)], 3)
        wt = wt.reshape(len(idx), 3, 3, 128, cout // 16, 16).permute(0, 4, 1, 2, 3, 5).contiguous()
        hi = wt.to(torch.float16)
        lo = ((wt - hi.to(torch.float32)) * 2048.0).to(torch.float16)
        for j, i in enumerate(idx):
            out[i] = (hi[j], lo[j])
    return out


def adam_chainer(params, grads, ms, vs, alpha_t, beta1, beta2, eps, weight_decay, steps=None):
    """iago_adam_chainer: one launch for all (float32, contiguous, CUDA) parameters; m / v in place, the parameters
    too unless `steps` (tensors that receive alpha_t m / (sqrt(v) + eps), for the caller to subtract) is given."""
    if len(params) > _lib.ADAM_MAX_TENSORS:
        raise ValueError("adam_chainer: at most %d tensors" % _lib.ADAM_MAX_TENSORS)
    A = _lib.AdamArgs()
    for k, (p, g, m, v) in enumerate(zip(params, grads, ms, vs)):
        A.p[k], A.g[k] = _dev(p, torch.float32, "p").value, _dev(g, torch.float32, "g").value
        A.m[k], A.v[k] = _dev(m, torch.float32, "m").value, _dev(v, torch.float32, "v").value
        A.count[k] = p.numel()
        if steps is not None:
            A.step[k] = _dev(steps[k], torch.float32, "step").value
    A.n_tensors = len(params)
    A.alpha_t, A.one_minus_beta1, A.one_minus_beta2 = alpha_t, 1.0 - beta1, 1.0 - beta2
    A.eps, A.weight_decay = eps, weight_decay
    check(_lib.lib().iago_adam_chainer(C.byref(A), _stream()), "iago_adam_chainer")


_grad_workspace = {}


def policy_grad_workspace(device, n):
    """The scratch of iago_policy_reinforce_grad for n rows on `device`: kept, grown in steps of 256 rows."""
    need = int(_lib.lib().iago_policy_grad_workspace_bytes((n + 255) // 256 * 256))
    ws = _grad_workspace.get(str(device))
    if ws is None or ws.numel() < need:
        _grad_workspace[str(device)] = ws = None      # (release before the larger allocation)
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _grad_workspace[str(device)] = ws
    return ws


def release_grad_workspace():
    """Free the scratch policy_grad_workspace keeps (308 KB per row of the largest batch seen + 247 MB)."""
    _grad_workspace.clear()


def policy_reinforce_grad(own, opp, action, reward, n_mean, w1, b1, layers, layers_t, w9, b10, grads, probs=None,
                          overflow=None):
    """iago_policy_reinforce_grad (include/iago_hip.h): the gradients of mean(softmax_cross_entropy(model(x), a) * r)
    (src/train_rl.py:61-65) written to `grads` = dict(w1, b1, w=[7], b=[7], w9, b10) of float32 tensors in the
    parameters' shapes.  layers: 7 x (w_hi, w_lo, bias) as for conv3x3_split; layers_t: 7 x (wt_hi, wt_lo) from
    split_weights_transposed.  Returns the loss (0-dim float32 device tensor)."""
    n = own.numel()
    dev = own.device
    ws = policy_grad_workspace(dev, n)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    A = _lib.PolicyGradArgs()
    A.own, A.opp = _dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp")
    A.action, A.reward = _dev(action, torch.int32, "action"), _dev(reward, torch.float32, "reward")
    A.n, A.n_mean = n, int(n_mean)
    A.w1, A.b1 = _dev(w1, torch.float32, "w1"), _dev(b1, torch.float32, "b1")
    for k in range(7):
        A.w_hi[k], A.w_lo[k] = _dev(layers[k][0], torch.float16, "w_hi").value, _dev(layers[k][1], torch.float16, "w_lo").value
        A.bias[k] = _dev(layers[k][2], torch.float32, "bias").value
        A.wt_hi[k], A.wt_lo[k] = (_dev(layers_t[k][0], torch.float16, "wt_hi").value,
                                  _dev(layers_t[k][1], torch.float16, "wt_lo").value)
        A.g_w[k], A.g_b[k] = _dev(grads["w"][k], torch.float32, "g_w").value, _dev(grads["b"][k], torch.float32, "g_b").value
    A.w9, A.b10 = _dev(w9, torch.float32, "w9"), _dev(b10, torch.float32, "b10")
    A.g_w1, A.g_b1 = _dev(grads["w1"], torch.float32, "g_w1"), _dev(grads["b1"], torch.float32, "g_b1")
    A.g_w9, A.g_b10 = _dev(grads["w9"], torch.float32, "g_w9"), _dev(grads["b10"], torch.float32, "g_b10")
    A.loss = _dev(loss, torch.float32, "loss")
    A.probs = _dev(probs, torch.float32, "probs") if probs is not None else None
    A.workspace, A.workspace_bytes = ws.data_ptr(), ws.numel()
    A.overflow = _flag(overflow)
    check(_lib.lib().iago_policy_reinforce_grad(C.byref(A), _stream()), "iago_policy_reinforce_grad")
    return loss


def conv3x3_bwd_data_split(dy, scale_exp, wt_hi, wt_lo, saved, max_bits=None):
    """Gradient at a block's input through the ReLU of the block below (saved: its SplitActs output).  Returns
    (dx float32 (n, C/16, 64, 16), max_bits)."""
    n, ch = dy.n, saved.channels
    dx = torch.empty((n, ch // 16, 64, 16), dtype=torch.float32, device=dy.hi.device)
    if max_bits is None:
        max_bits = torch.zeros(1, dtype=torch.int32, device=dy.hi.device)
    check(_lib.lib().iago_conv3x3_bwd_data_split(_dev(dy.hi, torch.float16, "dy_hi"), _dev(dy.lo, torch.float16, "dy_lo"),
                                                 _dev(scale_exp, torch.int32, "scale_exp"),
                                                 _dev(wt_hi, torch.float16, "wt_hi"), _dev(wt_lo, torch.float16, "wt_lo"),
                                                 _dev(saved.hi, torch.float16, "mask_hi"), _dev(saved.lo, torch.float16, "mask_lo"),
                                                 ch, _dev(dx, torch.float32, "dx"), _dev(max_bits, torch.int32, "max_bits"),
                                                 n, _stream()), "iago_conv3x3_bwd_data_split")
    return dx, max_bits


def split_scaled(x, max_bits, bias_grad=False):
    """float32 channel blocks (n, C/16, 64, 16) -> (SplitActs of x * 2**e, e as an int32 device word[, the sums
    over boards and cells per channel])."""
    n, nb = x.shape[0], x.shape[1]
    hi = torch.empty((n, nb, 64, 16), dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi)
    e = torch.empty(1, dtype=torch.int32, device=x.device)
    part = db = None
    if bias_grad:
        part = torch.empty(((n * nb + 1) // 2) * 32, dtype=torch.float32, device=x.device)
        db = torch.empty(nb * 16, dtype=torch.float32, device=x.device)
    check(_lib.lib().iago_split_scaled(_dev(x, torch.float32, "x"), _dev(max_bits, torch.int32, "max_bits"),
                                       _dev(hi, torch.float16, "hi"), _dev(lo, torch.float16, "lo"),
                                       _dev(e, torch.int32, "scale_exp"), n, nb * 16,
                                       _dev(part, torch.float32, "bias_part") if bias_grad else None,
                                       _dev(db, torch.float32, "bias_grad") if bias_grad else None,
                                       _stream()), "iago_split_scaled")
    return (SplitActs(hi, lo, nb * 16), e, db) if bias_grad else (SplitActs(hi, lo, nb * 16), e)


def blocks_to_nchw(x):
    """float32 channel blocks (n, C/16, 64, 16) -> (n, C, 8, 8)."""
    n, nb = x.shape[0], x.shape[1]
    return x.permute(0, 1, 3, 2).reshape(n, nb * 16, 8, 8)


def nchw_to_blocks(x):
    """(n, C, 8, 8) float32 -> channel blocks (n, C/16, 64, 16)."""
    n, c = x.shape[0], x.shape[1]
    return x.reshape(n, c // 16, 16, 64).permute(0, 1, 3, 2).contiguous()


# ---- float32 small-batch convolution stack (policy net on expansions) ----------------

def f32_weights(weight):
    """(128, cin, 3, 3) float32 conv weight -> [4][9][cin][32] as iago_conv3x3_f32 reads it."""
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cout != 128 or cin not in (64, 128):
        raise ValueError("f32_weights: need a (128, 64|128, 3, 3) weight")
    w = weight.detach().to(torch.float32).permute(2, 3, 1, 0).reshape(9, cin, 4, 32)
    return w.permute(2, 0, 1, 3).contiguous()


def conv3x3_f32(x, w4, bias, n_dev=None):
    """relu(conv3x3(x, w) + bias) in float32 on the matrix units; x (n, cin, 8, 8)."""
    n, cin = x.shape[0], x.shape[1]
    if tuple(w4.shape) != (4, 9, cin, 32):
        raise ValueError("weight layout %s does not match %d input channels" % (tuple(w4.shape), cin))
    y = torch.empty((n, 128, 8, 8), dtype=torch.float32, device=x.device)
    check(_lib.lib().iago_conv3x3_f32(_dev(x, torch.float32, "x"), _dev(w4, torch.float32, "w"),
                                      _dev(bias, torch.float32, "bias"), _dev(y, torch.float32, "y"),
                                      n, cin, 128, _count(n_dev), _stream()), "iago_conv3x3_f32")
    return y


def stem_f32(planes, w1, b1, n_dev=None):
    n = planes.shape[0]
    if tuple(planes.shape[1:]) != (2, 8, 8) or tuple(w1.shape) != (64, 2, 3, 3):
        raise ValueError("stem_f32: planes (n,2,8,8) and w1 (64,2,3,3) expected")
    y = torch.empty((n, 64, 8, 8), dtype=torch.float32, device=planes.device)
    check(_lib.lib().iago_stem_f32(_dev(planes, torch.float32, "planes"), _dev(w1, torch.float32, "w1"),
                                   _dev(b1, torch.float32, "b1"), _dev(y, torch.float32, "y"), n,
                                   _count(n_dev), _stream()), "iago_stem_f32")
    return y


def stem_f32_boards(own, opp, index, w1, b1, n, n_dev=None):
    """stem_f32 of boards index[0..n) (int64 tensor or None = the first n boards) without the
    planes tensor; own = side to move."""
    if tuple(w1.shape) != (64, 2, 3, 3):
        raise ValueError("stem_f32_boards: w1 (64,2,3,3) expected")
    if index is not None and index.numel() < n:
        raise ValueError("index is shorter than n")
    y = torch.empty((n, 64, 8, 8), dtype=torch.float32, device=own.device)
    check(_lib.lib().iago_stem_f32_boards(_dev(own, torch.int64, "own"), _dev(opp, torch.int64, "opp"),
                                          _dev(index, torch.int64, "index") if index is not None else None,
                                          _dev(w1, torch.float32, "w1"), _dev(b1, torch.float32, "b1"),
                                          _dev(y, torch.float32, "y"), n, _count(n_dev), _stream()),
          "iago_stem_f32_boards")
    return y


def policy_head(x, w9, b10, n_dev=None):
    """softmax(conv1x1(x, w9) + b10) (network.py:29-47): x (n, 128, 8, 8) -> (n, 64)."""
    n = x.shape[0]
    if tuple(x.shape[1:]) != (128, 8, 8) or w9.numel() != 128 or b10.numel() != 64:
        raise ValueError("policy_head: unexpected shapes")
    probs = torch.empty((n, 64), dtype=torch.float32, device=x.device)
    check(_lib.lib().iago_policy_head(_dev(x, torch.float32, "x"), _dev(w9.reshape(128), torch.float32, "w9"),
                                      _dev(b10, torch.float32, "b10"), _dev(probs, torch.float32, "probs"),
                                      n, _count(n_dev), _stream()), "iago_policy_head")
    return probs


def sample_moves(probs, legal, uniforms=None, seed=0, id_base=0, step=0, stream_id=0):
    """Masked inverse-CDF sampling (src/rl_self_play.py:111-122); int8 actions,
    -1 where there is no legal move.  uniforms: optional float64 (n,)."""
    n = legal.numel()
    if probs.numel() != n * 64:
        raise ValueError("probs must be (n, 64)")
    out = torch.empty(n, dtype=torch.int8, device=legal.device)
    up = _dev(uniforms, torch.float64, "uniforms") if uniforms is not None else None
    check(_lib.lib().iago_sample_moves(_dev(probs, torch.float32, "probs"),
                                       _dev(legal, torch.int64, "legal"), up,
                                       int(seed) & 0xFFFFFFFFFFFFFFFF, int(id_base) & 0xFFFFFFFF,
                                       int(step) & 0xFFFFFFFF, int(stream_id) & 0xFFFFFFFF,
                                       _dev(out, torch.int8, "action"), n, _stream()),
          "iago_sample_moves")
    return out


class RolloutWeights(object):
    """Device-resident RolloutPolicy parameters (network.py:49-64) in the form
    the rollout kernel consumes (the table blob of iago_rollout_build_table).
    w = None builds the uniform policy."""

    def __init__(self, w=None, b=None, device="cuda"):
        blob = np.empty(IAGO_ROLLOUT_TABLE_FLOATS, dtype=np.float32)
        if w is None:
            self.w = self.b = None
            wp = bp = None
        else:
            self.w = np.ascontiguousarray(np.asarray(w, dtype=np.float32).reshape(18))
            self.b = np.ascontiguousarray(np.asarray(b, dtype=np.float32).reshape(64))
            wp, bp = C.c_void_p(self.w.ctypes.data), C.c_void_p(self.b.ctypes.data)
        check(_lib.lib().iago_rollout_build_table(wp, bp, C.c_void_p(blob.ctypes.data)),
              "iago_rollout_build_table")
        self.log_form = 0 if blob[_lib.ROLLOUT_MODE_INDEX] == 1.0 else 1
        self.table = torch.from_numpy(blob).to(device)


_UNIFORM = {}


def uniform_weights(device="cuda"):
    key = str(device)
    if key not in _UNIFORM:
        _UNIFORM[key] = RolloutWeights(None, None, device)
    return _UNIFORM[key]


class RolloutResult(object):
    __slots__ = ("z", "final_own", "final_opp", "n_turns", "trace")

    def __init__(self):
        self.z = self.final_own = self.final_opp = self.n_turns = self.trace = None


class PreparedRollout(object):
    """A fully marshalled iago_rollout call: `launch(stream)` costs one ctypes
    call.  Holds references to every tensor the launch touches."""
    __slots__ = ("args", "ref", "result", "_keep", "_fn")

    def launch(self, stream_ptr=None):
        """Enqueue on `stream_ptr` (a hipStream_t as int / c_void_p; None =
        torch's current stream); returns the iago status code."""
        if stream_ptr is None:
            stream_ptr = _stream()
        return self._fn(self.ref, stream_ptr)


def rollout_prepare(own, opp, weights=None, seed=0, id_base=0, stream_id=0, uniforms=None,
                    want_final=False, want_turns=False, want_trace=False, out=None,
                    throughput_hint=False, stream_id_dev=None):
    """Marshal a rollout launch once (see `rollout` for the arguments)."""
    n = own.numel()
    res = out if out is not None else RolloutResult()
    dev = own.device
    if res.z is None:
        res.z = torch.empty(n, dtype=torch.int8, device=dev)
    if want_final and res.final_own is None:
        res.final_own = torch.empty(n, dtype=torch.int64, device=dev)
        res.final_opp = torch.empty(n, dtype=torch.int64, device=dev)
    if want_turns and res.n_turns is None:
        res.n_turns = torch.empty(n, dtype=torch.uint8, device=dev)
    if want_trace and res.trace is None:
        res.trace = torch.full((IAGO_MAX_TURNS, n), 0xFE, dtype=torch.uint8, device=dev)
    if weights is None:
        weights = uniform_weights(dev)
    a = RolloutArgs()
    a.own = _dev(own, torch.int64, "own")
    a.opp = _dev(opp, torch.int64, "opp")
    a.n = n
    a.table = _dev(weights.table, torch.float32, "table")
    a.log_form = weights.log_form
    a.throughput_hint = int(throughput_hint)  # False / 0: automatic, True / 1: lane per board, 2: 8 lanes per board
    if uniforms is not None:
        if tuple(uniforms.shape) != (IAGO_MAX_TURNS, n):
            raise ValueError("uniforms must have shape (%d, n)" % IAGO_MAX_TURNS)
        a.uniforms = _dev(uniforms, torch.float32, "uniforms")
    a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    a.id_base = int(id_base) & 0xFFFFFFFF
    a.stream_id = int(stream_id) & 0xFFFFFFFF
    if stream_id_dev is not None:  # int32 CUDA tensor with one element, added on the device
        a.stream_id_dev = _dev(stream_id_dev, torch.int32, "stream_id_dev")
    a.z = _dev(res.z, torch.int8, "z")
    if res.final_own is not None:
        a.final_own = _dev(res.final_own, torch.int64, "final_own")
        a.final_opp = _dev(res.final_opp, torch.int64, "final_opp")
    if res.n_turns is not None:
        a.n_turns = _dev(res.n_turns, torch.uint8, "n_turns")
    if res.trace is not None:
        a.trace = _dev(res.trace, torch.uint8, "trace")
    p = PreparedRollout()
    p.args, p.ref, p.result = a, C.byref(a), res
    p._keep = (own, opp, weights, uniforms, stream_id_dev)
    p._fn = _lib.lib().iago_rollout
    return p


def rollout(own, opp, weights=None, seed=0, id_base=0, stream_id=0, uniforms=None,
            want_final=False, want_turns=False, want_trace=False, out=None,
            throughput_hint=False, stream_id_dev=None):
    """Simulate(state)(color) for every board (mcts_self_play.py:9-134).

    weights=None plays uniformly random legal moves.  `uniforms`
    (IAGO_MAX_TURNS, n) float32 replaces the Philox stream (parity tests).
    `out` may be a RolloutResult with preallocated tensors to reuse.
    throughput_hint: the caller overlaps this launch with others (see
    iago_rollout_args.throughput_hint).
    """
    p = rollout_prepare(own, opp, weights, seed, id_base, stream_id, uniforms, want_final,
                        want_turns, want_trace, out, throughput_hint, stream_id_dev)
    check(p.launch(), "iago_rollout")
    return p.result
