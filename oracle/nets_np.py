"""float64 numpy restatement of the reference networks (network.py:5-96).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED BY THE REFERENCE: the arithmetic
lives in third-party Chainer (unpinned: readme.md:13 `pip install chainer`, no
lockfile), which is not installable here, and the reference holds no test or
golden output for its nets.  This module restates Chainer's published
semantics at the call sites network.py:9,12-13,31-32,44-46,56-57,80-81,93-95:

  * L.Convolution2D(in, out, ksize, pad): cross-correlation, W (O,I,kh,kw),
    stride 1, zero padding `pad`, bias b (O) unless nobias.
  * L.Bias(shape=(64,)): adds a learned length-64 vector along axis 1.
  * L.Linear(None, out, nobias=True): flattens trailing dims, y = x W^T.
  * F.relu, F.softmax(axis=1); F.dropout is the identity at inference
    (MCTS.py:86 sets chainer.config.train = False).

Parameter dicts use the reference's npz key names (models/*.npz):
  block{k}/conv/W, block{k}/conv/b, conv9/W, bias10/b, fc10/W, fc11/W,
  conv1/W, bias2/b.
"""
import numpy as np


def conv2d(x, W, b=None, pad=1):
    """x (B,I,8,8), W (O,I,k,k) -> (B,O,8,8) in float64."""
    x = np.asarray(x, np.float64)
    W = np.asarray(W, np.float64)
    B, I, H, Wd = x.shape
    O, _, k, _ = W.shape
    xp = np.zeros((B, I, H + 2 * pad, Wd + 2 * pad))
    xp[:, :, pad:pad + H, pad:pad + Wd] = x
    out = np.zeros((B, O, H, Wd))
    for ky in range(k):
        for kx in range(k):
            out += np.einsum("bihw,oi->bohw", xp[:, :, ky:ky + H, kx:kx + Wd], W[:, :, ky, kx])
    if b is not None:
        out += np.asarray(b, np.float64).reshape(1, O, 1, 1)
    return out


def softmax(h):
    e = np.exp(h - h.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def _trunk(x, p):
    h = np.asarray(x, np.float64)
    for k in range(1, 9):  # network.py:35-42 / 84-91: Block = conv3x3 pad1 + ReLU
        h = np.maximum(conv2d(h, p["block%d/conv/W" % k], p["block%d/conv/b" % k], 1), 0.0)
    return h


def sl_policy(x, p):
    """network.py:34-47 -> (B,64) probabilities."""
    h = _trunk(x, p)
    h = conv2d(h, p["conv9/W"], None, 0).reshape(-1, 64)
    h = h + np.asarray(p["bias10/b"], np.float64).reshape(1, 64)
    return softmax(h)


def value(x, p):
    """network.py:83-96 -> (B,) values (no tanh, no activation after fc10)."""
    h = _trunk(x, p)
    h = np.maximum(conv2d(h, p["block9/conv/W"], p["block9/conv/b"], 1), 0.0).reshape(-1, 64)
    h = h @ np.asarray(p["fc10/W"], np.float64).T
    return (h @ np.asarray(p["fc11/W"], np.float64).T).reshape(-1)


def rollout_policy(x, p):
    """network.py:59-64 -> (B,64) probabilities."""
    h = conv2d(x, p["conv1/W"], None, 1).reshape(-1, 64)
    h = h + np.asarray(p["bias2/b"], np.float64).reshape(1, 64)
    return softmax(h)


def random_params(kind, seed):
    """Seeded stand-in parameters with the reference's shapes.  Chainer's
    default initialiser (LeCunNormal for W, zeros for b) is from memory and
    unpinned; biases get small noise here so that tests exercise them."""
    rs = np.random.RandomState(seed)
    p = {}

    def lecun(shape):
        fan_in = int(np.prod(shape[1:]))
        return (rs.randn(*shape) / np.sqrt(fan_in)).astype(np.float32)

    if kind == "rollout":
        p["conv1/W"] = lecun((1, 2, 3, 3))
        p["bias2/b"] = (0.1 * rs.randn(64)).astype(np.float32)
        return p
    chans = [2, 64, 128, 128, 128, 128, 128, 128, 128]
    for k in range(1, 9):
        p["block%d/conv/W" % k] = lecun((chans[k], chans[k - 1], 3, 3))
        p["block%d/conv/b" % k] = (0.05 * rs.randn(chans[k])).astype(np.float32)
    if kind == "sl":
        p["conv9/W"] = lecun((1, 128, 1, 1))
        p["bias10/b"] = (0.1 * rs.randn(64)).astype(np.float32)
    elif kind == "value":
        p["block9/conv/W"] = lecun((1, 128, 3, 3))
        p["block9/conv/b"] = (0.05 * rs.randn(1)).astype(np.float32)
        p["fc10/W"] = lecun((128, 64))
        p["fc11/W"] = lecun((1, 128))
    else:
        raise ValueError(kind)
    return p
