"""Game-level sharding and the one collective of the path.

Self-play games are independent, so they shard over ranks with no data-path
collective; the only exchange is the all-gather of finished training tuples at
the end of a self-play round (the reference's analogue is the list
concatenation of src/train_rl.py:48-51).  One process per GPU,
torch.distributed backend "nccl" (= RCCL over xGMI); "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_range(n_games, rank=None, world=None):
    """Contiguous block of global game ids owned by `rank`: [lo, hi)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(n_games, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _staged(group=None):
    """True when the collectives of `group` cannot take device tensors (gloo: world-size > 1
    rehearsals on ONE GPU, where RCCL refuses two ranks on the same device; gloo implements
    all-gather for host memory only): payloads then travel through host copies.  The nccl
    (= RCCL) backend takes the device tensors as they are."""
    return dist.get_backend(group) == "gloo"


def all_gather_into(recv, send, group=None):
    """dist.all_gather_into_tensor; under gloo (one-GPU rehearsals) device payloads travel through host copies."""
    if send.is_cuda and _staged(group):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def gather_tuples(fields, group=None):
    """All-gather per-rank tuple arrays (dict name -> tensor, same length along
    dim 0 on a rank, lengths may differ between ranks).  Returns a dict of
    tensors holding rank 0's rows first, then rank 1's, ...

    Two collectives regardless of the number of fields: one all-gather of the
    row counts, one all-gather of a byte buffer padded to the longest shard
    (payloads are MBs: latency-, not bandwidth-bound on 7 x 153 GB/s xGMI links).
    """
    names = sorted(fields)
    if not dist.is_initialized():
        return {k: fields[k] for k in names}
    world = dist.get_world_size(group)
    first = fields[names[0]]
    n = first.shape[0]
    dev = first.device
    for k in names:
        if fields[k].shape[0] != n:
            raise ValueError("field %s has %d rows, expected %d" % (k, fields[k].shape[0], n))
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    all_gather_into(counts, torch.tensor([n], dtype=torch.int64, device=dev), group)
    counts = counts.cpu().tolist()
    nmax = max(counts)
    # pack every field's rows as bytes, 8-byte aligned segments
    segs, offs, off = [], [], 0
    for k in names:
        t = fields[k].contiguous()
        row_elems = 1
        for d in t.shape[1:]:
            row_elems *= int(d)
        row_bytes = t.element_size() * row_elems
        seg = (row_bytes * nmax + 7) // 8 * 8
        segs.append((k, t, row_bytes, seg))
        offs.append(off)
        off += seg
    send = torch.zeros(off, dtype=torch.uint8, device=dev)
    for (k, t, row_bytes, seg), o in zip(segs, offs):
        send[o:o + row_bytes * n] = t.view(-1).view(torch.uint8)
    recv = torch.empty(world * off, dtype=torch.uint8, device=dev)
    all_gather_into(recv, send, group)
    out = {}
    for (k, t, row_bytes, seg), o in zip(segs, offs):
        parts = []
        for r in range(world):
            base = r * off + o
            parts.append(recv[base:base + row_bytes * counts[r]])
        flat = torch.cat(parts).view(t.dtype)
        out[k] = flat.view((sum(counts),) + tuple(t.shape[1:]))
    return out


def broadcast_tensors(tensors, src=0, group=None):
    """In place: every rank's `tensors` (same shapes / dtypes everywhere) become rank
    `src`'s, with ONE collective over a flat byte buffer (a model is a few MB)."""
    tensors = list(tensors)
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or not tensors:
        return
    flat = torch.cat([t.detach().contiguous().view(-1).view(torch.uint8) for t in tensors])
    if flat.is_cuda and _staged(group):
        host = flat.cpu()
        dist.broadcast(host, src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            nb = t.numel() * t.element_size()
            t.copy_(flat[off:off + nb].view(t.dtype).view(t.shape))
            off += nb


def broadcast_object(obj, src=0, group=None):
    """rank `src`'s python object on every rank (seeds, small configuration)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src, group=group)
    return box[0]


def barrier(group=None):
    if dist.is_initialized():
        dist.barrier(group=group)
