"""Multi-GPU merge of the Quick-ADC scan: one process per GPU, the code list sharded in contiguous
ranges in rank order, ONE gather of the per-shard candidate PUSH STREAMS per query batch (RCCL over
xGMI with backend "nccl"; gloo in the CPU tests), replay in shard order.

Why streams and not per-shard top-R: int8 sums tie massively, and which tied codes survive in the
reference's heap depends on push order (binheap.hpp:75-116).  Replaying, in global scan order, a
superset of the successful pushes reproduces the sequential heap array for array; re-pushing only
each shard's final top-R does not.  A shard's stream is such a superset
because its running bound starts from 127 and is therefore never below the global running bound.

The payload is small (a few thousand 5-byte entries per query and shard): the collective is
latency-bound, so it is issued once per batch, not per query, and it also carries the NEXT batch's
sharded pre-scan values (every rank pre-scans 1/world of the starts; qadc.h, qadc_prescan_submit) so
that the pre-scan costs no collective of its own.  Replay work is spread over the ranks (query q is
replayed by rank q % world, natively: qadc_merge_streams_i8) and the R-entry heaps are combined with
one all-reduce.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import merge_streams_i8


def shard_ranges(n, world, align=16):
    """Contiguous ranges in rank order; every range but the last a multiple of `align` codes."""
    per = (n // world) // align * align
    out = []
    for r in range(world):
        first = r * per
        out.append((first, (n - first) if r == world - 1 else per))
    return out


_bufs = {}
_cap_hint = {}


def _staging(words, world, device, tag="gather"):
    """Persistent (pinned, when the collective runs on the GPU) staging tensors: no per-step allocation.
    `tag` keeps buffers of different roles apart even when their sizes coincide (the result buffers of
    merge_batch must never alias the gather buffers whose `extra` payload is sliced out afterwards)."""
    key = (tag, words, world, str(device))
    b = _bufs.get(key)
    if b is None:
        pin = device.type == "cuda"
        h_in = torch.zeros(words, dtype=torch.int32, pin_memory=pin)
        h_out = torch.zeros(world * words, dtype=torch.int32, pin_memory=pin)
        d_in = h_in if not pin else torch.zeros(words, dtype=torch.int32, device=device)
        d_out = h_out if not pin else torch.zeros(world * words, dtype=torch.int32, device=device)
        b = _bufs[key] = (h_in, h_out, d_in, d_out)
        if len(_bufs) > 8:
            _bufs.pop(next(iter(_bufs)))
    return b


def _layout(nq, cap, extra_words, ma=1):
    nv, ns = (cap + 3) // 4, ((cap + 1) // 2 if ma > 1 else 0)   # no assign slots to tell apart with one probe
    return nv, ns, nq + cap + nv + ns + extra_words


def pack_stream(buf, local, nq, cap, ma=1, extra=None):
    """One rank's contribution to the gather (layout: qadc.h, qadc_merge_streams_i8):
    [nq counts][cap keys][cap int8 values, packed][ma > 1: cap u16 assign slots, packed][extra floats]."""
    ew = 0 if extra is None else extra.size
    nv, ns, words = _layout(nq, cap, ew, ma)
    assert buf.size == words
    counts = np.diff(local["offsets"]).astype(np.int64)
    total = int(counts.sum())
    buf[:nq] = counts
    if total <= cap:                                            # else: counts only, the caller retries larger
        buf[nq:nq + total] = local["keys"][:total].view(np.int32)
        buf[nq + cap:nq + cap + nv].view(np.int8)[:total] = local["vals"][:total]
        if ma > 1:
            buf[nq + cap + nv:nq + cap + nv + ns].view(np.uint16)[:total] = local["slots"][:total]
    if ew:
        buf[words - ew:] = np.ascontiguousarray(extra, np.float32).reshape(-1).view(np.int32)
    return buf


def _all_gather(h_in, h_out, d_in, d_out, device):
    if device.type == "cuda":
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        dist.all_gather_into_tensor(h_out, h_in)


def gather_prescan(vals, device):
    """vals float32 [nq][R] of this rank -> float32 [nq][world*R] of all ranks (first batch of a run; later
    batches ride on merge_batch's gather)."""
    device = torch.device(device)
    world = dist.get_world_size()
    nq, R = vals.shape
    h_in, h_out, d_in, d_out = _staging(nq * R, world, device)
    h_in.numpy()[:] = np.ascontiguousarray(vals, np.float32).reshape(-1).view(np.int32)
    _all_gather(h_in, h_out, d_in, d_out, device)
    g = h_out.numpy().view(np.float32).reshape(world, nq, R)
    return np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(nq, world * R)


def merge_batch(local, nq, R, status, device, cap=1 << 15, ma=1, extra=None):
    """All ranks call with their local ordered candidate stream
    local = dict(keys=u32[...], vals=i8[...], offsets=i64[nq+1][, slots=u16[...]]).
    ma > 1 (IVF, every probed partition range-sharded over the ranks): the global scan order is
    (assign slot, rank, position), so the per-rank streams are interleaved by their slot boundaries.
    extra: float32 [nq2][R2] payload gathered alongside (the next batch's sharded pre-scan values).
    Returns (keys u32[nq][R], vals i8[nq][R], sizes i32[nq]) identical on every rank — the heap arrays
    the reference's single sequential scan over the whole list would leave — and, with `extra`, a fourth
    item: the gathered payload float32 [nq2][world*R2]."""
    device = torch.device(device)
    world = dist.get_world_size()
    rank = dist.get_rank()
    ex = None if extra is None else np.ascontiguousarray(extra, np.float32)
    ew = 0 if ex is None else ex.size
    cap = max(cap, _cap_hint.get((nq, ma), 0))                 # what the previous batches of this shape needed
    while True:
        nv, ns, words = _layout(nq, cap, ew, ma)
        h_in, h_out, d_in, d_out = _staging(words, world, device)
        pack_stream(h_in.numpy(), local, nq, cap, ma, ex)
        _all_gather(h_in, h_out, d_in, d_out, device)
        allb = h_out.numpy().reshape(world, words)
        totals = allb[:, :nq].astype(np.int64).sum(1)
        if int(totals.max()) <= cap:
            break
        # every rank sees the same totals: same retry, with 1/8 head-room, remembered for the next batches
        cap = (int(totals.max()) * 9 // 8 + 4095) // 4096 * 4096
        _cap_hint[(nq, ma)] = cap
    # this rank replays queries rank, rank + world, ...; the others stay zero for the sum below
    res_h, res_o, res_d, _ = _staging(2 * nq * R + nq, 1, device, tag="result")
    res = res_h.numpy()
    res[:] = 0
    keys = res[:nq * R].view(np.uint32).reshape(nq, R)
    vals32 = res[nq * R:2 * nq * R].reshape(nq, R)
    sizes = res[2 * nq * R:]
    kq = np.zeros((nq, R), np.uint32)
    vq = np.zeros((nq, R), np.int8)
    sq = np.zeros(nq, np.int32)
    merge_streams_i8(allb, world, nq, R, cap, ma, rank, world, status, kq, vq, sq)
    keys[:] = kq
    vals32[:] = vq
    sizes[:] = sq
    if device.type == "cuda":
        res_d.copy_(res_h, non_blocking=True)
        dist.all_reduce(res_d, op=dist.ReduceOp.SUM)   # each query is non-zero on exactly one rank
        res_o.copy_(res_d, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        out = res_o.numpy()
    else:
        dist.all_reduce(res_h, op=dist.ReduceOp.SUM)
        out = res_h.numpy()
    keys = out[:nq * R].view(np.uint32).reshape(nq, R).copy()
    vals = out[nq * R:2 * nq * R].astype(np.int8).reshape(nq, R)
    sizes = out[2 * nq * R:].astype(np.int32)
    if ex is None:
        return keys, vals, sizes
    nq2, R2 = ex.shape
    g = allb[:, words - ew:].view(np.float32).reshape(world, nq2, R2)
    return keys, vals, sizes, np.ascontiguousarray(g.transpose(1, 0, 2)).reshape(nq2, world * R2)
