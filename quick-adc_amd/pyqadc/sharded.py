"""Multi-GPU merge of the Quick-ADC scan: one process per GPU, the code list sharded in contiguous
ranges in rank order, ONE gather of the per-shard candidate PUSH STREAMS per query batch (RCCL over
xGMI with backend "nccl"; gloo in the CPU tests), replay in shard order.

Why streams and not per-shard top-R: int8 sums tie massively, and which tied codes survive in the
reference's heap depends on push order (binheap.hpp:75-116).  Replaying, in global scan order, a
superset of the successful pushes reproduces the sequential heap array for array; re-pushing only
each shard's final top-R does not.  A shard's stream is such a superset
because its running bound starts from 127 and is therefore never below the global running bound.

The payload is tiny (a few thousand 5-byte entries per query and shard): the collective is
latency-bound, so it is issued once per batch, not per query.  Replay work is spread over the ranks
(query q is replayed by rank q % world) and the R-entry heaps are combined with one all-reduce.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import replay_i8


def shard_ranges(n, world, align=16):
    """Contiguous ranges in rank order; every range but the last a multiple of `align` codes."""
    per = (n // world) // align * align
    out = []
    for r in range(world):
        first = r * per
        out.append((first, (n - first) if r == world - 1 else per))
    return out


def _pack(local, nq, cap):
    """int32 buffer: [nq counts][cap keys][cap/4 packed int8 vals][cap/2 packed u16 assign slots]."""
    counts = np.diff(local["offsets"]).astype(np.int64)
    total = int(counts.sum())
    nv = (cap + 3) // 4
    buf = np.zeros(nq + cap + nv + (cap + 1) // 2, np.int32)
    buf[:nq] = counts
    if total <= cap:
        buf[nq:nq + total] = local["keys"][:total].view(np.int32)
        buf[nq + cap:nq + cap + nv].view(np.int8)[:total] = local["vals"][:total]
        if local.get("slots") is not None:
            buf[nq + cap + nv:].view(np.uint16)[:total] = local["slots"][:total]
    return buf, total


def merge_batch(local, nq, R, status, device, cap=1 << 15, ma=1):
    """All ranks call with their local ordered candidate stream
    local = dict(keys=u32[...], vals=i8[...], offsets=i64[nq+1][, slots=u16[...]]).
    ma > 1 (IVF, every probed partition range-sharded over the ranks): the global scan order is
    (assign slot, rank, position), so the per-rank streams are interleaved by their slot boundaries.
    Returns (keys u32[nq][R], vals i8[nq][R], sizes i32[nq]) identical on every rank: the heap arrays
    the reference's single sequential scan over the whole list would leave."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    while True:
        buf, total = _pack(local, nq, cap)
        t = torch.from_numpy(buf).to(device)
        allb = torch.empty(world * buf.size, dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(allb, t)
        allb = allb.cpu().numpy().reshape(world, buf.size)
        totals = allb[:, :nq].astype(np.int64).sum(1)
        if int(totals.max()) <= cap:
            break
        cap = int(2 ** np.ceil(np.log2(totals.max() + 1)))  # every rank sees the same totals: same retry
    keys = np.zeros((nq, R), np.uint32)
    vals = np.zeros((nq, R), np.int8)
    sizes = np.zeros(nq, np.int32)
    offs = np.concatenate([np.zeros((world, 1), np.int64), np.cumsum(allb[:, :nq].astype(np.int64), 1)], 1)
    for q in range(rank, nq, world):
        if status is not None and status[q]:
            continue
        ks, vs = [], []
        nv = (cap + 3) // 4
        segs = []
        for g in range(world):
            a, b = int(offs[g, q]), int(offs[g, q + 1])
            k = allb[g, nq + a:nq + b].view(np.uint32)
            v = allb[g, nq + cap:nq + cap + nv].view(np.int8)[a:b]
            if ma == 1:
                segs.append((k, v, None))
            else:
                sl = allb[g, nq + cap + nv:].view(np.uint16)[a:b]          # ascending: a rank scans in assign order
                segs.append((k, v, np.searchsorted(sl, np.arange(ma + 1))))
        for slot in range(ma):
            for k, v, bnd in segs:
                if bnd is None:
                    ks.append(k)
                    vs.append(v)
                else:
                    ks.append(k[bnd[slot]:bnd[slot + 1]])
                    vs.append(v[bnd[slot]:bnd[slot + 1]])
        k, v = replay_i8(np.concatenate(ks), np.concatenate(vs), R, sentinel=True)   # db_query_4.cpp:276
        keys[q, :len(k)] = k
        vals[q, :len(v)] = v
        sizes[q] = len(k)
    res = np.concatenate([keys.view(np.int32).reshape(-1), vals.astype(np.int32).reshape(-1), sizes])
    rt = torch.from_numpy(res).to(device)
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)   # each query is non-zero on exactly one rank
    res = rt.cpu().numpy()
    keys = res[:nq * R].view(np.uint32).reshape(nq, R).copy()
    vals = res[nq * R:2 * nq * R].astype(np.int8).reshape(nq, R)
    sizes = res[2 * nq * R:].astype(np.int32)
    return keys, vals, sizes
