"""Seeded multi-rank cases shared by tests/dist_worker.py (one process per rank, all on ONE GPU) and
tests/test_gpu_dist_multiproc.py (the parent: starts the ranks, computes the unsharded oracle answers).

Every case is one database + one query batch + the options that steer the native merge (qadc_dist_collect) through a
particular branch: host-share replay with a second gather, lane-per-query device replay, a forced regrow of the gather
block, the extra payload, R above the lane replay's limit, a query the device cannot order, a rank that fails locally.
"""
import numpy as np

from helpers import float_tables, rand_codes


def shard_ranges(n, world, align=16):
    """Contiguous ranges in rank order, every range but the last a multiple of 16 codes (pyqadc/sharded.py)."""
    per = (n // world) // align * align
    return [(r * per, (n - r * per) if r == world - 1 else per) for r in range(world)]


def start_size(n, keep):
    return min(n, max(1, int(np.float32(n) * np.float32(keep)))) if n else 0


def build_case(name):
    """-> dict(M, parts, labels, keep, R, assign, tables, options, index_options, placement, extra_n)."""
    c = dict(name=name, keep=0.01, R=100, labels=None, options={}, index_options={}, placement="range", extra_n=0,
             slots=(0,))
    if name == "flat32":
        # bench.py's multi-rank step in small: 32 queries over one flat list, host-share replay + second gather;
        # the gather block starts far too small (every rank regrows alike), an extra payload rides along, two batches in flight
        rng = np.random.default_rng(501)
        c.update(M=16, parts=[rand_codes(rng, 400003, 16)], assign=np.zeros((32, 1), np.int32))
        c["tables"] = float_tables(rng, 32, 1, 16)
        c["options"] = {"dist_cap_entries": 64}
        c["extra_n"] = 37
        c["slots"] = (0, 1)
        c["repeat"] = 2                                    # the second pass fits the regrown block: merges enqueued with the batches
    elif name in ("ivf_lanes", "ivf_whole"):
        # >= 256 queries: the gathered streams are replayed on the GPU, one lane per query.  Ragged labelled partitions,
        # an empty one, some too small for every rank to hold a piece, tie-heavy tables.
        M = 16 if name == "ivf_lanes" else 32
        rng = np.random.default_rng(502 + M)
        sizes = [int(x) for x in rng.integers(50, 9000, 44)]
        sizes[3], sizes[7], sizes[11], sizes[20] = 0, 16, 37, 1
        parts = [rand_codes(rng, s, M) for s in sizes]
        perm = rng.permutation(sum(sizes)).astype(np.uint32)
        labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
        nq, ma = (300, 6) if name == "ivf_lanes" else (260, 9)
        live = [p for p, s in enumerate(sizes) if s]
        assign = np.stack([rng.choice(live, ma, replace=False) for _ in range(nq)]).astype(np.int32)
        c.update(M=M, parts=parts, labels=labels, keep=0.05, assign=assign, tables=float_tables(rng, nq, ma, M, scale=0.3), repeat=2)
        if name == "ivf_whole":
            c["placement"] = "whole"                      # whole partitions per rank, size-balanced
            c["index_options"] = {"wgq_group": 2, "wgq_group_head": 2}    # partition-major second phase under the merge
            # every rank replays EVERY query of the enqueued merge (round 4's form); all other cases take the default: the replay
            # sharded by query + a second all-gather of the heap shares (300 / 260 queries: ragged shares at 8, 3 and 7... ranks)
            c["options"] = {"dist_shard_replay": 0}
    elif name == "big_r":
        # R above the lane replay's 288: host-share replay whatever the batch size
        rng = np.random.default_rng(503)
        c.update(M=16, parts=[rand_codes(rng, 90001, 16)], assign=np.zeros((270, 1), np.int32), R=300)
        c["tables"] = float_tables(rng, 270, 1, 16)
        c["options"] = {"dist_device_nq": 1}
        c["repeat"] = 2
    elif name == "unordered":
        # query 0: constant float tables -> all-zero int8 tables; with one bound level over the whole list every code of
        # every rank's range is emitted (> 16384 candidates per rank): the ranks order it on the host and ship it anyway
        rng = np.random.default_rng(504)
        c.update(M=16, parts=[rand_codes(rng, 300007, 16)], assign=np.zeros((3, 1), np.int32))
        t = float_tables(rng, 3, 1, 16)
        t[0] = 1.0
        c["tables"] = t
        c["index_options"] = {"wgq": 0, "head_level": 0, "level_base": 1 << 22}
    elif name in ("ivf_search", "ivf_search_whole", "ivf_search_regrow", "ivf_search_fallback", "ivf_search_fewstarts",
                  "ivf_search_onerank"):
        # QUERIES in (qadc_search_submit): under the merge the front of the batch — coarse assignment, residual tables,
        # pre-scan, quantizer — is itself sharded over the ranks and all-gathered before the sharded scan.  nq not a multiple
        # of the world sizes (ragged last share), an exact tie between two coarse centroids, one query far from everything.
        M = 32 if name == "ivf_search_whole" else 16
        rng = np.random.default_rng(506 + M + len(name))
        K, dim, nq, ma = 40, 32, 277, 5
        sizes = [int(x) for x in rng.integers(300, 6000, K)]
        sizes[9] = 0
        parts = [rand_codes(rng, s_, M) for s_ in sizes]
        perm = rng.permutation(sum(sizes)).astype(np.uint32)
        labels = list(np.split(perm, np.cumsum(sizes)[:-1]))
        cb = rng.normal(size=(M, 16, dim // M)).astype(np.float32)
        coarse = rng.normal(size=(K, dim)).astype(np.float32)
        coarse[5] = coarse[3]
        queries = rng.normal(size=(nq, dim)).astype(np.float32)
        queries[17] *= 50.0
        c.update(M=M, parts=parts, labels=labels, keep=0.05, codebooks=cb, coarse=coarse, queries=queries, search_ma=ma,
                 assign=None, tables=None)
        c["index_options"] = {"wgq_group": 2, "wgq_group_head": 2}
        if name == "ivf_search_whole":
            c["placement"] = "whole"
        if name == "ivf_search_regrow":                    # stream regions far too small: the batch is re-run inside collect with the
            c["index_options"]["wgq_capacity"] = 16        # gathered tables (no second front gather), every rank redoes the merge
        if name == "ivf_search_fallback":                  # more candidates than the in-workgroup sort is allowed: the batch falls back to
            c["index_options"]["wgq_cand_cap"] = 48        # the level path, which takes the gathered int8 tables as they lie on the device
        if name == "ivf_search_onerank":                   # AUTO grouping, and ONLY RANK 0 overflows (dist_worker.py lowers its cap): the two
            c["index_options"]["wgq_group"] = 1            # strikes that end the second phase — and with it the front's all-gather — must be
                                                           # counted on every rank, from the gathered headers
        if name == "ivf_search_fewstarts":                 # fewer than R - 1 starts per query: qmax too high (status 1) — the verdict of the
            c["keep"] = 0.0004                             # rank that ran the query's front reaches every rank with the gather
    elif name == "inject":
        # the LAST rank's batch fails before the gather: every rank must return an error (no rank left in the
        # collective), and the next batch must go through
        rng = np.random.default_rng(505)
        c.update(M=16, parts=[rand_codes(rng, 120001, 16)], assign=np.zeros((8, 1), np.int32))
        c["tables"] = float_tables(rng, 8, 1, 16)
    else:
        raise KeyError(name)
    return c


CASES = ["flat32", "ivf_lanes", "ivf_whole", "ivf_search", "ivf_search_whole", "ivf_search_regrow", "ivf_search_fallback",
         "ivf_search_onerank", "ivf_search_fewstarts", "big_r", "unordered", "inject"]


def search_inputs(case):
    """assign[] and float tables of a queries-in case, evaluated by the same sequential float loops as the device feeders
    (and host/query_driver.hpp): expansion-form coarse distances through find_k_neighbors' selection; BLAS-expansion tables for ma > 1 (the
    oracle's orc_tables_expansion)."""
    q, coarse, cb, ma, M = case["queries"], case["coarse"], case["codebooks"], case["search_ma"], case["M"]
    K, dim = coarse.shape
    ds = dim // M

    import pyoracle                                              # (the checker's side only: the workers never call this function)
    nq = q.shape[0]
    assign = np.zeros((nq, ma), np.int32)
    tables = np.zeros((nq, ma, M * 16), np.float32)
    for i in range(nq):
        dist = pyoracle.cross_dists(coarse, q[i][None, :])[0]         # the reference's coarse distances (expansion form)
        assign[i] = pyoracle.select_k_neighbors(dist, ma)[0][0]     # find_k_neighbors' heaps (exact ties: as the reference leaves them)
        resid = (q[i][None, :] - coarse[assign[i]]).astype(np.float32)
        for a in range(ma):
            tables[i, a] = pyoracle.tables_expansion(cb, resid[a])[0]
    return assign, tables
