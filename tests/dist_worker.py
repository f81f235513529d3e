"""One rank of the multi-process test of the native multi-GPU merge (tests/test_gpu_dist_multiproc.py).

    python dist_worker.py <rank> <world> <shm_name> <out.npz> <case> [<case> ...]

All ranks share GPU 0; the transport is the library's shared-memory all-gather (qadc_shm_transport_*), everything else is
the unmodified qadc_dist_collect.  No torch here: a rank is numpy + ctypes + libqadc_hip.so.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "quick-adc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import pyqadc  # noqa: E402
from dist_cases import build_case, shard_ranges, start_size  # noqa: E402


def build_index(case, rank, world):
    M, keep = case["M"], case["keep"]
    idx = pyqadc.Index(M, 0)
    sizes = [p.shape[0] for p in case["parts"]]
    owner = pyqadc.place_partitions(sizes, world) if case["placement"] == "whole" else None
    for p, codes in enumerate(case["parts"]):
        n = sizes[p]
        lab = None if case["labels"] is None else case["labels"][p]
        if n == 0:
            idx.add_partitions([np.zeros((0, M // 2), np.uint8)], None if lab is None else [np.zeros(0, np.uint32)])
            continue
        if owner is None:
            first, ln = shard_ranges(n, world)[rank]
        else:
            first, ln = (0, n) if owner[p] == rank else (0, 0)
        idx.add_partition_shard(codes[first:first + ln], first, n, labels=None if (lab is None or ln == 0) else lab[first:first + ln],
                                starts=codes[:start_size(n, keep)])
    idx.finalize(keep)
    if "queries" in case:
        idx.set_pq(case["codebooks"])
        idx.set_coarse(case["coarse"])
    for k, v in case["index_options"].items():
        idx.set_option(k, v)
    return idx


def run_case(name, rank, world, transport):
    case = build_case(name)
    idx = build_index(case, rank, world)
    idx.dist_init_transport(transport)
    for k, v in case["options"].items():
        idx.set_option(k, v)
    idx.set_option("profile", 1)
    out = {}
    extra = None
    if case["extra_n"]:
        extra = (np.arange(case["extra_n"], dtype=np.float32) + 1000.0 * rank)
    R = case["R"]
    if name == "inject":
        if rank == world - 1:
            idx.set_option("dist_inject_failure", 1)
        idx.submit(0, case["assign"], case["tables"].copy(), R)
        try:
            idx.dist_collect(0)
            out["inject_error"] = np.array([0])
        except pyqadc.QadcError as e:
            out["inject_error"] = np.array([1])
            out["inject_message"] = np.frombuffer(str(e).encode()[:200].ljust(200), np.uint8)
    if name == "ivf_search_onerank" and rank == 0:
        idx.set_option("wgq_cand_cap", 48)                 # this rank's grouped batches overflow and fall back; the others' do not
    if "queries" in case:                                  # queries in: two batches in flight, the second with the front unsharded
        idx.search_submit(0, case["queries"], case["search_ma"], R)
        idx.search_submit(1, case["queries"][::-1].copy(), case["search_ma"], R)
        for s, rev in ((0, False), (1, True)):
            got = idx.dist_collect(s)
            for k in ("keys", "values", "sizes", "status"):
                out["%s_slot%d" % (k, s)] = got[k][::-1].copy() if rev else got[k]
            a = idx.slot_assign(s, case["queries"].shape[0], case["search_ma"])
            out["assign_slot%d" % s] = a[::-1].copy() if rev else a
        if name != "ivf_search_onerank":                   # (that case: the front stays sharded if the index still groups)
            idx.set_option("dist_shard_front", 0)
        idx.search_submit(2, case["queries"], case["search_ma"], R)
        got = idx.dist_collect(2)
        for k in ("keys", "values", "sizes", "status"):
            out["%s_slot2" % k] = got[k]
        out["assign_slot2"] = idx.slot_assign(2, case["queries"].shape[0], case["search_ma"])
        out["group_launches"] = np.array([idx.profile()["group_launches"]])
        out["front_sharded_batches"] = np.array([idx.profile()["front_sharded_batches"]])
        out["regrows"] = np.array([idx.profile()["regrows"]])
        out["group_fallbacks"] = np.array([idx.profile()["group_fallbacks"]])
        out["dist_async_collects"] = np.array([idx.profile()["dist_async_collects"]])
        idx.close()
        return out
    for rep in range(case.get("repeat", 1)):               # (a repeat: the buffers the first pass had to regrow now fit)
        for s in case["slots"]:                            # (two batches in flight where the case says so)
            idx.submit(s, case["assign"], case["tables"].copy(), R)
        for s in case["slots"]:
            got = idx.dist_collect(s, extra=extra)
            for k in ("keys", "values", "sizes", "status"):
                out["%s_slot%d" % (k, s)] = got[k]
            if extra is not None:
                out["extra_slot%d" % s] = got["extra"]
    prof = idx.profile()
    out["regrows"] = np.array([prof["regrows"]])
    out["group_launches"] = np.array([prof["group_launches"]])
    out["host_sorted_queries"] = np.array([prof["host_sorted_queries"]])
    out["dist_async_collects"] = np.array([prof["dist_async_collects"]])
    idx.close()
    return out


def main():
    rank, world, shm_name, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    names = sys.argv[5:]
    transport = pyqadc.ShmTransport(shm_name, rank, world, slot_bytes=32 << 20, timeout_s=90.0)
    res = {}
    for name in names:
        for k, v in run_case(name, rank, world, transport).items():
            res["%s.%s" % (name, k)] = v
    transport.close()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main()
