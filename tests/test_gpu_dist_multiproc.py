"""The native multi-GPU merge (qadc_dist_collect) with world > 1 — on ONE GPU.

2, 3, 4 and 8 PROCESSES (fresh children, one rank each, all on GPU 0) hold the shards of one database and run the unmodified
qadc_dist_collect: pack kernel -> all-gather -> replay in global scan order (assign slot, rank, position).  The transport
is the library's shared-memory all-gather (qadc_dist_init_transport + qadc_shm_transport_*) instead of RCCL, which needs one
GPU per rank; everything around it — header protocol, identical regrow on every rank, host-share replay + second gather,
lane-per-query device replay, extra payload, host-ordered queries, failure flag — is the code an 8-GPU RCCL run executes.
Every rank's heaps must equal the UNSHARDED oracle's, array for array.

No reference counterpart: the reference is one process (query_common.hpp:351-365); the heap semantics that make the merge
order matter are binheap.hpp:75-116.
"""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from dist_cases import CASES, build_case, search_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dist_worker.py")


def run_world(world, names, tmp_path, timeout=420):
    shm = "/qadc_test_%s" % uuid.uuid4().hex[:12]
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("rank%d.npz" % r))
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(r), str(world), shm, out] + list(names),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    logs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=timeout)
            logs.append((p.returncode, se.decode(errors="replace")[-1500:]))
    finally:
        for p in procs:                                    # exactly the children started here
            if p.poll() is None:
                p.kill()
    for r, (rc, err) in enumerate(logs):
        assert rc == 0, "rank %d of %d failed:\n%s" % (r, world, err)
    return [np.load(o) for o in outs]


def oracle_heaps(po, case):
    want = []
    for q in range(case["assign"].shape[0]):
        w = po.query_scan(case["M"], case["parts"], case["labels"], case["keep"], case["assign"][q], case["tables"][q].copy(), case["R"])
        want.append(w)
    return want


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_native_merge_with_several_processes_on_one_gpu(po, world, tmp_path, scan_path):
    names = CASES
    res = run_world(world, names, tmp_path)
    for name in names:
        case = build_case(name)
        if "queries" in case:                              # queries in: the oracle gets assign[] / tables from the same float loops
            case["assign"], case["tables"] = search_inputs(case)
            case["slots"] = (0, 1, 2)                      # sharded front (two in flight, the second reversed), then unsharded
        want = oracle_heaps(po, case)
        nq = case["assign"].shape[0]
        for r in range(world):
            d = res[r]
            for s in case["slots"]:
                if "queries" in case:
                    assert np.array_equal(d["%s.assign_slot%d" % (name, s)], case["assign"]), (name, r, s)
                sizes, status = d["%s.sizes_slot%d" % (name, s)], d["%s.status_slot%d" % (name, s)]
                keys, vals = d["%s.keys_slot%d" % (name, s)], d["%s.values_slot%d" % (name, s)]
                for q in range(nq):
                    assert status[q] == want[q]["rc"], (name, r, q)
                    if want[q]["rc"]:
                        continue
                    n = int(sizes[q])
                    assert np.array_equal(keys[q, :n], want[q]["keys"]) and np.array_equal(vals[q, :n], want[q]["values"]), \
                        (name, "rank", r, "slot", s, "query", q)
                if case["extra_n"]:
                    ex = d["%s.extra_slot%d" % (name, s)]
                    assert ex.shape == (world, case["extra_n"])
                    for g in range(world):
                        assert np.array_equal(ex[g], np.arange(case["extra_n"], dtype=np.float32) + 1000.0 * g)
        if name == "flat32":
            assert all(int(res[r]["flat32.regrows"][0]) >= 1 for r in range(world))          # the 64-entry block was regrown everywhere
            # (32 queries < dist_device_nq: merged at collect time, the ranks replaying shares on the host)
            assert all(int(res[r]["flat32.dist_async_collects"][0]) == 0 for r in range(world))
        if name in ("ivf_lanes", "big_r"):                 # second pass: merges enqueued with the batch, level path and query kernel alike
            assert all(int(res[r]["%s.dist_async_collects" % name][0]) >= 1 for r in range(world)), \
                (name, [int(res[r]["%s.dist_async_collects" % name][0]) for r in range(world)])
        if name == "unordered":                            # a query the device could not order: every rank redid the merge at collect time
            assert all(int(res[r]["unordered.dist_async_collects"][0]) == 0 for r in range(world))
        if name == "unordered":
            assert all(int(res[r]["unordered.host_sorted_queries"][0]) >= 1 for r in range(world))
        if name == "ivf_whole" and scan_path != "levels" and scan_path != "levels_head":
            assert all(int(res[r]["ivf_whole.group_launches"][0]) >= 1 for r in range(world))  # grouped second phase under the merge
        if name == "ivf_search_fewstarts":
            assert all(w["rc"] == 1 for w in want)             # (what the case is for)
        if name == "ivf_search_onerank" and scan_path == "wgq":
            # rank 0 alone fell back, twice (slots 0 and 1); every rank counted both strikes from the gathered headers, so slot 2
            # neither grouped nor sharded its front on ANY rank (before: rank 0 stopped issuing the front's all-gather by itself)
            assert int(res[0]["%s.group_fallbacks" % name][0]) == 2
            assert all(int(res[r]["%s.group_fallbacks" % name][0]) == 0 for r in range(1, world))
            assert all(int(res[r]["%s.group_launches" % name][0]) == 2 for r in range(world)), \
                [int(res[r]["%s.group_launches" % name][0]) for r in range(world)]
        if "queries" in case and scan_path == "wgq":
            # slots 0 and 1 took the sharded front (every rank ran the front of 1/world of the queries), slot 2 did not
            assert all(int(res[r]["%s.front_sharded_batches" % name][0]) == 2 for r in range(world))
        if name == "ivf_search_regrow" and scan_path == "wgq":
            assert all(int(res[r]["%s.regrows" % name][0]) >= 1 for r in range(world))        # re-run inside collect, merge redone
        if name == "ivf_search_fallback" and scan_path == "wgq":
            assert all(int(res[r]["%s.group_fallbacks" % name][0]) >= 1 for r in range(world))   # level path on the gathered tables
        if name == "inject":
            # the failure of ONE rank reached every rank through the gathered headers
            assert all(int(res[r]["inject.inject_error"][0]) == 1 for r in range(world))
