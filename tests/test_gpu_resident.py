"""GPU: the resident query kernel (option "resident"): a lone synchronous query_scan — what nns_engine issues per query,
query_common.hpp:278-307 — answered through a bell in mapped memory by workgroups that stay on the GPU, instead of a launch.
Same bar as every other path: heaps bit-equal to the reference-made fixtures and to the oracle; plus the protocol's corners —
the kernel leaving by its idle limit, other batches in between, options changing under it, capacities overflowing."""
import time

import numpy as np
import pytest

import golden_cases
from helpers import float_tables, heaps_equal, rand_codes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


def _lone(idx, po, M, parts, labels, keep, assign, tables, R, **modes):
    res = idx.query_scan(assign.reshape(1, -1), tables.copy().reshape(1, len(assign), -1), R)
    want = po.query_scan(M, parts, labels, keep, assign, tables.copy(), R, **modes)
    if want["rc"] != 0:
        assert res["status"][0] == 1
        return
    assert res["status"][0] == 0
    assert heaps_equal(res["heaps"][0], (want["keys"], want["values"]))


def test_resident_kernel_on_the_reference_fixtures(pyqadc, po):
    """The reference-made query_scan fixtures (tests/golden/ref_query_scan_cases.npz) through the resident kernel: exit status,
    qmin, qmax, the in-place clamp and the final heap.  wgq_split_codes at its minimum, so that the fixtures' short lists are
    spread over several workgroups (the shape the resident kernel takes)."""
    g = golden_cases.load_query_scan()
    indexes, n, served = {}, 0, 0
    for c in golden_cases.query_scan_cases(g, po):
        key = (id(c["parts"]), float(c["keep"]))
        if key not in indexes:
            idx = pyqadc.Index(c["M"])
            idx.add_partitions(c["parts"], c["labels"])
            idx.finalize(float(c["keep"]))
            idx.set_option("wgq", 2)
            idx.set_option("wgq_split_codes", 1024)
            idx.set_option("resident", 1)
            indexes[key] = idx
        idx = indexes[key]
        assign = c["assign"]
        tb = c["tables"].copy().reshape(1, len(assign), -1)
        res = idx.query_scan(assign.reshape(1, -1), tb, c["R"])
        assert res["status"][0] == c["exit"], c["cid"]
        assert res["qmax"][0] == c["qmax"], c["cid"]
        if c["exit"]:
            continue
        assert res["qmin"][0] == c["qmin"], c["cid"]
        assert np.array_equal(tb[0], np.where(c["tables"] < 0, np.float32(0), c["tables"])), c["cid"]
        assert heaps_equal(res["heaps"][0], (c["keys"], c["vals"])), c["cid"]
        n += 1
    for idx in indexes.values():
        served += idx.profile()["resident_queries"]
        idx.close()
    assert n >= 20
    assert served >= 1, "no fixture took the resident kernel: the test does not test it"


@pytest.mark.parametrize("M,labelled", [(16, False), (16, True), (32, False)])
def test_resident_query_loop_matches_oracle(pyqadc, po, M, labelled):
    """A synchronous query loop with everything that can happen to the resident kernel between two queries: R and ma change,
    the caller goes quiet for longer than the idle limit, a batch of several queries is submitted (the kernel leaves first),
    options change, the stream capacity overflows (regrow through an ordinary launch)."""
    rng = np.random.default_rng(77 + M + labelled)
    sizes = [150001, 40000, 5000, 17, 0, 90000]
    parts = [rand_codes(rng, n, M) for n in sizes]
    labels = [rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) for n in sizes] if labelled else None
    keep = 0.01
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels=labels)
    idx.finalize(keep)
    idx.set_option("resident", 1)
    idx.set_option("resident_idle_us", 3000)
    modes = dict(quant_mode=1, sum_mode=1)
    for it in range(48):
        ma = int(rng.integers(1, 4))
        assign = rng.permutation(len(sizes))[:ma].astype(np.int32)
        if it % 3 == 0:
            assign[0] = 0                                        # a long first partition: the register-resident first block
        R = int(rng.choice([1, 10, 100, 100, 257]))
        tables = float_tables(rng, 1, ma, M, scale=float(rng.choice([0.2, 1.0])))[0]
        if it % 4 == 1:
            tables = np.round(tables * 2) / 2                    # tie-heavy
        _lone(idx, po, M, parts, labels, keep, assign, tables, R, **modes)
        if it == 10:
            time.sleep(0.02)                                     # the kernel has left by now: relaunched by the next query
        if it == 20:                                             # a batch in between: the kernel is told to leave first
            a4 = np.stack([rng.permutation(len(sizes))[:2] for _ in range(4)]).astype(np.int32)
            t4 = float_tables(rng, 4, 2, M)
            r4 = idx.query_scan(a4, t4.copy(), 100)
            for q in range(4):
                want = po.query_scan(M, parts, labels, keep, a4[q], t4[q].copy(), 100, **modes)
                assert heaps_equal(r4["heaps"][q], (want["keys"], want["values"]))
        if it == 28:
            modes = dict(quant_mode=0, sum_mode=0)
            idx.set_option("quant_mode", 0)
            idx.set_option("sum_mode", 0)
        if it == 34:
            idx.set_option("wgq_capacity", 64)                   # streams overflow: regrown by an ordinary launch
        if it == 38:
            idx.set_option("wgq_capacity", 4096)
            idx.set_option("wgq_split", 5)
    pr = idx.profile()
    idx.close()
    assert pr["resident_queries"] >= 30 and pr["resident_launches"] >= 3, pr


def test_resident_kernel_leaving_under_the_caller(pyqadc, po):
    """An idle limit as short as the gaps between the calls: workgroups leave while queries arrive.  Every answer is still
    the oracle's (a query that finds a workgroup gone is served by an ordinary launch), and nothing hangs."""
    rng = np.random.default_rng(5)
    M, n = 16, 100000
    parts = [rand_codes(rng, n, M)]
    idx = pyqadc.Index(M)
    idx.add_partitions(parts, labels=None)
    idx.finalize(0.01)
    idx.set_option("resident", 1)
    idx.set_option("resident_idle_us", 50)
    assign = np.zeros(1, np.int32)
    tabs = [float_tables(rng, 1, 1, M)[0] for _ in range(8)]
    wants = [po.query_scan(M, parts, None, 0.01, assign, t.copy(), 100, quant_mode=1, sum_mode=1) for t in tabs]
    t_end = time.time() + 60
    for it in range(400):
        k = it % 8
        res = idx.query_scan(assign.reshape(1, 1), tabs[k].copy().reshape(1, 1, -1), 100)
        assert res["status"][0] == 0
        assert heaps_equal(res["heaps"][0], (wants[k]["keys"], wants[k]["values"])), it
        if it % 5 == 0:
            time.sleep(float(rng.choice([0, 20e-6, 60e-6, 200e-6])))
        assert time.time() < t_end, "the loop takes far too long: something waits for a workgroup that is gone"
    pr = idx.profile()
    idx.close()
    assert pr["resident_queries"] + pr["resident_fallbacks"] >= 400, pr
