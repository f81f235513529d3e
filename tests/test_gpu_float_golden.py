"""GPU: the float half of the HIP path (pre-scan sums, qmax select, qmin / clamp, QuantizerMAX, the direct table form),
called through the C-ABI, against

  * tests/golden/ref_query_scan_cases.npz — outputs of the reference's OWN scanner_4 / QuantizerMAX / scan_4 / scan_avx_4
    as compiled with its flags (oracle/gen_golden_float.py), and
  * the oracle, whose float half is pinned to that same build by tests/test_oracle_float_ref.py.

SURVEY.md section 8 rows A5 / A6 / A7 / A8 / A10 and (c).  Bit-exact bar throughout."""
import numpy as np
import pytest

import golden_cases
from helpers import rand_codes, float_tables, heaps_equal

pytestmark = pytest.mark.gpu

FMAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def pyqadc():
    import pyqadc
    return pyqadc


@pytest.fixture(scope="module")
def g():
    return golden_cases.load_query_scan()


def test_gpu_query_scan_matches_reference_golden(pyqadc, po, g, scan_path):
    """qadc_query_scan on the reference-made fixtures: start sizes, exit status, qmin, qmax, int8 tables, the in-place clamp
    and the final heap, on every scan path of the library (conftest's scan_path)."""
    n = nexit = 0
    indexes = {}
    for c in golden_cases.query_scan_cases(g, po):
        key = (id(c["parts"]), float(c["keep"]))
        if key not in indexes:
            idx = pyqadc.Index(c["M"])
            idx.add_partitions(c["parts"], c["labels"])
            idx.finalize(float(c["keep"]))
            indexes[key] = idx
        idx = indexes[key]
        assign = c["assign"]
        for a, p in enumerate(assign):
            assert idx.start_size(int(p)) == c["starts"][a], c["cid"]
        tb = c["tables"].copy().reshape(1, len(assign), -1)
        res = idx.query_scan(assign.reshape(1, -1), tb, c["R"], want_qtables=True)
        assert res["status"][0] == c["exit"], c["cid"]
        assert res["qmax"][0] == c["qmax"], (c["cid"], res["qmax"][0], c["qmax"])
        if c["exit"]:
            assert res["sizes"][0] == 0 and res["qmax"][0] > 1e30
            nexit += 1
            continue
        assert res["qmin"][0] == c["qmin"], c["cid"]
        assert np.array_equal(res["qtables"][0], c["qt"]), c["cid"]
        assert np.array_equal(tb[0], np.where(c["tables"] < 0, np.float32(0), c["tables"])), c["cid"]
        assert heaps_equal(res["heaps"][0], (c["keys"], c["vals"])), c["cid"]
        assert np.array_equal(pyqadc.sort_keys_i8(*res["heaps"][0]), c["sorted"]), c["cid"]     # kv_binheap::sort_keys
        n += 1
    for idx in indexes.values():
        idx.close()
    assert n >= 20 and nexit >= 3


def test_gpu_per_code_values_match_reference_golden(pyqadc, po, g):
    """The per-code arrays of the fixtures: int8 `cand` of every code (qadc_candidates_i8), and the float pre-scan sums of
    the starts as a sorted multiset — qadc_scan_start with heap capacity k returns the k-th smallest start sum
    (tmp_bh.max() after k or more pushes), k = 1 .. starts."""
    n = 0
    for c in golden_cases.query_scan_cases(g, po):
        if "fcand" not in c:
            continue
        idx = pyqadc.Index(c["M"])
        idx.add_partitions(c["parts"], c["labels"])
        idx.finalize(float(c["keep"]))
        p0 = int(c["assign"][0])
        assert np.array_equal(idx.candidates_i8(p0, c["qt"][0]), c["cand"]), c["cid"]
        want = np.sort(c["fcand"])
        for path in ("one", "mq"):
            nq = 8 if path == "mq" else 1                    # (the multi-query pre-scan kernel takes groups of queries that share their starts)
            tb = np.repeat(c["tables"][None, :1, :], nq, 0).copy()
            got = np.array([idx.scan_start(np.full((nq, 1), p0, np.int32), tb, k)[nq - 1] for k in range(1, len(want) + 1)])
            assert np.array_equal(got, want), (c["cid"], path)
        assert idx.scan_start(np.full((1, 1), p0, np.int32), c["tables"][None, :1, :].copy(), len(want) + 1)[0] == FMAX
        idx.close()
        n += 1
    assert n == 4


@pytest.mark.parametrize("M", [16, 32])
def test_gpu_sum_mode_0_is_the_source_order(pyqadc, po, M, scan_path):
    """Option "sum_mode" = 0 keeps scan_4's source order (query_common.hpp:72-80); the default (1) is the reference as
    compiled.  The two give different qmax on ordinary inputs, and each equals the oracle's same mode."""
    rng = np.random.default_rng(90 + M)
    codes = rand_codes(rng, 60000, M)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(0.02)
    nq = 8
    tables = float_tables(rng, nq, 1, M, scale=3.7)
    differ = 0
    qmax = {}
    for mode in (1, 0, 1):
        idx.set_option("sum_mode", mode)
        res = idx.query_scan(np.zeros((nq, 1), np.int32), tables.copy(), 100, want_qtables=True)
        for q in range(nq):
            want = po.query_scan(M, [codes], None, 0.02, [0], tables[q].copy(), 100, sum_mode=mode)
            assert res["qmax"][q] == np.float32(want["qmax"]), (mode, q)
            assert np.array_equal(res["qtables"][q], want["qtables"])
            assert heaps_equal(res["heaps"][q], (want["keys"], want["values"]))
        qmax[mode] = res["qmax"].copy()
    differ = int((qmax[0] != qmax[1]).sum())
    assert differ > 0
    idx.close()


@pytest.mark.parametrize("M,dim", [(16, 128), (32, 128), (16, 256), (32, 256), (16, 64), (16, 480), (32, 960), (16, 960),
                                   (32, 96), (16, 96), (16, 1024)])
def test_gpu_direct_tables_add_like_the_reference(pyqadc, po, M, dim):
    """The direct table form on the device (flat database, ma = 1: nns_engine's "single" form, query_common.hpp:292-294)
    adds like fmanorm as compiled wherever the reference has an instance of the sub-vector width (sq_dim 4 / 8 / 16 / 30 / 32 /
    60 / 64 here), and sequentially for the widths it has none of (2, 3, 6, 15): the oracle's orc_tables_direct follows the
    same rule and is pinned to the reference build on the CPU.  Compared through everything the tables decide: qmax-scaled
    int8 tables and the heap."""
    rng = np.random.default_rng(M * 7 + dim)
    ds = dim // M
    codes = rand_codes(rng, 30000, M)
    cb = rng.normal(size=(M, 16, ds)).astype(np.float32)
    idx = pyqadc.Index(M)
    idx.add_partitions([codes])
    idx.finalize(0.02)
    idx.set_pq(cb)
    nq, R = 24, 100
    queries = rng.normal(size=(nq, dim)).astype(np.float32)
    queries[3, :ds] = cb[0, 5]                                           # an exact zero entry
    for mode in (1, 0):
        idx.set_option("sum_mode", mode)
        idx.search_submit(0, queries, 1, R)
        res = idx.search_collect(0)
        qts = idx.slot_qtables(0, 0, nq, 1)
        ndiff = 0
        for q in range(nq):
            tables = po.tables_direct(cb, queries[q], sum_mode=mode).reshape(1, M * 16)
            ndiff += int((tables != po.tables_direct(cb, queries[q], sum_mode=1 - mode).reshape(1, M * 16)).sum())
            want = po.query_scan(M, [codes], None, 0.02, [0], tables.copy(), R, sum_mode=mode)
            assert res["status"][q] == want["rc"] == 0
            assert np.array_equal(qts[q], want["qtables"]), (mode, q)
            sz = res["sizes"][q]
            assert heaps_equal((res["keys"][q, :sz], res["values"][q, :sz]), (want["keys"], want["values"])), (mode, q)
        if ds % 8 in (0, 4, 6) and ds >= 4:
            assert ndiff > 0                                             # (the two groupings really differ on these inputs)
    idx.close()


def test_gpu_labels_mode_is_read_off_partition_0_even_if_empty(pyqadc):
    """compute_sizes takes has_labels from partition 0 whatever its size (db_query_4.cpp:105-110); an index_db whose
    partition 0 is empty hands out a null labels pointer for it, so the reference refuses the database ("Some partitions
    have labels and some have not", 118-124; reproduced with the reference build in tests/test_oracle_float_ref.py)."""
    rng = np.random.default_rng(5)
    a, b = rand_codes(rng, 50, 16), rand_codes(rng, 60, 16)
    la, lb = np.arange(50, dtype=np.uint32), np.arange(60, dtype=np.uint32)
    e = np.zeros((0, 8), np.uint8)
    idx = pyqadc.Index(16)
    with pytest.raises(pyqadc.QadcError, match="Some partitions have labels and some have not"):
        idx.add_partitions([e, a, b], [None, la, lb])
    idx.close()
    idx = pyqadc.Index(16)
    idx.add_partitions([a, e, b], [la, None, lb])                        # an empty partition elsewhere is only a warning
    idx.finalize(0.5)
    idx.close()


def test_stream_layout_check_of_a_fresh_process():
    """qadc_stream_layout / qadc_stream_probe: in a process that created nothing on the GPU before the
    library's stream set, the ten pairs that matter are unobstructed ("... | ok"), an obstruction the layout has BY DESIGN is
    there (a CU-hungry launch on the highest-priority scan stream of the query-kernel batches holds up the level path's
    lowest-priority scan stream: they share a pipe, and a batch uses one or the other) and a marker on another pipe starts within
    a few microseconds.  Run in a child process: the test process itself has a history (torch, other indexes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import pyqadc; pyqadc.device_prepare(0); print(pyqadc.stream_layout(0)); "
            "print(min(pyqadc.stream_probe(4, 0)[0] for _ in range(3)), min(pyqadc.stream_probe(0, 1)[0] for _ in range(3)))"
            % os.path.join(root, "quick-adc_amd"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("QADC_")}
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().split("\n")
    assert out[-2].startswith("S,C,O,F,W,L,M0 | "), out
    blocked, free = [float(x) for x in out[-1].split()]
    assert free < 40.0, out
    if out[-2].endswith("| ok"):                                 # (a box shared with other jobs may obstruct a probe: then only the format is checked)
        assert blocked > 60.0, out
