"""CPU: the C restatement (oracle/) against the golden vectors produced by the reference's own
kernels (oracle/gen_golden.py), and against the live reference build when it is present."""
import numpy as np
import pytest

import golden_cases
from helpers import rand_codes, rand_qtables


@pytest.fixture(scope="module")
def g():
    return golden_cases.load()


def test_scan_i8_matches_reference_golden(po, g):
    ncases = 0
    for c in golden_cases.scan_cases(g, po):
        for layout in ("rowmajor", "interleaved"):
            if layout == "rowmajor":
                keys, vals = po.scan_i8(c["M"], c["parts"], c["labels"], c["qt"], c["R"])
            else:
                inter = [po.interleave(p) for p in c["parts"]]
                keys, vals = po.scan_i8_interleaved(c["M"], inter, [p.shape[0] for p in c["parts"]], c["labels"],
                                                    c["qt"], c["R"])
            assert np.array_equal(keys, c["keys"]) and np.array_equal(vals, c["vals"]), (c["cid"], layout)
        if c["inter0"] is not None:
            assert np.array_equal(po.interleave(c["parts"][0]), c["inter0"]), c["cid"]
            n, cs = c["parts"][0].shape
            assert np.array_equal(po.deinterleave(c["inter0"], n, cs), c["parts"][0])
        ncases += 1
    assert ncases >= 50


def test_heap_replay_matches_reference_golden(po, g):
    for i in range(int(g["n_heap_cases"])):
        R = int(g["h%d_R" % i])
        k, v = po.heap_replay_i8(g["h%d_in_keys" % i], g["h%d_in_vals" % i], R)
        assert np.array_equal(k, g["h%d_keys" % i]) and np.array_equal(v, g["h%d_vals" % i])
        k, v = po.heap_replay_f32(g["h%d_in_keys" % i], g["f%d_in_vals" % i], R)
        assert np.array_equal(k, g["f%d_keys" % i]) and np.array_equal(v, g["f%d_vals" % i])


def test_padding_quirk_duplicates_last_code(po):
    # N = 37, R = 100: 1 sentinel + 37 codes + 11 replicas of code 36 (SURVEY.md §8 A1 iii)
    rng = np.random.default_rng(3)
    codes = rand_codes(rng, 37, 16)
    qt = rand_qtables(rng, (1,), 16, 3)
    keys, vals = po.scan_i8(16, [codes], None, qt, 100)
    assert len(keys) == 49 and (keys == 36).sum() == 12


def test_oracle_vs_live_reference_build(po):
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(77)
    for M in (16, 32):
        for n in (1, 16, 31, 33, 4096, 50001):
            codes = rand_codes(rng, n, M)
            # arbitrary int8 tables, negatives included: the block restatement keeps the reference's
            # saturating-add order, so it must agree even outside QuantizerMAX's [0,127] range
            for lo, hi in ((0, 5), (0, 127), (-128, 127)):
                qt = rng.integers(lo, hi + 1, (1, M, 16)).astype(np.int8)
                for R in (1, 100):
                    want = po.ref_scan(M, [codes], None, qt, R)
                    got = po.scan_i8_interleaved(M, [po.interleave(codes)], [n], None, qt, R)
                    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (M, n, lo, hi, R)
                    if lo >= 0:
                        got = po.scan_i8(M, [codes], None, qt, R)
                        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.array_equal(po.interleave(codes), po.ref_interleave(codes))


def test_candidates_helper_is_min127_sum(po):
    rng = np.random.default_rng(5)
    codes = rand_codes(rng, 1000, 16)
    qt = rand_qtables(rng, (), 16, 40)
    want = np.zeros(1000, np.int64)
    for b in range(8):
        want += qt[2 * b][codes[:, b] & 15].astype(np.int64) + qt[2 * b + 1][codes[:, b] >> 4].astype(np.int64)
    assert np.array_equal(po.candidates_i8(16, codes, qt), np.minimum(want, 127).astype(np.int8))


def test_pack4_nibble_order(po):
    # quantizers.hpp:49-68: even sub-quantizer -> low nibble, odd -> high nibble of byte sq/2
    assign = np.arange(32, dtype=np.int32).reshape(2, 16) % 16
    codes = po.pack4(assign, 16)
    assert codes[0, 0] == (0 | (1 << 4)) and codes[0, 7] == (14 | (15 << 4))


def test_start_size_float_product(po):
    assert po.start_size(1000000, 0.01) == max(1, int(np.float32(1000000) * np.float32(0.01)))
    assert po.start_size(5, 0.01) == 1 and po.start_size(0, 0.5) == 0


def test_quantizer_modes(po):
    t = np.array([0.0, 1.0, 2.5, 9.99, 10.0, 11.0], np.float32)
    for mode in (0, 1):
        q = po.quantize_tables(t, 0.0, 10.0, mode)
        assert q[0] == 0 and q[-1] == 127 and q[-2] == 127 and q[1] == 12 and q[2] == 31


def test_query_scan_qmax_too_high(po):
    rng = np.random.default_rng(1)
    codes = rand_codes(rng, 50, 16)          # starts = max(1, 50*0.01) = 1 < R-1 -> heap never fills
    tables = rng.random((1, 256)).astype(np.float32)
    res = po.query_scan(16, [codes], None, 0.01, [0], tables, 100)
    assert res["rc"] == 1 and res["qmax"] > 1e30


def test_scan_standard_8x8(po):
    # BASELINE config 1 (db_query.cpp scanner_simple): R sentinels then float ADC over row-major bytes
    rng = np.random.default_rng(2)
    codes = rng.integers(0, 256, (5000, 8), dtype=np.uint8)
    tables = rng.random((1, 8, 256)).astype(np.float32)
    d = tables[0][np.arange(8)[None, :], codes].astype(np.float32)
    # sum_mode 0: the source's sequential sum (query_common.hpp:106-108)
    keys, vals = po.scan_standard_u8(8, [codes], None, tables, 100, sum_mode=0)
    s = np.zeros(5000, np.float32)
    for m in range(8):
        s = (s + d[:, m]).astype(np.float32)
    assert len(keys) == 100 and np.array_equal(np.sort(vals), np.sort(s)[:100])
    # sum_mode 1 (default): the grouping of the reference as compiled, ((t1+t2)+(t3+t4)) + ((t5+t6)+(t7+t0))
    # (pinned to the reference build in tests/test_oracle_float_ref.py)
    keys, vals = po.scan_standard_u8(8, [codes], None, tables, 100)
    t = [d[:, m] for m in range(8)]
    s = ((t[1] + t[2]) + (t[3] + t[4])) + ((t[5] + t[6]) + (t[7] + t[0]))
    assert len(keys) == 100 and np.array_equal(np.sort(vals), np.sort(s)[:100])


def test_oracle_sort_keys_matches_reference_golden(po):
    """The oracle's restatement of libstdc++'s introsort (orc_sort_keys_i8) reproduces the reference build's
    bh.sort_keys() on every golden case — including the order of tied values."""
    g = golden_cases.load()
    ties = 0
    for c in golden_cases.scan_cases(g, po):
        got = po.sort_keys_i8(c["keys"], c["vals"])
        assert np.array_equal(got, c["sorted"]), c["cid"]
        ties += int(len(np.unique(c["vals"])) < len(c["vals"]))
    assert ties > 10          # the pin is only meaningful if ties occur


def test_oracle_sort_keys_vs_reference_build_random(po):
    if not po.have_ref():
        pytest.skip("reference build absent")
    rng = np.random.default_rng(11)
    for n, R, vmax in ((3, 5, 2), (16, 16, 3), (17, 17, 1), (100, 100, 5), (400, 257, 3), (5000, 1000, 2), (3000, 2048, 126)):
        keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        vals = rng.integers(0, vmax + 1, n).astype(np.int8)
        k, v, want = po.ref_heap_replay_i8(keys, vals, R, want_sorted=True)
        assert np.array_equal(po.sort_keys_i8(k, v), want), (n, R, vmax)
