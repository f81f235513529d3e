"""the reference's on-disk formats (host/qadc_io.hpp).  CPU tests: files written by numpy in the
reference's layouts are read by the C++ readers and written back byte for byte; error behaviour.  GPU test: the
reference's own command line (db_query_4 -r -m -k -b DB QUERIES GROUNDTRUTH) end to end on such files."""
import os
import subprocess

import numpy as np
import pytest

import io_formats as iof
from test_scanner_hip_cpp import DRIVER, _parse_dump, build_driver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "cpp", "io_roundtrip")


def build_tool():
    """host/qadc_io.hpp's test driver, built with AddressSanitizer + UndefinedBehaviorSanitizer (CPU code: the readers
    parse untrusted files) — any report aborts the tool and fails the round-trip tests below."""
    tmp = "%s.%d.tmp" % (TOOL, os.getpid())
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-Wall", "-Werror", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", os.path.join(ROOT, "tests", "cpp", "io_roundtrip.cpp"), "-o", tmp])
    os.replace(tmp, TOOL)


def run_tool(*args):
    p = subprocess.run([TOOL] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode().strip(), p.stderr.decode().strip()


def test_vecs_files_round_trip(tmp_path):
    build_tool()
    rng = np.random.default_rng(0)
    f = rng.normal(size=(37, 24)).astype(np.float32)
    b = rng.integers(0, 256, (50, 128), dtype=np.uint8)
    i = rng.integers(-5, 1000, (9, 100)).astype(np.int32)
    for name, a in (("a.fvecs", f), ("b.bvecs", b), ("c.ivecs", i)):
        src, dst = str(tmp_path / name), str(tmp_path / (name + ".out.fvecs"))
        iof.write_vecs(src, a)
        rc, out, err = run_tool("vecs", src, dst)
        assert rc == 0 and out == "vecs dim=%d count=%d" % (a.shape[1], a.shape[0]), err
        back = np.fromfile(dst, np.uint8).reshape(a.shape[0], 4 + 4 * a.shape[1])
        assert np.all(back[:, :4].view(np.int32) == a.shape[1])
        assert np.array_equal(back[:, 4:].copy().view(np.float32), a.astype(np.float32))   # everything becomes float
    # errors: unknown extension, a record with another dimension (message + exit 1, vector_io.cpp:20-38)
    rc, _, err = run_tool("vecs", str(tmp_path / "a.fvecs") + ".txt", str(tmp_path / "x"))
    assert rc == 1 and "Unknown extension" in err
    bad = str(tmp_path / "bad.fvecs")
    iof.write_vecs(bad, f)
    raw = bytearray(open(bad, "rb").read())
    raw[(4 + 4 * 24) * 3:(4 + 4 * 24) * 3 + 4] = (25).to_bytes(4, "little")
    open(bad, "wb").write(bytes(raw))
    rc, _, err = run_tool("vecs", bad, str(tmp_path / "x"))
    assert rc == 1 and "Vector 3 has 25 dimensions" in err


def test_quantizer_data_files_round_trip(tmp_path):
    build_tool()
    rng = np.random.default_rng(1)
    cb = rng.normal(size=(16, 16, 8)).astype(np.float32)
    rot = rng.normal(size=(128, 128)).astype(np.float32)
    for name, r in (("q.pq.data", None), ("q.opq.data", rot)):
        src, dst = str(tmp_path / name), str(tmp_path / ("out." + name))
        iof.write_pq_data(src, cb, r)
        rc, out, err = run_tool("pq", src, dst)
        assert rc == 0 and out == "pq dim=128 m=16 b=4 opq=%d" % (r is not None), err
        assert open(src, "rb").read() == open(dst, "rb").read()
    rc, _, err = run_tool("pq", str(tmp_path / "q.pq.data") + "x", str(tmp_path / "x"))
    assert rc == 1 and "Filename must end with: .pq.data or .opq.data" in err


def test_database_archives_round_trip(tmp_path):
    build_tool()
    rng = np.random.default_rng(2)
    cb = rng.normal(size=(16, 16, 8)).astype(np.float32)
    rot = np.linalg.qr(rng.normal(size=(128, 128)))[0].astype(np.float32)
    codes = rng.integers(0, 256, (1000, 8), dtype=np.uint8)
    sizes = [10, 0, 333, 1]
    parts = [rng.integers(0, 256, (n, 8), dtype=np.uint8) for n in sizes]
    labels = [rng.integers(0, 1 << 31, n).astype(np.uint32) for n in sizes]
    cents = rng.normal(size=(4, 128)).astype(np.float32)
    cases = []
    for r in (None, rot):
        p = str(tmp_path / ("flat%d.db" % (r is not None)))
        iof.write_flat_db(p, cb, codes, r)
        cases.append((p, "db indexed=0 dim=128 m=16 b=4 opq=%d parts=0 codes_count=1000 code_bytes=8000 labels=0" % (r is not None)))
        p = str(tmp_path / ("index%d.db" % (r is not None)))
        iof.write_index_db(p, cb, cents, parts, labels, r)
        cases.append((p, "db indexed=1 dim=128 m=16 b=4 opq=%d parts=4 codes_count=0 code_bytes=%d labels=%d"
                      % (r is not None, 8 * sum(sizes), sum(sizes))))
    for p, summary in cases:
        rc, out, err = run_tool("db", p, p + ".out")
        assert rc == 0 and out == summary, (out, err)
        assert open(p, "rb").read() == open(p + ".out", "rb").read()
    raw = open(cases[0][0], "rb").read()
    open(str(tmp_path / "cut.db"), "wb").write(raw[:len(raw) // 2])
    rc, _, err = run_tool("db", str(tmp_path / "cut.db"), str(tmp_path / "x"))
    assert rc == 1 and "truncated" in err


def _need_ref_io(po):
    if not po.have_ref_io():
        pytest.skip("oracle/_ref/libqadc_ref_io.so not built (no /root/reference here and no prebuilt copy)")


def test_quantizer_data_files_against_the_reference_readers(po, tmp_path):
    """N3, .pq.data / .opq.data pinned: files written by the numpy writer (convert-quantizer.py's layout) and written back by
    host/qadc_io.hpp's pq_to_data_file are read by the REFERENCE's own pq_from_data_file factory (quantizers.cpp:27-46,
    89-103, compiled from line ranges into oracle/_ref/libqadc_ref_float.so) to the same header, codebooks and rotation bit
    for bit; and parse_data_filename (58-87) accepts / refuses exactly the names our reader does."""
    if not po.have_ref_float():
        pytest.skip("oracle/_ref/libqadc_ref_float.so not built")
    build_tool()
    rng = np.random.default_rng(11)
    n = 0
    for m, k, sq_dim, opq in ((16, 16, 8, False), (16, 16, 8, True), (32, 16, 3, True), (8, 256, 16, False), (32, 16, 4, False),
                              (16, 16, 60, True)):
        cb = rng.normal(size=(m, k, sq_dim)).astype(np.float32)
        dim = m * sq_dim
        rot = rng.normal(size=(dim, dim)).astype(np.float32) if opq else None
        name = "q%d.%s.data" % (n, "opq" if opq else "pq")
        src, dst = str(tmp_path / name), str(tmp_path / ("out." + name))
        iof.write_pq_data(src, cb, rot)
        rc, out, err = run_tool("pq", src, dst)
        assert rc == 0 and out == "pq dim=%d m=%d b=%d opq=%d" % (dim, m, int(np.log2(k)), opq), err
        for f in (src, dst):
            got = po.reff_pq_from_data_file(f)
            assert (got["dim"], got["m"], got["b"], got["is_opq"]) == (dim, m, int(np.log2(k)), opq), f
            assert np.array_equal(got["centroids"].view(np.uint32), cb.reshape(-1).view(np.uint32)), f
            if opq:
                assert np.array_equal(got["rotation"].view(np.uint32), rot.view(np.uint32)), f
        n += 1
    # names: the last extension must be ".data", the one before it ".pq" or ".opq" (case-sensitive; directories are not looked at)
    cb = rng.normal(size=(16, 16, 2)).astype(np.float32)
    rot = rng.normal(size=(32, 32)).astype(np.float32)
    for name, want in (("a.pq.data", 0), ("a.opq.data", 1), ("a.b.c.opq.data", 1), (".pq.data", 0), ("a.data", 101), ("apq.data", 101),
                       ("a.pq.dat", 101), ("a.pq.data.bak", 101), ("noext", 101), ("a.PQ.data", 101), ("a.opq.pq.data", 0),
                       ("a.pq.opq.data", 1), ("a..data", 101)):
        f = str(tmp_path / name)
        iof.write_pq_data(f, cb, rot if want == 1 else None)
        assert po.reff_parse_data_filename(f) == want, name
        rc, out, err = run_tool("pq", f, str(tmp_path / "o.pq.data" if want != 1 else tmp_path / "o.opq.data"))
        if want == 101:
            assert rc == 1 and "Invalid data filename: " + f in err and "Filename must end with: .pq.data or .opq.data" in err, name
        else:
            assert rc == 0 and out == "pq dim=32 m=16 b=4 opq=%d" % want, (name, err)
    # (a short file: the reference's fstream reads leave the tail of the codebooks uninitialised without a word,
    # quantizers.cpp:17-33 checks nothing; ours refuses the file — stricter on purpose, stated in DESIGN.md section 8)
    short = str(tmp_path / "short.pq.data")
    open(short, "wb").write(open(str(tmp_path / "a.pq.data"), "rb").read()[:100])
    rc, _, err = run_tool("pq", short, str(tmp_path / "o2.pq.data"))
    assert rc == 1 and "Invalid quantizer file: " + short in err


def test_vecs_files_against_the_reference_reader_and_writer(po, tmp_path):
    """host/qadc_io.hpp's vecs side pinned to the reference's own vector_io.cpp / vector_io.hpp compiled
    into oracle/_ref (vector_io.cpp:40-91, vector_io.hpp:69-166): files written by the REFERENCE's save_vectors are read by
    our readers bit for bit, files written by OUR save_vectors are read by the reference's load_vectors_by_extension bit for
    bit, both readers agree on a file with trailing garbage (count = size / record size) and on both error cases."""
    _need_ref_io(po)
    build_tool()
    rng = np.random.default_rng(11)
    data = {"fvecs": rng.normal(size=(41, 24)).astype(np.float32),
            "bvecs": rng.integers(0, 256, (53, 128), dtype=np.uint8),
            "ivecs": rng.integers(-(1 << 30), 1 << 30, (9, 100)).astype(np.int32)}
    data["fvecs"][3, 5], data["fvecs"][4, 0], data["fvecs"][5, 1] = np.float32("nan"), np.float32("-inf"), np.float32(-0.0)
    kind = {"fvecs": "f32", "bvecs": "u8", "ivecs": "i32"}
    for ext, a in data.items():
        by_ref, by_us = str(tmp_path / ("ref." + ext)), str(tmp_path / ("us." + ext))
        # the reference's writer -> our reader (io_roundtrip vecs re-writes what it read as .fvecs)
        po.ref_save_vectors(by_ref, a)
        back = str(tmp_path / ("ref.%s.back.fvecs" % ext))
        rc, out, err = run_tool("vecs", by_ref, back)
        assert rc == 0 and out == "vecs dim=%d count=%d" % (a.shape[1], a.shape[0]), err
        ours = np.fromfile(back, np.uint8).reshape(a.shape[0], 4 + 4 * a.shape[1])[:, 4:].copy().view(np.uint32)
        theirs = po.ref_load_vectors(by_ref)                      # the reference reading its own file
        assert np.array_equal(ours, theirs.view(np.uint32))       # bit for bit (NaN, -0.0 included)
        assert np.array_equal(theirs.view(np.uint32), a.astype(np.float32).view(np.uint32))
        # our writer -> the reference's reader; and both writers produce the same bytes
        raw = str(tmp_path / ("raw." + ext))
        a.tofile(raw)
        rc, out, err = run_tool("save", kind[ext], raw, str(a.shape[1]), by_us)
        assert rc == 0 and out == "saved dim=%d count=%d" % (a.shape[1], a.shape[0]), err
        assert open(by_us, "rb").read() == open(by_ref, "rb").read()
        assert po.ref_try_load(by_us)[0] == 0
        assert np.array_equal(po.ref_load_vectors(by_us).view(np.uint32), a.astype(np.float32).view(np.uint32))
        # the file format helper the other tests write their inputs with is the same format
        iof.write_vecs(str(tmp_path / ("np." + ext)), a)
        assert open(str(tmp_path / ("np." + ext)), "rb").read() == open(by_ref, "rb").read()
    # ground truth as recall_file reads it (load_vectors<int>, recall.hpp:37-39) == load_ivecs
    assert np.array_equal(po.ref_load_ivecs(str(tmp_path / "ref.ivecs")), data["ivecs"])
    # trailing bytes shorter than a record: count_vectors truncates (vector_io.hpp:69-76) on both sides
    tail = str(tmp_path / "tail.fvecs")
    open(tail, "wb").write(open(str(tmp_path / "ref.fvecs"), "rb").read() + b"\x18\x00\x00\x00" + b"\x01" * 40)
    rc, out, _ = run_tool("vecs", tail, str(tmp_path / "tail.out.fvecs"))
    assert rc == 0 and out == "vecs dim=24 count=41" and po.ref_load_vectors(tail).shape == (41, 24)
    # errors: both sides exit 1 with the same message lines (vector_io.cpp:20-38)
    bad = str(tmp_path / "bad.bvecs")
    raw = bytearray(open(str(tmp_path / "ref.bvecs"), "rb").read())
    raw[(4 + 128) * 7:(4 + 128) * 7 + 4] = (127).to_bytes(4, "little")
    open(bad, "wb").write(bytes(raw))
    rc_ref, err_ref = po.ref_try_load(bad)
    rc, _, err = run_tool("vecs", bad, str(tmp_path / "x"))
    assert rc_ref == 1 and rc == 1
    assert [l.strip() for l in err_ref.strip().splitlines()] == [l.strip() for l in err.strip().splitlines()] == [
        "Error while reading vectors.", "Vector 7 has 127 dimensions while other vectors have 128 dimensions",
        "All vectors must have the same number of dimensions"]
    unk = str(tmp_path / "ref.fvecs") + ".txt"
    open(unk, "wb").write(b"x")
    rc_ref, err_ref = po.ref_try_load(unk)
    rc, _, err = run_tool("vecs", unk, str(tmp_path / "x"))
    assert rc_ref == 1 and rc == 1
    assert [l.strip() for l in err_ref.strip().splitlines()] == [l.strip() for l in err.strip().splitlines()] == [
        "Could not load vectors from " + unk, "Unknown extension", "Known extensions: .bvecs, .ivecs, .fvecs"]


def test_chunked_reader_against_the_reference_reader(po, tmp_path):
    """The reader thread behind db_add (vector_io.hpp:187-290, vector_io.cpp:60-91, consumed as in db_add.cpp:52-82): the same
    chunk offsets / counts and the same floats as the reference's vectors_reader, for chunk sizes that divide the file, do not
    divide it, exceed it and equal 1, on all three element types."""
    _need_ref_io(po)
    build_tool()
    rng = np.random.default_rng(12)
    files = {"c.fvecs": rng.normal(size=(103, 16)).astype(np.float32), "c.bvecs": rng.integers(0, 256, (64, 32), dtype=np.uint8),
             "c.ivecs": rng.integers(-9, 9, (7, 5)).astype(np.int32)}
    for name, a in files.items():
        path = str(tmp_path / name)
        po.ref_save_vectors(path, a)
        for chunk in (1, 7, 16, 64, 1000):
            if chunk == 1 and a.shape[0] > 64:
                continue
            want_data, want_chunks = po.ref_read_chunked(path, chunk, a.shape[0], a.shape[1])
            out = str(tmp_path / (name + ".chunks"))
            rc, txt, err = run_tool("chunks", path, str(chunk), out)
            assert rc == 0, err
            lines = txt.splitlines()
            assert lines[-1] == "total dim=%d count=%d" % (a.shape[1], a.shape[0])
            assert [tuple(int(x) for x in l.split()) for l in lines[:-1]] == want_chunks
            got = np.fromfile(out, np.float32).reshape(a.shape)
            assert np.array_equal(got.view(np.uint32), want_data.view(np.uint32))
            assert np.array_equal(want_data, a.astype(np.float32))


def test_recall_rule_against_the_reference_recall_file(po, tmp_path):
    """A9: the recall column of process_queries<> = recall_file::check_labels(q, keys, keys + R, t = 1) (query_common.hpp:360-361,
    recall.hpp:21-61), pinned to the reference's recall.hpp compiled into oracle/_ref — on key arrays as the heaps hold them:
    unsorted, with duplicate keys (the padding-lane replays), with the surviving sentinel key 0 (which counts as label 0),
    ids above 2^31 (int ground truth against unsigned keys) and t up to the ground truth's width."""
    _need_ref_io(po)
    build_tool()
    rng = np.random.default_rng(13)
    nq, R, T = 64, 100, 5
    gt = rng.integers(0, 5000, (nq, T)).astype(np.int32)
    keys = rng.integers(0, 5000, (nq, R)).astype(np.uint32)
    for q in range(nq):
        kind = q % 8
        if kind == 0:                                            # the true neighbour is there, once
            keys[q, rng.integers(0, R)] = gt[q, 0]
        elif kind == 1:                                          # ... several times (padding-lane duplicates)
            keys[q, rng.choice(R, 5, replace=False)] = gt[q, 0]
        elif kind == 2:                                          # absent
            keys[q][keys[q] == np.uint32(gt[q, 0])] = 7777
        elif kind == 3:                                          # the true neighbour IS label 0 and only the sentinel's key 0 is there
            gt[q, 0] = 0
            keys[q] = rng.integers(1, 5000, R)
            keys[q, 0] = 0
        elif kind == 4:                                          # all T ground-truth ids present (t > 1 succeeds)
            keys[q, rng.choice(R, T, replace=False)] = gt[q].astype(np.uint32)
        elif kind == 5:                                          # id >= 2^31: int -1 in the file, 0xffffffff among the keys
            gt[q, 0] = -1
            keys[q, 17] = 0xFFFFFFFF
        elif kind == 6:                                          # same id, key missing
            gt[q, 0] = -2
        # kind 7: whatever the draw says
    gt_path, keys_path = str(tmp_path / "gt.ivecs"), str(tmp_path / "keys.bin")
    po.ref_save_vectors(gt_path, gt)
    keys.tofile(keys_path)
    seen = set()
    for t in (1, 2, T):
        want = po.ref_check_labels(gt_path, keys, t)
        rc, out, err = run_tool("recall", gt_path, keys_path, str(R), str(t))
        assert rc == 0, err
        assert [int(c) for c in out.strip()] == want.tolist(), t
        seen |= set(want.tolist())
        if t == 1:
            assert want[0] == 1 and want[1] == 1 and want[2] == 0 and want[3] == 1 and want[5] == 1 and want[6] == 0
        if t == T:
            assert want[4] == 1
    assert seen == {0, 1}


@pytest.mark.gpu
@pytest.mark.parametrize("kind,opq,qext", [("flat", False, ".fvecs"), ("index", True, ".bvecs")])
def test_reference_command_line_on_reference_files(po, tmp_path, kind, opq, qext):
    """db_query_4's command line on a database archive + vecs files: the loader hands the scanner exactly the
    codes / labels that were written, OPQ rotates the residuals as x @ R^T, the heaps equal the oracle's, and the
    recall column is computed from the .ivecs ground truth."""
    build_driver()
    rng = np.random.default_rng(5)
    M, dim, N, nq, R, K, ma = 16, 128, 40000, 10, 100, 32, 5
    centres = 3 * rng.normal(size=(500, dim)).astype(np.float32)
    base = (centres[rng.integers(0, 500, N)] + rng.normal(size=(N, dim))).astype(np.float32)
    queries = (centres[rng.integers(0, 500, nq)] + rng.normal(size=(nq, dim))).astype(np.float32)
    if qext == ".bvecs":                                             # byte vectors (SIFT1B style): small non-negative ints
        base = np.clip(np.round(base * 8 + 128), 0, 255).astype(np.float32)
        queries = np.clip(np.round(queries * 8 + 128), 0, 255).astype(np.float32)
    rot = np.linalg.qr(rng.normal(size=(dim, dim)))[0].astype(np.float32) if opq else None
    gt = np.stack([np.argsort(((base - q) ** 2).sum(1))[:5] for q in queries]).astype(np.int32)
    db_path, q_path, gt_path = str(tmp_path / "test.db"), str(tmp_path / ("q" + qext)), str(tmp_path / "gt.ivecs")
    iof.write_vecs(q_path, queries.astype(np.uint8) if qext == ".bvecs" else queries)
    off = 1000 if kind == "index" else 0                             # labels of the indexed database = base id + 1000
    iof.write_vecs(gt_path, gt + off)
    ds = dim // M
    if kind == "flat":
        src = base if rot is None else base @ rot.T
        cb = np.stack([src[rng.integers(0, N, 16), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
        codes = iof.pq_encode(cb, base, rot)
        iof.write_flat_db(db_path, cb, codes, rot)
        want_parts, want_labels, ma = [codes], None, 1
    else:
        coarse = base[rng.integers(0, N, K)].copy()
        assign = ((base ** 2).sum(1)[:, None] - 2 * base @ coarse.T + (coarse ** 2).sum(1)[None]).argmin(1)
        res = base - coarse[assign]
        src = res if rot is None else res @ rot.T
        cb = np.stack([src[rng.integers(0, N, 16), m * ds:(m + 1) * ds] for m in range(M)]).astype(np.float32)
        codes = iof.pq_encode(cb, res, rot)
        want_parts = [codes[assign == k] for k in range(K)]
        want_labels = [np.nonzero(assign == k)[0].astype(np.uint32) + off for k in range(K)]
        iof.write_index_db(db_path, cb, coarse, want_parts, want_labels, rot)
    dump = str(tmp_path / "dump.bin")
    keep_pct = 3.0
    out = subprocess.check_output([DRIVER, "-r%d" % R, "-m", str(ma), "-k%g" % keep_pct, "-b4", db_path, q_path, gt_path, dump],
                                  stderr=subprocess.DEVNULL).decode().strip().split("\n")
    assert out[-2] == "r,recall,ma,adc_type,keep,index_us,rotate_us,table_us,scan_us"
    cols = out[-1].split(",")
    assert int(cols[0]) == R and int(cols[2]) == ma and cols[3] == "qadc"
    parts, labels, dq = _parse_dump(dump, M, ma, nq)
    assert len(parts) == len(want_parts) and all(np.array_equal(a, b) for a, b in zip(parts, want_parts))
    if want_labels is not None:
        assert all(np.array_equal(a, b) for a, b in zip(labels, want_labels))
    keep = float(np.float32(keep_pct) * np.float32(0.01))
    hits = 0
    for q, (assign_q, tables, keys, vals) in enumerate(dq):
        want = po.query_scan(M, parts, labels, keep, assign_q, np.ascontiguousarray(tables.reshape(ma, M * 16)), R)
        assert np.array_equal(keys, want["keys"]) and np.array_equal(vals, want["values"]), q
        # the tables the driver built = ||rotate(residual) - centroid||^2 with rotate(x) = x @ R^T
        x = queries[q] - (coarse[assign_q[0]] if kind == "index" else 0)
        x = x if rot is None else x @ rot.T
        t0 = ((x.reshape(M, 1, ds) - cb) ** 2).sum(-1).reshape(-1)
        assert np.allclose(tables[:M * 16], t0, rtol=2e-4, atol=1e-3)
        hits += int(gt[q, 0] + off in set(keys.tolist()))
    assert abs(float(cols[1]) - hits / nq) < 1e-6 and hits >= nq // 2    # recall column = ground-truth rank 0 among the keys
