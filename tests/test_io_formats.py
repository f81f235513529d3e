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
