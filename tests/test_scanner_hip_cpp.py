"""The C++14 host mirror (quick-adc_amd/host/scanner_hip.hpp) of the reference's ScannerType:
builds with plain g++ -std=c++14 against the C-ABI (CPU), and on the GPU its query_scan fills the
caller's heap exactly as the oracle's restatement of scanner_4::query_scan does."""
import os
import subprocess

import numpy as np
import pytest

from helpers import path_independent, splitmix64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "scanner_hip_demo")


def _compile(src, exe, link=True):
    """g++ -std=c++14 against the C-ABI library (link=False: a header-only driver); written under a per-process name and
    renamed into place, so that parallel test workers never execute (or overwrite) a half-linked binary."""
    libdir = os.path.join(ROOT, "quick-adc_amd")
    if link and not os.path.exists(os.path.join(libdir, "libqadc_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    tmp = "%s.%d.tmp" % (exe, os.getpid())
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-pthread", src, "-o", tmp] +
                          (["-L" + libdir, "-lqadc_hip", "-Wl,-rpath," + libdir] if link else []))
    os.replace(tmp, exe)



def build_demo():
    _compile(os.path.join(ROOT, "tests", "cpp", "scanner_hip_demo.cpp"), EXE)


def test_scanner_hip_builds_as_cxx14():
    build_demo()
    assert os.path.exists(EXE)


REFHEAP_EXE = os.path.join(ROOT, "oracle", "_ref", "scanner_hip_demo_refheap")


def test_scanner_hip_builds_with_the_reference_kv_binheap():
    """scanner_hip<mini_db, kv_binheap<unsigned, int8_t>> compiled (-Wall -Werror, C++14) against the reference's own
    binheap.hpp where it lies — the BhType nns_engine<scanner_4> uses (db_query_4.cpp:244).  Only possible where
    /root/reference exists (oracle/Makefile `ref`); the binary travels to the GPU box in oracle/_ref/."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("/root/reference absent (GPU box): the prebuilt binary is used")
    import __graft_entry__
    __graft_entry__.build()
    assert os.path.exists(REFHEAP_EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("flavour", ["kv_heap", "reference_kv_binheap"])
@pytest.mark.parametrize("M,sizes,labeled,ma", [(16, [60000], 0, 1), (16, [20000, 37, 9000, 15001], 1, 3),
                                                 (32, [30000, 12345], 1, 2)])
def test_scanner_hip_fills_heap_like_oracle(po, M, sizes, labeled, ma, flavour):
    """Both heap types the scanner can be instantiated with — the library's kv_heap and the reference's own
    kv_binheap (binheap.hpp compiled from /root/reference) — end in the oracle's heap arrays and sort_keys order."""
    build_demo()
    exe = EXE
    if flavour == "reference_kv_binheap":
        exe = REFHEAP_EXE
        if not os.path.exists(exe):
            pytest.skip("oracle/_ref/scanner_hip_demo_refheap was not built (needs /root/reference at build time)")
    keep, R, nq, seed = 0.02, 100, 3, 4242
    out = subprocess.check_output([exe, str(M), str(len(sizes))] + [str(s) for s in sizes] +
                                  [str(labeled), str(keep), str(R), str(nq), str(ma), str(seed)]).decode().split("\n")
    cs = M // 2
    parts = [po.fill_codes(0, (s * cs + 7) // 8, seed + p)[:s * cs].reshape(s, cs).copy() for p, s in enumerate(sizes)]
    labels = [(1000000 * (p + 1) + 7 * np.arange(s)).astype(np.uint32) for p, s in enumerate(sizes)] if labeled else None
    line = 0
    for q in range(nq):
        assign = [(q + i * 3) % len(sizes) for i in range(ma)]
        i = np.arange(ma * M * 16, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = splitmix64(np.uint64(seed * 31 + q * 1000003) + i)
        tables = ((h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(4.0))
        want = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(tables.reshape(ma, M * 16)), R)
        assert want["rc"] == 0
        hdr = out[line].split()
        assert hdr[0] == "q" and int(hdr[1]) == q
        size = int(hdr[3])
        rows = np.array([[int(x) for x in l.split()] for l in out[line + 1:line + 1 + size]], np.int64).reshape(size, 2)
        srt = out[line + 1 + size].split()
        assert srt[0] == "sorted"
        line += 2 + size
        assert size == len(want["keys"])
        assert np.array_equal(rows[:, 0].astype(np.uint32), want["keys"]) and np.array_equal(rows[:, 1].astype(np.int8), want["values"])
        assert np.array_equal(np.array(srt[1:], np.uint64).astype(np.uint32), po.sort_keys_i8(want["keys"], want["values"]))


DRIVER = os.path.join(ROOT, "tests", "cpp", "db_query_4_hip")


def build_driver():
    _compile(os.path.join(ROOT, "tests", "cpp", "db_query_4_hip.cpp"), DRIVER)


def test_db_query_4_driver_builds_as_cxx14():
    build_driver()
    assert os.path.exists(DRIVER)


def _parse_dump(path, M, ma, nq):
    raw = np.fromfile(path, np.uint8)
    o = 0

    def take(n, dt):
        nonlocal o
        a = raw[o:o + n * np.dtype(dt).itemsize].view(dt)
        o += n * np.dtype(dt).itemsize
        return a

    nparts = int(take(1, np.int32)[0])
    parts, labels = [], []
    for _ in range(nparts):
        n = int(take(1, np.uint32)[0])
        labeled = int(take(1, np.int32)[0])
        parts.append(take(n * (M // 2), np.uint8).reshape(n, M // 2).copy())
        labels.append(take(n, np.uint32).copy() if labeled else None)
    queries = []
    for _ in range(nq):
        assign = take(ma, np.int32).copy()
        tables = take(ma * M * 16, np.float32).copy()
        size = int(take(1, np.int32)[0])
        keys = take(size, np.uint32).copy()
        vals = take(size, np.int8).copy()
        queries.append((assign, tables, keys, vals))
    assert o == len(raw)
    return parts, (labels if labels[0] is not None else None), queries


@pytest.mark.gpu
@pytest.mark.parametrize("mode,M,N,ma,K,keep_pct", [("flat", 16, 60000, 1, 0, 2.0), ("ivf", 16, 60000, 6, 48, 2.0),
                                                      ("ivf", 32, 30000, 4, 24, 8.0)])
def test_db_query_4_driver_end_to_end(po, tmp_path, mode, M, N, ma, K, keep_pct):
    """Real PQ encodings of clustered vectors (tie-heavy int8 sums, labelled IVF partitions of ragged size):
    the heap every query leaves behind equals the oracle's scanner_4::query_scan on the same inputs; the
    CSV line has the reference's columns and a sane recall."""
    build_driver()
    nq, R = 12, 100
    dump = str(tmp_path / "dump.bin")
    out = subprocess.check_output([DRIVER, mode, str(M), str(N), str(nq), str(R), str(keep_pct), str(ma), str(K), "7",
                                   dump]).decode().strip().split("\n")
    assert out[-2] == "r,recall,ma,adc_type,keep,index_us,rotate_us,table_us,scan_us"
    cols = out[-1].split(",")
    assert int(cols[0]) == R and cols[3] == "qadc" and int(cols[2]) == ma
    assert 0.5 <= float(cols[1]) <= 1.0          # true nearest neighbour found for most queries
    parts, labels, queries = _parse_dump(dump, M, ma, nq)
    keep = float(np.float32(keep_pct) * np.float32(0.01))
    for q, (assign, tables, keys, vals) in enumerate(queries):
        want = po.query_scan(M, parts, labels, keep, assign, np.ascontiguousarray(tables.reshape(ma, M * 16)), R)
        assert want["rc"] == 0
        assert np.array_equal(keys, want["keys"]) and np.array_equal(vals, want["values"]), (mode, M, q)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,M,N,ma,K", [("flat", 16, 50000, 1, 0), ("ivf", 16, 50000, 5, 40)])
def test_batched_engine_equals_per_query_engine(po, tmp_path, mode, M, N, ma, K):
    """nns_engine_batch-style driving (-b 5: one GPU call per 5 queries, per-query replay of the cached
    streams, last batch ragged) leaves exactly the heaps of the per-query engine and of the oracle."""
    build_driver()
    nq, R, keep_pct = 13, 100, 3.0
    dumps = []
    for batch in (1, 5):
        dump = str(tmp_path / ("dump%d.bin" % batch))
        out = subprocess.check_output([DRIVER, mode, str(M), str(N), str(nq), str(R), str(keep_pct), str(ma), str(K), "11",
                                       dump, str(batch)]).decode().strip().split("\n")
        assert out[-2].startswith("r,recall,ma,adc_type,keep")
        dumps.append(_parse_dump(dump, M, ma, nq))
    (parts, labels, q1), (_, _, q5) = dumps
    keep = float(np.float32(keep_pct) * np.float32(0.01))
    for q in range(nq):
        assert np.array_equal(q1[q][0], q5[q][0])                                               # same assignments
        if ma > 1:
            # both engines build BLAS-expansion tables for ma > 1 (query_common.hpp:209-213, 295-297): same inputs,
            # so the batched scan must leave the very same heaps
            assert np.array_equal(q1[q][1], q5[q][1])
            assert np.array_equal(q1[q][2], q5[q][2]) and np.array_equal(q1[q][3], q5[q][3]), q
        else:
            # ma == 1: nns_engine uses the direct form, nns_engine_batch the expansion — the reference's "two numerically
            # different table paths"; close, not equal
            assert np.allclose(q1[q][1], q5[q][1], rtol=1e-4, atol=1e-3)
        for dump_q in (q1[q], q5[q]):                                                           # each equals the oracle on ITS tables
            want = po.query_scan(M, parts, labels, keep, dump_q[0], np.ascontiguousarray(dump_q[1].reshape(ma, M * 16)), R)
            assert np.array_equal(dump_q[2], want["keys"]) and np.array_equal(dump_q[3], want["values"]), q


DIST_DEMO = os.path.join(ROOT, "tests", "cpp", "dist_demo")


def build_dist_demo():
    _compile(os.path.join(ROOT, "tests", "cpp", "dist_demo.cpp"), DIST_DEMO)


def test_dist_demo_builds_as_cxx14():
    build_dist_demo()
    assert os.path.exists(DIST_DEMO)


@pytest.mark.gpu
def test_dist_demo_world_1_equals_plain_scan(po, tmp_path):
    """The C++14 multi-process driver of the native merge, with the one rank a 1-GPU box allows: its heap checksum is
    the checksum of the oracle's heaps for the same list and tables (on an 8-GPU node every rank prints this value)."""
    build_dist_demo()
    n, nq, R, M = 300000, 6, 100, 16
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.check_output([DIST_DEMO, str(tmp_path / "id.bin"), str(n), str(nq), str(R)], env=env).decode()
    assert "extra payload ok" in out
    got = int(out.split("heap checksum ")[1].split(",")[0], 16)
    codes = po.fill_codes(0, (n * 8 + 7) // 8, 0x5EED0001)[:n * 8].reshape(n, 8)
    i = np.arange(nq * M * 16, dtype=np.uint64)
    with np.errstate(over="ignore"):
        tables = ((splitmix64(np.uint64(977) + i) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(4.0))
    tables = tables.reshape(nq, M * 16)
    acc = np.uint64(0)
    for q in range(nq):
        want = po.query_scan(M, [codes], None, 0.01, [0], np.ascontiguousarray(tables[q:q + 1].copy()), R)
        assert want["rc"] == 0
        for k, v in zip(want["keys"], want["values"]):
            with np.errstate(over="ignore"):
                acc = splitmix64(np.uint64(acc) ^ ((np.uint64(k) << np.uint64(8)) | np.uint64(np.uint8(v))))
    assert got == int(acc)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_dist_demo_several_ranks_on_one_gpu_equal_the_unsharded_scan(po, tmp_path, world):
    """The same C++14 driver with world > 1: `world` processes on GPU 0, shards of one list, the library's shared-memory
    transport in place of RCCL, the unmodified qadc_dist_collect (32-query batch: host-share replay + second gather).
    Every rank prints the checksum of the unsharded oracle's heaps."""
    import uuid
    build_dist_demo()
    n, nq, R, M = 500003, 32, 100, 16
    name = "shm:/qadc_demo_%s" % uuid.uuid4().hex[:10]
    procs = [subprocess.Popen([DIST_DEMO, name, str(n), str(nq), str(R)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0")) for r in range(world)]
    outs = []
    try:
        for p in procs:
            so, se = p.communicate(timeout=300)
            assert p.returncode == 0, se.decode()[-800:]
            outs.append(so.decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = po.fill_codes(0, (n * 8 + 7) // 8, 0x5EED0001)[:n * 8].reshape(n, 8)
    i = np.arange(nq * M * 16, dtype=np.uint64)
    with np.errstate(over="ignore"):
        tables = ((splitmix64(np.uint64(977) + i) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0) * np.float32(4.0))
    tables = tables.reshape(nq, M * 16)
    acc = np.uint64(0)
    for q in range(nq):
        want = po.query_scan(M, [codes], None, 0.01, [0], np.ascontiguousarray(tables[q:q + 1].copy()), R)
        assert want["rc"] == 0
        for k, v in zip(want["keys"], want["values"]):
            with np.errstate(over="ignore"):
                acc = splitmix64(np.uint64(acc) ^ ((np.uint64(k) << np.uint64(8)) | np.uint64(np.uint8(v))))
    for out in outs:
        assert "extra payload ok" in out
        assert int(out.split("heap checksum ")[1].split(",")[0], 16) == int(acc), out


DB_BUILD = os.path.join(ROOT, "tests", "cpp", "db_build_demo")


def test_db_build_demo_builds_as_cxx14():
    _compile(os.path.join(ROOT, "tests", "cpp", "db_build_demo.cpp"), DB_BUILD)
    assert os.path.exists(DB_BUILD)


@pytest.mark.gpu
@path_independent
@pytest.mark.parametrize("M,dim,K,n,opq", [(16, 32, 37, 5000, 0), (32, 64, 300, 40000, 1), (16, 128, 1000, 70000, 0)])
def test_gpu_database_build_equals_the_host_build(M, dim, K, n, opq, tmp_path):
    """N4: host/db_build.hpp (index_db::add_vectors / flat_db::add_vectors compute and the k-means fast iterations on
    the GPU, from C++14) against the sequential host loops: same partitions, codes, labels, centroids, bit for bit
    (chunked adds with offsets, an exact centroid tie, OPQ rotation, batches on both sides of the 256-vector switch of
    the coarse kernels)."""
    _compile(os.path.join(ROOT, "tests", "cpp", "db_build_demo.cpp"), DB_BUILD)
    out = subprocess.check_output([DB_BUILD, str(M), str(dim), str(K), str(n), str(opq), str(tmp_path / "base")]).decode()
    # (the last figure: db_add's streamed form — reader thread + two-chunk queue from an .fvecs file, db_add.cpp:52-82)
    assert "ivf_partitions_differing 0 flat_differs 0 kmeans_differs 0 streamed_add_differs 0" in out, out


DBQ_SIMPLE = os.path.join(ROOT, "tests", "cpp", "db_query_simple")


@pytest.mark.parametrize("M,bits,batch", [(8, 8, 1), (16, 8, 1), (4, 8, 4), (16, 4, 1), (32, 4, 3)])
def test_db_query_cpu_front_end_matches_the_oracle(po, tmp_path, M, bits, batch):
    """BASELINE configs[0]: the reference's plain `db_query` path (scanner_simple, db_query.cpp:17-46; scan_standard /
    scan_4, query_common.hpp:59-118) restated in host/scanner_simple.hpp and driven by the engines of
    host/query_driver.hpp — CPU only.  The driver dumps its codes, the float tables each query was scanned with and
    the resulting float heaps; the oracle's orc_scan_standard_u8 (whole-byte codes) or its float ADC values replayed
    through the float heap (4-bit codes) must give the same arrays."""
    _compile(os.path.join(ROOT, "tests", "cpp", "db_query_simple.cpp"), DBQ_SIMPLE)
    dim, n, nq, R = 64, 6001, 7, 40
    dump = str(tmp_path / "dump.bin")
    out = subprocess.check_output([DBQ_SIMPLE, "-r", str(R), "-b", str(batch), str(M), str(bits), str(dim), str(n), str(nq), "11", dump]).decode()
    lines = out.strip().split("\n")
    assert lines[0] == "r,recall,ma,adc_type,index_us,rotate_us,table_us,scan_us"      # db_query.cpp:112-115
    assert lines[1].split(",")[0] == str(R) and lines[1].split(",")[3] == "adc" and 0.0 <= float(lines[1].split(",")[1]) <= 1.0
    raw = np.fromfile(dump, np.uint8)
    hdr = raw[:24].view(np.int32)
    assert list(hdr) == [M, bits, dim, n, nq, R]
    cs = M * bits // 8
    o = 24
    codes = raw[o:o + n * cs].reshape(n, cs).copy()
    o += n * cs
    td = M * (1 << bits)
    tables = raw[o:o + nq * td * 4].view(np.float32).reshape(nq, td).copy()
    o += nq * td * 4
    for q in range(nq):
        size = int(raw[o:o + 4].view(np.int32)[0])
        keys = raw[o + 4:o + 4 + 4 * size].view(np.uint32)
        vals = raw[o + 4 + 4 * size:o + 4 + 8 * size].view(np.float32)
        o += 4 + 8 * size
        assert size == R
        if bits == 8:
            wk, wv = po.scan_standard_u8(M, [codes], None, tables[q].reshape(1, M, 256), R)
        else:
            cand = po.candidates_f32(M, codes, tables[q])          # scan_4's sums (query_common.hpp:72-80)
            fmax = np.float32(np.finfo(np.float32).max)
            sent = np.array([fmax - np.float32(t) for t in range(R)], np.float32)
            wk, wv = po.heap_replay_f32(np.concatenate([np.zeros(R, np.uint32), np.arange(n, dtype=np.uint32)]),
                                        np.concatenate([sent, cand]), R)
        assert np.array_equal(keys, wk) and np.array_equal(vals, wv), q


ENCODE_DEMO = os.path.join(ROOT, "tests", "cpp", "encode_demo")


def _run_encode_demo(tmp_path, cb, v, rot=None, bits=4, form=1):
    M, nc, ds = cb.shape
    n, dim = v.shape
    fin, fout = str(tmp_path / "enc_in.bin"), str(tmp_path / "enc_out.bin")
    with open(fin, "wb") as fh:
        fh.write(np.array([M, bits, dim, n, form, 0 if rot is None else 1], np.int32).tobytes() + cb.tobytes() +
                 (b"" if rot is None else rot.tobytes()) + v.tobytes())
    subprocess.check_call([ENCODE_DEMO, fin, fout])
    raw = np.fromfile(fout, np.uint8)
    cs = M * bits // 8
    return raw[:n * cs].reshape(n, cs), raw[n * cs:].view(np.float32).reshape(n, M * nc)


def test_host_encoder_and_expansion_tables_match_the_oracle(po, tmp_path):
    """N4 / A10 on the CPU: pq4::encode and pq4::tables_blas (host/query_driver.hpp), the host twins of pq_encode_kernel and
    build_tables_kernel, against the golden codes made from the reference's own text (tests/golden/ref_encode_cases.npz) and
    the oracle's orc_tables_expansion; encode_form 0 = the oracle's direct form; pq_bytes::encode (BASELINE configs[0]'s
    quantizers, 256 centroids) = find_k_neighbors' selection on the oracle's expansion distances."""
    import golden_cases
    _compile(os.path.join(ROOT, "tests", "cpp", "encode_demo.cpp"), ENCODE_DEMO, link=False)
    for c in golden_cases.encode_cases():
        codes, tables = _run_encode_demo(tmp_path, c["codebooks"], c["vectors"], c["rotation"])
        assert np.array_equal(codes, c["codes"]), c["cid"]
        if c["rotation"] is None:
            assert np.array_equal(tables, po.tables_expansion(c["codebooks"], c["vectors"])), c["cid"]
        codes0, _ = _run_encode_demo(tmp_path, c["codebooks"], c["vectors"], c["rotation"], form=0)
        assert np.array_equal(codes0, po.pq_encode(c["codebooks"], c["vectors"], c["rotation"], form=0)), c["cid"]
    rng = np.random.default_rng(8)
    for M, dim, kind in ((8, 64, "n"), (4, 32, "grid"), (16, 64, "offset")):
        ds = dim // M
        if kind == "grid":
            cb, v = rng.integers(0, 3, (M, 256, ds)).astype(np.float32), rng.integers(0, 3, (300, dim)).astype(np.float32)
        elif kind == "offset":
            cb = (50 + 0.01 * rng.normal(size=(M, 256, ds))).astype(np.float32)
            v = (50 + 0.01 * rng.normal(size=(300, dim))).astype(np.float32)
        else:
            cb, v = rng.normal(size=(M, 256, ds)).astype(np.float32), rng.normal(size=(300, dim)).astype(np.float32)
        codes, _ = _run_encode_demo(tmp_path, cb, v, bits=8)
        want = np.stack([po.select_k_neighbors(po.cross_dists(cb[m], v[:, m * ds:(m + 1) * ds]), 1)[0][:, 0] for m in range(M)], 1)
        assert np.array_equal(codes, want.astype(np.uint8)), (M, kind)


NEAREST = os.path.join(ROOT, "tests", "cpp", "nearest_demo")


@pytest.mark.parametrize("K,dim,ma,levels", [(700, 16, 9, 3), (4100, 48, 33, 2), (300, 16, 2, 2), (900, 32, 64, 3), (40, 16, 40, 2), (513, 16, 256, 4)])
def test_host_coarse_selection_with_exact_ties_is_find_k_neighbors(po, tmp_path, K, dim, ma, levels):
    """The host twin of the device's coarse selection (ivf_database::nearest, host/query_driver.hpp) on integer-valued centroids
    and queries — distances tie exactly, inside the ma nearest and across the ma-th — against the REFERENCE's own heaps
    (find_k_neighbors' selection half compiled from its text, oracle/_ref) and against the oracle's restatement, fed with the same
    sequential squared distances.  CPU only."""
    _compile(os.path.join(ROOT, "tests", "cpp", "nearest_demo.cpp"), NEAREST, link=False)
    rng = np.random.default_rng(K + ma)
    nq = 25
    coarse = rng.integers(0, levels, (K, dim)).astype(np.float32)
    queries = rng.integers(0, levels, (nq, dim)).astype(np.float32)
    f = str(tmp_path / "in.bin")
    with open(f, "wb") as fh:
        fh.write(np.array([K, dim, ma, nq], np.int32).tobytes() + coarse.tobytes() + queries.tobytes())
    got = np.array([[int(x) for x in l.split()] for l in subprocess.check_output([NEAREST, f]).decode().strip().split("\n")], np.int32)
    d = po.cross_dists(coarse, queries)                          # the reference's coarse distances (expansion form: dist2 of the twin)
    want, _ = po.select_k_neighbors(d, ma)
    assert np.array_equal(got, want)
    if po.have_ref_float():
        ref, _ = po.reff_select_k_neighbors(d, ma)
        assert np.array_equal(got, ref)
    rule = np.stack([np.lexsort((np.arange(K), row))[:ma] for row in d])
    assert ma == K or not np.array_equal(got, rule)              # (the plain (distance, index) rule is not what the heaps leave)
