"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle (oracle/libqadc_oracle.so, the C
restatement) and, when present, the reference build (oracle/_ref/libqadc_ref.so, the
reference's own kernels compiled from /root/reference by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (quick-adc_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libqadc_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libqadc_ref.so")
_REF_IO_SO = os.path.join(_HERE, "_ref", "libqadc_ref_io.so")
_REF_FLOAT_SO = os.path.join(_HERE, "_ref", "libqadc_ref_float.so")

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


def build(force=False):
    """Compile the C restatement (and the reference build when /root/reference exists)."""
    if force or not os.path.exists(_ORACLE_SO) or \
            os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "qadc_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libqadc_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF_SO) or not os.path.exists(_REF_IO_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF_FLOAT_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref_float"], stdout=subprocess.DEVNULL)


def _p(a, t):
    return a.ctypes.data_as(t)


def _ptr_array(arrs, t):
    if arrs is None:
        return None, None
    keep = [np.ascontiguousarray(a) for a in arrs]
    arr = (t * len(keep))(*[_p(a, t) for a in keep])
    return arr, keep


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_ORACLE_SO)
        _lib.orc_interleaved_size.restype = C.c_long
        _lib.orc_interleaved_size.argtypes = [C.c_uint32, C.c_int]
        _lib.orc_start_size.restype = C.c_uint32
        _lib.orc_start_size.argtypes = [C.c_uint32, C.c_float]
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_SO)
        _ref.qadc_ref_interleaved_size.restype = C.c_long
        _ref.qadc_ref_interleaved_size.argtypes = [C.c_uint, C.c_int]
    return _ref


# ----------------------------------------------------------------------------- C restatement
def heap_replay_i8(keys, vals, R):
    keys = np.ascontiguousarray(keys, np.uint32)
    vals = np.ascontiguousarray(vals, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    lib().orc_heap_replay_i8(C.c_long(len(keys)), _p(keys, u32p), _p(vals, i8p), R,
                             _p(ok, u32p), _p(ov, i8p), C.byref(osz))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def heap_replay_f32(keys, vals, R):
    keys = np.ascontiguousarray(keys, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    lib().orc_heap_replay_f32(C.c_long(len(keys)), _p(keys, u32p), _p(vals, f32p), R,
                              _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def sort_keys_i8(heap_keys, heap_vals):
    """kv_binheap::sort_keys (binheap.hpp:129-137) of a heap in the given array state."""
    k = np.ascontiguousarray(heap_keys, np.uint32)
    v = np.ascontiguousarray(heap_vals, np.int8)
    out = np.zeros(len(k), np.uint32)
    lib().orc_sort_keys_i8(len(k), _p(k, u32p), _p(v, i8p), _p(out, u32p))
    return out


def pack4(assign, M):
    assign = np.ascontiguousarray(assign, np.int32)
    n = assign.shape[0]
    codes = np.zeros((n, M // 2), np.uint8)
    lib().orc_pack4(_p(assign, i32p), C.c_long(n), M, _p(codes, u8p))
    return codes


def interleave(codes):
    codes = np.ascontiguousarray(codes, np.uint8)
    n, cs = codes.shape
    out = np.zeros(lib().orc_interleaved_size(n, cs), np.uint8)
    lib().orc_interleave(_p(out, u8p), _p(codes, u8p), n, cs)
    return out


def deinterleave(inter, n, cs):
    inter = np.ascontiguousarray(inter, np.uint8)
    out = np.zeros((n, cs), np.uint8)
    lib().orc_deinterleave(_p(out, u8p), _p(inter, u8p), n, cs)
    return out


def scan_i8(M, parts, labels, qtables, R, sentinel=True, layout="rowmajor"):
    """parts: list of arrays (row-major [n][cs] or interleaved bytes); labels: list or None;
    qtables: int8 [nparts][M][16].  Returns heap (keys, values) arrays of length size."""
    if layout != "rowmajor":
        raise ValueError("use scan_i8_interleaved")
    parts = [np.ascontiguousarray(p, np.uint8) for p in parts]
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    qt = np.ascontiguousarray(qtables, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    rc = lib().orc_scan_i8(M, len(parts), pa, la, _p(sizes, u32p), _p(qt, i8p), R, int(sentinel), 1,
                           _p(ok, u32p), _p(ov, i8p), C.byref(osz))
    assert rc == 0
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def scan_i8_interleaved(M, inter_parts, sizes, labels, qtables, R, sentinel=True):
    sizes = np.ascontiguousarray(sizes, np.uint32)
    pa, keep1 = _ptr_array(inter_parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    qt = np.ascontiguousarray(qtables, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    rc = lib().orc_scan_i8(M, len(inter_parts), pa, la, _p(sizes, u32p), _p(qt, i8p), R, int(sentinel), 0,
                           _p(ok, u32p), _p(ov, i8p), C.byref(osz))
    assert rc == 0
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def shard_stream(M, codes, labels, global_n, first_pos, qt, R, cap=1 << 20):
    """Ordered push stream of one shard (see orc_scan_i8_shard_stream)."""
    codes = np.ascontiguousarray(codes, np.uint8)
    lab = None if labels is None else np.ascontiguousarray(labels, np.uint32)
    qt = np.ascontiguousarray(qt, np.int8)
    ok, ov = np.zeros(cap, np.uint32), np.zeros(cap, np.int8)
    lib().orc_scan_i8_shard_stream.restype = C.c_long
    cnt = lib().orc_scan_i8_shard_stream(M, _p(codes, u8p), None if lab is None else _p(lab, u32p), codes.shape[0],
                                         global_n, first_pos, _p(qt, i8p), R, _p(ok, u32p), _p(ov, i8p), C.c_long(cap))
    assert cnt <= cap
    return ok[:cnt].copy(), ov[:cnt].copy()


def shards_stream(M, pieces, labels, global_n, first_pos, qt, R, cap=1 << 20):
    """Ordered push stream of one rank that holds one range of every probed partition (see
    orc_scan_i8_shards_stream).  pieces[a] = row-major codes of the local range; labels[a] or None."""
    ma = len(pieces)
    pieces = [np.ascontiguousarray(p, np.uint8) for p in pieces]
    pa, keep1 = _ptr_array(pieces, u8p)
    la = None
    if labels is not None:
        la, keep2 = _ptr_array([np.ascontiguousarray(l, np.uint32) for l in labels], u32p)
    n = np.array([p.shape[0] for p in pieces], np.uint32)
    gn = np.ascontiguousarray(global_n, np.uint32)
    fp = np.ascontiguousarray(first_pos, np.uint32)
    qt = np.ascontiguousarray(qt, np.int8)
    ok, ov, os_ = np.zeros(cap, np.uint32), np.zeros(cap, np.int8), np.zeros(cap, np.uint16)
    lib().orc_scan_i8_shards_stream.restype = C.c_long
    cnt = lib().orc_scan_i8_shards_stream(M, ma, pa, la, _p(n, u32p), _p(gn, u32p), _p(fp, u32p), _p(qt, i8p), R,
                                          _p(ok, u32p), _p(ov, i8p), os_.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_long(cap))
    assert cnt <= cap
    return ok[:cnt].copy(), ov[:cnt].copy(), os_[:cnt].copy()


def candidates_i8(M, codes, qt):
    codes = np.ascontiguousarray(codes, np.uint8)
    qt = np.ascontiguousarray(qt, np.int8)
    out = np.zeros(codes.shape[0], np.int8)
    lib().orc_candidates_i8(M, _p(codes, u8p), C.c_long(codes.shape[0]), _p(qt, i8p), _p(out, i8p))
    return out


def candidates_f32(M, codes, dists, sum_mode=1):
    codes = np.ascontiguousarray(codes, np.uint8)
    dists = np.ascontiguousarray(dists, np.float32)
    out = np.zeros(codes.shape[0], np.float32)
    lib().orc_candidates_f32_mode(M, _p(codes, u8p), C.c_long(codes.shape[0]), _p(dists, f32p), sum_mode, _p(out, f32p))
    return out


def scan4_start(M, parts, labels, tables, R, sum_mode=1):
    """query_scan_start alone: push(0, FLT_MAX) + scan_4<M> over each run with its table -> float heap (keys, values)."""
    parts = [np.ascontiguousarray(p, np.uint8) for p in parts]
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    tb = np.ascontiguousarray(tables, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    lib().orc_scan4_start(M, len(parts), pa, la, _p(sizes, u32p), _p(tb, f32p), R, sum_mode,
                          _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def scan_standard_u8(NSQ, parts, labels, tables, R, sum_mode=1):
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    tb = np.ascontiguousarray(tables, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    lib().orc_scan_standard_u8(NSQ, len(parts), pa, la, _p(sizes, u32p), _p(tb, f32p), R, sum_mode,
                               _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def quantize_tables(tables, qmin, qmax, mode=1):
    tb = np.ascontiguousarray(tables, np.float32)
    out = np.zeros(tb.shape, np.int8)
    lib().orc_quantize_tables(_p(tb, f32p), C.c_long(tb.size), C.c_float(qmin), C.c_float(qmax), mode,
                              _p(out, i8p))
    return out


def tables_direct(centroids, vector, sum_mode=1):
    """Direct table form (compute_dists_single_simd_cg): centroids [M][16][dsq], vector [M*dsq] -> [M*16] float32."""
    cf = np.ascontiguousarray(centroids, np.float32)
    M, _, dsq = cf.shape
    v = np.ascontiguousarray(vector, np.float32)
    out = np.zeros(M * 16, np.float32)
    lib().orc_tables_direct(dsq, M, _p(cf, f32p), _p(v, f32p), sum_mode, _p(out, f32p))
    return out


def cross_dists(centroids, vectors, sum_mode=1, with_product=True):
    """compute_cross_dists_blas restated (orc_cross_dists): centroids [cent_count][dsq], vectors [count][dsq] ->
    [count][cent_count] float32 = fma(-2, v.c, ||v||^2 + ||c||^2); with_product=False: the norms matrix sgemm receives."""
    c = np.ascontiguousarray(centroids, np.float32)
    v = np.ascontiguousarray(vectors, np.float32)
    out = np.zeros((v.shape[0], c.shape[0]), np.float32)
    lib().orc_cross_dists(c.shape[1], _p(c, f32p), c.shape[0], _p(v, f32p), C.c_long(v.shape[0]), C.c_long(c.shape[0]),
                          sum_mode, 1 if with_product else 0, _p(out, f32p))
    return out


def tables_expansion(centroids, vectors, sum_mode=1):
    """BLAS-expansion table form (compute_dists_multiple_blas_cg restated): centroids [M][16][dsq], vectors
    [count][M*dsq] (or one vector) -> [count][M*16] float32."""
    cf = np.ascontiguousarray(centroids, np.float32)
    M, _, dsq = cf.shape
    v = np.ascontiguousarray(vectors, np.float32).reshape(-1, M * dsq)
    out = np.zeros((v.shape[0], M * 16), np.float32)
    lib().orc_tables_expansion(dsq, M, _p(cf, f32p), _p(v, f32p), C.c_long(v.shape[0]), sum_mode, _p(out, f32p))
    return out


def pq_encode(codebooks, vectors, rotation=None, form=1, sum_mode=1):
    """base_pq / opq encode_multiple_vectors restated (orc_pq_encode): codebooks [M][16][dsq], vectors [n][dim] ->
    codes [n][M/2] uint8.  form 1 = the reference's (expansion distances, find_k_neighbors with k = 1), 0 = direct."""
    cf = np.ascontiguousarray(codebooks, np.float32)
    M, _, dsq = cf.shape
    v = np.ascontiguousarray(vectors, np.float32).reshape(-1, M * dsq)
    rot = None if rotation is None else np.ascontiguousarray(rotation, np.float32)
    out = np.zeros((v.shape[0], M // 2), np.uint8)
    lib().orc_pq_encode(M, M * dsq, _p(cf, f32p), _p(rot, f32p) if rot is not None else None, _p(v, f32p),
                        C.c_long(v.shape[0]), form, sum_mode, _p(out, u8p))
    return out


def start_size(size, keep):
    return int(lib().orc_start_size(int(size), C.c_float(keep)))


def query_scan(M, parts, labels, keep, assign, tables, R, quant_mode=1, sum_mode=1):
    """Whole scanner_4::query_scan on row-major database partitions.  `tables` [ma][M*16] float32 is
    modified in place (negative clamp).  Returns dict(rc, qmin, qmax, qtables, keys, values)."""
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    assign = np.ascontiguousarray(assign, np.int32)
    ma = len(assign)
    assert tables.dtype == np.float32 and tables.flags.c_contiguous
    qmin, qmax = C.c_float(0), C.c_float(0)
    qt = np.zeros((ma, M, 16), np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    rc = lib().orc_query_scan(M, pa, la, _p(sizes, u32p), C.c_float(keep), _p(assign, i32p), ma,
                              _p(tables, f32p), R, quant_mode, sum_mode, C.byref(qmin), C.byref(qmax),
                              _p(qt, i8p), _p(ok, u32p), _p(ov, i8p), C.byref(osz))
    return dict(rc=rc, qmin=qmin.value, qmax=qmax.value, qtables=qt,
                keys=ok[:osz.value].copy(), values=ov[:osz.value].copy())


def fill_codes(first_word, nwords, seed):
    out = np.zeros(nwords * 8, np.uint8)
    lib().orc_fill_codes(_p(out, u8p), C.c_uint64(first_word), C.c_uint64(nwords), C.c_uint64(seed))
    return out


# ----------------------------------------------------------------------------- reference build
def ref_interleave(codes):
    codes = np.ascontiguousarray(codes, np.uint8)
    n, cs = codes.shape
    out = np.zeros(ref().qadc_ref_interleaved_size(n, cs), np.uint8)
    ref().qadc_ref_interleave(_p(out, u8p), _p(codes, u8p), n, cs)
    return out


def ref_scan(M, parts_rowmajor, labels, qtables, R, sentinel=True, want_sorted=False):
    """Runs the reference's scan_avx_4<M> over partitions given row-major (they are interleaved
    with the reference's own interleave_partition_4 first)."""
    inter = [ref_interleave(p) for p in parts_rowmajor]
    return ref_scan_interleaved(M, inter, [p.shape[0] for p in parts_rowmajor], labels, qtables, R,
                                sentinel, want_sorted)


def ref_scan_interleaved(M, inter_parts, sizes, labels, qtables, R, sentinel=True, want_sorted=False):
    sizes = np.ascontiguousarray(sizes, np.uint32)
    pa, keep1 = _ptr_array(inter_parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    qt = np.ascontiguousarray(qtables, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    sk = np.zeros(R, np.uint32)
    rc = ref().qadc_ref_scan(M, len(inter_parts), pa, la, _p(sizes, u32p), _p(qt, i8p), R, int(sentinel),
                             _p(ok, u32p), _p(ov, i8p), C.byref(osz), _p(sk, u32p) if want_sorted else None)
    assert rc == 0
    if want_sorted:
        return ok[:osz.value].copy(), ov[:osz.value].copy(), sk[:osz.value].copy()
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def host_topology():
    """(cpu model, [one logical cpu per physical core, restricted to the cpus this process may run on])."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    seen, cpus = set(), []
    for c in allowed:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            cpus.append(c)
    # a container may see every core of the host but be allowed only a CPU-time quota (cgroup v2 cpu.max / v1
    # cfs_quota): more runnable threads than that are throttled, so the "all cores" leg uses at most that many
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None and quota < len(cpus):
        k = max(1, int(quota))
        # spread the threads over the whole package (core complexes share an L3 slice and a memory link: the first k
        # cores would sit on one or two of them and measure that link, not the CPU)
        cpus = [cpus[(i * len(cpus)) // k] for i in range(k)]
    return model, cpus


def ref_scan_mt(M, inter, size, qtables, R, cpus, seconds):
    """All-cores CPU leg (qadc_ref_scan_mt): one pinned thread per entry of `cpus`, each with its own copy of the
    interleaved partition.  Returns (whole queries finished, wall seconds)."""
    inter = np.ascontiguousarray(inter, np.uint8)
    qt = np.ascontiguousarray(qtables, np.int8).reshape(-1, M, 16)
    cp = np.ascontiguousarray(cpus, np.int32)
    nq, dt = C.c_long(0), C.c_double(0)
    rc = ref().qadc_ref_scan_mt(M, _p(inter, u8p), C.c_uint32(size), _p(qt, i8p), qt.shape[0], R, len(cp),
                                _p(cp, i32p), C.c_double(seconds), C.byref(nq), C.byref(dt))
    assert rc == 0
    return nq.value, dt.value


def ref_heap_replay_i8(keys, vals, R, want_sorted=False):
    keys = np.ascontiguousarray(keys, np.uint32)
    vals = np.ascontiguousarray(vals, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
    sk = np.zeros(R, np.uint32)
    ref().qadc_ref_heap_replay_i8(C.c_long(len(keys)), _p(keys, u32p), _p(vals, i8p), R,
                                  _p(ok, u32p), _p(ov, i8p), C.byref(osz), _p(sk, u32p) if want_sorted else None)
    if want_sorted:
        return ok[:osz.value].copy(), ov[:osz.value].copy(), sk[:osz.value].copy()
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def ref_heap_replay_f32(keys, vals, R):
    keys = np.ascontiguousarray(keys, np.uint32)
    vals = np.ascontiguousarray(vals, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    ref().qadc_ref_heap_replay_f32(C.c_long(len(keys)), _p(keys, u32p), _p(vals, f32p), R,
                                   _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


# ----------------------------------------------------------------------------- reference build: vecs I/O + recall rule
# oracle/_ref/libqadc_ref_io.so = the reference's vector_io.cpp / vector_io.hpp / recall.hpp compiled as they are
# (oracle/ref_io_harness.cpp).  Pins host/qadc_io.hpp's vecs side and host/query_driver.hpp's recall rule.
_ref_io = None


def have_ref_io():
    return os.path.exists(_REF_IO_SO)


def ref_io():
    global _ref_io
    if _ref_io is None:
        _ref_io = C.CDLL(_REF_IO_SO)
        _ref_io.qadc_ref_io_free.argtypes = [C.c_void_p]
        _ref_io.qadc_ref_io_free.restype = None
    return _ref_io


def _ref_io_take(ptr, ctype, dim, count, dtype):
    n = dim.value * count.value
    out = np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dtype, copy=True).reshape(count.value, dim.value)
    ref_io().qadc_ref_io_free(C.cast(ptr, C.c_void_p))
    return out


def ref_load_vectors(filename):
    """load_vectors_by_extension (vector_io.cpp:40-58) -> float32 [count][dim].  Exits the process on bad input: probe
    with ref_try_load first."""
    ptr, dim, count = f32p(), C.c_int(0), C.c_long(0)
    rc = ref_io().qadc_ref_io_load(filename.encode(), C.byref(ptr), C.byref(dim), C.byref(count))
    assert rc == 0
    return _ref_io_take(ptr, C.c_float, dim, count, np.float32)


def ref_load_ivecs(filename):
    """load_vectors<int> as recall_file reads its ground truth (recall.hpp:37-39) -> int32 [count][dim]."""
    ptr, dim, count = i32p(), C.c_int(0), C.c_long(0)
    rc = ref_io().qadc_ref_io_load_ivecs(filename.encode(), C.byref(ptr), C.byref(dim), C.byref(count))
    assert rc == 0
    return _ref_io_take(ptr, C.c_int32, dim, count, np.int32)


def ref_save_vectors(filename, arr):
    """save_vectors<T> (vector_io.hpp:153-166), T by dtype: float32 / uint8 / int32."""
    a = np.ascontiguousarray(arr)
    fn = {np.dtype(np.float32): ("qadc_ref_io_save_f32", f32p), np.dtype(np.uint8): ("qadc_ref_io_save_u8", u8p),
          np.dtype(np.int32): ("qadc_ref_io_save_i32", i32p)}[a.dtype]
    getattr(ref_io(), fn[0])(filename.encode(), _p(a, fn[1]), C.c_int(a.shape[1]), C.c_long(a.shape[0]))


def ref_try_load(filename):
    """-> (exit code, stderr text) of a child process calling load_vectors_by_extension: the reference reports errors by
    message + std::exit(1)."""
    buf = C.create_string_buffer(4096)
    rc = ref_io().qadc_ref_io_try_load(filename.encode(), buf, 4096)
    return rc, buf.value.decode(errors="replace")


def ref_read_chunked(filename, chunk_count, max_vectors, max_dim):
    """The reference's vectors_reader driven like db_add.cpp:52-82 -> (data [count][dim], [(offset, count), ...]).  (The harness
    also collects what the reference's own loop leaves in the queue when its done() fires early: see ref_io_harness.cpp.)"""
    out = np.zeros(max_vectors * max_dim, np.float32)
    offs, cnts = np.zeros(4096, np.uint32), np.zeros(4096, np.uint32)
    nch, dim, total, early = C.c_int(0), C.c_int(0), C.c_uint(0), C.c_int(0)
    rc = ref_io().qadc_ref_io_read_chunked(filename.encode(), C.c_uint(chunk_count), _p(out, f32p), C.c_long(out.size),
                                           _p(offs, u32p), _p(cnts, u32p), 4096, C.byref(nch), C.byref(dim), C.byref(total), C.byref(early))
    assert rc == 0, rc
    return (out[:total.value * dim.value].reshape(total.value, dim.value).copy(),
            [(int(offs[i]), int(cnts[i])) for i in range(nch.value)])


def ref_check_labels(gt_filename, keys, t):
    """recall_file(gt).check_labels(q, keys[q], keys[q] + n, t) for every query (recall.hpp:46-54) -> int32 [nq]."""
    k = np.ascontiguousarray(keys, np.uint32)
    out = np.zeros(k.shape[0], np.int32)
    rc = ref_io().qadc_ref_io_check_labels(gt_filename.encode(), k.shape[0], _p(k, u32p), k.shape[1], t, _p(out, i32p))
    assert rc == 0
    return out


# ----------------------------------------------------------------------------- reference build: the float half
# oracle/_ref/libqadc_ref_float.so = QuantizerMAX, scan_4, scanner_4 (whole), scanner_simple, base_pq, multiple_set_bits_4,
# fmanorm / compute_dists_single_simd_cg and substract_vectors_from_unique compiled from line ranges of the reference's own
# files (oracle/ref_extract.sh, oracle/ref_float_harness.cpp).  Pins rows A5 / A6 / A7 / A8 and the direct form of A10.
_ref_float = None


def have_ref_float():
    return os.path.exists(_REF_FLOAT_SO)


def ref_float():
    global _ref_float
    if _ref_float is None:
        _ref_float = C.CDLL(_REF_FLOAT_SO)
        _ref_float.qadc_reff_scanner4_create.restype = C.c_void_p
    return _ref_float


def reff_quantize_tables(tables, qmin, qmax):
    """QuantizerMAX<int8_t>(qmin, qmax).quantize_tables as compiled here; tables [..., 16] float32."""
    tb = np.ascontiguousarray(tables, np.float32)
    assert tb.size % 16 == 0
    out = np.zeros(tb.shape, np.int8)
    ref_float().qadc_reff_quantize_tables(_p(tb, f32p), tb.size // 16, C.c_float(qmin), C.c_float(qmax), _p(out, i8p))
    return out


def reff_scan4_start(M, parts, labels, tables, R):
    """push(0, FLT_MAX) + scan_4<M> over each run with its table -> float heap (keys, values); qmax = values[0]."""
    parts = [np.ascontiguousarray(p, np.uint8) for p in parts]
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    tb = np.ascontiguousarray(tables, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    rc = ref_float().qadc_reff_scan4_start(M, len(parts), pa, la, _p(sizes, u32p), _p(tb, f32p), R,
                                           _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    assert rc == 0
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def reff_scan_standard_u8(NSQ, parts, labels, tables, R):
    """scanner_simple::query_scan (db_query.cpp:26-45) over an in-memory database -> float heap."""
    parts = [np.ascontiguousarray(p, np.uint8) for p in parts]
    sizes = np.array([p.shape[0] for p in parts], np.uint32)
    pa, keep1 = _ptr_array(parts, u8p)
    la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
    tb = np.ascontiguousarray(tables, np.float32)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
    rc = ref_float().qadc_reff_scan_standard_u8(NSQ, len(parts), pa, la, _p(sizes, u32p), _p(tb, f32p), R,
                                                _p(ok, u32p), _p(ov, f32p), C.byref(osz))
    assert rc == 0
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def reff_pack4(assign, M):
    assign = np.ascontiguousarray(assign, np.int32)
    n = assign.shape[0]
    codes = np.zeros((n, M // 2), np.uint8)
    ref_float().qadc_reff_pack4(_p(assign, i32p), C.c_long(n), M, _p(codes, u8p))
    return codes


def reff_tables_direct(centroids, vector):
    """compute_dists_single_simd_cg<DSQ>: centroids [M][16][DSQ], vector [M*DSQ] -> [M*16] float32."""
    cf = np.ascontiguousarray(centroids, np.float32)
    M, _, dsq = cf.shape
    v = np.ascontiguousarray(vector, np.float32)
    out = np.zeros(M * 16, np.float32)
    rc = ref_float().qadc_reff_tables_direct(dsq, M, _p(cf, f32p), _p(v, f32p), _p(out, f32p))
    assert rc == 0, "the reference's dispatch has no case for sq_dim %d" % dsq
    return out


def reff_substract_from_unique(vector, base_vectors, assign):
    v = np.ascontiguousarray(vector, np.float32)
    b = np.ascontiguousarray(base_vectors, np.float32)
    a = np.ascontiguousarray(assign, np.int32)
    out = np.zeros((len(a), len(v)), np.float32)
    ref_float().qadc_reff_substract_from_unique(_p(v, f32p), len(v), _p(b, f32p), _p(a, i32p), len(a), _p(out, f32p))
    return out


def select_k_neighbors(dists, k):
    """The oracle's restatement of find_k_neighbors' selection half (orc_select_k_neighbors; pinned to the reference's own
    heaps, exact ties included) -> (assign [count][k] int32, sorted distances [count][k])."""
    d = np.ascontiguousarray(dists, np.float32)
    if d.ndim == 1:
        d = d[None, :]
    count, nn = d.shape
    a = np.zeros((count, k), np.int32)
    sd = np.zeros((count, k), np.float32)
    lib().orc_select_k_neighbors(_p(d, f32p), C.c_long(count), nn, k, _p(a, i32p), _p(sd, f32p))
    return a, sd


def reff_kmeans_update(vectors, assign, K):
    """kmeans_fast_iterations_thread's centroid-update loops (databases.cpp:70-88) as compiled with the reference's flags ->
    centroids [K][dim]."""
    v = np.ascontiguousarray(vectors, np.float32)
    a = np.ascontiguousarray(assign, np.int32)
    c = np.zeros((K, v.shape[1]), np.float32)
    ref_float().qadc_reff_kmeans_update(_p(v, f32p), C.c_long(v.shape[0]), v.shape[1], K, _p(a, i32p), _p(c, f32p))
    return c


def reff_parse_data_filename(filename):
    """parse_data_filename (quantizers.cpp:58-87) in a child process: 0 = .pq.data, 1 = .opq.data, 101 = its exit(1)."""
    return int(ref_float().qadc_reff_parse_data_filename(str(filename).encode()))


def reff_pq_from_data_file(filename):
    """The reference's pq_from_data_file factory (quantizers.cpp:27-46, 89-103) -> dict(dim, m, b, is_opq, centroids, rotation)."""
    L = ref_float()
    L.qadc_reff_pq_from_data_file.restype = C.c_long
    dim, m, b, o = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    fn = str(filename).encode()
    need = L.qadc_reff_pq_from_data_file(fn, C.byref(dim), C.byref(m), C.byref(b), C.byref(o), None, C.c_long(0), None, C.c_long(0))
    cent = np.zeros(need, np.float32)
    rot = np.zeros(dim.value * dim.value if o.value else 0, np.float32)
    L.qadc_reff_pq_from_data_file(fn, C.byref(dim), C.byref(m), C.byref(b), C.byref(o), _p(cent, f32p), C.c_long(need),
                                  _p(rot, f32p) if o.value else None, C.c_long(len(rot)))
    return dict(dim=dim.value, m=m.value, b=b.value, is_opq=bool(o.value), centroids=cent,
                rotation=rot.reshape(dim.value, dim.value) if o.value else None)


def reff_select_k_neighbors(dists, k):
    """The selection half of find_k_neighbors (neighbors.cpp:18-28, 47-71: add_candidates_heaps block by block, then
    kv_binheap::sort) on given distances [count][neighbor_count] -> (assign [count][k] int32, sorted distances)."""
    d = np.ascontiguousarray(dists, np.float32)
    count, nn = d.shape
    a = np.zeros((count, k), np.int32)
    sd = np.zeros((count, k), np.float32)
    ref_float().qadc_reff_select_k_neighbors(_p(d, f32p), count, nn, k, _p(a, i32p), _p(sd, f32p))
    return a, sd


def reff_cross_norms(centroids, vectors):
    """compute_cross_dists_blas<DSQ> (distances.hpp:151-215) UP TO its cblas_sgemm call: [count][cent_count] float32 =
    ||v||^2 + ||c||^2 as the reference's own text computes it with its flags.  None for a DSQ outside its dispatch."""
    c = np.ascontiguousarray(centroids, np.float32)
    v = np.ascontiguousarray(vectors, np.float32)
    out = np.zeros((v.shape[0], c.shape[0]), np.float32)
    rc = ref_float().qadc_reff_cross_norms(c.shape[1], _p(c, f32p), c.shape[0], _p(v, f32p), v.shape[0], _p(out, f32p))
    return out if rc == 0 else None


def reff_extract_subvectors(vectors, subq_dim, sq_i):
    v = np.ascontiguousarray(vectors, np.float32)
    out = np.zeros((v.shape[0], subq_dim), np.float32)
    ref_float().qadc_reff_extract_subvectors(_p(v, f32p), v.shape[1], v.shape[0], subq_dim, sq_i, _p(out, f32p))
    return out


class RefScanner4:
    """The reference's scanner_4, whole (db_query_4.cpp:73-310), over row-major partitions held in memory."""

    def __init__(self, M, parts, labels, keep):
        self.M = M
        parts = [np.ascontiguousarray(p, np.uint8).reshape(-1, M // 2) for p in parts]
        self.nparts = len(parts)
        sizes = np.array([p.shape[0] for p in parts], np.uint32)
        pa, keep1 = _ptr_array(parts, u8p)
        la, keep2 = _ptr_array(labels, u32p) if labels is not None else (None, None)
        self._h = C.c_void_p(ref_float().qadc_reff_scanner4_create(M, C.c_float(keep), self.nparts, pa, la, _p(sizes, u32p)))
        assert self._h

    @staticmethod
    def try_prepare(M, parts, labels, keep):
        """exit status of prepare_database in a child process; labels: list with None entries allowed (mixed)."""
        parts = [np.ascontiguousarray(p, np.uint8).reshape(-1, M // 2) for p in parts]
        sizes = np.array([p.shape[0] for p in parts], np.uint32)
        pa, keep1 = _ptr_array(parts, u8p)
        la = None
        if labels is not None:
            keep2 = [None if l is None else np.ascontiguousarray(l, np.uint32) for l in labels]
            la = (u32p * len(keep2))(*[None if l is None else _p(l, u32p) for l in keep2])
        return ref_float().qadc_reff_scanner4_try_prepare(M, C.c_float(keep), len(parts), pa, la, _p(sizes, u32p))

    def close(self):
        if self._h:
            ref_float().qadc_reff_scanner4_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def sizes(self):
        st, ps, hl = np.zeros(self.nparts, np.uint32), np.zeros(self.nparts, np.uint32), C.c_int(0)
        ref_float().qadc_reff_scanner4_sizes(self._h, _p(st, u32p), _p(ps, u32p), C.byref(hl))
        return st, ps, bool(hl.value)

    def query_start(self, assign, tables, R):
        a = np.ascontiguousarray(assign, np.int32)
        tb = np.ascontiguousarray(tables, np.float32)
        ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.float32), C.c_int(0)
        ref_float().qadc_reff_scanner4_query_start(self._h, _p(a, i32p), len(a), _p(tb, f32p), R,
                                                   _p(ok, u32p), _p(ov, f32p), C.byref(osz))
        return ok[:osz.value].copy(), ov[:osz.value].copy()

    def try_query(self, assign, tables, R):
        a = np.ascontiguousarray(assign, np.int32)
        tb = np.ascontiguousarray(tables, np.float32)
        return ref_float().qadc_reff_scanner4_try_query(self._h, _p(a, i32p), len(a), _p(tb, f32p), R)

    def query_scan(self, assign, tables, R, want_sorted=False):
        """`tables` float32 [ma][M*16], C-contiguous, modified in place (negative clamp).  EXITS the process if
        qmax > 1e30: call try_query first where that can happen."""
        a = np.ascontiguousarray(assign, np.int32)
        assert tables.dtype == np.float32 and tables.flags.c_contiguous
        ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int(0)
        sk = np.zeros(R, np.uint32)
        ref_float().qadc_reff_scanner4_query_scan(self._h, _p(a, i32p), len(a), _p(tables, f32p), R,
                                                  _p(ok, u32p), _p(ov, i8p), C.byref(osz),
                                                  _p(sk, u32p) if want_sorted else None)
        if want_sorted:
            return ok[:osz.value].copy(), ov[:osz.value].copy(), sk[:osz.value].copy()
        return ok[:osz.value].copy(), ov[:osz.value].copy()
