"""ctypes binding of libqadc_hip.so (the C-ABI in include/qadc.h) for tests and bench.py.

Thin by design: numpy in, numpy out, no computation here.  Loading fails loudly when the HIP
library has not been built; there is no CPU fallback anywhere on this path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libqadc_hip.so")
if os.environ.get("QADC_TEST_HOOKS") == "1" and os.environ.get("QADC_LIB_PATH"):   # same-box A/B of two builds (tools/)
    LIB_PATH = os.environ["QADC_LIB_PATH"]

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)

# every symbol include/qadc.h declares
SYMBOLS = [
    "qadc_last_error", "qadc_version", "qadc_index_create", "qadc_index_destroy",
    "qadc_index_add_partitions", "qadc_index_add_partition_interleaved",
    "qadc_index_add_partition_device", "qadc_index_add_partition_synthetic",
    "qadc_index_add_partition_shard", "qadc_index_add_partition_synthetic_shard", "qadc_query_scan_collect_candidates",
    "qadc_index_set_key_base", "qadc_index_finalize", "qadc_index_partition_count",
    "qadc_index_partition_size", "qadc_index_start_size", "qadc_set_option", "qadc_option_names",
    "qadc_index_read_codes", "qadc_query_scan", "qadc_query_scan_candidates", "qadc_scan_i8",
    "qadc_scan_i8_candidates", "qadc_scan_start", "qadc_query_scan_submit", "qadc_prescan_submit",
    "qadc_prescan_collect", "qadc_query_scan_submit_prescanned",
    "qadc_query_scan_collect", "qadc_index_set_pq", "qadc_index_set_rotation", "qadc_index_set_coarse", "qadc_search", "qadc_search_submit",
    "qadc_search_collect", "qadc_device_prepare", "qadc_stream_probe", "qadc_stream_layout", "qadc_pq_encode", "qadc_pq_encode_host", "qadc_ivf_encode_host", "qadc_pq_encode_mode", "qadc_pq_encode_host_mode", "qadc_ivf_encode_host_mode", "qadc_kmeans_iterations_host", "qadc_kmeans_iterations_host_mode", "qadc_replay_i8", "qadc_sort_keys_i8", "qadc_merge_streams_i8", "qadc_candidates_i8", "qadc_float_top1", "qadc_profile_read", "qadc_profile_reset",
    "qadc_dist_unique_id", "qadc_dist_init", "qadc_dist_collect", "qadc_dist_shutdown", "qadc_dist_merge_blocks", "qadc_dist_merge_blocks_host",
    "qadc_dist_init_transport", "qadc_dist_init_loopback", "qadc_shm_transport_open", "qadc_shm_transport_allgather", "qadc_shm_transport_allgather_host",
    "qadc_shm_transport_close", "qadc_shm_transport_error", "qadc_slot_assign", "qadc_slot_qtables", "qadc_place_partitions",
]


class Profile(C.Structure):
    _fields_ = [("scan_launches", C.c_uint64), ("scan_codes", C.c_uint64), ("scan_ms", C.c_double),
                ("small_launches", C.c_uint64), ("small_codes", C.c_uint64), ("small_ms", C.c_double),
                ("start_codes", C.c_uint64), ("start_ms", C.c_double), ("candidates", C.c_uint64),
                ("regrows", C.c_uint64), ("host_replay_ms", C.c_double), ("host_plan_ms", C.c_double),
                ("host_heap_ms", C.c_double), ("host_sorted_queries", C.c_uint64), ("mq_launches", C.c_uint64),
                ("pass_codes", C.c_uint64), ("wgq_launches", C.c_uint64), ("wgq_queries", C.c_uint64),
                ("wgq_codes", C.c_uint64), ("wgq_ms", C.c_double), ("wgq_front_cycles", C.c_uint64),
                ("wgq_scan_cycles", C.c_uint64), ("wgq_sort_cycles", C.c_uint64), ("head_launches", C.c_uint64),
                ("group_launches", C.c_uint64), ("group_fallbacks", C.c_uint64),
                ("group_head_ms", C.c_double), ("group_scan_ms", C.c_double), ("group_order_ms", C.c_double),
                ("group_head_codes", C.c_uint64), ("group_pairs", C.c_uint64), ("group_seats", C.c_uint64),
                ("group_pass_codes8", C.c_uint64), ("group_pass_codes4", C.c_uint64), ("group_batches", C.c_uint64),
                ("front_sharded_batches", C.c_uint64), ("dist_async_collects", C.c_uint64),
                ("lone_front_launches", C.c_uint64)]


class QadcError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QadcError("libqadc_hip.so is not built (run __graft_entry__.build() or make -C quick-adc_amd); "
                            "the Quick-ADC engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.qadc_last_error.restype = C.c_char_p
        L.qadc_version.restype = C.c_char_p
        L.qadc_index_partition_size.restype = C.c_uint32
        L.qadc_index_start_size.restype = C.c_uint32
        L.qadc_index_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        L.qadc_index_destroy.argtypes = [C.c_void_p]
        L.qadc_index_add_partitions.argtypes = [C.c_void_p, C.c_int, C.POINTER(u8p), C.POINTER(u32p), u32p]
        L.qadc_index_add_partition_interleaved.argtypes = [C.c_void_p, u8p, u32p, C.c_uint32]
        L.qadc_index_add_partition_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.qadc_index_add_partition_synthetic.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]
        L.qadc_index_add_partition_shard.argtypes = [C.c_void_p, u8p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u8p,
                                                     C.c_uint32]
        L.qadc_index_add_partition_synthetic_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                               C.c_uint64, C.c_uint32]
        L.qadc_query_scan_collect_candidates.argtypes = [C.c_void_p, C.c_int, C.c_uint64, u32p, i8p, C.POINTER(C.c_uint16),
                                                         u64p, i32p, f32p, f32p]
        L.qadc_index_set_key_base.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.qadc_index_finalize.argtypes = [C.c_void_p, C.c_float]
        L.qadc_index_partition_count.argtypes = [C.c_void_p]
        L.qadc_index_partition_size.argtypes = [C.c_void_p, C.c_int]
        L.qadc_index_start_size.argtypes = [C.c_void_p, C.c_int]
        L.qadc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.qadc_index_read_codes.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, u8p]
        L.qadc_query_scan.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, f32p, C.c_int, u32p, i8p, i32p, i32p,
                                      f32p, f32p, i8p]
        L.qadc_query_scan_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, f32p, C.c_int, C.c_uint64,
                                                 u32p, i8p, u64p, i32p, f32p, f32p]
        L.qadc_scan_i8.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i8p, C.c_int, u32p, i8p, i32p]
        L.qadc_scan_i8_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i8p, C.c_int, C.c_uint64,
                                              u32p, i8p, u64p]
        L.qadc_scan_start.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, f32p, C.c_int, f32p]
        L.qadc_query_scan_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_int]
        L.qadc_prescan_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_int, C.c_int, C.c_int]
        L.qadc_prescan_collect.argtypes = [C.c_void_p, C.c_int, f32p]
        L.qadc_query_scan_submit_prescanned.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_int,
                                                        f32p, C.c_int]
        L.qadc_query_scan_collect.argtypes = [C.c_void_p, C.c_int, u32p, i8p, i32p, i32p, f32p, f32p, i8p]
        L.qadc_index_set_pq.argtypes = [C.c_void_p, C.c_int, f32p]
        L.qadc_index_set_rotation.argtypes = [C.c_void_p, f32p]
        L.qadc_index_set_coarse.argtypes = [C.c_void_p, C.c_int, f32p]
        L.qadc_search.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, C.c_int, u32p, i8p, i32p, i32p, i32p]
        L.qadc_search_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, C.c_int, C.c_int]
        L.qadc_search_collect.argtypes = [C.c_void_p, C.c_int, u32p, i8p, i32p, i32p, i32p]
        L.qadc_pq_encode.argtypes = [C.c_int, C.c_int, f32p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.qadc_pq_encode_host.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_uint64, u8p, C.c_int]
        L.qadc_ivf_encode_host.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_int, f32p, f32p, C.c_uint64, i32p, u8p, C.c_int]
        L.qadc_pq_encode_mode.argtypes = [C.c_int, C.c_int, f32p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.qadc_pq_encode_host_mode.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_uint64, u8p, C.c_int, C.c_int, C.c_int]
        L.qadc_ivf_encode_host_mode.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_int, f32p, f32p, C.c_uint64, i32p, u8p, C.c_int, C.c_int,
                                                C.c_int]
        L.qadc_kmeans_iterations_host.argtypes = [f32p, C.c_uint64, C.c_int, C.c_int, f32p, C.c_int, i32p, C.c_int]
        L.qadc_kmeans_iterations_host_mode.argtypes = [f32p, C.c_uint64, C.c_int, C.c_int, f32p, C.c_int, i32p, C.c_int, C.c_int]
        L.qadc_replay_i8.argtypes = [C.c_uint64, u32p, i8p, C.c_int, C.c_int, u32p, i8p, i32p]
        L.qadc_sort_keys_i8.argtypes = [C.c_int, u32p, i8p, u32p]
        L.qadc_merge_streams_i8.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, i32p, C.c_uint64, C.c_int,
                                            C.c_int, i32p, u32p, i8p, i32p]
        L.qadc_candidates_i8.argtypes = [C.c_void_p, C.c_int, i8p, i8p]
        L.qadc_float_top1.argtypes = [C.c_void_p, C.c_int, f32p, u32p, u32p, f32p]
        L.qadc_dist_unique_id.argtypes = [u8p]
        L.qadc_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p]
        L.qadc_dist_collect.argtypes = [C.c_void_p, C.c_int, u32p, i8p, i32p, i32p, f32p, C.c_int, f32p]
        L.qadc_dist_shutdown.argtypes = [C.c_void_p]
        L.qadc_dist_merge_blocks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_uint64, u32p, i8p, i32p]
        L.qadc_dist_merge_blocks_host.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_uint64, u32p, i8p, i32p]
        L.qadc_dist_init_transport.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.qadc_dist_init_loopback.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qadc_shm_transport_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_double, C.POINTER(C.c_void_p)]
        L.qadc_shm_transport_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.qadc_shm_transport_allgather_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.qadc_shm_transport_close.argtypes = [C.c_void_p]
        L.qadc_shm_transport_error.restype = C.c_char_p
        L.qadc_slot_assign.argtypes = [C.c_void_p, C.c_int, i32p]
        L.qadc_slot_qtables.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, i8p]
        L.qadc_place_partitions.argtypes = [C.c_int, u32p, C.c_int, i32p]
        L.qadc_profile_read.argtypes = [C.c_void_p, C.POINTER(Profile)]
        L.qadc_profile_reset.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _check(rc):
    if rc != 0:
        raise QadcError("qadc error %d: %s" % (rc, lib().qadc_last_error().decode()))


def replay_i8(keys, vals, R, sentinel=False):
    """Host-only heap replay (no GPU needed)."""
    keys = np.ascontiguousarray(keys, np.uint32)
    vals = np.ascontiguousarray(vals, np.int8)
    ok, ov, osz = np.zeros(R, np.uint32), np.zeros(R, np.int8), C.c_int32(0)
    _check(lib().qadc_replay_i8(len(keys), _p(keys, u32p), _p(vals, i8p), R, int(sentinel), _p(ok, u32p),
                                _p(ov, i8p), C.byref(osz)))
    return ok[:osz.value].copy(), ov[:osz.value].copy()


def device_prepare(device=0):
    """qadc_device_prepare: the library's per-device stream set, created now — call before any communicator is initialised."""
    _check(lib().qadc_device_prepare(int(device)))


def option_names():
    """The names qadc_set_option accepts (qadc_option_names)."""
    f = lib().qadc_option_names
    f.restype = C.c_char_p
    return f().decode().split(",")


def stream_layout(device=0):
    """qadc_stream_layout: '<stream creation order kept> | <ok or the obstructed pairs>'."""
    f = lib().qadc_stream_layout
    f.restype = C.c_char_p
    return f(int(device)).decode()


def stream_probe(a, b, device=0):
    """qadc_stream_probe: (microseconds a one-wave marker on stream b waited behind a CU-hungry launch on stream a, that launch's
    duration).  Streams: 0 scan, 1 copy, 2 ordering, 3 front, 4 alternative scan, 5 collectives, 6 merge."""
    w, sp = C.c_double(0), C.c_double(0)
    _check(lib().qadc_stream_probe(int(device), int(a), int(b), C.byref(w), C.byref(sp)))
    return w.value, sp.value


def sort_keys_i8(heap_keys, heap_vals):
    """Host-only: kv_binheap::sort_keys of a heap array (binheap.hpp:129-137)."""
    k = np.ascontiguousarray(heap_keys, np.uint32)
    v = np.ascontiguousarray(heap_vals, np.int8)
    out = np.zeros(len(k), np.uint32)
    _check(lib().qadc_sort_keys_i8(len(k), _p(k, u32p), _p(v, i8p), _p(out, u32p)))
    return out


def dist_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the other ranks receive it by any means)."""
    uid = np.zeros(128, np.uint8)
    _check(lib().qadc_dist_unique_id(_p(uid, u8p)))
    return uid


class ShmTransport:
    """The library's built-in host-staged all-gather over a POSIX shared-memory segment (qadc_shm_transport_*): the
    transport of ranks that share one GPU or have no RCCL.  name: "/something", unique per run."""

    def __init__(self, name, rank, world, slot_bytes=64 << 20, timeout_s=120.0):
        self.rank, self.world = rank, world
        self._ctx = C.c_void_p()
        rc = lib().qadc_shm_transport_open(name.encode(), rank, world, slot_bytes, timeout_s, C.byref(self._ctx))
        if rc != 0:
            raise QadcError("qadc error %d: %s" % (rc, lib().qadc_shm_transport_error().decode()))

    def allgather_host(self, arr):
        """Host buffers only (no GPU): every rank's `arr` -> [world][...]."""
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + a.shape, a.dtype)
        rc = lib().qadc_shm_transport_allgather_host(self._ctx, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.nbytes)
        if rc != 0:
            raise QadcError("qadc error %d: %s" % (rc, lib().qadc_shm_transport_error().decode()))
        return out

    def close(self):
        if self._ctx:
            lib().qadc_shm_transport_close(self._ctx)
            self._ctx = C.c_void_p()


def place_partitions(sizes, world):
    """Size-balanced owner rank of every partition (qadc_place_partitions)."""
    sz = np.ascontiguousarray(sizes, np.uint32)
    owner = np.zeros(len(sz), np.int32)
    _check(lib().qadc_place_partitions(len(sz), _p(sz, u32p), world, _p(owner, i32p)))
    return owner


def dist_merge_blocks(streams, nq, ma, R, device=0, host=False):
    """streams[g] = dict(keys, vals, slots, offsets) of virtual rank g (Index.query_scan_shard_streams): assembles the
    blocks qadc_dist_collect would gather and runs the device-side merge (host=True: the host-share replay the ranks
    run for few-query batches, all ranks' shares in turn).  -> list of (keys, values) heaps."""
    world = len(streams)
    cap = max(int(st["offsets"][-1]) for st in streams) + 3
    bw = 2 * nq + cap
    g = np.zeros((world, bw), np.uint64)
    for r, st in enumerate(streams):
        hdr = g[r, :2 * nq].view(np.uint32).reshape(nq, 4)
        off = st["offsets"].astype(np.int64)
        hdr[:, 0] = off[:-1]
        hdr[:, 1] = np.diff(off)
        hdr[:, 2] = 4
        tot = int(off[-1])
        ent = st["keys"][:tot].astype(np.uint64) | (st["vals"][:tot].astype(np.uint8).astype(np.uint64) << np.uint64(32)) | \
            (st["slots"][:tot].astype(np.uint64) << np.uint64(40))
        g[r, 2 * nq:2 * nq + tot] = ent
    keys = np.zeros((nq, R), np.uint32)
    vals = np.zeros((nq, R), np.int8)
    sizes = np.zeros(nq, np.int32)
    if host:
        _check(lib().qadc_dist_merge_blocks_host(world, nq, ma, R, _p(g, u64p), bw, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p)))
    else:
        _check(lib().qadc_dist_merge_blocks(device, world, nq, ma, R, _p(g, u64p), bw, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p)))
    return [(keys[q, :sizes[q]].copy(), vals[q, :sizes[q]].copy()) for q in range(nq)]


def merge_streams_i8(gathered, world, nq, R, cap, ma, q_first, q_step, status, keys, vals, sizes):
    """Host-only: replay the queries q_first, q_first+q_step, ... of the gathered per-rank streams (sharded.py)."""
    g = np.ascontiguousarray(gathered, np.int32).reshape(world, -1)
    st = None if status is None else np.ascontiguousarray(status, np.int32)
    _check(lib().qadc_merge_streams_i8(world, nq, R, cap, ma, _p(g, i32p), g.shape[1], q_first, q_step, _p(st, i32p),
                                       _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p)))


def ivf_encode(codebooks, vectors, coarse=None, rotation=None, device=0, encode_form=1, sum_mode=1):
    """index_db::add_vectors' compute on the GPU: nearest coarse centroid, residual, optional OPQ rotation, PQ encode.
    -> (assign int32 [n] or None for a flat database, codes uint8 [n][M/2]).  encode_form / sum_mode: see pq_encode."""
    cb = np.ascontiguousarray(codebooks, np.float32)
    v = np.ascontiguousarray(vectors, np.float32)
    M, dim, n = cb.shape[0], v.shape[1], v.shape[0]
    co = None if coarse is None else np.ascontiguousarray(coarse, np.float32)
    rot = None if rotation is None else np.ascontiguousarray(rotation, np.float32)
    assert rot is None or rot.shape == (dim, dim)
    K = 0 if co is None else co.shape[0]
    assign = np.zeros(n, np.int32) if K else None
    codes = np.zeros((n, M // 2), np.uint8)
    _check(lib().qadc_ivf_encode_host_mode(M, dim, _p(cb, f32p), _p(rot, f32p), K, _p(co, f32p), _p(v, f32p), n, _p(assign, i32p),
                                           _p(codes, u8p), encode_form, sum_mode, device))
    return assign, codes


def kmeans_iterations(vectors, centroids, iters, device=0, div_mode=1):
    """kmeans_fast_iterations_thread on the GPU: -> (centroids float32 [K][dim], assign int32 [n] of the last round).
    div_mode 1: centroid = sum * (1 / count) as the reference is compiled (-ffast-math); 0: the source's division."""
    v = np.ascontiguousarray(vectors, np.float32)
    c = np.array(centroids, np.float32, order="C", copy=True)
    assign = np.zeros(v.shape[0], np.int32)
    _check(lib().qadc_kmeans_iterations_host_mode(_p(v, f32p), v.shape[0], v.shape[1], c.shape[0], _p(c, f32p), iters, _p(assign, i32p),
                                                  div_mode, device))
    return c, assign


def pq_encode(codebooks, vectors, device=0, encode_form=1, sum_mode=1):
    """PQ-encode host vectors [n][dim] on the GPU -> uint8 codes [n][M/2].  encode_form 1 (default) = the reference's form
    (find_k_neighbors with k = 1 on the BLAS-expansion distances), 0 = direct sum (x - c)^2; sum_mode 1 = norms as compiled."""
    cb = np.ascontiguousarray(codebooks, np.float32)
    v = np.ascontiguousarray(vectors, np.float32)
    M, dim = cb.shape[0], v.shape[1]
    codes = np.zeros((v.shape[0], M // 2), np.uint8)
    _check(lib().qadc_pq_encode_host_mode(M, dim, _p(cb, f32p), _p(v, f32p), v.shape[0], _p(codes, u8p), encode_form, sum_mode, device))
    return codes


def pq_encode_device(codebooks, d_vectors_ptr, n, dim, d_codes_ptr, device=0):
    cb = np.ascontiguousarray(codebooks, np.float32)
    _check(lib().qadc_pq_encode(cb.shape[0], dim, _p(cb, f32p), C.c_void_p(d_vectors_ptr), n, C.c_void_p(d_codes_ptr), device))


class Index:
    """One GPU-resident Quick-ADC database (the role of scanner_4 after prepare_database)."""

    def __init__(self, M, device=0):
        self.M = M
        self.cs = M // 2
        self._h = C.c_void_p()
        _check(lib().qadc_index_create(C.byref(self._h), M, device))
        self._keepalive = []

    def close(self):
        if self._h:
            lib().qadc_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- database --------------------------------------------------------------------------
    def add_partitions(self, codes, labels=None):
        codes = [np.ascontiguousarray(c, np.uint8).reshape(-1, self.cs) for c in codes]
        sizes = np.array([c.shape[0] for c in codes], np.uint32)
        ca = (u8p * len(codes))(*[_p(c, u8p) for c in codes])
        la = None
        if labels is not None:
            labels = [None if l is None else np.ascontiguousarray(l, np.uint32) for l in labels]
            la = (u32p * len(labels))(*[None if l is None else _p(l, u32p) for l in labels])
        _check(lib().qadc_index_add_partitions(self._h, len(codes), ca, la, _p(sizes, u32p)))

    def add_partition_interleaved(self, inter, size, labels=None):
        inter = np.ascontiguousarray(inter, np.uint8)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint32)
        _check(lib().qadc_index_add_partition_interleaved(self._h, _p(inter, u8p), _p(lab, u32p), size))

    def add_partition_device(self, d_codes_ptr, size, d_labels_ptr=None, keepalive=None):
        self._keepalive.append(keepalive)
        _check(lib().qadc_index_add_partition_device(self._h, C.c_void_p(d_codes_ptr),
                                                     C.c_void_p(d_labels_ptr) if d_labels_ptr else None, size))

    def add_partition_synthetic(self, size, seed, first_word=0):
        _check(lib().qadc_index_add_partition_synthetic(self._h, size, seed, first_word))

    def add_partition_shard(self, codes, local_first, global_n, labels=None, starts=None):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, self.cs)
        lab = None if labels is None else np.ascontiguousarray(labels, np.uint32)
        st = None if starts is None else np.ascontiguousarray(starts, np.uint8).reshape(-1, self.cs)
        _check(lib().qadc_index_add_partition_shard(self._h, _p(codes, u8p), _p(lab, u32p), codes.shape[0], global_n,
                                                    local_first, _p(st, u8p), 0 if st is None else st.shape[0]))

    def add_partition_synthetic_shard(self, global_n, first_pos, local_n, seed, starts_count):
        _check(lib().qadc_index_add_partition_synthetic_shard(self._h, global_n, first_pos, local_n, seed,
                                                              starts_count))

    def set_key_base(self, part, base):
        _check(lib().qadc_index_set_key_base(self._h, part, base))

    def finalize(self, keep):
        _check(lib().qadc_index_finalize(self._h, keep))

    def set_option(self, name, value):
        _check(lib().qadc_set_option(self._h, name.encode(), float(value)))

    def partition_count(self):
        return lib().qadc_index_partition_count(self._h)

    def partition_size(self, p):
        return lib().qadc_index_partition_size(self._h, p)

    def start_size(self, p):
        return lib().qadc_index_start_size(self._h, p)

    def read_codes(self, part, first, count):
        out = np.zeros((count, self.cs), np.uint8)
        _check(lib().qadc_index_read_codes(self._h, part, first, count, _p(out, u8p)))
        return out

    # ---- queries ---------------------------------------------------------------------------
    @staticmethod
    def _prep(assign, nq=None):
        assign = np.ascontiguousarray(assign, np.int32)
        if assign.ndim == 1:
            assign = assign.reshape(1, -1) if nq is None else assign.reshape(nq, -1)
        return assign

    def _heaps(self, nq, R, keys, vals, sizes):
        return [(keys[q, :sizes[q]].copy(), vals[q, :sizes[q]].copy()) for q in range(nq)]

    def query_scan(self, assign, tables, R, want_qtables=False):
        """tables: float32 [nq][ma][M*16], mutated in place.  Returns dict with heaps etc."""
        assign = self._prep(assign)
        nq, ma = assign.shape
        assert tables.dtype == np.float32 and tables.flags.c_contiguous and tables.size == nq * ma * self.M * 16
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        status = np.zeros(nq, np.int32)
        qmin = np.zeros(nq, np.float32)
        qmax = np.zeros(nq, np.float32)
        qt = np.zeros((nq, ma, self.M, 16), np.int8) if want_qtables else None
        _check(lib().qadc_query_scan(self._h, nq, ma, _p(assign, i32p), _p(tables, f32p), R, _p(keys, u32p),
                                     _p(vals, i8p), _p(sizes, i32p), _p(status, i32p), _p(qmin, f32p),
                                     _p(qmax, f32p), _p(qt, i8p)))
        return dict(heaps=self._heaps(nq, R, keys, vals, sizes), status=status, qmin=qmin, qmax=qmax, qtables=qt,
                    keys=keys, values=vals, sizes=sizes)

    def submit(self, slot, assign, tables, R, prescan=None):
        """prescan: float32 [nq][w*R], the gathered output of prescan_collect on all w ranks (sharded pre-scan)."""
        assign = self._prep(assign)
        nq, ma = assign.shape
        assert tables.dtype == np.float32 and tables.flags.c_contiguous
        self._pending = getattr(self, "_pending", {})
        self._pending[slot] = (nq, R, tables, assign)
        if prescan is None:
            _check(lib().qadc_query_scan_submit(self._h, slot, nq, ma, _p(assign, i32p), _p(tables, f32p), R))
        else:
            pv = np.ascontiguousarray(prescan, np.float32).reshape(nq, -1)
            _check(lib().qadc_query_scan_submit_prescanned(self._h, slot, nq, ma, _p(assign, i32p), _p(tables, f32p), R,
                                                           _p(pv, f32p), pv.shape[1]))

    def prescan_submit(self, slot, assign, tables, R, slice_index, nslices):
        """Sharded pre-scan, first half: this rank's slice of the starts (own buffers: may overlap a pending batch)."""
        assign = self._prep(assign)
        nq, ma = assign.shape
        assert tables.dtype == np.float32 and tables.flags.c_contiguous
        self._pre_pending = getattr(self, "_pre_pending", {})
        self._pre_pending[slot] = (nq, R, tables, assign)
        _check(lib().qadc_prescan_submit(self._h, slot, nq, ma, _p(assign, i32p), _p(tables, f32p), R, slice_index, nslices))

    def prescan_collect(self, slot):
        """-> float32 [nq][R]: the R smallest pre-scan distances of the slice per query (FLT_MAX-padded)."""
        nq, R, tables, assign = self._pre_pending.pop(slot)
        vals = np.zeros((nq, R), np.float32)
        _check(lib().qadc_prescan_collect(self._h, slot, _p(vals, f32p)))
        return vals

    def collect(self, slot):
        nq, R, tables, assign = self._pending.pop(slot)
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        status = np.zeros(nq, np.int32)
        _check(lib().qadc_query_scan_collect(self._h, slot, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p),
                                             _p(status, i32p), None, None, None))
        return dict(keys=keys, values=vals, sizes=sizes, status=status)

    def collect_candidates(self, slot, capacity=1 << 18):
        nq, R, tables, assign = self._pending.pop(slot)
        bufs = getattr(self, "_cand_bufs", None)
        u16p = C.POINTER(C.c_uint16)
        if bufs is None or len(bufs[0]) < capacity:        # reused across calls: no per-step page faults
            bufs = self._cand_bufs = (np.zeros(capacity, np.uint32), np.zeros(capacity, np.int8),
                                      np.zeros(capacity, np.uint16))
        ck, cv, cs_ = bufs
        off = np.zeros(nq + 1, np.uint64)
        status = np.zeros(nq, np.int32)
        qmin = np.zeros(nq, np.float32)
        qmax = np.zeros(nq, np.float32)
        rc = lib().qadc_query_scan_collect_candidates(self._h, slot, len(ck), _p(ck, u32p), _p(cv, i8p), _p(cs_, u16p),
                                                      _p(off, u64p), _p(status, i32p), _p(qmin, f32p), _p(qmax, f32p))
        if rc == -3:                                        # QADC_E_CAPACITY: the result is kept, retry larger
            need = int(off[nq]) + 1024
            ck, cv, cs_ = self._cand_bufs = (np.zeros(need, np.uint32), np.zeros(need, np.int8), np.zeros(need, np.uint16))
            rc = lib().qadc_query_scan_collect_candidates(self._h, slot, need, _p(ck, u32p), _p(cv, i8p), _p(cs_, u16p),
                                                          _p(off, u64p), _p(status, i32p), _p(qmin, f32p), _p(qmax, f32p))
        _check(rc)
        return dict(keys=ck, vals=cv, slots=cs_, offsets=off.astype(np.int64), status=status, qmin=qmin, qmax=qmax)

    # ---- native multi-GPU merge (RCCL inside the library) ----------------------------------
    def dist_init(self, rank, world, unique_id):
        uid = np.ascontiguousarray(unique_id, np.uint8)
        assert uid.size == 128
        _check(lib().qadc_dist_init(self._h, rank, world, _p(uid, u8p)))
        self._dist_world = world

    def dist_init_transport(self, transport):
        """The merge over the library's shared-memory transport (ShmTransport) instead of RCCL."""
        fn = C.cast(lib().qadc_shm_transport_allgather, C.c_void_p)
        _check(lib().qadc_dist_init_transport(self._h, transport.rank, transport.world, fn, transport._ctx))
        self._dist_world = transport.world
        self._transport = transport

    def dist_init_loopback(self, rank, world):
        """Measurement aid: this process stands in for rank `rank` of `world` (qadc_dist_init_loopback)."""
        _check(lib().qadc_dist_init_loopback(self._h, rank, world))
        self._dist_world = world

    def dist_shutdown(self):
        _check(lib().qadc_dist_shutdown(self._h))

    def slot_qtables(self, slot, q_first, q_count, ma):
        """int8 tables [q_count][ma][M][16] of the batch last collected from `slot` (qadc_slot_qtables)."""
        out = np.zeros((q_count, ma, self.M, 16), np.int8)
        _check(lib().qadc_slot_qtables(self._h, slot, q_first, q_count, _p(out, i8p)))
        return out

    def slot_assign(self, slot, nq, ma):
        out = np.zeros((nq, ma), np.int32)
        _check(lib().qadc_slot_assign(self._h, slot, _p(out, i32p)))
        return out

    def dist_collect(self, slot, extra=None):
        """Replaces collect(): one ncclAllGather of the ranks' push streams + device-side replay in global scan order.
        Returns dict(keys, values, sizes, status[, extra = float32 [world][n]])."""
        if slot in getattr(self, "_spending", {}):          # a search_submit batch (queries in)
            q, _, R = self._spending.pop(slot)
            nq = q.shape[0]
        else:
            nq, R, tables, assign = self._pending.pop(slot)
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        status = np.zeros(nq, np.int32)
        ex = None if extra is None else np.ascontiguousarray(extra, np.float32).reshape(-1)
        world = getattr(self, "_dist_world", 1)
        ex_out = None if ex is None else np.zeros((world, ex.size), np.float32)
        _check(lib().qadc_dist_collect(self._h, slot, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p), _p(status, i32p),
                                       _p(ex, f32p), 0 if ex is None else ex.size, _p(ex_out, f32p)))
        out = dict(keys=keys, values=vals, sizes=sizes, status=status)
        if ex is not None:
            out["extra"] = ex_out
        return out

    def set_pq(self, codebooks):
        cb = np.ascontiguousarray(codebooks, np.float32)
        assert cb.shape[0] == self.M and cb.shape[1] == 16
        self.dim = self.M * cb.shape[2]
        _check(lib().qadc_index_set_pq(self._h, self.dim, _p(cb, f32p)))

    def set_rotation(self, rotation):
        r = None if rotation is None else np.ascontiguousarray(rotation, np.float32)
        _check(lib().qadc_index_set_rotation(self._h, _p(r, f32p)))

    def set_coarse(self, centroids):
        c = np.ascontiguousarray(centroids, np.float32)
        _check(lib().qadc_index_set_coarse(self._h, c.shape[0], _p(c, f32p)))

    def search(self, queries, ma, R):
        q = np.ascontiguousarray(queries, np.float32)
        nq = q.shape[0]
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        status = np.zeros(nq, np.int32)
        assign = np.zeros((nq, ma), np.int32)
        _check(lib().qadc_search(self._h, nq, _p(q, f32p), ma, R, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p),
                                 _p(status, i32p), _p(assign, i32p)))
        return dict(heaps=self._heaps(nq, R, keys, vals, sizes), status=status, assign=assign, keys=keys,
                    values=vals, sizes=sizes)

    def search_submit(self, slot, queries, ma, R):
        """Asynchronous (slot 0..3) form of search(): enqueue a batch, collect it later (overlaps the host replay of
        one batch with the GPU work of the next)."""
        q = np.ascontiguousarray(queries, np.float32)
        self._spending = getattr(self, "_spending", {})
        self._spending[slot] = (q, ma, R)
        _check(lib().qadc_search_submit(self._h, slot, q.shape[0], _p(q, f32p), ma, R))

    def search_collect(self, slot):
        q, ma, R = self._spending.pop(slot)
        nq = q.shape[0]
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        status = np.zeros(nq, np.int32)
        assign = np.zeros((nq, ma), np.int32)
        _check(lib().qadc_search_collect(self._h, slot, _p(keys, u32p), _p(vals, i8p), _p(sizes, i32p), _p(status, i32p),
                                         _p(assign, i32p)))
        return dict(keys=keys, values=vals, sizes=sizes, status=status, assign=assign)

    def scan_i8(self, assign, qtables, R):
        assign = self._prep(assign)
        nq, ma = assign.shape
        qt = np.ascontiguousarray(qtables, np.int8)
        assert qt.size == nq * ma * self.M * 16
        keys = np.zeros((nq, R), np.uint32)
        vals = np.zeros((nq, R), np.int8)
        sizes = np.zeros(nq, np.int32)
        _check(lib().qadc_scan_i8(self._h, nq, ma, _p(assign, i32p), _p(qt, i8p), R, _p(keys, u32p), _p(vals, i8p),
                                  _p(sizes, i32p)))
        return self._heaps(nq, R, keys, vals, sizes)

    def scan_i8_candidates(self, assign, qtables, R, capacity=1 << 20):
        assign = self._prep(assign)
        nq, ma = assign.shape
        qt = np.ascontiguousarray(qtables, np.int8)
        ck = np.zeros(capacity, np.uint32)
        cv = np.zeros(capacity, np.int8)
        off = np.zeros(nq + 1, np.uint64)
        _check(lib().qadc_scan_i8_candidates(self._h, nq, ma, _p(assign, i32p), _p(qt, i8p), R, capacity,
                                             _p(ck, u32p), _p(cv, i8p), _p(off, u64p)))
        return [(ck[int(off[q]):int(off[q + 1])].copy(), cv[int(off[q]):int(off[q + 1])].copy()) for q in range(nq)]

    def query_scan_shard_streams(self, assign, tables, R, capacity=1 << 18):
        """submit + collect_candidates: this rank's ordered streams with assign slots (multi-GPU callers)."""
        self.submit(0, assign, tables, R)
        res = self.collect_candidates(0, capacity)
        total = int(res["offsets"][-1])
        return dict(keys=res["keys"][:total].copy(), vals=res["vals"][:total].copy(), slots=res["slots"][:total].copy(),
                    offsets=res["offsets"].copy(), status=res["status"], qmin=res["qmin"], qmax=res["qmax"])

    def query_scan_candidates(self, assign, tables, R, capacity=1 << 20):
        assign = self._prep(assign)
        nq, ma = assign.shape
        ck = np.zeros(capacity, np.uint32)
        cv = np.zeros(capacity, np.int8)
        off = np.zeros(nq + 1, np.uint64)
        status = np.zeros(nq, np.int32)
        qmin = np.zeros(nq, np.float32)
        qmax = np.zeros(nq, np.float32)
        _check(lib().qadc_query_scan_candidates(self._h, nq, ma, _p(assign, i32p), _p(tables, f32p), R, capacity,
                                                _p(ck, u32p), _p(cv, i8p), _p(off, u64p), _p(status, i32p),
                                                _p(qmin, f32p), _p(qmax, f32p)))
        streams = [(ck[int(off[q]):int(off[q + 1])].copy(), cv[int(off[q]):int(off[q + 1])].copy())
                   for q in range(nq)]
        return dict(streams=streams, status=status, qmin=qmin, qmax=qmax)

    def scan_start(self, assign, tables, R):
        assign = self._prep(assign)
        nq, ma = assign.shape
        tb = np.ascontiguousarray(tables, np.float32)
        qmax = np.zeros(nq, np.float32)
        _check(lib().qadc_scan_start(self._h, nq, ma, _p(assign, i32p), _p(tb, f32p), R, _p(qmax, f32p)))
        return qmax

    def candidates_i8(self, part, qtable):
        qt = np.ascontiguousarray(qtable, np.int8)
        out = np.zeros(self.partition_size(part), np.int8)
        _check(lib().qadc_candidates_i8(self._h, part, _p(qt, i8p), _p(out, i8p)))
        return out

    def float_top1(self, part, table):
        tb = np.ascontiguousarray(table, np.float32).reshape(-1)
        key, pos, dist = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
        _check(lib().qadc_float_top1(self._h, part, _p(tb, f32p), C.byref(key), C.byref(pos), C.byref(dist)))
        return key.value, pos.value, dist.value

    def profile(self):
        pr = Profile()
        _check(lib().qadc_profile_read(self._h, C.byref(pr)))
        return {f: getattr(pr, f) for f, _ in Profile._fields_}

    def profile_reset(self):
        _check(lib().qadc_profile_reset(self._h))
