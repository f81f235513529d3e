// Host side of libqadc_hip.so: the C-ABI declared in include/qadc.h.
//
// Mirrors scanner_4 (db_query_4.cpp:73-310): prepare (upload partitions, starts sizes) and
// query_scan (float pre-scan -> qmax, qmin/clamp, int8 quantization, scan of every probed
// partition in assign order into one heap).  All per-query arithmetic runs on the GPU in one
// stream-ordered chain (no host round trip between pre-scan, quantizer, scan and the ordering of the
// candidates); the host only plans the work items and replays the returned, already ordered candidate
// streams through a heap with the reference's push semantics (host/qadc_heap.hpp).
//
// The product path never touches oracle/: if the HIP runtime or the GPU is missing every entry
// point fails loudly with QADC_E_HIP.
#include "../../include/qadc.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../host/qadc_heap.hpp"
#include "../host/worker_pool.hpp"
#include "qadc_kernels.h"

using namespace qadc;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(QADC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n, unsigned flags = hipHostMallocDefault) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T), flags);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Part {
    uint8_t* d_codes = nullptr;    // row-major codes of the local range
    uint32_t* d_labels = nullptr;  // labels of the local range (or null)
    uint8_t* d_starts = nullptr;   // replica of the global partition's first codes (null: d_codes, first_pos == 0)
    uint32_t n = 0;                // codes held here
    uint32_t global_n = 0;         // codes of the whole partition (== n unless sharded)
    uint32_t first_pos = 0;        // global position of local code 0
    uint32_t starts_cap = 0;       // codes available for the pre-scan
    uint32_t start_n = 0;
    uint32_t key_base = 0;
    bool own = true;
};

constexpr int kSlots = 8;   // batches in flight: one being collected, one scanning, the others queued behind it with their fronts running
                            // ahead.  Three or four cover every loop measured so far (deeper pipelines of the multi-GPU IVF loop — six,
                            // eight batches — were tried and are no faster: its batches share the GPU, they do not wait for it)

struct LevelLaunch {
    size_t first;   // first item
    int nitems;
    int wgs;
    uint64_t codes;
    bool small;     // small-run kernel (runs below idx->small_run codes)
    bool shared;    // every run of the launch covers the same codes (one run per query): sibling-major launch
    bool mq;        // ... and groups of 8 of them share one pass (scan_i8_mq_kernel)
    uint64_t maxn;  // longest run of the launch
    bool early;     // launched on the front stream, under the previous batch's long levels: counted, not event-timed
    int ev = -1;    // index of the HIP event recorded before the launch (the next one follows it), -1 = not timed
};

struct Slot {
    bool busy = false;
    bool has_result = false;            // collected streams not yet handed to the caller (capacity retry)
    bool float_path = false;
    int nq = 0, ma = 0, R = 0;
    float* tables = nullptr;            // caller's float tables (float path)
    std::vector<int32_t> assign;
    std::vector<int8_t> qtables_in;     // int8 path input copy
    uint32_t cap_q = 0;                 // candidate region entries per query
    uint32_t out_cap = 0;               // entries of the device-sorted output

    // one upload block per batch: [ScanItem items][StartItem starts][u32 fc_init[2 nq]][float or int8 tables]
    DevBuf<unsigned char> d_in;
    PinBuf<unsigned char> h_in;
    ScanItem* d_items = nullptr;
    StartItem* d_sitems = nullptr;
    uint32_t* d_fc_init = nullptr;
    float* d_ftables_in = nullptr;      // float tables inside d_in (host-table path)
    // one state block, cleared by ONE memset: [CandHeader (64 B)][QueryState[nq]]
    DevBuf<unsigned char> d_state;
    CandHeader* d_hdr = nullptr;
    QueryState* d_qs = nullptr;
    // one result block in pinned, device-mapped HOST memory: [QueryOut[nq]][u64 entries[out_cap]].  The ordering
    // kernel stores into it directly; there is no result copy (see plan_and_launch)
    PinBuf<unsigned char> h_result;
    unsigned char* h_result_mapped = nullptr;   // the allocation d_result_mapped was looked up for
    unsigned char* d_result_mapped = nullptr;
    QueryOut* d_qout = nullptr;         // device-side addresses of h_qout / h_entries
    uint64_t* d_entries = nullptr;
    const int8_t* d_qt = nullptr;       // int8 tables the scan reads (d_qtables, or the uploaded ones)
    // device-side heap replay (large batches): the ordered stream also stays in device memory, one wave per query
    // pushes it through the reference's heap, and the host block receives [heaps u64[nq][R]][sizes u32[nq]] as well
    bool dev_replay = false;
    DevBuf<uint64_t> d_stream;
    uint64_t* h_heaps = nullptr;
    uint32_t* h_heap_sizes = nullptr;
    bool heaps_ready = false;           // a replay kernel ran for the batch: h_heaps / h_heap_sizes hold its result
    bool rerun = false;                 // the batch is being re-run from inside a collect call (no merge is enqueued with it)
    // Front sharded over the ranks of the multi-GPU merge (qadc_search batches): this rank ran feeders + pre-scan + quantizer
    // for queries [front_q0, front_q0 + front_n) only; the int8 tables, assign[] and (flags, qmin, qmax) of ALL queries come
    // from one all-gather and the scan takes them like an int8 batch.
    bool front_sharded = false;
    int front_q0 = 0, front_n = 0, front_per = 0;
    DevBuf<unsigned char> d_fblock, d_fgathered;
    DevBuf<uint32_t> d_front_all;       // [nq][4] gathered {flags, qmin, qmax, 0}
    PinBuf<unsigned char> h_fmap;       // mapped: assign i32[nq][ma], then front u32[nq][4] (written by front_unpack_kernel)
    unsigned char* d_fmap = nullptr;
    unsigned char* h_fmap_mapped = nullptr;
    hipEvent_t ev_fa = nullptr, ev_fb = nullptr;
    bool skipped_streams = false;       // collect_common left device-replayed queries' streams unassembled
    QueryOut* h_qout = nullptr;
    uint64_t* h_entries = nullptr;
    DevBuf<float> d_ftables;            // float tables built on the device (qadc_search)
    DevBuf<int8_t> d_qtables;
    DevBuf<Cand> d_cands;
    bool wgq_grouped = false;           // the batch took the partition-major second phase
    int group_head_slots = 0;           // ... after a head of this many local probes per query
    DevBuf<float> d_fc;

    // one-workgroup-per-query path (qadc_query_kernel.hip): no planner, no levels, no sort
    bool wgq = false;
    bool dist_batch = false;            // launched with the native multi-GPU merge active: streams kept in device memory
    uint32_t wgq_cap = 0;               // stream entries per query workgroup (regrown on overflow)
    bool poll = false;                  // collect watches the workgroups' done bits instead of the completion event
    int wgq_G = 1;                      // workgroups per query (small batches: the scan order of a query is split)
    uint64_t wgq_codes = 0;             // codes a query probes (exact maximum, or an estimate) — sizes wgq_G
    uint64_t head_codes = 0;            // level path: codes of every query's scan order covered by the head launch (0 = none)
    uint64_t wgq_fcap = 0;              // pre-scan values per query in the global scratch (0 = they fit LDS)
    DevBuf<uint32_t> d_qflags;          // [nq][4]: {flags, entries} for replay_heap_lanes_kernel
    DevBuf<float> d_fvals;
    DevBuf<QCand> d_qcands;             // unordered candidates of the query workgroups (scratch)
    PinBuf<uint64_t> h_fetch;           // streams fetched on demand when they were left in device memory
    bool assign_on_device = false;      // qadc_search: assign[] was produced on the GPU and copied back asynchronously
    hipEvent_t ev_assign = nullptr;
    bool full_prescan = false;          // survivor buffer overflowed: pre-scan everything unfiltered
    // sharded pre-scan (multi-GPU): mode 1 = pre-scan ONLY, of slice pre_slice of pre_nslices of every probed
    // partition's starts, exporting the R smallest values per query; mode 2 = a full batch whose pre-scan is
    // replaced by the gathered values inj_vals[nq][inj_n] of all ranks
    int mode = 0;
    int pre_slice = 0, pre_nslices = 1;
    std::vector<float> inj_vals;
    uint32_t inj_n = 0;
    float* h_export = nullptr;          // mode 1 results in the pinned result block: [nq][R] floats, then [nq] flags
    uint32_t* h_export_flags = nullptr;
    PinBuf<Cand> h_cands;               // host-sort fallback only
    // device-side feeders (qadc_search): queries in, tables never leave the GPU
    bool device_tables = false;
    DevBuf<float> d_queries;
    DevBuf<int32_t> d_assign;
    DevBuf<float> d_cdist;
    PinBuf<float> h_queries;
    PinBuf<int32_t> h_assign;
    hipEvent_t ev_feed = nullptr;

    std::vector<LevelLaunch> launches;
    uint64_t start_codes = 0;
    hipEvent_t ev_done = nullptr;
    hipEvent_t ev_front = nullptr;      // pre-scan + quantizer finished (front stream)
    hipEvent_t ev_up = nullptr;         // this batch's upload finished (copy stream)
    hipEvent_t ev_scanned = nullptr;    // last scan level finished (main stream)
    std::vector<hipEvent_t> prof_ev;    // pairs: [2i] before, [2i+1] after; pair 0 = pre-scan chain
    size_t prof_used = 0;

    // collect() results: ordered candidate streams, entry = key | value << 32 | assign slot << 40
    std::vector<uint64_t> out_entries;
    std::vector<uint64_t> out_off;
};

// ---- native multi-GPU merge (qadc_dist_*): RCCL through dlopen, so that the library has no link-time dependency
// on it and a single-GPU user never loads it ----
struct QadcNcclId { char internal[128]; };                  // layout of ncclUniqueId (rccl.h)
// A merge enqueued together with its batch (qadc_dist_*; one-workgroup-per-query batches without an extra payload): pack,
// all-gather, interleave and replay follow the scan on the merge's stream with no host in between, so the collect call
// only waits for one event.  One set of buffers per submission slot (several batches are in flight).
struct DistSlot {
    DevBuf<uint64_t> d_block, d_gathered, d_merged, d_moff;
    DevBuf<uint32_t> d_mcnt;
    DevBuf<uint32_t> d_src;                                  // level-path batches: {offset, count, flags}[nq] of the ordered streams
    PinBuf<unsigned char> h_out;                             // mapped: heaps u64[nq][R], sizes u32[nq], status u32[4]
    unsigned char* d_out = nullptr;
    unsigned char* h_out_mapped = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr, ev_gathered = nullptr;
    bool enqueued = false;
    bool pending = false;                                    // scan enqueued, merge not yet: it is issued BEHIND the next batch's front
    uint64_t seq = 0;                                        // gather (flush_merges), so that that gather never waits for this batch's scan
    void release() {
        d_block.release(); d_gathered.release(); d_merged.release(); d_moff.release(); d_mcnt.release(); h_out.release();
        d_src.release();
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (ev_gathered) (void)hipEventDestroy(ev_gathered);
        ev_ready = ev_done = ev_gathered = nullptr;
    }
};

struct DistState {
    void* lib = nullptr;
    int (*GetUniqueId)(QadcNcclId*) = nullptr;
    int (*CommInitRank)(void**, int, QadcNcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* comm = nullptr;
    qadc_allgather_fn user_fn = nullptr;                     // qadc_dist_init_transport: the caller's all-gather instead of RCCL
    void* user_ctx = nullptr;
    hipStream_t stream = nullptr;                            // own high-priority stream: the merge of batch s must not queue
                                                             // behind the scan kernels of batches s+1.. on the main stream
    hipStream_t merge_stream[kSlots] = {};                   // interleave + replay of a merge enqueued with its batch: a millisecond of
                                                             // latency that must neither sit in front of the NEXT batch's collectives
                                                             // nor behind the PREVIOUS batch's replay (one stream per slot)
    int rank = 0, world = 1;
    uint32_t cap_entries = 1u << 16;                         // entries per rank block; regrown (by every rank alike) on overflow
    DevBuf<uint64_t> d_block, d_gathered;
    DevBuf<uint32_t> d_src;                                  // [3][nq]: offset, count, flags of this rank's streams
    PinBuf<uint32_t> h_src;
    DevBuf<uint64_t> d_moff, d_merged;                       // merge scratch: per-query offsets, the world's streams in global scan order
    DevBuf<uint32_t> d_mcnt;                                 // [2][nq]: merged entries per query, replay flags
    DevBuf<uint64_t> d_fix;                                  // streams of the queries this rank had to order on the host
    PinBuf<uint64_t> h_fix;
    DevBuf<float> d_extra;
    DevBuf<uint64_t> d_extra_all;                            // the payload of a batch whose merge was enqueued with it, gathered [world][w]
    PinBuf<float> h_extra;
    PinBuf<unsigned char> h_out;                             // mapped: heaps u64[nq][R], sizes u32[nq]
    unsigned char* d_out = nullptr;
    unsigned char* h_out_mapped = nullptr;
    PinBuf<uint32_t> h_hdr;                                  // gathered headers [world][nq][4]
    PinBuf<float> h_extra_all;                               // gathered extra payload [world][extra_n]
    // few-query batches: the gathered streams come back to the host, rank r replays queries q = r (mod world) there,
    // and a second, tiny all-gather shares the heaps (a lane-per-query device replay of ~10^4 sequential pushes per
    // query would take milliseconds when only a few dozen lanes have work)
    PinBuf<uint64_t> h_gathered;
    PinBuf<uint64_t> h_myheaps;                              // [per][R + 1]: heap entries, then the size
    DevBuf<uint64_t> d_myheaps, d_allheaps;
    PinBuf<uint64_t> h_allheaps;
    int device_nq = 256;                                     // batches of at least this many queries replay on the device
    int inject_failure = 0;                                  // test hook: the next qadc_dist_collect of this rank fails locally
    int async_merge = 1;                                     // enqueue the merge with the batch where possible (option "dist_async")
    uint64_t next_seq = 1;
    int shard_front = 1;                                     // qadc_search batches: every rank runs the front of 1/world of the queries (option "dist_shard_front")
    DistSlot slot[kSlots];
    // One all-gather of `words` u64 per rank on `st`: RCCL (enqueued, stream-ordered) or the caller's transport
    // (complete on return).  0 = ok; else the message is in `err`.
    int gather(const void* d_send, void* d_recv, size_t words, hipStream_t st, std::string& err) {
        if (user_fn) {
            const int rc = user_fn(user_ctx, d_send, d_recv, (uint64_t)words * sizeof(uint64_t), st);
            if (rc != 0) err = "the transport's all-gather failed (code " + std::to_string(rc) + ")";
            return rc;
        }
        const int rc = AllGather(d_send, d_recv, words, /*ncclUint64*/ 5, comm, st);
        if (rc != 0) err = std::string("ncclAllGather: ") + (GetErrorString ? GetErrorString(rc) : "error");
        return rc;
    }
};

int load_rccl(DistState& d, std::string& err) {
    if (d.lib) return 0;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        d.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (d.lib) break;
    }
    if (!d.lib) { err = std::string("cannot load RCCL: ") + dlerror(); return -1; }
    d.GetUniqueId = reinterpret_cast<int (*)(QadcNcclId*)>(dlsym(d.lib, "ncclGetUniqueId"));
    d.CommInitRank = reinterpret_cast<int (*)(void**, int, QadcNcclId, int)>(dlsym(d.lib, "ncclCommInitRank"));
    d.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(d.lib, "ncclAllGather"));
    d.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(d.lib, "ncclCommDestroy"));
    d.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(d.lib, "ncclGetErrorString"));
    if (!d.GetUniqueId || !d.CommInitRank || !d.AllGather || !d.CommDestroy) { err = "RCCL lacks an expected symbol"; return -1; }
    return 0;
}

}  // namespace

struct qadc_index {
    int M = 16, cs = 8, device = 0;
    hipStream_t stream = nullptr;
    hipStream_t front_stream = nullptr; // a batch's pre-scan/quantize chain, under the previous batch's streaming launches
    hipStream_t copy_stream = nullptr;  // uploads and on-demand copies: issued where they depend on nothing (see plan_and_launch)
    uint32_t replay_seq = 0;            // device replays alternate between two side streams in submission order
    hipStream_t sort_stream = nullptr;  // candidate ordering of batch s (stores into pinned host memory) overlaps batch s+1
    std::vector<Part> parts;
    int labeled = -1;  // -1 unknown, 0 flat, 1 labels
    bool finalized = false;
    float keep = 0.01f;
    // options
    int quant_mode = 1;
    uint32_t cand_capacity = kSortCap;  // candidate region entries per query
    uint64_t level_base = 512;
    uint64_t level_growth = 4;
    int wgs_per_item = 0;  // 0 = auto
    int share_variant = 0x41;            // streaming-kernel variant for shared launches: sibling-major, U=2, cached loads
    uint32_t share_codes_per_wg = 1u << 20;
    int mq = 1;                          // shared launches use the 8-queries-per-pass kernel
    int device_replay_nq = 64;           // batches of at least this many queries replay their streams on the device (0 = never)
    int device_replay_alone_nq = 512;    // ... a batch with nothing else in flight (a synchronous call): from this many
    uint64_t front_run_max = 2u << 20;   // leading levels whose runs are at most this long join the front (0 = none); they are
                                         // counted with the small launches, not event-timed.  125M x 32: 2 Mi -4 %, 8 Mi +1 %
    uint64_t front_min_batch = 3000000000ull;   // ... in batches of at least this many (code, query) pairs: under a last level
                                         // of a few hundred microseconds the early levels hide; under a shorter one they only make the front
                                         // stream the longest chain of the step (32 queries per step: 3e7 codes 0.42 -> 0.37 ms with
                                         // everything on the main stream, 6e7 0.64 -> 0.62, 1.25e8 1.12 -> 1.14: tools/front_run_ab.sh)
    int prescan_mq = 1;                  // ... and so does the float pre-scan when every query pre-scans the same starts
    uint32_t mq_codes_per_wg = 1u << 16;
    uint32_t mq_min_wgs = 4096;          // workgroups a multi-query launch should have at least (2 rounds of the chip)
    uint32_t mq_min_tiles = 4;           // ... but never fewer than this many 4 KiB tiles per workgroup
    int front_dist = 1;    // early levels also for the multi-GPU loop's batches (pre-scan injected)
    uint32_t wgq_split_codes = 8192;    // a query is split over several workgroups only down to this many codes each
    int wgq_poll = 1;      // ... and its completion is read from the result block, not from the event
    int group_strikes = 0; // grouped batches whose candidate regions overflowed (data whose later probes fall below the head's bound)
    int wgq_group = 1;     // partition-major second phase for large IVF batches: 0 never, 1 auto, 2 whenever possible
    int wgq_group_head = 3;   // ... after a head of this many probes per query (one workgroup per query; 4 until the ordering pass took 8192 candidates)
    int wgq_group_head_dist = 4;   // ... under the multi-GPU merge (probes with codes on this rank; option "wgq_group_head_dist")
    int wgq_inline = 1;    // a lone small query's input rides in the kernel arguments (no upload)
    int mq_narrow = 1;     // IVF second phase: groups whose upper four seats are empty run the 4-seat form (the two-body build of the kernel)
    int replay_wave = 1;   // device replay of the query kernel's streams: 1 = one wave per query (heap in registers), 0 = one lane per query
                           // (C3 shape, 1024-query batches: lanes 0.78 us per query, waves 0.75; C5 shape: 4.83 vs 4.67 — since
                           // the wave heap sifts all levels at once; with its element-by-element sift the waves lost,
                           // 0.93 vs 0.80.  The multi-GPU merge replays by waves.)
    int head_early = 1;    // the head launch joins the front stream (under the previous batch's long levels)
    int overlap_front = 1; // pre-scan chain of batch s+1 on its own stream, under batch s's scan
    uint32_t prescan_sample = 1u << 16;  // starts pre-scanned unfiltered before the survivor filter kicks in
    int replay_threads = 0;            // 0 = auto
    WorkerPool pool;                   // host replay workers (started on first use)
    uint32_t small_vec_per_wg = 512;  // 16-byte vectors one small-run workgroup covers
    uint32_t small_run = 1u << 17;  // runs shorter than this use the small-run kernel
    int variant = 0x0d;    // kernel tuning variant (see launch_scan_i8): U=2, non-temporal loads, chunked tiles
    // one workgroup per query (IVF batches, small lists): 0 = never, 1 = auto, 2 = whenever structurally possible
    int wgq = 1;
    int wgq_min_nq = 128;                // auto: batches of at least this many queries ...
    uint64_t wgq_max_codes = 1ull << 24; //   ... probing at most this many codes per query (estimate), or
    uint64_t wgq_small_codes = 1ull << 18;   // any batch probing at most this many codes per query
    uint32_t wgq_capacity = 4096;        // stream entries per query to start with
    int wgq_split = 12;                 // workgroups a small batch may spread one query's scan order over
    int head_level = 5;                  // level path: bound levels 0..head_level-1 (the first 512 Ki codes of every query) are
                                         // scanned by ONE launch of the query kernel in head mode instead of head_level dependent
                                         // level launches (0 = off): -5 % per step on a 125M-code shard, neutral at 1B
    int wgq_variant = 0;                 // kernel tuning variant (launch_scan_query)
    int table_form = 2;                  // float tables of qadc_search: 0 direct, 1 BLAS expansion, 2 the reference's nns_engine rule
    uint32_t wgq_cand_cap = kQueryCandCap;   // candidates per query before the batch falls back to the level path (test knob)
    uint32_t wgq_group_cand_cap = kOrderCandCap;   // ... of a batch with the partition-major second phase (option "wgq_group_cand_cap")
    DevBuf<PartDesc> d_partdesc;         // device partition table (qadc_index_finalize)
    std::vector<PartDesc> h_partdesc;    // its host copy (a lone small query carries the descriptors it needs in its launch)
    uint32_t max_start_n = 0;
    uint64_t total_codes = 0;
    uint32_t max_part_n = 0;
    bool profile = false;
    // N1: host feeders on the device
    int dim = 0;                 // vector dimension (0 = qadc_index_set_pq not called)
    DevBuf<float> d_codebooks;   // [M][16][dim/M]
    DevBuf<float> d_rotation;    // [dim][dim] OPQ rotation (empty = plain PQ)
    bool has_rotation = false;
    int K = 0;                   // coarse centroids (0 = flat)
    DevBuf<float> d_coarse;      // [K][dim]
    Slot slot[kSlots];
    Slot pre_slot[2];                   // sharded pre-scan passes (mode 1): own buffers, so that one can run
                                        // while slot[i] still holds an uncollected batch
    qadc_profile prof{};
    DistState* dist = nullptr;          // qadc_dist_init
};

namespace {

// Which float-table form qadc_search builds (option "table_form"): 0 = direct ||x - c||^2 always
// (compute_dists_single_simd_cg, distances.hpp:294-311), 1 = BLAS expansion always (nns_engine_batch,
// query_common.hpp:194-213), 2 = the rule of nns_engine (query_common.hpp:292-297): direct for ma == 1, expansion otherwise.
int table_expansion(const qadc_index* idx, int ma);

int use_device(const qadc_index* idx) {
    HIPCHECK(hipSetDevice(idx->device));
    return QADC_OK;
}

int table_expansion(const qadc_index* idx, int ma) {
    return idx->table_form == 1 || (idx->table_form == 2 && ma > 1);
}

hipError_t prof_event(Slot& s, hipStream_t st) {
    if (s.prof_used == s.prof_ev.size()) {
        hipEvent_t e;
        hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess) return r;
        s.prof_ev.push_back(e);
    }
    return hipEventRecord(s.prof_ev[s.prof_used++], st);
}

// Level boundaries in the concatenated scan position space of one query.
void level_bounds(const qadc_index* idx, uint64_t* L) {
    L[0] = 0;
    uint64_t b = std::max<uint64_t>(idx->level_base, 16);
    for (int k = 1; k < kMaxLevels; ++k) {
        L[k] = b;
        b = (b > (UINT64_MAX >> 8)) ? UINT64_MAX : b * std::max<uint64_t>(idx->level_growth, 2);
    }
    L[kMaxLevels] = UINT64_MAX;
}

struct ScopedMs {
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedMs(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedMs() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// What the planner hands to the launcher: the work items of a batch in upload order.
struct BatchPlan {
    std::vector<ScanItem> all_items;             // runs, grouped by bound level (Slot::launches indexes into it)
    std::vector<StartItem> sitems_a, sitems_b;   // pre-scan: phase A = unfiltered sample, phase B = filtered remainder
    std::vector<uint32_t> fc_init;               // per query: {sample values, capacity} of its pre-scan buffer
    uint64_t fc_stride = 1;
};

// Host planning of one batch: cuts every query's scan order into bound levels, emits the runs (ScanItem) and the
// pre-scan items (StartItem), and decides kernel and grid per level launch (Slot::launches).  No GPU calls.
int plan_batch(qadc_index* idx, Slot& s, BatchPlan& plan) {
    const int M = idx->M, cs = idx->cs, nq = s.nq, ma = s.ma;
    const uint32_t cpl = 16 / cs;
    uint64_t L[kMaxLevels + 1];
    level_bounds(idx, L);
    std::vector<std::vector<ScanItem>> per_level(kMaxLevels);
    const int k0 = (s.mode != 1 && idx->head_level > 0) ? idx->head_level : 0;   // levels < k0 belong to the head launch
    s.head_codes = k0 ? L[k0] : 0;
    std::vector<StartItem>& sitems_a = plan.sitems_a;
    std::vector<StartItem>& sitems_b = plan.sitems_b;
    std::vector<uint32_t>& fc_init = plan.fc_init;
    uint64_t& fc_stride = plan.fc_stride;
    fc_init.assign(2 * (size_t)nq, 0);
    fc_stride = 1;
    s.start_codes = 0;
    for (int q = 0; q < nq; ++q) {
        uint64_t c = 0;
        uint64_t stotal = 0;
        // the starts of a partition this call pre-scans: all of them, or (mode 1) this rank's slice, cut at
        // multiples of 16 codes so that every slice starts on a 16-byte boundary; (mode 2) none
        auto starts_range = [&](const Part& pt, uint64_t& lo, uint64_t& len) {
            lo = 0;
            len = s.mode == 2 ? 0 : pt.start_n;
            if (s.mode == 1 && s.pre_nslices > 1) {
                lo = ((uint64_t)pt.start_n * s.pre_slice / s.pre_nslices) & ~15ull;
                const uint64_t hi = s.pre_slice + 1 == s.pre_nslices
                                        ? pt.start_n : (((uint64_t)pt.start_n * (s.pre_slice + 1) / s.pre_nslices) & ~15ull);
                len = hi > lo ? hi - lo : 0;
            }
        };
        if (s.float_path)
            for (int a = 0; a < ma; ++a) {
                const int p = s.assign[(size_t)q * ma + a];
                if (p >= 0 && p < (int)idx->parts.size()) {
                    uint64_t lo, len;
                    starts_range(idx->parts[p], lo, len);
                    stotal += len;
                }
            }
        // two-phase pre-scan only pays (and is only needed) when the starts are many
        uint64_t sample = (s.full_prescan || stotal <= 2ull * idx->prescan_sample) ? stotal : idx->prescan_sample;
        uint64_t soff = 0;
        for (int a = 0; a < ma; ++a) {
            const int p = s.assign[(size_t)q * ma + a];
            if (p < 0 || p >= (int)idx->parts.size())
                return fail(QADC_E_ARG, "assign[] names a partition that does not exist");
            const Part& pt = idx->parts[p];
            if (pt.global_n == 0) continue;  // empty partition: db_query_4.cpp:291-293
            uint64_t slo = 0, slen = 0;
            if (s.float_path) starts_range(pt, slo, slen);
            if (slen) {
                const uint8_t* sc = (pt.d_starts ? pt.d_starts : pt.d_codes) + slo * cs;
                const uint64_t in_a = soff < sample ? std::min<uint64_t>(slen, sample - soff) : 0;
                StartItem si;
                si.table = (uint32_t)((size_t)q * ma + a);
                si.query = (uint32_t)q;
                if (in_a) {
                    si.codes = sc;
                    si.n = (uint32_t)in_a;
                    si.out_off = (uint32_t)soff;
                    si.filter = 0;
                    sitems_a.push_back(si);
                }
                if (in_a < slen) {
                    si.codes = sc + in_a * cs;
                    si.n = (uint32_t)(slen - in_a);
                    si.out_off = 0;
                    si.filter = 1;
                    sitems_b.push_back(si);
                }
                soff += slen;
                s.start_codes += slen;
            }
            if (pt.n == 0 || s.mode == 1) continue;   // no codes of the partition here (only its starts replica) / pre-scan only
            uint64_t prev = 0;
            for (int k = 0; k < kMaxLevels && prev < pt.n; ++k) {
                uint64_t cut = pt.n;
                if (L[k + 1] < c + pt.n) {
                    cut = L[k + 1] > c ? L[k + 1] - c : 0;
                    cut -= cut % cpl;  // keep every run 16-byte aligned
                }
                if (cut <= prev) continue;
                if (k < k0) {                                 // scanned by the head launch (same cut: scan_query_kernel, HEAD)
                    prev = cut;
                    continue;
                }
                // runs longer than 2^31 codes are cut so that 32-bit vector indices cannot wrap
                for (uint64_t b0 = prev; b0 < cut;) {
                    const uint64_t len = std::min<uint64_t>(cut - b0, 1ull << 31);
                    ScanItem it;
                    it.codes = pt.d_codes + b0 * cs;
                    it.labels = pt.d_labels;
                    it.n = (uint32_t)len;
                    it.pos0 = (uint32_t)b0;
                    it.key_base = pt.key_base + pt.first_pos;
                    it.table = (uint32_t)((size_t)q * ma + a);
                    it.query = (uint32_t)q;
                    it.order = ((uint32_t)k << 16) | (uint32_t)a;
                    // padding-lane replay of the partition's last code (simd_layout.hpp:46-50, simd_scan.hpp:67)
                    it.dup_pos = (pt.first_pos + pt.n == pt.global_n) ? pt.n - 1u : 0xffffffffu;
                    it.dup_reps = (16u - pt.global_n % 16u) % 16u;
                    per_level[k].push_back(it);
                    b0 += len;
                }
                prev = cut;
            }
            c += pt.n;
        }
        // survivors of the filter: expected R * stotal / sample; 16x head-room, the overflow flag catches the rest
        uint64_t cap = sample;
        if (sample < stotal)
            cap += std::min<uint64_t>(stotal - sample, std::max<uint64_t>(16ull * s.R * ((stotal + sample - 1) / sample), 4096));
        if (s.mode == 2) sample = cap = s.inj_n;              // the gathered values are the whole "pre-scan output"
        fc_init[2 * q] = (uint32_t)sample;
        fc_init[2 * q + 1] = (uint32_t)cap;
        fc_stride = std::max<uint64_t>(fc_stride, cap);
    }
    size_t nitems = 0;
    for (auto& v : per_level) nitems += v.size();
    std::vector<ScanItem>& all_items = plan.all_items;
    all_items.assign(nitems, ScanItem());
    s.launches.clear();
    size_t off = 0;
    const int wgs_cap = idx->wgs_per_item > 0 ? idx->wgs_per_item : (M == 16 ? 1024 : 512);   // (r02 sweep: 1024 reaches the streaming ceiling of the "probe" variant, 512 is 1.6 % below)
    for (int k = 0; k < kMaxLevels; ++k) {
        if (per_level[k].empty()) continue;
        // one launch for the short runs of the level, one for the long ones
        for (int small = 1; small >= 0; --small) {
            uint64_t maxn = 0, codes = 0;
            size_t cnt = 0;
            bool same = true;
            for (auto& it : per_level[k]) {
                if ((it.n < idx->small_run) != (small == 1)) continue;
                if (cnt) {
                    const ScanItem& f = all_items[off];
                    same = same && it.codes == f.codes && it.n == f.n && it.pos0 == f.pos0 && it.labels == f.labels &&
                           it.key_base == f.key_base && it.dup_pos == f.dup_pos && it.dup_reps == f.dup_reps;
                }
                all_items[off + cnt++] = it;
                maxn = std::max<uint64_t>(maxn, it.n);
                codes += it.n;
            }
            if (!cnt) continue;
            const uint64_t nvec = (maxn + cpl - 1) / cpl;
            LevelLaunch ll;
            ll.first = off;
            ll.nitems = (int)cnt;
            ll.small = small == 1;
            ll.maxn = maxn;
            ll.early = false;
            ll.shared = !ll.small && same && cnt >= 2 && idx->share_variant != 0;
            ll.mq = ll.shared && idx->mq;
            if (ll.mq) {
                // 8 queries per pass (scan_i8_mq_kernel): 256-thread workgroups, ~64 Ki codes each, groups of 8
                // queries as L2-sharing siblings
                const uint64_t tiles = std::max<uint64_t>((nvec + 255) / 256, 1);
                uint64_t w = idx->wgs_per_item > 0 ? (uint64_t)idx->wgs_per_item
                                                   : (maxn + idx->mq_codes_per_wg - 1) / idx->mq_codes_per_wg;
                const uint64_t ngroups = (cnt + 7) / 8;
                w = std::max<uint64_t>(w, (idx->mq_min_wgs + ngroups - 1) / ngroups);   // >= 2 rounds of the 2048 resident workgroups
                w = std::min<uint64_t>(std::min<uint64_t>(w, 65536), std::max<uint64_t>(tiles / idx->mq_min_tiles, 1));
                if (w >= 8) w &= ~7ull;
                ll.wgs = (int)w;
            } else if (ll.shared) {
                // Queries of a batch over the same codes (flat database; IVF queries probing the same cell): the
                // sibling-major launch makes them share every tile through one XCD's L2, so the codes cross the
                // HBM interface about once per LAUNCH, not once per query, and the launch is bound by the LDS
                // lookup rate instead.  Workgroups per run: ~2M codes each (amortises the table build, leaves the
                // dispatcher room to balance), a multiple of 8 so that the XCD decode applies.
                uint64_t w = idx->wgs_per_item > 0 ? (uint64_t)idx->wgs_per_item
                                                   : (maxn + idx->share_codes_per_wg - 1) / idx->share_codes_per_wg;
                w = std::min<uint64_t>(std::max<uint64_t>(w, 64), 512);
                w = std::min<uint64_t>(w, std::max<uint64_t>((nvec + 4095) / 4096, 1));
                if (w >= 8) w &= ~7ull;
                ll.wgs = (int)w;
            } else if (ll.small) {
                // enough workgroups to fill the chip, but no more: each one pays a table build + bound fetch
                const uint64_t want = std::max<uint64_t>(1, 4096 / cnt);
                ll.wgs = (int)std::min<uint64_t>(std::max<uint64_t>((nvec + idx->small_vec_per_wg - 1) / idx->small_vec_per_wg, 1), want);
            }
            else {
                // each streaming workgroup builds a 64-128 KiB table: with many runs in the launch, give every
                // workgroup more tiles instead of more workgroups per run
                const uint64_t want = std::max<uint64_t>(1, 8192 / cnt);
                ll.wgs = (int)std::min<uint64_t>(std::max<uint64_t>((nvec + 4095) / 4096, 1), std::min<uint64_t>(wgs_cap, want));
            }
            ll.codes = codes;
            s.launches.push_back(ll);
            off += cnt;
        }
    }

    return QADC_OK;
}

int launch_wgq_batch(qadc_index* idx, Slot& s);
int enqueue_merge(qadc_index* idx, Slot& s, hipStream_t scan_stream);
int flush_merges(qadc_index* idx, uint64_t upto);

// Plans the batch in slot s and enqueues all of its GPU work (front, levels, ordering, optional device replay).
int plan_and_launch(qadc_index* idx, Slot& s) {
    if (s.wgq) return launch_wgq_batch(idx, s);
    ScopedMs timer(idx->prof.host_plan_ms);
    const int M = idx->M, cs = idx->cs, nq = s.nq, ma = s.ma;
    hipStream_t st = idx->stream;
    const size_t table_dim = (size_t)M * 16;
    BatchPlan plan;
    if (int rc = plan_batch(idx, s, plan)) return rc;
    const std::vector<ScanItem>& all_items = plan.all_items;
    const std::vector<StartItem>&sitems_a = plan.sitems_a, &sitems_b = plan.sitems_b;
    const std::vector<uint32_t>& fc_init = plan.fc_init;
    uint64_t fc_stride = plan.fc_stride;
    const size_t nitems = all_items.size();

    // ---- upload: ONE block, ONE copy ----------------------------------------------------------
    const size_t nt = (size_t)nq * ma * table_dim;
    const size_t na = sitems_a.size(), nb = sitems_b.size();
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t off_items = 0;
    const size_t off_sitems = align16(off_items + nitems * sizeof(ScanItem));
    const size_t off_init = align16(off_sitems + (na + nb) * sizeof(StartItem));
    const size_t off_tables = align16(off_init + fc_init.size() * sizeof(uint32_t));
    const size_t tables_bytes = s.float_path ? (s.device_tables ? 0 : nt * sizeof(float)) : (s.front_sharded ? 0 : nt);
    const size_t off_inj = align16(off_tables + tables_bytes);
    const size_t inj_bytes = s.mode == 2 ? sizeof(float) * (size_t)nq * s.inj_n : 0;
    const size_t off_hassign = align16(off_inj + inj_bytes);  // assign[] for the head launch (it walks the partition table itself)
    const size_t hassign_bytes = s.head_codes ? sizeof(int32_t) * (size_t)nq * ma : 0;
    const size_t in_bytes = align16(off_hassign + hassign_bytes);
    HIPCHECK(s.h_in.ensure(in_bytes));
    HIPCHECK(s.d_in.ensure(in_bytes));
    if (nitems) std::memcpy(s.h_in.p + off_items, all_items.data(), nitems * sizeof(ScanItem));
    if (na) std::memcpy(s.h_in.p + off_sitems, sitems_a.data(), na * sizeof(StartItem));
    if (nb) std::memcpy(s.h_in.p + off_sitems + na * sizeof(StartItem), sitems_b.data(), nb * sizeof(StartItem));
    std::memcpy(s.h_in.p + off_init, fc_init.data(), fc_init.size() * sizeof(uint32_t));
    if (s.float_path && !s.device_tables) std::memcpy(s.h_in.p + off_tables, s.tables, nt * sizeof(float));
    if (!s.float_path && !s.front_sharded) std::memcpy(s.h_in.p + off_tables, s.qtables_in.data(), nt);
    if (inj_bytes) std::memcpy(s.h_in.p + off_inj, s.inj_vals.data(), inj_bytes);
    if (hassign_bytes) std::memcpy(s.h_in.p + off_hassign, s.assign.data(), hassign_bytes);
    s.d_items = reinterpret_cast<ScanItem*>(s.d_in.p + off_items);
    s.d_sitems = reinterpret_cast<StartItem*>(s.d_in.p + off_sitems);
    s.d_fc_init = reinterpret_cast<uint32_t*>(s.d_in.p + off_init);
    s.d_ftables_in = reinterpret_cast<float*>(s.d_in.p + off_tables);

    // state block: [CandHeader, 64 B][QueryState[nq]]; result block: [QueryOut[nq]][u64 entries[out_cap]]
    const size_t state_bytes = 64 + sizeof(QueryState) * (size_t)nq;
    bool lone = s.mode != 1;                                    // nothing else in flight: a synchronous call (see launch_wgq_batch)
    for (int i = 0; i < kSlots; ++i) lone = lone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    lone = lone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    const int replay_from = lone ? std::max(idx->device_replay_nq, idx->device_replay_alone_nq) : idx->device_replay_nq;
    s.dev_replay = s.mode != 1 && idx->device_replay_nq > 0 && nq >= replay_from && s.R <= 4096;
    const size_t off_heaps = sizeof(QueryOut) * (size_t)nq + sizeof(uint64_t) * (size_t)s.out_cap;
    const size_t heaps_bytes = s.dev_replay ? (sizeof(uint64_t) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq : 0;
    const size_t result_bytes = std::max(off_heaps + heaps_bytes, (sizeof(float) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq);
    HIPCHECK(s.d_state.ensure(state_bytes));
    HIPCHECK(s.h_result.ensure(result_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    if (s.h_result.p != s.h_result_mapped) {               // (looked up once per allocation: the call costs ~0.1 ms)
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_result_mapped), s.h_result.p, 0));
        s.h_result_mapped = s.h_result.p;
    }
    unsigned char* d_result = s.d_result_mapped;
    s.d_hdr = reinterpret_cast<CandHeader*>(s.d_state.p);
    s.d_qs = reinterpret_cast<QueryState*>(s.d_state.p + 64);
    s.d_qout = reinterpret_cast<QueryOut*>(d_result);
    s.d_entries = reinterpret_cast<uint64_t*>(d_result + sizeof(QueryOut) * (size_t)nq);
    s.h_qout = reinterpret_cast<QueryOut*>(s.h_result.p);
    s.h_entries = reinterpret_cast<uint64_t*>(s.h_result.p + sizeof(QueryOut) * (size_t)nq);
    s.h_heaps = reinterpret_cast<uint64_t*>(s.h_result.p + off_heaps);
    s.h_heap_sizes = reinterpret_cast<uint32_t*>(s.h_result.p + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
    s.dist_batch = idx->dist != nullptr;
    const bool dev_stream = s.dev_replay || s.dist_batch;          // the native multi-GPU merge gathers from device memory
    if (dev_stream) HIPCHECK(s.d_stream.ensure(s.out_cap));
    s.h_export = reinterpret_cast<float*>(s.h_result.p);
    s.h_export_flags = reinterpret_cast<uint32_t*>(s.h_result.p + sizeof(float) * (size_t)s.R * nq);
    HIPCHECK(s.d_cands.ensure((size_t)nq * s.cap_q));
    HIPCHECK(s.d_qtables.ensure(nt));
    // The front of the batch (state clear, table build, float pre-scan, selects, quantizer) depends on nothing the
    // previous batch produces: it runs on its own high-priority stream, under that batch's streaming launches, and
    // the first scan level waits for it.  Its short single-workgroup selects are pure latency; hidden this way they
    // stop being a fixed cost per batch (which is what limits strong scaling when the per-GPU shard gets small).
    // A batch submitted while nothing else is in flight (a synchronous call, the first batch of a pipeline) has
    // nothing to overlap with: it runs front, levels and ordering on ONE stream, which spares it three cross-stream
    // event hops (~15 us of a ~130 us single query).
    bool alone = s.mode != 1;
    for (int i = 0; i < kSlots; ++i) alone = alone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    alone = alone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    hipStream_t main_stream = st;
    if ((idx->overlap_front && !alone) || s.mode == 1) st = idx->front_stream;
    HIPCHECK(hipMemsetAsync(s.d_state.p, 0, state_bytes, st));
    // The upload goes on the copy stream, where it depends on nothing (the slot's previous batch was collected),
    // and the main stream waits for it.  Issued on the main stream it would sit in the DMA engine's queue until the
    // PREVIOUS batch's kernels finish, and every other copy of the process (the caller's streams, the RCCL gather of
    // the multi-GPU merge) would queue behind it: copies wait in engine order, not stream order.
    if (alone) {
        HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, st));
    } else {
        HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, idx->copy_stream));
        if (!s.ev_up) HIPCHECK(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_up, idx->copy_stream));
        HIPCHECK(hipStreamWaitEvent(st, s.ev_up, 0));
    }
    s.prof_used = 0;

    s.d_qt = s.d_qtables.p;
    if (s.float_path) {
        float* d_ft = s.d_ftables_in;
        if (s.device_tables) {
            // residuals + float tables built on the GPU from the queries uploaded by search_submit
            HIPCHECK(s.d_ftables.ensure(nt));
            d_ft = s.d_ftables.p;
            HIPCHECK(hipStreamWaitEvent(st, s.ev_feed, 0));
            launch_build_tables(s.d_queries.p, idx->K ? idx->d_coarse.p : nullptr, s.d_assign.p, idx->d_codebooks.p,
                                idx->has_rotation ? idx->d_rotation.p : nullptr, nq, ma, M, idx->dim, table_expansion(idx, ma), d_ft, st);
        }
        HIPCHECK(s.d_fc.ensure((size_t)nq * fc_stride));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        auto wgs_for = [](const std::vector<StartItem>& v) {
            uint32_t maxs = 0;
            for (auto& si : v) maxs = std::max(maxs, si.n);
            return (int)std::min<uint32_t>(std::max<uint32_t>((maxs + 4095) / 4096, 1), 512);
        };
        const int tda = (int)(ma * table_dim);
        // phase A: the sample, unfiltered -> its R-th smallest; phase B: the rest, keeping only values <= that.
        // The LAST select of the chain also quantizes the query's tables (QuantizerMAX) in the same workgroup.
        // every query pre-scans the same starts (flat database, or one shared probe): 8 queries per pass
        auto shared_items = [&](const std::vector<StartItem>& v) {
            if (!idx->prescan_mq || v.size() < 2) return false;
            for (auto& si : v)
                if (si.codes != v[0].codes || si.n != v[0].n || si.out_off != v[0].out_off || si.filter != v[0].filter) return false;
            return true;
        };
        auto start_scan = [&](const std::vector<StartItem>& v, const StartItem* d_v) {
            if (shared_items(v))
                launch_start_scan_mq(M, d_v, (int)v.size(), std::min(2 * wgs_for(v), 1024), d_ft, s.d_fc.p, fc_stride, s.d_fc_init,
                                     s.d_qs, st);
            else
                launch_start_scan_f32(M, d_v, (int)v.size(), wgs_for(v), d_ft, s.d_fc.p, fc_stride, s.d_fc_init, s.d_qs, st);
        };
        if (na) start_scan(sitems_a, s.d_sitems);
        if (nb) {
            // (with a phase B the first select only has to bound the R-th smallest from above: 2 digit passes)
            launch_select_kth(s.d_fc.p, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 2, nullptr, nullptr, tda, 0, st);
            start_scan(sitems_b, s.d_sitems + na);
        }
        if (s.mode == 1) {
            // pre-scan only: the exact R-th smallest of this rank's slice, and the R smallest values themselves,
            // stored straight into the pinned result block; nothing else runs
            float* d_exp = reinterpret_cast<float*>(d_result);
            uint32_t* d_expf = reinterpret_cast<uint32_t*>(d_result + sizeof(float) * (size_t)s.R * nq);
            launch_select_kth(s.d_fc.p, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 4, nullptr, nullptr, tda, 0, st,
                              d_exp, d_expf);
            HIPCHECK(hipGetLastError());
            if (idx->profile) HIPCHECK(prof_event(s, st));
            if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_done, st));
            return QADC_OK;
        }
        const float* d_sel = s.d_fc.p;
        if (s.mode == 2) {
            // the ranks' gathered smallest values stand in for the pre-scan output
            d_sel = reinterpret_cast<const float*>(s.d_in.p + off_inj);
            fc_stride = s.inj_n;
            launch_prescan_minmax(d_sel, s.inj_n, nq, s.d_qs, st);
        }
        launch_select_kth(d_sel, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 4, d_ft, s.d_qtables.p, tda,
                          idx->quant_mode, st);
        if (idx->profile) HIPCHECK(prof_event(s, st));
    } else {
        s.d_qt = s.front_sharded ? s.d_qtables.p                             // (a sharded-front batch redone here: the gathered tables)
                                 : reinterpret_cast<const int8_t*>(s.d_in.p + off_tables);     // caller's int8 tables, as uploaded
        if (idx->profile) { HIPCHECK(prof_event(s, st)); HIPCHECK(prof_event(s, st)); }
    }

    // ---- scan levels ------------------------------------------------------------------------
    // a database that fits the 256 MiB Infinity Cache is re-read from it by every query: keep the default
    // cache policy there; non-temporal loads only pay for lists that stream from HBM anyway
    uint64_t db_bytes = 0;
    for (auto& p : idx->parts) db_bytes += (uint64_t)p.n * cs;
    const int variant = db_bytes <= (200ull << 20) ? (idx->variant & ~4) : idx->variant;
    auto launch_level = [&](LevelLaunch& ll, hipStream_t str) {
        if (ll.small)
            launch_scan_i8_small(M, s.d_items + ll.first, ll.nitems, ll.wgs, s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p,
                                 s.cap_q, (uint32_t)s.R, str);
        else if (ll.mq)
            launch_scan_i8_mq(M, s.d_items + ll.first, ll.nitems, ll.wgs, s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p, s.cap_q,
                              (uint32_t)s.R, str, /*narrow=*/0);   // (the 8-seat build: see scan_i8_mq_kernel)
        else
            launch_scan_i8(M, ll.shared ? idx->share_variant : (variant & ~64), s.d_items + ll.first, ll.nitems, ll.wgs,
                           s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p, s.cap_q, (uint32_t)s.R, str);
    };
    // The first levels of a batch are short launches in a dependent chain (each level's bound needs the previous
    // levels' candidates): latency, not work.  The head launch and the levels with short runs join the front — same
    // stream, after the quantizer — and so run under the previous batch's long levels instead of in front of this
    // batch's; only the long levels stay on the main stream.  Front-stream kernels only find room as workgroups of
    // the long launches retire, so a level with real work is slower there than on the main stream: runs of <= 2 Mi
    // codes go early (125M x 32 queries: 1.143 -> 1.10 ms per step; 1B x 32: -1 %), 8 Mi already costs more than it hides.
    auto launch_head = [&](hipStream_t str) -> int {
        // ONE launch scans the first head_codes codes of every query (bound levels 0..k0-1): every query's scan order
        // split over G workgroups that refresh their bound in LDS, instead of k0 dependent launches of a few
        // microseconds of work each.  It emits into the same candidate regions / level-0 histogram the levels use.
        QueryKernelArgs H{};
        H.parts = idx->d_partdesc.p;
        H.assign = reinterpret_cast<const int32_t*>(s.d_in.p + off_hassign);
        H.ma = ma;
        H.qtables = const_cast<int8_t*>(s.d_qt);
        H.R = (uint32_t)s.R;
        H.head_codes = s.head_codes;
        H.qstates = s.d_qs;
        H.cand_regions = s.d_cands.p;
        H.cand_cap = s.cap_q;
        H.hdr = s.d_hdr;
        H.nontemporal = 0;                                   // the queries of a batch share the head of a flat list through L2
        int G = std::min<int>(idx->wgq_split, std::max(1, 256 / nq));
        G = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)G, s.head_codes / 16384));
        H.G = G;
        HIPCHECK(launch_scan_query(M, idx->wgq_variant, nq, H, str));
        idx->prof.head_launches++;
        return QADC_OK;
    };
    size_t n_early = 0;
    uint64_t batch_codes = 0;
    for (auto& ll : s.launches) batch_codes += ll.codes;
    if (st != main_stream && (s.mode == 0 || (s.mode == 2 && idx->front_dist)) && batch_codes >= idx->front_min_batch)
        while (n_early < s.launches.size() && s.launches[n_early].maxn <= idx->front_run_max) ++n_early;
    if (n_early == s.launches.size() && n_early) --n_early;        // the last level closes the batch on the main stream
    // the head precedes every level: with the early levels (or on request) it joins the front as well
    bool head_pending = s.head_codes != 0;
    if (head_pending && st != main_stream && (n_early || idx->head_early)) {
        if (int rc = launch_head(st)) return rc;
        head_pending = false;
    }
    for (size_t li = 0; li < n_early; ++li) {
        s.launches[li].early = true;
        s.launches[li].ev = -1;
        launch_level(s.launches[li], st);
    }
    if (st != main_stream) {
        if (!s.ev_front) HIPCHECK(hipEventCreateWithFlags(&s.ev_front, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_front, st));
        st = main_stream;
        HIPCHECK(hipStreamWaitEvent(st, s.ev_front, 0));
    }
    if (head_pending)
        if (int rc = launch_head(st)) return rc;
    // HIP events cost ~10 us of stream time each: with profiling on, every run of consecutive streaming-kernel
    // launches (the roofline figure) shares ONE event pair; small-run launches are counted, not timed
    for (size_t li = n_early; li < s.launches.size(); ++li) {
        LevelLaunch& ll = s.launches[li];
        const bool timed = idx->profile && !ll.small;
        const bool group_start = timed && (li == n_early || s.launches[li - 1].small);
        const bool group_end = timed && (li + 1 == s.launches.size() || s.launches[li + 1].small);
        ll.ev = -1;
        if (group_start) { ll.ev = (int)s.prof_used; HIPCHECK(prof_event(s, st)); }
        launch_level(ll, st);
        if (group_end) HIPCHECK(prof_event(s, st));
    }
    // the ordering pass runs on a side stream: it only occupies nq CUs, and the main stream is free to start the
    // next batch's kernels meanwhile (it uses the other slot's buffers).  It stores [QueryOut[nq]][entries] straight
    // into the slot's pinned host block, so no device-to-host copy waits in the DMA queue behind this batch (a
    // queued copy with an unmet dependency stalls every later copy of the process, see the upload above).
    if (!alone) {
        hipStream_t main_st = st;
        if (!s.ev_scanned) HIPCHECK(hipEventCreateWithFlags(&s.ev_scanned, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_scanned, main_st));
        st = idx->sort_stream;
        HIPCHECK(hipStreamWaitEvent(st, s.ev_scanned, 0));
    }
    launch_sort_cands(s.d_qs, s.d_cands.p, s.cap_q, nq, s.d_qout, s.d_entries, s.out_cap, s.d_hdr, st,
                      dev_stream ? s.d_stream.p : nullptr);
    s.heaps_ready = s.dev_replay && !s.dist_batch;           // (a merge batch is replayed after the gather, not here)
    if (s.heaps_ready) {
        uint64_t* d_heaps = reinterpret_cast<uint64_t*>(d_result + off_heaps);
        uint32_t* d_sizes = reinterpret_cast<uint32_t*>(d_result + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
        if (idx->replay_wave && (uint32_t)s.R <= replay_wave_max_R())     // one wave per query, all lanes at work (heap in registers)
            HIPCHECK(launch_replay_heap_wave_states(s.d_qs, s.d_stream.p, s.out_cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st));
        else                                                              // one wave per query, lane 0 pushing into an LDS heap (any R)
            launch_replay_heap(s.d_qs, s.d_stream.p, s.out_cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(take_launch_error());
    if (s.dist_batch && !s.rerun && s.mode != 1)
        if (int rc = enqueue_merge(idx, s, st)) return rc;

    if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(s.ev_done, st));
    return QADC_OK;
}

// Decides whether a batch takes the one-workgroup-per-query path.  codes_per_query: exact maximum when the host
// knows assign[], an estimate (ma x mean partition size) when assign[] is produced on the GPU.
bool wgq_eligible(const qadc_index* idx, int nq, int ma, int R, int mode, uint64_t codes_per_query) {
    if (idx->wgq == 0 || mode != 0 || ma > 4096 || R <= 0) return false;
    if (idx->wgq >= 2) return true;
    if (codes_per_query <= idx->wgq_small_codes) return true;
    // an IVF batch (several partitions, several probes per query: the queries walk different codes) — the query kernel is
    // ahead of the level path at every batch size (C3 shape, synchronous: 1 query 217 -> 175 us, 64 queries 0.85 -> 0.73 ms)
    if (ma > 1 && idx->parts.size() > 1 && codes_per_query <= idx->wgq_max_codes) return true;
    return nq >= idx->wgq_min_nq && codes_per_query <= idx->wgq_max_codes;
}

// One launch per batch: scan_query_kernel (one workgroup per query), then — for batches large enough to replay on
// the device — replay_heap_lanes_kernel on the side stream.  No host planning: the kernel walks assign[] and the
// device partition table itself.
int launch_wgq_batch(qadc_index* idx, Slot& s) {
    ScopedMs timer(idx->prof.host_plan_ms);
    const int M = idx->M, nq = s.nq, ma = s.ma;
    const size_t table_dim = (size_t)M * 16, nt = (size_t)nq * ma * table_dim;
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    s.launches.clear();
    s.start_codes = 0;
    // ---- upload block: [assign i32 nq*ma (host-assign path)][float or int8 tables (host-table path)] ----
    const size_t assign_bytes = s.assign_on_device ? 0 : sizeof(int32_t) * (size_t)nq * ma;
    const size_t off_tables = align16(assign_bytes);
    const size_t tables_bytes = s.float_path ? (s.device_tables ? 0 : nt * sizeof(float)) : nt;
    const size_t in_bytes = align16(off_tables + tables_bytes);
    HIPCHECK(s.h_in.ensure(std::max<size_t>(in_bytes, 16)));
    HIPCHECK(s.d_in.ensure(std::max<size_t>(in_bytes, 16)));
    if (assign_bytes) std::memcpy(s.h_in.p, s.assign.data(), assign_bytes);
    if (s.float_path && !s.device_tables) std::memcpy(s.h_in.p + off_tables, s.tables, nt * sizeof(float));
    if (!s.float_path) std::memcpy(s.h_in.p + off_tables, s.qtables_in.data(), nt);

    // ---- result block in pinned, device-mapped host memory: [QueryOut[nq]][streams u64[nq][cap] unless they stay
    // on the device][heaps u64[nq][R]][sizes u32[nq]] ----
    const uint32_t cap = s.wgq_cap;
    bool alone = true;
    for (int i = 0; i < kSlots; ++i) alone = alone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    alone = alone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    // The lane-per-query replay takes ~1.3 ms whatever the batch size (one query's pushes are sequential): in a pipeline
    // that latency hides under the next batches and the host stays free, but a batch submitted while nothing else is in
    // flight — a synchronous call — is answered sooner by the host's threads up to a few hundred queries (C3 shape,
    // synchronous: 64 queries 1.41 -> 0.87 ms, 256: 1.89 -> 1.44, 512: 2.26 vs 2.41)
    const int replay_from = alone ? std::max(idx->device_replay_nq, idx->device_replay_alone_nq) : idx->device_replay_nq;
    s.dev_replay = idx->device_replay_nq > 0 && nq >= replay_from && (uint32_t)s.R <= (idx->replay_wave ? replay_wave_max_R() : replay_lanes_max_R());
    s.dist_batch = idx->dist != nullptr;
    s.heaps_ready = s.dev_replay && !s.dist_batch;
    if (s.dist_batch) s.dev_replay = true;                   // streams stay on the device for the gather (qadc_dist_collect)
    // A batch too small to fill the GPU splits every query's scan order over G workgroups (each tightens its bound on
    // the query's first block, then scans its own chunk); the sub-streams are concatenated in workgroup order.
    int G = 1;
    if (!s.dev_replay && nq * 2 <= 256) {
        G = std::min<int>(idx->wgq_split, 256 / nq);
        G = (int)std::min<uint64_t>((uint64_t)G, s.wgq_codes / idx->wgq_split_codes);   // at least wgq_split_codes codes per workgroup
        G = std::max(G, 1);
    }
    s.wgq_G = G;
    const int nsub = nq * G;
    const size_t stream_entries = (size_t)nsub * cap;
    if (stream_entries >= (1ull << 32)) return fail(QADC_E_CAPACITY, "candidate stream capacity exceeds 2^32 entries");
    s.out_cap = (uint32_t)stream_entries;
    const size_t host_stream_bytes = s.dev_replay ? 0 : sizeof(uint64_t) * stream_entries;
    const size_t off_heaps = sizeof(QueryOut) * (size_t)nsub + host_stream_bytes;
    const size_t heaps_bytes = s.dev_replay ? (sizeof(uint64_t) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq : 0;
    HIPCHECK(s.h_result.ensure(off_heaps + heaps_bytes + 16, hipHostMallocMapped | hipHostMallocCoherent));
    if (s.h_result.p != s.h_result_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_result_mapped), s.h_result.p, 0));
        s.h_result_mapped = s.h_result.p;
    }
    unsigned char* d_result = s.d_result_mapped;
    s.d_qout = reinterpret_cast<QueryOut*>(d_result);
    s.h_qout = reinterpret_cast<QueryOut*>(s.h_result.p);
    s.d_entries = reinterpret_cast<uint64_t*>(d_result + sizeof(QueryOut) * (size_t)nsub);
    s.h_entries = reinterpret_cast<uint64_t*>(s.h_result.p + sizeof(QueryOut) * (size_t)nsub);
    s.h_heaps = reinterpret_cast<uint64_t*>(s.h_result.p + off_heaps);
    s.h_heap_sizes = reinterpret_cast<uint32_t*>(s.h_result.p + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
    if (s.dev_replay) {
        HIPCHECK(s.d_stream.ensure(stream_entries));
        HIPCHECK(s.d_qflags.ensure((size_t)nq * 4));
    }
    HIPCHECK(s.d_qtables.ensure(nt));
    // pre-scan values beyond the kernel's LDS budget go to a global scratch
    uint64_t fcap = 0;
    if (s.float_path) {
        uint64_t worst = (uint64_t)ma * idx->max_start_n;
        if (!s.assign_on_device) {
            worst = 0;
            for (int q = 0; q < nq; ++q) {
                uint64_t t = 0;
                for (int a = 0; a < ma; ++a) t += idx->parts[s.assign[(size_t)q * ma + a]].start_n;
                worst = std::max(worst, t);
            }
        }
        if (worst > query_kernel_lds_values(M)) fcap = worst;
        for (int q = 0; q < nq && !s.assign_on_device; ++q)
            for (int a = 0; a < ma; ++a) s.start_codes += idx->parts[s.assign[(size_t)q * ma + a]].start_n;
    }
    s.wgq_fcap = fcap;
    if (fcap) HIPCHECK(s.d_fvals.ensure((size_t)nsub * fcap));
    const uint32_t ccap = std::min<uint32_t>(idx->wgq_cand_cap, kQueryCandCap);
    HIPCHECK(s.d_qcands.ensure((size_t)nsub * ccap));

    hipStream_t st = idx->stream;
    // A lone small query (the synchronous single-query call): its input — which partitions, their descriptors, the
    // float tables — fits the kernel-argument segment and rides in the dispatch packet; no copy precedes the launch.
    alignas(16) unsigned char inl[kInlineBytes];
    size_t inl_bytes = 0, inl_off_parts = 0, inl_off_tables = 0;
    if (alone && G > 1 && idx->wgq_inline && s.float_path && !s.device_tables && !s.assign_on_device) {
        const size_t na = (size_t)nq * ma;
        inl_off_parts = align16(sizeof(int32_t) * na);
        inl_off_tables = align16(inl_off_parts + sizeof(PartDesc) * na);
        const size_t total = inl_off_tables + nt * sizeof(float);
        if (total <= kInlineBytes) {
            for (size_t i = 0; i < na; ++i) {
                reinterpret_cast<int32_t*>(inl)[i] = (int32_t)i;
                std::memcpy(inl + inl_off_parts + sizeof(PartDesc) * i, &idx->h_partdesc[s.assign[i]], sizeof(PartDesc));
            }
            std::memcpy(inl + inl_off_tables, s.tables, nt * sizeof(float));
            inl_bytes = total;
        }
    }
    if (in_bytes && !inl_bytes) {
        if (alone) {
            HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, st));
        } else {                                            // never queue a copy behind the previous batch's kernels
            HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, idx->copy_stream));
            if (!s.ev_up) HIPCHECK(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_up, idx->copy_stream));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_up, 0));
        }
    }
    s.prof_used = 0;
    QueryKernelArgs A{};
    A.parts = idx->d_partdesc.p;
    A.assign = s.assign_on_device ? s.d_assign.p : reinterpret_cast<const int32_t*>(s.d_in.p);
    A.ma = ma;
    A.ftables = nullptr;
    A.qtables = s.d_qtables.p;
    if (s.front_sharded) {
        // (tables come out of the sharded front below)
    } else if (s.float_path) {
        if (s.device_tables) {
            HIPCHECK(s.d_ftables.ensure(nt));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_feed, 0));
            launch_build_tables(s.d_queries.p, idx->K ? idx->d_coarse.p : nullptr, s.d_assign.p, idx->d_codebooks.p,
                                idx->has_rotation ? idx->d_rotation.p : nullptr, nq, ma, M, idx->dim, table_expansion(idx, ma), s.d_ftables.p, st);
            A.ftables = s.d_ftables.p;
        } else {
            A.ftables = reinterpret_cast<float*>(s.d_in.p + off_tables);
        }
    } else {
        A.qtables = reinterpret_cast<int8_t*>(s.d_in.p + off_tables);    // the caller's int8 tables, as uploaded
    }
    if (s.front_sharded) {
        // ---- sharded front: this rank's share -> gather -> the whole batch as an int8 batch with device-resident inputs ----
        DistState& d = *idx->dist;
        const size_t tab = table_dim * (size_t)ma;
        const size_t block = ((size_t)s.front_per * (tab + (size_t)ma * 4 + 16) + 15) & ~(size_t)15;
        int32_t* d_assign_share = reinterpret_cast<int32_t*>(s.d_fblock.p + (size_t)s.front_per * tab);
        uint32_t* d_front_share = reinterpret_cast<uint32_t*>(s.d_fblock.p + (size_t)s.front_per * (tab + (size_t)ma * 4));
        if (!s.rerun) {                                          // (a re-run from inside collect reuses the gathered arrays)
            idx->prof.front_sharded_batches++;
            // The share's front runs on the FRONT stream, under the previous batch's scan (it depends on nothing that batch
            // produces), and its gather is issued IN FRONT of that batch's merge gather (flush_merges below).
            hipStream_t fs = alone ? st : idx->front_stream;
            HIPCHECK(hipStreamWaitEvent(fs, s.ev_feed, 0));
            if (s.front_n) {
                const size_t nt_share = (size_t)s.front_n * tab;
                HIPCHECK(s.d_ftables.ensure(nt_share));
                launch_build_tables(s.d_queries.p, idx->d_coarse.p, d_assign_share, idx->d_codebooks.p,
                                    idx->has_rotation ? idx->d_rotation.p : nullptr, s.front_n, ma, M, idx->dim, table_expansion(idx, ma),
                                    s.d_ftables.p, fs);
                QueryKernelArgs F{};
                F.parts = idx->d_partdesc.p;
                F.assign = d_assign_share;
                F.ma = ma;
                F.ftables = s.d_ftables.p;
                F.qtables = reinterpret_cast<int8_t*>(s.d_fblock.p);
                F.fvals = fcap ? s.d_fvals.p : nullptr;
                F.fcap = (uint32_t)fcap;
                F.R = (uint32_t)s.R;
                F.quant_mode = idx->quant_mode;
                F.head_codes = ~0ull;
                F.head_slots = 1;
                F.G = 1;
                F.front_only = 1;
                F.front_out = d_front_share;
                HIPCHECK(launch_scan_query(M, idx->wgq_variant, s.front_n, F, fs));
            }
            // the collectives of the merge live on ONE stream, in the order the host issues them (the same on every rank)
            if (!s.ev_fa) HIPCHECK(hipEventCreateWithFlags(&s.ev_fa, hipEventDisableTiming));
            if (!s.ev_fb) HIPCHECK(hipEventCreateWithFlags(&s.ev_fb, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_fa, fs));
            HIPCHECK(hipStreamWaitEvent(d.stream, s.ev_fa, 0));
            std::string gerr;
            if (d.gather(s.d_fblock.p, s.d_fgathered.p, block / 8, d.stream, gerr)) return fail(QADC_E_HIP, gerr);
            HIPCHECK(launch_front_unpack(s.d_fgathered.p, block, d.world, s.front_per, nq, ma, tab, s.d_qtables.p, s.d_assign.p,
                                         s.d_front_all.p, reinterpret_cast<int32_t*>(s.d_fmap),
                                         reinterpret_cast<uint32_t*>(s.d_fmap + (size_t)nq * ma * 4), d.stream));
            HIPCHECK(hipEventRecord(s.ev_fb, d.stream));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_fb, 0));
            if (int rc = flush_merges(idx, ~0ull)) return rc;   // the older batches' merges: behind this front gather
        }
        A.assign = s.d_assign.p;
        A.ftables = nullptr;                                     // from here on: an int8 batch
        A.qtables = s.d_qtables.p;
        A.front_in = s.d_front_all.p;
    }
    s.d_qt = A.qtables;
    A.fvals = fcap ? s.d_fvals.p : nullptr;
    A.fcap = (uint32_t)fcap;
    A.stream = s.dev_replay ? s.d_stream.p : s.d_entries;
    A.cap = cap;
    A.cands = s.d_qcands.p;
    A.ccap = ccap;
    A.qout = s.d_qout;
    A.qstate_flags = s.dev_replay ? s.d_qflags.p : nullptr;
    A.R = (uint32_t)s.R;
    A.quant_mode = idx->quant_mode;
    A.nontemporal = idx->total_codes * (uint64_t)idx->cs > (200ull << 20);   // (same rule as the level path)
    A.G = G;
    if (idx->profile) HIPCHECK(prof_event(s, st));
    s.poll = alone && G > 1 && !s.dev_replay && !idx->profile && idx->wgq_poll;
    if (s.poll)
        for (int i = 0; i < nsub; ++i) s.h_qout[i].flags = 0;
    A.inline_off_parts = (uint32_t)inl_off_parts;
    A.inline_off_tables = (uint32_t)inl_off_tables;
    // Large IVF batches: a (query, probe) pair lands on a partition several other queries of the batch probe too.  The
    // kernel then only walks the first probes of every query (head: front + a tight bound); the other pairs are
    // regrouped by partition on the device and scanned 8 queries per pass (see launch_ivf_plan), and a third kernel
    // orders every query's candidates into the stream layout the plain launch produces.
    // (under the merge the head counts probes WITH CODES ON THIS RANK; its length is an option of its own: whole-partition
    // placement is better off with 2 — one of 8 ranks, C5 shape: 1.53 vs 1.78 ms per batch — the range split with 4:
    // 1.35 vs 1.52, and at 2048-query batches a head of 2 short pieces bounds too loosely and the batches fall back)
    const int head_slots = std::min(idx->dist ? idx->wgq_group_head_dist : idx->wgq_group_head, ma);
    const size_t pairs = (size_t)nq * (size_t)(ma - head_slots);
    const size_t nparts = idx->parts.size();
    s.wgq_grouped = s.dev_replay && G == 1 && pairs > 0 && nparts < (1u << 24) &&
                    (idx->wgq_group == 2 || (idx->wgq_group == 1 && idx->group_strikes < 2 && nq >= 256 && pairs >= 2 * nparts));
    if (s.wgq_grouped) {
        const size_t state_bytes = 64 + sizeof(QueryState) * (size_t)nq;
        const size_t ngroups = ivf_max_groups(pairs, nparts);
        if (ngroups * 8 >= (1ull << 31)) return fail(QADC_E_CAPACITY, "too many (query, probe) pairs for one batch");
        // query states, plan counters and group items in ONE allocation: one clear instead of three (every launch between
        // two batches' scans costs the scan stream ~10 us)
        auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t gplan_off = up256(state_bytes), gplan_bytes = sizeof(uint32_t) * (3 * nparts + 1);
        const size_t gitems_off = up256(gplan_off + gplan_bytes), gitems_bytes = sizeof(ScanItem) * ngroups * 8;
        HIPCHECK(s.d_state.ensure(gitems_off + gitems_bytes));
        uint32_t* d_gplan = reinterpret_cast<uint32_t*>(s.d_state.p + gplan_off);
        ScanItem* d_gitems = reinterpret_cast<ScanItem*>(s.d_state.p + gitems_off);
        // (the ordering pass of this path sorts up to kOrderCandCap candidates per query — twice what the query kernel's own tail
        // takes; a cap the caller lowered — the tests' way to force the fallback — is honoured)
        const uint32_t gcap = idx->wgq_cand_cap < kQueryCandCap ? idx->wgq_cand_cap : std::min<uint32_t>(idx->wgq_group_cand_cap, kOrderCandCap);
        HIPCHECK(s.d_cands.ensure((size_t)nq * gcap));
        s.d_hdr = reinterpret_cast<CandHeader*>(s.d_state.p);
        s.d_qs = reinterpret_cast<QueryState*>(s.d_state.p + 64);
        HIPCHECK(hipMemsetAsync(s.d_state.p, 0, gitems_off + gitems_bytes, st));
        // (tried: the plan — two clears + count / offsets / scatter, needed by the second phase only — on a stream of its own
        // under the head launch: its workgroups then wait for head workgroups to retire and the second phase for them;
        // C3 0.75 -> 1.05 us per query, one of 8 ranks 0.72 -> 1.51 ms per batch.  Tried as well: table build, clears and
        // plan of a qadc_search batch on the front stream, enqueued under the PREVIOUS batch's scan — ~100 us of the scan
        // stream per batch to win, but the dozen small launches trickle through that scan so slowly that the next head
        // ends up waiting for them: C3 0.75 -> 0.91 us per query, C5 4.63 -> 4.82.)
        launch_ivf_plan(A.assign, idx->d_partdesc.p, nq, ma, head_slots, (int)nparts, d_gplan, d_gplan + 2 * nparts,
                        d_gplan + nparts, d_gitems, st);
        QueryKernelArgs H = A;
        H.head_codes = ~0ull;
        H.head_slots = (uint32_t)head_slots;
        H.qstates = s.d_qs;
        H.cand_regions = s.d_cands.p;
        H.cand_cap = gcap;
        H.hdr = s.d_hdr;
        H.G = 1;
        // (profile: one event before and after each of the three launches — prof_ev[1..4]; ~10 us of stream time each)
        if (idx->profile) HIPCHECK(prof_event(s, st));
        HIPCHECK(launch_scan_query(M, idx->wgq_variant, nq, H, st));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        const int wgs = (int)std::max<uint64_t>(1, ((uint64_t)idx->max_part_n + idx->mq_codes_per_wg - 1) / idx->mq_codes_per_wg);
        launch_scan_i8_mq(M, d_gitems, (int)(ngroups * 8), wgs, A.qtables, s.d_qs, s.d_hdr, s.d_cands.p, gcap, (uint32_t)s.R, st,
                          idx->mq_narrow);
        if (idx->profile) HIPCHECK(prof_event(s, st));
        HIPCHECK(launch_order_cands(s.d_qs, s.d_cands.p, gcap, gcap, nq, s.d_stream.p, cap, s.d_qout, s.d_qflags.p, st));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        s.group_head_slots = head_slots;
        idx->prof.group_launches++;
    } else {
        HIPCHECK(launch_scan_query(M, idx->wgq_variant, nq, A, st, inl_bytes ? inl : nullptr, inl_bytes));
    }
    if (idx->profile) HIPCHECK(prof_event(s, st));
    if (s.dist_batch && !s.rerun)
        if (int rc = enqueue_merge(idx, s, st)) return rc;
    if (s.dev_replay) {
        if (!alone) {
            // replay on a side stream, under the next batches' scans.  A 1024-query replay (16 waves, one lane per
            // query, ~1.5 K dependent pushes each) lasts about as long as the batch's scan: consecutive batches use two
            // side streams alternately, so that a replay never waits for the previous batch's replay.
            if (!s.ev_scanned) HIPCHECK(hipEventCreateWithFlags(&s.ev_scanned, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_scanned, st));
            // Alternate by SUBMISSION order, not by slot: with three batches in flight slots 2 and 0 follow each other, and
            // on one stream the second replay would wait out the first (every third batch lost 0.7 ms that way).
            // (a replay stream of its own per slot was tried: 1.2 -> 1.9 us per query at the IVF shape)
            st = (idx->replay_seq++ & 1) ? idx->front_stream : idx->sort_stream;
            HIPCHECK(hipStreamWaitEvent(st, s.ev_scanned, 0));
        }
        if (s.heaps_ready) {
            uint64_t* d_heaps = reinterpret_cast<uint64_t*>(d_result + off_heaps);
            uint32_t* d_sizes = reinterpret_cast<uint32_t*>(d_result + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
            if (idx->replay_wave)                               // one wave per query, heap in registers
                HIPCHECK(launch_replay_heap_wave_qflags(s.d_qflags.p, s.d_stream.p, cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st));
            else                                                // one lane per query, heaps in LDS
                HIPCHECK(launch_replay_heap_lanes(s.d_qflags.p, s.d_stream.p, cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st));
        }
    }
    if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(s.ev_done, st));
    return QADC_OK;
}

// The merge of a one-workgroup-per-query batch, enqueued behind its scan: pack (from the kernels' own records in
// device memory) -> all-gather -> interleave -> replay; the collectives on the merge's stream, the compute on a stream
// of its own; the heaps and a status word land in pinned host memory.  Every rank enqueues the same collectives in the
// same order (the ranks submit and collect the same batches in the same order).
// Not taken (the collect-time merge runs instead): few-query batches (host-share replay), R or ma x world beyond the device
// merge, option dist_async = 0.
// Two steps.  enqueue_merge (at the end of the batch's launch) records where the scan ends and marks the merge PENDING;
// flush_merges issues the pending merges in submission order.  A batch with a sharded front flushes the OLDER merges right
// after issuing its own front gather: on the collectives' stream that gather then lies in front of the previous batch's
// merge gather (which waits for that batch's scan), so a front that runs under the previous scan is not held up by it.
int enqueue_merge_now(qadc_index* idx, Slot& s) {
    DistState& d = *idx->dist;
    const int slot_i = (int)(&s - idx->slot);
    DistSlot& ds = d.slot[slot_i];
    ds.pending = false;
    const int nq = s.nq, R = s.R, world = d.world;
    const size_t bw = dist_block_words(nq, d.cap_entries, 0);
    HIPCHECK(ds.d_block.ensure(bw));
    HIPCHECK(ds.d_gathered.ensure(bw * world));
    HIPCHECK(ds.d_merged.ensure((size_t)d.cap_entries * world));
    HIPCHECK(ds.d_moff.ensure(nq));
    HIPCHECK(ds.d_mcnt.ensure(2 * (size_t)nq));
    const size_t heaps_bytes = (sizeof(uint64_t) * (size_t)R + sizeof(uint32_t)) * (size_t)nq;
    HIPCHECK(ds.h_out.ensure(heaps_bytes + 32, hipHostMallocMapped | hipHostMallocCoherent));
    if (ds.h_out.p != ds.h_out_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ds.d_out), ds.h_out.p, 0));
        ds.h_out_mapped = ds.h_out.p;
    }
    if (!ds.ev_done) HIPCHECK(hipEventCreateWithFlags(&ds.ev_done, hipEventDisableTiming));
    hipStream_t st = d.stream;
    HIPCHECK(hipStreamWaitEvent(st, ds.ev_ready, 0));
    if (s.wgq) {
        HIPCHECK(launch_dist_pack_qflags(s.d_qflags.p, nq, s.d_stream.p, s.wgq_cap, d.cap_entries, ds.d_block.p, st));
    } else {
        // level path: the ordering pass left {out_off, count, reps, flags} in the query states and the compact ordered
        // streams in d_stream; whatever the collect call would have to redo first (a query the device did not order, an
        // overflowed region / output / pre-scan buffer) raises bit7 and the merge is redone at collect time
        HIPCHECK(ds.d_src.ensure(3 * (size_t)nq));
        HIPCHECK(launch_dist_src_from_states(s.d_qs, nq, s.out_cap, ds.d_src.p, st));
        HIPCHECK(launch_dist_pack(ds.d_src.p, ds.d_src.p + nq, ds.d_src.p + 2 * (size_t)nq, nq, s.d_stream.p, nullptr, d.cap_entries,
                                  nullptr, 0, ds.d_block.p, st));
    }
    std::string gerr;
    if (d.gather(ds.d_block.p, ds.d_gathered.p, bw, st, gerr)) return fail(QADC_E_HIP, gerr);
    // the collectives keep `st` to themselves (the next batch's front gather is issued right behind this one); the merge's
    // compute — a millisecond of replay latency — goes to a stream of its own
    if (!ds.ev_gathered) HIPCHECK(hipEventCreateWithFlags(&ds.ev_gathered, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(ds.ev_gathered, st));
    hipStream_t ms = d.merge_stream[slot_i] ? d.merge_stream[slot_i] : st;
    if (ms != st) HIPCHECK(hipStreamWaitEvent(ms, ds.ev_gathered, 0));
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(ds.d_out + sizeof(uint64_t) * (size_t)R * nq);
    HIPCHECK(launch_dist_merge(ds.d_gathered.p, bw, world, nq, s.ma, (uint32_t)R, ds.d_moff.p, ds.d_mcnt.p, ds.d_mcnt.p + nq,
                               ds.d_merged.p, reinterpret_cast<uint64_t*>(ds.d_out), d_sizes, ms, d_sizes + nq));
    HIPCHECK(hipEventRecord(ds.ev_done, ms));
    ds.enqueued = true;
    return QADC_OK;
}

// Issues the pending merges with seq <= upto, oldest first.
int flush_merges(qadc_index* idx, uint64_t upto) {
    if (!idx->dist) return QADC_OK;
    DistState& d = *idx->dist;
    for (;;) {
        int best = -1;
        for (int i = 0; i < kSlots; ++i)
            if (d.slot[i].pending && d.slot[i].seq <= upto && (best < 0 || d.slot[i].seq < d.slot[best].seq)) best = i;
        if (best < 0) return QADC_OK;
        if (int rc = enqueue_merge_now(idx, idx->slot[best])) return rc;
    }
}

int enqueue_merge(qadc_index* idx, Slot& s, hipStream_t scan_stream) {
    DistState& d = *idx->dist;
    const int slot_i = (int)(&s - idx->slot);
    if (slot_i < 0 || slot_i >= kSlots) return QADC_OK;
    DistSlot& ds = d.slot[slot_i];
    ds.enqueued = false;
    ds.pending = false;
    const int nq = s.nq, R = s.R, world = d.world;
    // (level-path batches as well: the ordering pass leaves what the pack needs in the query states.  Below dist_device_nq
    // queries the merge stays at collect time, where the ranks replay shares on the host's otherwise idle cores: enqueuing
    // such a batch's device merge — or just its pack, gather and copy-out — behind the scan was measured on bench.py's
    // 32-query flat steps, one of 8 ranks: 1.61-1.65 ms per step against 1.22; interleave + replay kernels under the long
    // scan launches take 1.5-1.9 ms per batch, the host 0.07)
    if (!d.async_merge || nq < d.device_nq || (uint32_t)R > replay_wave_max_R() || (size_t)s.ma * world > dist_interleave_max_cells())
        return QADC_OK;
    if (!ds.ev_ready) HIPCHECK(hipEventCreateWithFlags(&ds.ev_ready, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(ds.ev_ready, scan_stream));
    ds.pending = true;
    ds.seq = d.next_seq++;
    if (!s.front_sharded) return flush_merges(idx, ds.seq);     // no front gather to let pass: issue it (and anything older) now
    return QADC_OK;
}

int submit_common(qadc_index* idx, int slot_i, int nq, int ma, const int32_t* assign, float* tables,
                  const int8_t* qtables, int R, int mode = 0, int slice = 0, int nslices = 1,
                  const float* inj_vals = nullptr, int inj_n = 0) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    if (nq <= 0 || ma <= 0 || R <= 0 || !assign) return fail(QADC_E_ARG, "nq, ma, R must be > 0 and assign non-null");
    if (nq >= (1 << 24)) return fail(QADC_E_ARG, "nq must be < 2^24");
    if (ma >= (1 << 14)) return fail(QADC_E_ARG, "ma must be < 16384");
    if (!tables && !qtables) return fail(QADC_E_ARG, "tables is null");
    if (mode == 1 && (nslices < 1 || slice < 0 || slice >= nslices)) return fail(QADC_E_ARG, "need 0 <= slice < nslices");
    if (mode == 1 && slot_i > 1) return fail(QADC_E_ARG, "pre-scan slot must be 0 or 1");
    if (mode == 2 && (!inj_vals || inj_n < 1)) return fail(QADC_E_ARG, "prescan values missing");
    Slot& s = mode == 1 ? idx->pre_slot[slot_i] : idx->slot[slot_i];
    if (s.busy) return fail(QADC_E_STATE, "slot still holds an uncollected batch");
    if (int rc = use_device(idx)) return rc;
    s.mode = mode;
    s.pre_slice = slice;
    s.pre_nslices = nslices;
    s.inj_n = (uint32_t)inj_n;
    if (mode == 2) s.inj_vals.assign(inj_vals, inj_vals + (size_t)nq * inj_n);
    s.nq = nq;
    s.ma = ma;
    s.R = R;
    s.float_path = tables != nullptr;
    s.tables = tables;
    s.device_tables = false;
    s.assign.assign(assign, assign + (size_t)nq * ma);
    if (qtables) {
        const size_t nt = (size_t)nq * ma * idx->M * 16;
        for (size_t i = 0; i < nt; ++i)
            if (qtables[i] < 0)
                return fail(QADC_E_ARG, "int8 tables must lie in [0,127] (QuantizerMAX<int8_t> output, db_query_4.cpp:37-71)");
        s.qtables_in.assign(qtables, qtables + nt);
    }
    s.full_prescan = false;
    s.rerun = false;
    s.front_sharded = false;
    s.assign_on_device = false;
    {   // which path: levels (long shared lists) or one workgroup per query (IVF batches, small lists)
        uint64_t max_codes = 0;
        for (int q = 0; q < nq; ++q) {
            uint64_t c = 0;
            for (int a = 0; a < ma; ++a) {
                const int p = s.assign[(size_t)q * ma + a];
                if (p < 0 || p >= (int)idx->parts.size()) return fail(QADC_E_ARG, "assign[] names a partition that does not exist");
                c += idx->parts[p].n;
            }
            max_codes = std::max(max_codes, c);
        }
        s.wgq = wgq_eligible(idx, nq, ma, R, mode, max_codes);
        s.wgq_codes = max_codes;
    }
    if (s.wgq) s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
    s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
    if (!s.wgq) s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)nq * 8192u, 1ull << 30));
    if (int rc = plan_and_launch(idx, s)) return rc;
    s.busy = true;
    return QADC_OK;
}

// Whether launch_wgq_batch will send a batch of this shape through the partition-major second phase (same test as there).
bool will_group(const qadc_index* idx, int nq, int ma, bool dev_replay) {
    const int head_slots = std::min(idx->dist ? idx->wgq_group_head_dist : idx->wgq_group_head, ma);
    const size_t pairs = (size_t)nq * (size_t)(ma - head_slots), nparts = idx->parts.size();
    return dev_replay && pairs > 0 && nparts < (1u << 24) &&
           (idx->wgq_group == 2 || (idx->wgq_group == 1 && idx->group_strikes < 2 && nq >= 256 && pairs >= 2 * nparts));
}

// N1: queries in.  Coarse assignment runs on the copy stream (so it does not queue behind the previous
// batch's scan), the host reads assign[] back to plan the work items, residuals and float tables are built
// on the GPU by the main stream.
int search_submit(qadc_index* idx, int slot_i, int nq, const float* queries, int ma, int R) {
    if (!idx || !queries) return fail(QADC_E_ARG, "null argument");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    if (idx->dim == 0) return fail(QADC_E_STATE, "qadc_index_set_pq has not been called");
    if (nq <= 0 || ma <= 0 || R <= 0) return fail(QADC_E_ARG, "nq, ma, R must be > 0");
    if (nq >= (1 << 24) || ma >= (1 << 14)) return fail(QADC_E_ARG, "nq must be < 2^24 and ma < 16384");
    if (idx->K && (idx->K != (int)idx->parts.size() || ma > idx->K))
        return fail(QADC_E_ARG, "coarse centroids must match the partitions one to one and ma <= K");
    if (!idx->K && idx->parts.size() != 1) return fail(QADC_E_ARG, "a database without coarse centroids must be flat (1 partition)");
    Slot& s = idx->slot[slot_i];
    if (s.busy) return fail(QADC_E_STATE, "slot still holds an uncollected batch");
    if (int rc = use_device(idx)) return rc;
    s.mode = 0;
    const int dim = idx->dim;
    s.nq = nq;
    s.ma = ma;
    s.R = R;
    s.float_path = true;
    s.device_tables = true;
    s.tables = nullptr;
    HIPCHECK(s.h_queries.ensure((size_t)nq * dim));
    HIPCHECK(s.d_queries.ensure((size_t)nq * dim));
    HIPCHECK(s.h_assign.ensure((size_t)nq * ma));
    HIPCHECK(s.d_assign.ensure((size_t)nq * ma));
    std::memcpy(s.h_queries.p, queries, sizeof(float) * (size_t)nq * dim);
    hipStream_t cs = idx->copy_stream;
    // Under the multi-GPU merge every rank receives the same queries; what is per QUERY rather than per code — coarse
    // assignment, residual tables, pre-scan, select, quantizer — is then split over the ranks: rank r does it for queries
    // [r * per, (r + 1) * per) and one all-gather ships assign[] + int8 tables + (flags, qmin, qmax) to everybody
    // (launch_wgq_batch).  Only for batches that take the one-workgroup-per-query head + partition-major second phase.
    s.front_sharded = false;
    {
        const uint64_t est = idx->parts.empty() ? 0 : idx->total_codes / idx->parts.size() * (uint64_t)(idx->K ? ma : 1);
        if (idx->dist && idx->dist->shard_front && idx->dist->world > 1 && idx->K && nq >= 2 * idx->dist->world &&
            wgq_eligible(idx, nq, ma, R, 0, est) && will_group(idx, nq, ma, true)) {
            s.front_sharded = true;
            s.front_per = (nq + idx->dist->world - 1) / idx->dist->world;
            s.front_q0 = std::min(nq, idx->dist->rank * s.front_per);
            s.front_n = std::min(nq, s.front_q0 + s.front_per) - s.front_q0;
        }
    }
    if (s.front_sharded) {
        const size_t tab = (size_t)ma * idx->M * 16;
        const size_t block = ((size_t)s.front_per * (tab + (size_t)ma * 4 + 16) + 15) & ~(size_t)15;   // [qtables][assign][front] of `per` queries
        HIPCHECK(s.d_fblock.ensure(block));
        HIPCHECK(s.d_fgathered.ensure(block * idx->dist->world));
        HIPCHECK(s.d_front_all.ensure(4 * (size_t)nq));
        HIPCHECK(s.h_fmap.ensure((size_t)nq * ma * 4 + (size_t)nq * 16, hipHostMallocMapped | hipHostMallocCoherent));
        if (s.h_fmap.p != s.h_fmap_mapped) {
            HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_fmap), s.h_fmap.p, 0));
            s.h_fmap_mapped = s.h_fmap.p;
        }
        if (s.front_n) {
            HIPCHECK(hipMemcpyAsync(s.d_queries.p, s.h_queries.p + (size_t)s.front_q0 * dim, sizeof(float) * (size_t)s.front_n * dim,
                                    hipMemcpyHostToDevice, cs));
            HIPCHECK(s.d_cdist.ensure((size_t)s.front_n * idx->K));
            int32_t* d_assign_share = reinterpret_cast<int32_t*>(s.d_fblock.p + (size_t)s.front_per * tab);
            launch_coarse_assign(s.d_queries.p, idx->d_coarse.p, s.front_n, idx->K, dim, ma, s.d_cdist.p, d_assign_share, cs);
            HIPCHECK(hipGetLastError());
        }
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));
        s.wgq = true;
        s.wgq_codes = idx->total_codes / idx->parts.size() * (uint64_t)ma;
        s.assign_on_device = true;
        s.full_prescan = false;
        s.rerun = false;
        s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
        s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
        if (int rc = plan_and_launch(idx, s)) return rc;
        s.busy = true;
        return QADC_OK;
    }
    HIPCHECK(hipMemcpyAsync(s.d_queries.p, s.h_queries.p, sizeof(float) * (size_t)nq * dim, hipMemcpyHostToDevice, cs));
    if (idx->K) {
        HIPCHECK(s.d_cdist.ensure((size_t)nq * idx->K));
        launch_coarse_assign(s.d_queries.p, idx->d_coarse.p, nq, idx->K, dim, ma, s.d_cdist.p, s.d_assign.p, cs);
        HIPCHECK(hipGetLastError());
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));             // the tables need assign[] on the device, not the copy below
        HIPCHECK(hipMemcpyAsync(s.h_assign.p, s.d_assign.p, sizeof(int32_t) * (size_t)nq * ma, hipMemcpyDeviceToHost, cs));
    } else {
        HIPCHECK(hipMemsetAsync(s.d_assign.p, 0, sizeof(int32_t) * (size_t)nq * ma, cs));
        std::memset(s.h_assign.p, 0, sizeof(int32_t) * (size_t)nq * ma);
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));
    }
    if (!s.ev_assign) HIPCHECK(hipEventCreateWithFlags(&s.ev_assign, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(s.ev_assign, cs));
    // one workgroup per query: the kernel reads assign[] on the device, the host only wants it back for the caller
    const uint64_t est_codes = idx->parts.empty() ? 0 : idx->total_codes / idx->parts.size() * (uint64_t)(idx->K ? ma : 1);
    s.wgq = wgq_eligible(idx, nq, ma, R, 0, est_codes);
    s.wgq_codes = est_codes;
    s.assign_on_device = s.wgq;
    if (!s.wgq) {
        HIPCHECK(hipStreamSynchronize(cs));                  // the planner needs assign[] on the host
        s.assign.assign(s.h_assign.p, s.h_assign.p + (size_t)nq * ma);
    }
    s.full_prescan = false;
    s.rerun = false;
    if (s.wgq) s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
    s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
    if (!s.wgq) s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)nq * 8192u, 1ull << 30));
    if (int rc = plan_and_launch(idx, s)) return rc;
    s.busy = true;
    return QADC_OK;
}

// Waits for the batch, regrows and re-runs on overflow, and lays the ordered candidate streams
// (padding-lane replays expanded) out in s.out_*.  Queries the device could not sort (more than
// kSortCap candidates) are sorted here.
// from_dist: the caller is qadc_dist_collect, which consumes the ordered streams where they lie in device memory; a batch
// submitted under the multi-GPU merge but collected by a plain collect call has no device-side heaps and, on the
// one-workgroup-per-query path, no host copy of its streams: they are fetched and replayed on the host.
int collect_common(qadc_index* idx, int slot_i, bool need_stream = true, bool from_dist = false) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    Slot& s = idx->slot[slot_i];
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no batch");
    if (int rc = use_device(idx)) return rc;
    if (s.dist_batch && !from_dist) {
        need_stream = true;
        if (idx->dist && idx->dist->slot[slot_i].pending)
            if (int rc = flush_merges(idx, idx->dist->slot[slot_i].seq)) return rc;
        if (idx->dist && idx->dist->slot[slot_i].enqueued) {     // a merge was enqueued with the batch: let it finish (its result is unused)
            HIPCHECK(hipEventSynchronize(idx->dist->slot[slot_i].ev_done));
            idx->dist->slot[slot_i].enqueued = false;
        }
    }
    const uint32_t sort_limit_max = kSortCap;
    uint64_t total_sorted = 0;
    if (s.assign_on_device) {                                  // qadc_search: assign[] comes back for the caller (and the planner)
        if (s.front_sharded) {                                   // gathered from the ranks; front_unpack_kernel stored it in mapped memory
            HIPCHECK(hipEventSynchronize(s.ev_fb));
            const int32_t* ha = reinterpret_cast<const int32_t*>(s.h_fmap.p);
            s.assign.assign(ha, ha + (size_t)s.nq * s.ma);
        } else {
            HIPCHECK(hipEventSynchronize(s.ev_assign));
            s.assign.assign(s.h_assign.p, s.h_assign.p + (size_t)s.nq * s.ma);
        }
        s.assign_on_device = false;
    }
    for (int attempt = 0; s.wgq; ++attempt) {                  // one workgroup per query: per-query stream capacity only
        bool seen = false;
        if (s.poll) {
            // a lone small batch: the kernel's workgroups set the done bit of their records in the mapped result block as
            // they finish; watching those spares the completion-signal path (end-of-kernel release, signal, wake-up).
            // Bounded: a batch that takes longer is waited for the ordinary way.
            const auto t_poll = std::chrono::steady_clock::now();
            const int nsub = s.nq * s.wgq_G;
            for (uint32_t spins = 0; !seen; ++spins) {
                seen = true;
                for (int i = 0; i < nsub && seen; ++i)
                    seen = (reinterpret_cast<const volatile uint32_t*>(&s.h_qout[i].flags)[0] & 4u) != 0;
                if (!seen && (spins & 63u) == 63u &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count() > 300e-6)
                    break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIPCHECK(hipEventSynchronize(s.ev_done));
        uint64_t max_count = 0;
        bool cand_overflow = false;
        for (int q = 0; q < s.nq * s.wgq_G; ++q) {
            max_count = std::max<uint64_t>(max_count, s.h_qout[q].count);
            cand_overflow |= (s.h_qout[q].flags & 32u) != 0;
        }
        if (cand_overflow) {
            // some query emitted more candidates than a workgroup sorts in LDS (adversarial order, all-equal tables):
            // the level-structured path has the machinery for that (regrow, host sort) — run the batch through it
            idx->prof.regrows++;
            if (s.wgq_grouped) {
                idx->prof.group_fallbacks++;
                idx->group_strikes++;                            // (auto mode gives the second phase up after two such batches)
            }
            s.wgq_grouped = false;
            s.wgq = false;
            s.wgq_G = 1;
            if (s.front_sharded) s.float_path = false;           // the level path takes the gathered int8 tables as they lie on the device
            s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)s.nq * 8192u, 1ull << 30));
            s.rerun = true;
            if (int rc = plan_and_launch(idx, s)) {
                s.busy = false;
                return rc;
            }
            break;
        }
        if (max_count <= s.wgq_cap) break;
        if (attempt >= 2 || max_count + 64 > (1ull << 31)) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "candidate stream overflow persists");
        }
        s.wgq_cap = (uint32_t)(max_count + max_count / 8 + 64);  // every entry was counted: size for all of them, run again
        idx->prof.regrows++;
        s.rerun = true;
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    for (int attempt = 0; !s.wgq; ++attempt) {
        HIPCHECK(hipEventSynchronize(s.ev_done));
        const uint32_t limit = std::min<uint32_t>(s.cap_q, sort_limit_max);
        uint64_t max_count = 0;
        total_sorted = 0;
        for (int q = 0; q < s.nq; ++q) {
            const QueryOut& qs = s.h_qout[q];
            max_count = std::max<uint64_t>(max_count, qs.count);
            if (qs.count <= limit) total_sorted += (uint64_t)qs.count + qs.reps;
        }
        const bool region_overflow = max_count > s.cap_q;     // emit_candidate counts every request
        const bool out_overflow = total_sorted > s.out_cap;
        bool prescan_overflow = false;
        for (int q = 0; q < s.nq; ++q) prescan_overflow |= (s.h_qout[q].flags & 8u) != 0;
        if (!region_overflow && !out_overflow && !prescan_overflow) break;
        if (attempt >= 4 || max_count + 64 > (1ull << 31) || total_sorted > (1ull << 31)) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "candidate buffers overflow persists (adversarial scan order?)");
        }
        // every emitted candidate was counted: size the buffers for all of them and run the batch again
        if (region_overflow) s.cap_q = (uint32_t)std::max<uint64_t>(max_count + 64, (uint64_t)s.cap_q * 2);
        if (out_overflow) s.out_cap = (uint32_t)(total_sorted + total_sorted / 4 + 1024);
        if (prescan_overflow) s.full_prescan = true;   // adversarially ordered starts: pre-scan all of them unfiltered
        idx->prof.regrows++;
        s.rerun = true;                                // (no merge is enqueued behind a re-run: the collect-time merge follows)
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    s.busy = false;
    if (idx->profile && s.wgq) {
        float ms = 0;
        if (s.prof_used >= 2) HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[s.prof_used - 1]));
        idx->prof.wgq_ms += ms;
        if (s.wgq_grouped && s.prof_used >= 6) {                 // [0] start, [1..4] around head / grouped scan / ordering, [5] end
            float a = 0, b = 0, c = 0;
            HIPCHECK(hipEventElapsedTime(&a, s.prof_ev[1], s.prof_ev[2]));
            HIPCHECK(hipEventElapsedTime(&b, s.prof_ev[2], s.prof_ev[3]));
            HIPCHECK(hipEventElapsedTime(&c, s.prof_ev[3], s.prof_ev[4]));
            idx->prof.group_head_ms += a;
            idx->prof.group_scan_ms += b;
            idx->prof.group_order_ms += c;
            idx->prof.group_batches++;
            // the work in those launches, recounted from assign[] the way the device draws the line (ivf_for_grouped_pairs)
            std::vector<uint32_t> cnt(idx->parts.size(), 0);
            for (int q = 0; q < s.nq; ++q) {
                int seen = 0;
                for (int a_ = 0; a_ < s.ma; ++a_) {
                    const int p = s.assign[(size_t)q * s.ma + a_];
                    const uint32_t n = idx->parts[p].n;
                    if (!n) continue;
                    if (seen < s.group_head_slots) idx->prof.group_head_codes += n;
                    else cnt[p]++;
                    ++seen;
                }
            }
            for (size_t p = 0; p < cnt.size(); ++p) {
                if (!cnt[p]) continue;
                const uint64_t n = idx->parts[p].n, full = cnt[p] / 8, rem = cnt[p] % 8;
                idx->prof.group_pairs += cnt[p];
                idx->prof.group_pass_codes8 += full * n;
                idx->prof.group_seats += full * 8;
                if (rem) {
                    const bool narrow = idx->mq_narrow && rem <= 4;
                    (narrow ? idx->prof.group_pass_codes4 : idx->prof.group_pass_codes8) += n;
                    idx->prof.group_seats += narrow ? 4 : 8;
                }
            }
        }
        idx->prof.wgq_launches++;
        for (int q = 0; q < s.nq * s.wgq_G; ++q) {
            idx->prof.wgq_front_cycles += (uint64_t)(s.h_qout[q].pad[0] & 0xffffu) << 6;
            idx->prof.wgq_sort_cycles += (uint64_t)(s.h_qout[q].pad[0] >> 16) << 6;
            idx->prof.wgq_scan_cycles += (uint64_t)s.h_qout[q].pad[1] << 4;
        }
        idx->prof.wgq_queries += (uint64_t)s.nq;
        for (int q = 0; q < s.nq; ++q)
            for (int a = 0; a < s.ma; ++a) idx->prof.wgq_codes += idx->parts[s.assign[(size_t)q * s.ma + a]].n;
        if (s.float_path) idx->prof.start_codes += s.start_codes;
    }
    if (idx->profile && !s.wgq) {
        float ms = 0;
        if (s.prof_used >= 2) {
            HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[1]));
            if (s.float_path) idx->prof.start_ms += ms;
        }
        for (auto& ll : s.launches) {
            if (ll.small || ll.early) {                      // counted, not timed (see plan_and_launch)
                idx->prof.small_launches++;
                idx->prof.small_codes += ll.codes;
                continue;
            }
            idx->prof.scan_launches++;
            idx->prof.scan_codes += ll.codes;
            idx->prof.mq_launches += ll.mq ? 1 : 0;
            idx->prof.pass_codes += ll.mq ? ll.codes / (uint64_t)ll.nitems * (uint64_t)((ll.nitems + 7) / 8) : ll.codes;
            if (ll.ev < 0 || (size_t)ll.ev + 1 >= s.prof_used) continue;   // not the first launch of its timed group
            HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[ll.ev], s.prof_ev[ll.ev + 1]));
            idx->prof.scan_ms += ms;
        }
        if (s.float_path) idx->prof.start_codes += s.start_codes;
    }
    const auto t0 = std::chrono::steady_clock::now();
    s.out_off.assign((size_t)s.nq + 1, 0);                       // (out_entries keeps its size as a high-water mark: out_off[nq] is the length)
    s.skipped_streams = false;
    const bool streams_stay = s.dist_batch && from_dist;         // consumed in device memory by the gather's pack kernel
    if (s.wgq && s.dev_replay && !streams_stay) {
        // the streams stayed in device memory; fetch them only if the caller wants them (or a query could not be
        // replayed on the device): one strided copy of the used part of every query's region
        bool need = need_stream;
        uint32_t max_count = 0;
        for (int q = 0; q < s.nq; ++q) {
            need = need || (s.heaps_ready && s.h_heap_sizes[q] == 0xffffffffu);
            max_count = std::max(max_count, s.h_qout[q].count);
        }
        if (need && max_count) {
            HIPCHECK(s.h_fetch.ensure((size_t)s.nq * s.wgq_cap));
            HIPCHECK(hipMemcpy2DAsync(s.h_fetch.p, sizeof(uint64_t) * s.wgq_cap, s.d_stream.p, sizeof(uint64_t) * s.wgq_cap,
                                      sizeof(uint64_t) * max_count, s.nq, hipMemcpyDeviceToHost, idx->copy_stream));
            HIPCHECK(hipStreamSynchronize(idx->copy_stream));
            s.h_entries = s.h_fetch.p;
        }
    }
    // Common case — every query's stream lies ready in the pinned result block (ordered on the device; a split query's
    // sub-streams in workgroup order) or is not wanted (heap built on the device): lengths first, then the pool's threads
    // copy disjoint ranges (the block was just written by the GPU: a single thread reads it at ~9 GB/s).
    {
        const bool split = s.wgq && s.wgq_G > 1;
        bool simple = true, skipped = false;
        uint64_t total = 0, ncand = 0;
        for (int q = 0; q < s.nq && simple; ++q) {
            s.out_off[q] = total;
            if (split) {
                for (int g = 0; g < s.wgq_G; ++g) total += s.h_qout[(size_t)q * s.wgq_G + g].count;
                continue;
            }
            const QueryOut& qs = s.h_qout[q];
            if ((!need_stream && s.heaps_ready && s.h_heap_sizes[q] != 0xffffffffu) ||   // heap already built on the device
                (streams_stay && (qs.flags & 4u))) {                                     // ... or the gather takes it from there
                skipped = true;
                continue;
            }
            if (qs.flags & 4u) total += (uint64_t)qs.count + qs.reps;
            else simple = false;
        }
        if (simple) {
            s.out_off[s.nq] = total;
            s.skipped_streams = skipped;
            if (s.out_entries.size() < total) s.out_entries.resize(total);
            for (int q = 0; q < s.nq * (split ? s.wgq_G : 1); ++q) ncand += s.h_qout[q].count;
            idx->prof.candidates += ncand;
            auto copy = [&](int q0, int q1) {
                for (int q = q0; q < q1; ++q) {
                    uint64_t* dst = s.out_entries.data() + s.out_off[q];
                    if (split) {
                        for (int g = 0; g < s.wgq_G; ++g) {
                            const QueryOut& qs = s.h_qout[(size_t)q * s.wgq_G + g];
                            std::memcpy(dst, s.h_entries + qs.out_off, sizeof(uint64_t) * qs.count);
                            dst += qs.count;
                        }
                    } else if (s.out_off[q + 1] > s.out_off[q]) {
                        std::memcpy(dst, s.h_entries + s.h_qout[q].out_off, sizeof(uint64_t) * (s.out_off[q + 1] - s.out_off[q]));
                    }
                }
            };
            int nt = idx->replay_threads > 0 ? idx->replay_threads : (int)std::min<unsigned>(std::thread::hardware_concurrency(), 8);
            nt = std::max(1, std::min(nt, s.nq / 2));
            if (total < 16384) nt = 1;
            const int per = std::max(1, s.nq / (nt * 4));
            idx->pool.run((s.nq + per - 1) / per, nt, [&](int t) { copy(t * per, std::min(s.nq, (t + 1) * per)); });
            idx->prof.host_replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return QADC_OK;
        }
        s.out_off.assign((size_t)s.nq + 1, 0);                   // a query needs the host sort: the sequential path below
        s.out_entries.clear();
    }
    for (int q = 0; q < s.nq && s.wgq && s.wgq_G > 1; ++q) {   // sub-streams of a query's workgroups, in workgroup order
        s.out_off[q] = s.out_entries.size();
        for (int g = 0; g < s.wgq_G; ++g) {
            const QueryOut& qs = s.h_qout[(size_t)q * s.wgq_G + g];
            idx->prof.candidates += qs.count;
            s.out_entries.insert(s.out_entries.end(), s.h_entries + qs.out_off, s.h_entries + qs.out_off + qs.count);
        }
    }
    for (int q = 0; q < s.nq && !(s.wgq && s.wgq_G > 1); ++q) {
        const QueryOut& qs = s.h_qout[q];
        s.out_off[q] = s.out_entries.size();
        idx->prof.candidates += qs.count;
        if ((!need_stream && s.heaps_ready && s.h_heap_sizes[q] != 0xffffffffu) ||   // heap already built on the device
            (streams_stay && (qs.flags & 4u))) {
            s.skipped_streams = true;
            continue;
        }
        if (qs.flags & 4u) {                                  // ordered and expanded on the device
            const size_t n = (size_t)qs.count + qs.reps;
            s.out_entries.insert(s.out_entries.end(), s.h_entries + qs.out_off, s.h_entries + qs.out_off + n);
            continue;
        }
        // host fallback: fetch the raw region, sort by (level, assign slot, position), expand replays
        idx->prof.host_sorted_queries++;
        HIPCHECK(s.h_cands.ensure(qs.count));
        HIPCHECK(hipMemcpyAsync(s.h_cands.p, s.d_cands.p + (size_t)q * s.cap_q, sizeof(Cand) * qs.count,
                                hipMemcpyDeviceToHost, idx->copy_stream));
        HIPCHECK(hipStreamSynchronize(idx->copy_stream));
        Cand* c = s.h_cands.p;
        std::sort(c, c + qs.count, [](const Cand& a, const Cand& b) {
            const uint32_t oa = a.order & 0xfffffu, ob = b.order & 0xfffffu;
            if (oa != ob) return oa < ob;
            return a.pos < b.pos;
        });
        for (uint32_t i = 0; i < qs.count; ++i) {
            const uint32_t reps = 1u + ((c[i].order >> 20) & 15u);
            for (uint32_t r = 0; r < reps; ++r)
                s.out_entries.push_back((uint64_t)c[i].key | ((uint64_t)(c[i].val & 0xffu) << 32) |
                                        ((uint64_t)(c[i].order & 0x3fffu) << 40));
        }
    }
    s.out_off[s.nq] = s.out_entries.size();
    idx->prof.host_replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return QADC_OK;
}

void finish_float_outputs(qadc_index* idx, Slot& s, int32_t* status, float* qmin, float* qmax) {
    const size_t per_q = (size_t)s.ma * idx->M * 16;
    const int stride = s.wgq ? s.wgq_G : 1;                      // (a query's workgroups report the same qmin / qmax / flags)
    for (int q = 0; q < s.nq; ++q) {
        QueryOut qs = s.h_qout[(size_t)q * stride];
        if (s.front_sharded) {                                   // the front ran on rank q / per: its verdict came with the gather
            const uint32_t* fr = reinterpret_cast<const uint32_t*>(s.h_fmap.p + (size_t)s.nq * s.ma * 4) + 4 * (size_t)q;
            qs.flags = (qs.flags & ~3u) | (fr[0] & 3u);
            std::memcpy(&qs.qmin, &fr[1], 4);
            std::memcpy(&qs.qmax, &fr[2], 4);
        }
        if (status) status[q] = (qs.flags & 1u) ? 1 : 0;
        if (qmin) qmin[q] = qs.qmin;
        if (qmax) qmax[q] = qs.qmax;
        if ((qs.flags & 2u) && s.tables) {  // in-place clamp of the caller's tables (db_query_4.cpp:262-269)
            float* t = s.tables + (size_t)q * per_q;
            for (size_t i = 0; i < per_q; ++i)
                if (t[i] < 0) t[i] = 0;
        }
    }
}

int replay_outputs(qadc_index* idx, Slot& s, uint32_t* keys, int8_t* values, int32_t* sizes, const int32_t* status) {
    ScopedMs timer(idx->prof.host_heap_ms);
    auto work = [&](int q0, int q1) {
        kv_heap<uint32_t, int8_t> bh(s.R);
        for (int q = q0; q < q1; ++q) {
            bh.reset();
            if (status && status[q]) {
                if (sizes) sizes[q] = 0;
                continue;
            }
            if (s.heaps_ready && s.skipped_streams && s.h_heap_sizes[q] != 0xffffffffu) {   // replayed by replay_heap_kernel
                const uint32_t sz = s.h_heap_sizes[q];
                const uint64_t* hv = s.h_heaps + (size_t)q * s.R;
                if (sizes) sizes[q] = (int32_t)sz;
                for (uint32_t i = 0; i < sz; ++i) {
                    if (keys) keys[(size_t)q * s.R + i] = (uint32_t)hv[i];
                    if (values) values[(size_t)q * s.R + i] = (int8_t)(hv[i] >> 32);
                }
                continue;
            }
            bh.push(0, 127);  // db_query_4.cpp:276
            for (uint64_t i = s.out_off[q]; i < s.out_off[q + 1]; ++i)
                bh.push((uint32_t)s.out_entries[i], (int8_t)(s.out_entries[i] >> 32));
            if (sizes) sizes[q] = bh.size();
            if (keys) std::memcpy(keys + (size_t)q * s.R, bh.keys(), sizeof(uint32_t) * bh.size());
            if (values) std::memcpy(values + (size_t)q * s.R, bh.values(), bh.size());
        }
    };
    // queries are independent: large batches (IVF) are replayed by a few host threads, the caller still
    // drives the library from one thread
    const uint64_t pushes = s.out_off[s.nq];
    int nt = idx->replay_threads > 0 ? idx->replay_threads : (int)std::min<unsigned>(std::thread::hardware_concurrency(), 16);   // (10M-code list, 32 queries per step: 8 -> 16 threads 0.260 -> 0.231 ms per step)
    nt = std::max(1, std::min(nt, s.nq / 2));
    if (pushes < 4000) nt = 1;                                 // (waking the workers costs about as much as 4 K pushes)
    // tasks of a few queries each, handed out dynamically: candidate counts differ from query to query
    const int per = std::max(1, s.nq / (nt * 4));
    const int tasks = (s.nq + per - 1) / per;
    idx->pool.run(tasks, nt, [&](int t) { work(t * per, std::min(s.nq, (t + 1) * per)); });
    return QADC_OK;
}

}  // namespace

extern "C" {

const char* qadc_last_error(void) { return g_err.c_str(); }
const char* qadc_version(void) { return "qadc-mi355x 0.1 (gfx950)"; }

int qadc_index_create(qadc_index** out, int M, int device_id) {
    if (!out) return fail(QADC_E_ARG, "out is null");
    if (M != 16 && M != 32)
        return fail(QADC_E_ARG, "Unsupported (nsq,nsq_bits) configuration. Supported configurations are: (16,4) (32,4).");
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(QADC_E_HIP, "no HIP device: the Quick-ADC engine has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(QADC_E_ARG, "device_id out of range");
    HIPCHECK(hipSetDevice(device_id));
    qadc_index* idx = new qadc_index();
    idx->M = M;
    idx->cs = M / 2;
    idx->device = device_id;
    // Test hooks (tests/conftest.py runs every parity test through every scan path this way).  They only apply when
    // QADC_TEST_HOOKS=1 is set as well, so that a stray QADC_* variable in a deployment cannot change the scan path.
    const char* hooks = std::getenv("QADC_TEST_HOOKS");
    if (hooks && std::atoi(hooks) == 1) {
        if (const char* e = std::getenv("QADC_WGQ")) idx->wgq = std::atoi(e);   // force (2) / forbid (0) the one-workgroup-per-query path
        if (const char* e = std::getenv("QADC_WGQ_POLL")) idx->wgq_poll = std::atoi(e);     // the lone-small-batch shortcuts
        if (const char* e = std::getenv("QADC_WGQ_INLINE")) idx->wgq_inline = std::atoi(e);
        if (const char* e = std::getenv("QADC_WGQ_GROUP")) idx->wgq_group = std::max(0, std::min(std::atoi(e), 2));
        if (const char* e = std::getenv("QADC_REPLAY_WAVE")) idx->replay_wave = std::atoi(e) != 0;
        if (const char* e = std::getenv("QADC_HEAD_LEVEL")) idx->head_level = std::max(0, std::min(std::atoi(e), kMaxLevels - 1));
    }
    // The streaming launches fill every CU for milliseconds.  They go on the LOWEST-priority queue so that the
    // short work that must overlap them is dispatched as soon as a workgroup slot frees up instead of waiting for
    // the whole batch: the previous batch's candidate sort, the next batch's front (own streams, highest priority) and
    // the caller's own streams (the RCCL gather of the multi-GPU merge).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&idx->stream, hipStreamNonBlocking, prio_least);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&idx->copy_stream, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&idx->sort_stream, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&idx->front_stream, hipStreamNonBlocking, prio_greatest);
    if (e != hipSuccess) {
        delete idx;
        return fail(QADC_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = idx;
    return QADC_OK;
}

int qadc_index_destroy(qadc_index* idx) {
    if (!idx) return QADC_OK;
    (void)hipSetDevice(idx->device);
    // a pre-scan (front stream) or an on-demand copy may still be in flight: drain all four streams before freeing
    for (hipStream_t st : {idx->stream, idx->front_stream, idx->copy_stream, idx->sort_stream})
        if (st) (void)hipStreamSynchronize(st);
    (void)qadc_dist_shutdown(idx);      // drains the merge's stream and frees the communicator while the index's streams and
                                        // the slot buffers the pack kernel reads are still alive
    for (auto& p : idx->parts) {
        if (p.own) {
            if (p.d_codes) (void)hipFree(p.d_codes);
            if (p.d_labels) (void)hipFree(p.d_labels);
        }
        if (p.d_starts) (void)hipFree(p.d_starts);
    }
    idx->d_codebooks.release();
    idx->d_rotation.release();
    idx->d_coarse.release();
    idx->d_partdesc.release();
    Slot* all_slots[kSlots + 2];
    for (int i = 0; i < kSlots; ++i) all_slots[i] = &idx->slot[i];
    all_slots[kSlots] = &idx->pre_slot[0];
    all_slots[kSlots + 1] = &idx->pre_slot[1];
    for (Slot* sp : all_slots) {
        Slot& s = *sp;
        s.d_in.release(); s.h_in.release(); s.d_state.release(); s.h_result.release();
        s.d_ftables.release(); s.d_qtables.release(); s.d_cands.release(); s.d_fc.release();
        s.h_cands.release(); s.d_stream.release(); s.d_qflags.release(); s.d_fvals.release(); s.d_qcands.release(); s.h_fetch.release();
        s.d_fblock.release(); s.d_fgathered.release(); s.d_front_all.release(); s.h_fmap.release();
        if (s.ev_fa) (void)hipEventDestroy(s.ev_fa);
        if (s.ev_fb) (void)hipEventDestroy(s.ev_fb);
        if (s.ev_assign) (void)hipEventDestroy(s.ev_assign);
        s.d_queries.release(); s.d_assign.release(); s.d_cdist.release(); s.h_queries.release(); s.h_assign.release();
        if (s.ev_feed) (void)hipEventDestroy(s.ev_feed);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_up) (void)hipEventDestroy(s.ev_up);
        if (s.ev_front) (void)hipEventDestroy(s.ev_front);
        if (s.ev_scanned) (void)hipEventDestroy(s.ev_scanned);
        for (auto e : s.prof_ev) (void)hipEventDestroy(e);
    }
    (void)hipStreamDestroy(idx->stream);
    if (idx->copy_stream) (void)hipStreamDestroy(idx->copy_stream);
    if (idx->front_stream) (void)hipStreamDestroy(idx->front_stream);
    if (idx->sort_stream) (void)hipStreamDestroy(idx->sort_stream);
    delete idx;
    return QADC_OK;
}

static int check_labels_mode(qadc_index* idx, bool has_labels) {
    if (idx->labeled < 0) idx->labeled = has_labels ? 1 : 0;
    if ((idx->labeled == 1) != has_labels)
        return fail(QADC_E_ARG, "Cannot prepare database. Some partitions have labels and some have not");
    return QADC_OK;
}

static int alloc_part(qadc_index* idx, Part& pt, uint32_t n, bool labels) {
    pt.n = n;
    pt.global_n = n;
    pt.starts_cap = n;
    const size_t bytes = ((size_t)n * idx->cs + 15) / 16 * 16 + 64;  // tail padding for 16-byte vector reads
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_codes), bytes));
    HIPCHECK(hipMemsetAsync(pt.d_codes + (bytes - 80), 0, 80, idx->stream));
    if (labels) HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_labels), std::max<size_t>(n, 1) * sizeof(uint32_t)));
    return QADC_OK;
}

int qadc_index_add_partitions(qadc_index* idx, int part_count, const uint8_t* const* codes,
                              const uint32_t* const* labels, const uint32_t* sizes) {
    if (!idx || part_count < 0 || !sizes || (part_count && !codes)) return fail(QADC_E_ARG, "bad arguments");
    if (int rc = use_device(idx)) return rc;
    for (int p = 0; p < part_count; ++p) {
        Part pt;
        const bool has_labels = labels != nullptr && labels[p] != nullptr;
        if (sizes[p] == 0) {  // "Warning: Partition i is empty" (db_query_4.cpp:113-116)
            idx->parts.push_back(pt);
            continue;
        }
        if (int rc = check_labels_mode(idx, has_labels)) return rc;
        if (int rc = alloc_part(idx, pt, sizes[p], has_labels)) return rc;
        HIPCHECK(hipMemcpyAsync(pt.d_codes, codes[p], (size_t)sizes[p] * idx->cs, hipMemcpyHostToDevice, idx->stream));
        if (has_labels)
            HIPCHECK(hipMemcpyAsync(pt.d_labels, labels[p], (size_t)sizes[p] * 4, hipMemcpyHostToDevice, idx->stream));
        HIPCHECK(hipStreamSynchronize(idx->stream));
        idx->parts.push_back(pt);
    }
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_interleaved(qadc_index* idx, const uint8_t* interleaved, const uint32_t* labels, uint32_t size) {
    if (!idx || (size && !interleaved)) return fail(QADC_E_ARG, "bad arguments");
    if (int rc = use_device(idx)) return rc;
    Part pt;
    if (size == 0) {
        idx->parts.push_back(pt);
        return QADC_OK;
    }
    if (int rc = check_labels_mode(idx, labels != nullptr)) return rc;
    if (int rc = alloc_part(idx, pt, size, labels != nullptr)) return rc;
    const size_t ibytes = (size_t)((size + 15u) / 16u) * idx->cs * 16;
    uint8_t* d_tmp = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_tmp), ibytes));
    HIPCHECK(hipMemcpyAsync(d_tmp, interleaved, ibytes, hipMemcpyHostToDevice, idx->stream));
    launch_deinterleave(pt.d_codes, d_tmp, size, idx->cs, idx->stream);
    if (labels) HIPCHECK(hipMemcpyAsync(pt.d_labels, labels, (size_t)size * 4, hipMemcpyHostToDevice, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    HIPCHECK(hipFree(d_tmp));
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_device(qadc_index* idx, const void* d_codes, const void* d_labels, uint32_t size) {
    if (!idx || (size && !d_codes)) return fail(QADC_E_ARG, "bad arguments");
    if ((reinterpret_cast<uintptr_t>(d_codes) & 15u) != 0) return fail(QADC_E_ARG, "d_codes must be 16-byte aligned");
    Part pt;
    if (size) {
        if (int rc = check_labels_mode(idx, d_labels != nullptr)) return rc;
        pt.d_codes = const_cast<uint8_t*>(static_cast<const uint8_t*>(d_codes));
        pt.d_labels = const_cast<uint32_t*>(static_cast<const uint32_t*>(d_labels));
        pt.n = size;
        pt.global_n = size;
        pt.starts_cap = size;
        pt.own = false;
    }
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_synthetic(qadc_index* idx, uint32_t size, uint64_t seed, uint64_t first_word) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (int rc = use_device(idx)) return rc;
    Part pt;
    if (size) {
        if (int rc = check_labels_mode(idx, false)) return rc;
        if (int rc = alloc_part(idx, pt, size, false)) return rc;
        const uint64_t nwords = ((uint64_t)size * idx->cs + 7) / 8;
        launch_fill_codes(pt.d_codes, first_word, nwords, seed, idx->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

static int attach_starts(qadc_index* idx, Part& pt, const uint8_t* starts_host, uint32_t starts_count, uint64_t seed,
                         bool synthetic) {
    if (pt.n && pt.first_pos == 0 && starts_count <= pt.n) {  // the local range begins with the starts
        pt.starts_cap = pt.n;
        return QADC_OK;
    }
    if (starts_count == 0) return fail(QADC_E_ARG, "a shard that does not begin the partition needs a starts replica");
    const size_t bytes = ((size_t)starts_count * idx->cs + 15) / 16 * 16 + 64;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_starts), bytes));
    if (synthetic) {
        launch_fill_codes(pt.d_starts, 0, ((uint64_t)starts_count * idx->cs + 7) / 8, seed, idx->stream);
        HIPCHECK(hipGetLastError());
    } else {
        HIPCHECK(hipMemcpyAsync(pt.d_starts, starts_host, (size_t)starts_count * idx->cs, hipMemcpyHostToDevice, idx->stream));
    }
    HIPCHECK(hipStreamSynchronize(idx->stream));
    pt.starts_cap = starts_count;
    return QADC_OK;
}

int qadc_index_add_partition_shard(qadc_index* idx, const uint8_t* codes, const uint32_t* labels, uint32_t local_n,
                                   uint32_t global_n, uint32_t first_pos, const uint8_t* starts, uint32_t starts_count) {
    if (!idx || (local_n && !codes) || global_n == 0 || (uint64_t)first_pos + local_n > global_n)
        return fail(QADC_E_ARG, "bad shard range");
    if ((uint64_t)first_pos * idx->cs % 16 != 0) return fail(QADC_E_ARG, "first_pos must keep the shard 16-byte aligned");
    if (int rc = use_device(idx)) return rc;
    if (local_n)
        if (int rc = check_labels_mode(idx, labels != nullptr)) return rc;
    Part pt;
    if (local_n) {
        if (int rc = alloc_part(idx, pt, local_n, labels != nullptr)) return rc;
        HIPCHECK(hipMemcpyAsync(pt.d_codes, codes, (size_t)local_n * idx->cs, hipMemcpyHostToDevice, idx->stream));
        if (labels) HIPCHECK(hipMemcpyAsync(pt.d_labels, labels, (size_t)local_n * 4, hipMemcpyHostToDevice, idx->stream));
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    pt.global_n = global_n;      // local_n == 0: the rank holds only the partition's starts replica
    pt.first_pos = first_pos;
    if (int rc = attach_starts(idx, pt, starts, starts_count, 0, false)) return rc;
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_synthetic_shard(qadc_index* idx, uint32_t global_n, uint32_t first_pos, uint32_t local_n,
                                             uint64_t seed, uint32_t starts_count) {
    if (!idx || global_n == 0 || (uint64_t)first_pos + local_n > global_n) return fail(QADC_E_ARG, "bad shard range");
    if ((uint64_t)first_pos * idx->cs % 16 != 0) return fail(QADC_E_ARG, "first_pos must keep the shard 16-byte aligned");
    if (int rc = use_device(idx)) return rc;
    if (int rc = check_labels_mode(idx, false)) return rc;
    Part pt;
    if (local_n) {                                              // (local_n == 0: only the partition's starts replica lives here)
        if (int rc = alloc_part(idx, pt, local_n, false)) return rc;
        launch_fill_codes(pt.d_codes, (uint64_t)first_pos * idx->cs / 8, ((uint64_t)local_n * idx->cs + 7) / 8, seed, idx->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    pt.global_n = global_n;
    pt.first_pos = first_pos;
    if (int rc = attach_starts(idx, pt, nullptr, starts_count, seed, true)) return rc;
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_set_key_base(qadc_index* idx, int part, uint32_t key_base) {
    if (!idx || part < 0 || part >= (int)idx->parts.size()) return fail(QADC_E_ARG, "bad partition");
    idx->parts[part].key_base = key_base;
    if (idx->finalized && (size_t)part < idx->h_partdesc.size()) idx->h_partdesc[part].key_base = key_base;
    if (idx->finalized && idx->d_partdesc.p) {                  // keep the device partition table in step
        if (int rc = use_device(idx)) return rc;
        HIPCHECK(hipMemcpy(reinterpret_cast<unsigned char*>(idx->d_partdesc.p + part) + offsetof(PartDesc, key_base), &key_base,
                           sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return QADC_OK;
}

int qadc_index_finalize(qadc_index* idx, float keep) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (idx->parts.empty()) return fail(QADC_E_STATE, "no partitions");
    idx->keep = keep;
    for (auto& p : idx->parts) {
        if (p.global_n == 0) {
            p.start_n = 0;
            continue;
        }
        // std::max(1u, static_cast<unsigned>(size * keep)) with unsigned*float -> float (db_query_4.cpp:125-126)
        const float prod = static_cast<float>(p.global_n) * keep;
        const unsigned s = static_cast<unsigned>(prod);
        p.start_n = std::min<uint32_t>(std::max(1u, s), p.global_n);
        if (p.start_n > p.starts_cap)
            return fail(QADC_E_ARG, "shard holds fewer start codes than max(1, unsigned(global_size * keep))");
    }
    // device partition table for the one-workgroup-per-query kernel
    if (int rc = use_device(idx)) return rc;
    std::vector<PartDesc>& pd = idx->h_partdesc;
    pd.assign(idx->parts.size(), PartDesc{});
    idx->max_start_n = 0;
    idx->total_codes = 0;
    idx->max_part_n = 0;
    for (size_t i = 0; i < idx->parts.size(); ++i) {
        const Part& p = idx->parts[i];
        PartDesc& d = pd[i];
        d.codes = p.d_codes;
        d.labels = p.d_labels;
        d.starts = p.d_starts;
        d.n = p.n;
        d.global_n = p.global_n;
        d.first_pos = p.first_pos;
        d.key_base = p.key_base;
        d.start_n = p.start_n;
        d.pad = 0;
        idx->max_start_n = std::max(idx->max_start_n, p.start_n);
        idx->max_part_n = std::max(idx->max_part_n, p.n);
        idx->total_codes += p.n;
    }
    HIPCHECK(idx->d_partdesc.ensure(pd.size()));
    HIPCHECK(hipMemcpy(idx->d_partdesc.p, pd.data(), pd.size() * sizeof(PartDesc), hipMemcpyHostToDevice));
    idx->finalized = true;
    return QADC_OK;
}

int qadc_index_partition_count(const qadc_index* idx) { return idx ? (int)idx->parts.size() : 0; }
uint32_t qadc_index_partition_size(const qadc_index* idx, int part) {
    return (idx && part >= 0 && part < (int)idx->parts.size()) ? idx->parts[part].n : 0;
}
uint32_t qadc_index_start_size(const qadc_index* idx, int part) {
    return (idx && part >= 0 && part < (int)idx->parts.size()) ? idx->parts[part].start_n : 0;
}

int qadc_set_option(qadc_index* idx, const char* name, double value) {
    if (!idx || !name) return fail(QADC_E_ARG, "bad arguments");
    const std::string n(name);
    if (n == "quant_mode") idx->quant_mode = value != 0 ? 1 : 0;
    else if (n == "cand_capacity") idx->cand_capacity = (uint32_t)std::max(16.0, std::min(value, 2147483648.0 - 1));
    else if (n == "level_base") idx->level_base = (uint64_t)std::max(16.0, value);
    else if (n == "level_growth") idx->level_growth = (uint64_t)std::max(2.0, value);
    else if (n == "wgs_per_item") idx->wgs_per_item = (int)value;
    else if (n == "overlap_front") idx->overlap_front = value != 0;
    else if (n == "head_early") idx->head_early = value != 0;
    else if (n == "wgq_inline") idx->wgq_inline = value != 0;
    else if (n == "replay_wave") idx->replay_wave = value != 0;
    else if (n == "mq_narrow") idx->mq_narrow = value != 0;
    else if (n == "device_replay_alone_nq") idx->device_replay_alone_nq = (int)std::max(0.0, value);
    else if (n == "wgq_group") { idx->wgq_group = (int)std::max(0.0, std::min(value, 2.0)); idx->group_strikes = 0; }
    else if (n == "wgq_group_head") idx->wgq_group_head = idx->wgq_group_head_dist = (int)std::max(1.0, std::min(value, 4096.0));
    else if (n == "wgq_group_head_dist") idx->wgq_group_head_dist = (int)std::max(1.0, std::min(value, 4096.0));
    else if (n == "wgq_poll") idx->wgq_poll = value != 0;
    else if (n == "wgq_split_codes") idx->wgq_split_codes = (uint32_t)std::max(value, 1024.0);
    else if (n == "front_dist") idx->front_dist = value != 0;
    else if (n == "share_variant") idx->share_variant = (int)value;
    else if (n == "mq") idx->mq = value != 0;
    else if (n == "prescan_mq") idx->prescan_mq = value != 0;
    else if (n == "front_run_max") idx->front_run_max = (uint64_t)std::max(value, 0.0);
    else if (n == "device_replay_nq") idx->device_replay_nq = (int)std::max(value, 0.0);
    else if (n == "front_min_batch") idx->front_min_batch = (uint64_t)std::max(value, 0.0);
    else if (n == "mq_codes_per_wg") idx->mq_codes_per_wg = (uint32_t)std::max(value, 4096.0);
    else if (n == "mq_min_wgs") idx->mq_min_wgs = (uint32_t)std::max(value, 1.0);
    else if (n == "mq_min_tiles") idx->mq_min_tiles = (uint32_t)std::max(value, 1.0);
    else if (n == "share_codes_per_wg") idx->share_codes_per_wg = (uint32_t)std::max(value, 4096.0);
    else if (n == "variant") idx->variant = (int)value;
    else if (n == "prescan_sample") idx->prescan_sample = (uint32_t)std::max(256.0, value);
    else if (n == "replay_threads") idx->replay_threads = (int)value;
    else if (n == "small_vec_per_wg") idx->small_vec_per_wg = (uint32_t)std::max(256.0, value);
    else if (n == "small_run") idx->small_run = (uint32_t)std::max(0.0, value);
    else if (n == "wgq") idx->wgq = (int)value;
    else if (n == "wgq_variant") idx->wgq_variant = (int)value;
    else if (n == "dist_cap_entries") {                       // entries per rank block of the native gather (test knob)
        if (!idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
        idx->dist->cap_entries = (uint32_t)std::max(16.0, std::min(value, 1073741824.0));
    }
    else if (n == "dist_device_nq") {                         // batches of at least this many queries replay on the device
        if (!idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
        idx->dist->device_nq = (int)std::max(1.0, value);
    }
    else if (n == "dist_async") {                             // 1: enqueue the merge with the batch where possible; 0: always at collect time
        if (!idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
        idx->dist->async_merge = value != 0;
    }
    else if (n == "dist_shard_front") {                       // 1: feeders + pre-scan + quantizer of a qadc_search batch are split over the ranks
        if (!idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
        idx->dist->shard_front = value != 0;
    }
    else if (n == "dist_inject_failure") {                    // test hook: this rank's next qadc_dist_collect fails before the gather
        if (!idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
        idx->dist->inject_failure = value != 0;
    }
    else if (n == "table_form") idx->table_form = std::max(0, std::min((int)value, 2));
    else if (n == "wgq_group_cand_cap") idx->wgq_group_cand_cap = (uint32_t)std::max(64.0, std::min(value, (double)kOrderCandCap));
    else if (n == "wgq_cand_cap") idx->wgq_cand_cap = (uint32_t)std::max(1.0, std::min(value, (double)kQueryCandCap));
    else if (n == "wgq_min_nq") idx->wgq_min_nq = (int)std::max(value, 1.0);
    else if (n == "wgq_max_codes") idx->wgq_max_codes = (uint64_t)std::max(value, 0.0);
    else if (n == "wgq_small_codes") idx->wgq_small_codes = (uint64_t)std::max(value, 0.0);
    else if (n == "head_level") idx->head_level = (int)std::max(0.0, std::min(value, (double)(kMaxLevels - 1)));
    else if (n == "wgq_split") idx->wgq_split = (int)std::max(1.0, std::min(value, 64.0));
    else if (n == "wgq_capacity") idx->wgq_capacity = (uint32_t)std::max(16.0, std::min(value, 1048576.0));
    else if (n == "profile") idx->profile = value != 0;
    else return fail(QADC_E_ARG, "unknown option: " + n);
    if (n == "cand_capacity") for (auto& sl : idx->slot) sl.cap_q = 0;
    if (n == "wgq_capacity") for (auto& sl : idx->slot) sl.wgq_cap = 0;
    return QADC_OK;
}

int qadc_index_read_codes(qadc_index* idx, int part, uint32_t first, uint32_t count, uint8_t* out) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !out) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if ((uint64_t)first + count > p.n) return fail(QADC_E_ARG, "range outside the partition");
    if (int rc = use_device(idx)) return rc;
    HIPCHECK(hipMemcpy(out, p.d_codes + (size_t)first * idx->cs, (size_t)count * idx->cs, hipMemcpyDeviceToHost));
    return QADC_OK;
}

int qadc_query_scan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R);
}

/* ---- sharded pre-scan (multi-GPU): every rank pre-scans one slice of the starts, the R smallest values per query
 * are gathered by the caller (RCCL), and the batch is submitted with the gathered values in place of its own
 * pre-scan.  The R-th smallest of the union of the per-slice R smallest IS the R-th smallest of all starts, so
 * qmax — and everything after it — is bit-identical to the unsharded path. ---- */
int qadc_prescan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R,
                        int slice, int nslices) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R, 1, slice, nslices);
}

int qadc_prescan_collect(qadc_index* idx, int slot, float* vals) {
    if (!idx || slot < 0 || slot > 1 || !vals) return fail(QADC_E_ARG, "bad arguments");
    Slot& s = idx->pre_slot[slot];
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no pre-scan");
    if (int rc = use_device(idx)) return rc;
    for (int attempt = 0;; ++attempt) {
        HIPCHECK(hipEventSynchronize(s.ev_done));
        bool overflow = false;
        for (int q = 0; q < s.nq; ++q) overflow |= (s.h_export_flags[q] & 8u) != 0;
        if (!overflow) break;
        if (attempt >= 1) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "pre-scan survivor buffer overflow persists");
        }
        s.full_prescan = true;               // adversarially ordered starts: evaluate the slice unfiltered
        idx->prof.regrows++;
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    s.busy = false;
    if (idx->profile && s.prof_used >= 2) {
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[1]));
        idx->prof.start_ms += ms;
        idx->prof.start_codes += s.start_codes;
    }
    std::memcpy(vals, s.h_export, sizeof(float) * (size_t)s.nq * s.R);
    return QADC_OK;
}

int qadc_query_scan_submit_prescanned(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables,
                                      int R, const float* prescan_vals, int nvals) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R, 2, 0, 1, prescan_vals, nvals);
}

int qadc_query_scan_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                            float* qmin, float* qmax, int8_t* qtables) {
    if (int rc = collect_common(idx, slot, /*need_stream=*/false)) return rc;
    Slot& s = idx->slot[slot];
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(s.nq);
        status = st_local.data();
    }
    finish_float_outputs(idx, s, status, qmin, qmax);
    if (qtables) {
        HIPCHECK(hipMemcpyAsync(qtables, s.d_qt, (size_t)s.nq * s.ma * idx->M * 16, hipMemcpyDeviceToHost,
                                idx->copy_stream));
        HIPCHECK(hipStreamSynchronize(idx->copy_stream));
    }
    return replay_outputs(idx, s, keys, values, sizes, status);
}

static int copy_stream(Slot& s, uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                       uint16_t* cand_slots = nullptr);

int qadc_query_scan_collect_candidates(qadc_index* idx, int slot, uint64_t cand_capacity, uint32_t* cand_keys,
                                       int8_t* cand_vals, uint16_t* cand_slots, uint64_t* offsets, int32_t* status,
                                       float* qmin, float* qmax) {
    if (!idx || slot < 0 || slot >= kSlots) return fail(QADC_E_ARG, "bad arguments");
    Slot& s = idx->slot[slot];
    if (s.busy) {
        if (int rc = collect_common(idx, slot)) return rc;
        s.has_result = true;
    } else if (!s.has_result) {
        return fail(QADC_E_STATE, "slot holds no batch");
    }
    finish_float_outputs(idx, s, status, qmin, qmax);
    // QADC_E_CAPACITY keeps the result: call again with buffers of offsets[nq] entries
    const int rc = copy_stream(s, cand_capacity, cand_keys, cand_vals, offsets, cand_slots);
    if (rc == QADC_OK) s.has_result = false;
    return rc;
}

int qadc_query_scan(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R, uint32_t* keys,
                    int8_t* values, int32_t* sizes, int32_t* status, float* qmin, float* qmax, int8_t* qtables) {
    if (int rc = qadc_query_scan_submit(idx, 0, nq, ma, assign, tables, R)) return rc;
    return qadc_query_scan_collect(idx, 0, keys, values, sizes, status, qmin, qmax, qtables);
}

static int copy_stream(Slot& s, uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                       uint16_t* cand_slots) {
    if (!offsets) return fail(QADC_E_ARG, "offsets is null");
    std::memcpy(offsets, s.out_off.data(), sizeof(uint64_t) * (s.nq + 1));
    const uint64_t total = s.out_off[s.nq];
    if (total > cand_capacity) return fail(QADC_E_CAPACITY, "candidate output buffers too small (see offsets[nq])");
    for (uint64_t i = 0; i < total; ++i) {
        if (cand_keys) cand_keys[i] = (uint32_t)s.out_entries[i];
        if (cand_vals) cand_vals[i] = (int8_t)(s.out_entries[i] >> 32);
        if (cand_slots) cand_slots[i] = (uint16_t)((s.out_entries[i] >> 40) & 0x3fffu);
    }
    return QADC_OK;
}

int qadc_query_scan_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R,
                               uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                               int32_t* status, float* qmin, float* qmax) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, tables, nullptr, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    Slot& s = idx->slot[0];
    std::vector<int32_t> st_local(nq);
    finish_float_outputs(idx, s, st_local.data(), qmin, qmax);
    if (status) std::memcpy(status, st_local.data(), sizeof(int32_t) * nq);
    for (int q = 0; q < nq; ++q)
        if (st_local[q]) {  // the reference never reaches the scan for such a query
            // keep offsets monotone but expose no candidates: handled by the caller through status
        }
    return copy_stream(s, cand_capacity, cand_keys, cand_vals, offsets);
}

int qadc_scan_i8(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R, uint32_t* keys,
                 int8_t* values, int32_t* sizes) {
    if (!qtables) return fail(QADC_E_ARG, "qtables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, nullptr, qtables, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    return replay_outputs(idx, idx->slot[0], keys, values, sizes, nullptr);
}

int qadc_scan_i8_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R,
                            uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets) {
    if (!qtables) return fail(QADC_E_ARG, "qtables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, nullptr, qtables, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    return copy_stream(idx->slot[0], cand_capacity, cand_keys, cand_vals, offsets);
}

int qadc_scan_start(qadc_index* idx, int nq, int ma, const int32_t* assign, const float* tables, int R, float* qmax) {
    if (!idx || !tables || !qmax || !assign || nq <= 0 || ma <= 0 || R <= 0) return fail(QADC_E_ARG, "bad arguments");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    // Runs the float chain of a batch and reads back qmax only (copy: the caller's tables stay untouched).
    std::vector<float> copy(tables, tables + (size_t)nq * ma * idx->M * 16);
    std::vector<float> qm(nq);
    if (int rc = submit_common(idx, 0, nq, ma, assign, copy.data(), nullptr, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    for (int q = 0; q < nq; ++q) qmax[q] = idx->slot[0].h_qout[(size_t)q * (idx->slot[0].wgq ? idx->slot[0].wgq_G : 1)].qmax;
    return QADC_OK;
}

int qadc_replay_i8(uint64_t n, const uint32_t* keys, const int8_t* vals, int R, int push_sentinel, uint32_t* out_keys,
                   int8_t* out_vals, int32_t* out_size) {
    if (R <= 0 || (n && (!keys || !vals)) || !out_size) return fail(QADC_E_ARG, "bad arguments");
    kv_heap<uint32_t, int8_t> bh(R);
    if (push_sentinel) bh.push(0, 127);
    for (uint64_t i = 0; i < n; ++i) bh.push(keys[i], vals[i]);
    *out_size = bh.size();
    if (out_keys) std::memcpy(out_keys, bh.keys(), sizeof(uint32_t) * bh.size());
    if (out_vals) std::memcpy(out_vals, bh.values(), bh.size());
    return QADC_OK;
}

int qadc_sort_keys_i8(int size, const uint32_t* heap_keys, const int8_t* heap_vals, uint32_t* out_keys) {
    if (size < 0 || (size && (!heap_keys || !heap_vals || !out_keys))) return fail(QADC_E_ARG, "bad arguments");
    kv_heap<uint32_t, int8_t> bh(std::max(size, 1));
    bh.assign(heap_keys, heap_vals, size);
    bh.sort_keys(out_keys);
    return QADC_OK;
}

// Host half of the multi-GPU merge (pyqadc/sharded.py): `gathered` holds, rank after rank, the int32 buffers the
// ranks contributed to the all-gather: [nq counts][cap keys][ceil(cap/4) words of int8 values]
// [only when ma > 1: ceil(cap/2) words of u16 assign slots][...].  Queries q_first, q_first + q_step, ... are replayed here in
// global scan order (assign slot, rank, position) through the reference heap after the (0,127) sentinel.
int qadc_merge_streams_i8(int world, int nq, int R, uint64_t cap, int ma, const int32_t* gathered, uint64_t buflen,
                          int q_first, int q_step, const int32_t* status, uint32_t* keys, int8_t* vals, int32_t* sizes) {
    if (world <= 0 || nq <= 0 || R <= 0 || ma <= 0 || !gathered || !keys || !vals || !sizes || q_step <= 0 || q_first < 0)
        return fail(QADC_E_ARG, "bad arguments");
    const uint64_t nv = (cap + 3) / 4, ns = ma > 1 ? (cap + 1) / 2 : 0;
    if (buflen < (uint64_t)nq + cap + nv + ns) return fail(QADC_E_ARG, "gathered buffers shorter than their layout");
    // entry offsets of every (rank, query)
    std::vector<uint64_t> offs((size_t)world * (nq + 1), 0);
    for (int g = 0; g < world; ++g) {
        const int32_t* cnt = gathered + (uint64_t)g * buflen;
        uint64_t* o = offs.data() + (size_t)g * (nq + 1);
        for (int q = 0; q < nq; ++q) o[q + 1] = o[q] + (uint32_t)cnt[q];
        if (o[nq] > cap) return fail(QADC_E_CAPACITY, "a rank's stream exceeds the gathered capacity");
    }
    std::vector<int> mine;
    for (int q = q_first; q < nq; q += q_step) mine.push_back(q);
    auto work = [&](size_t i0, size_t i1) {
        kv_heap<uint32_t, int8_t> bh(R);
        std::vector<uint64_t> cur(world);
        for (size_t i = i0; i < i1; ++i) {
            const int q = mine[i];
            sizes[q] = 0;
            if (status && status[q]) continue;
            bh.reset();
            bh.push(0, 127);                                     // db_query_4.cpp:276
            for (int g = 0; g < world; ++g) cur[g] = offs[(size_t)g * (nq + 1) + q];
            for (int slot = 0; slot < ma; ++slot)
                for (int g = 0; g < world; ++g) {
                    const int32_t* base = gathered + (uint64_t)g * buflen;
                    const uint32_t* k = reinterpret_cast<const uint32_t*>(base + nq);
                    const int8_t* v = reinterpret_cast<const int8_t*>(base + nq + cap);
                    const uint16_t* sl = reinterpret_cast<const uint16_t*>(base + nq + cap + nv);
                    const uint64_t end = offs[(size_t)g * (nq + 1) + q + 1];
                    uint64_t& c = cur[g];
                    // a rank scans its partitions in assign order: its slots are ascending
                    while (c < end && (ma == 1 || sl[c] == (uint16_t)slot)) {
                        bh.push(k[c], v[c]);
                        ++c;
                    }
                }
            sizes[q] = bh.size();
            std::memcpy(keys + (size_t)q * R, bh.keys(), sizeof(uint32_t) * bh.size());
            std::memcpy(vals + (size_t)q * R, bh.values(), bh.size());
        }
    };
    const size_t nt = std::min<size_t>(std::min<size_t>(mine.size(), 4), std::max<unsigned>(std::thread::hardware_concurrency(), 1));
    if (nt <= 1) {
        work(0, mine.size());
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t) th.emplace_back(work, mine.size() * t / nt, mine.size() * (t + 1) / nt);
        for (auto& x : th) x.join();
    }
    return QADC_OK;
}

int qadc_candidates_i8(qadc_index* idx, int part, const int8_t* qtable, int8_t* out) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !qtable || !out) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if (p.n == 0) return QADC_OK;
    for (int i = 0; i < idx->M * 16; ++i)
        if (qtable[i] < 0) return fail(QADC_E_ARG, "int8 tables must lie in [0,127]");
    if (int rc = use_device(idx)) return rc;
    int8_t *d_t = nullptr, *d_o = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_t), idx->M * 16));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_o), p.n));
    HIPCHECK(hipMemcpyAsync(d_t, qtable, idx->M * 16, hipMemcpyHostToDevice, idx->stream));
    launch_candidates_i8(idx->M, p.d_codes, p.n, d_t, d_o, idx->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(take_launch_error());
    HIPCHECK(hipMemcpyAsync(out, d_o, p.n, hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    HIPCHECK(hipFree(d_t));
    HIPCHECK(hipFree(d_o));
    return QADC_OK;
}

int qadc_float_top1(qadc_index* idx, int part, const float* table, uint32_t* out_key, uint32_t* out_pos, float* out_dist) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !table) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if (p.n == 0) return fail(QADC_E_ARG, "empty partition");
    if (int rc = use_device(idx)) return rc;
    const int blocks = (int)std::min<uint32_t>((p.n + 255) / 256, 2048);
    float *d_t = nullptr, *d_v = nullptr;
    uint32_t* d_p = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_t), idx->M * 16 * sizeof(float)));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_v), blocks * sizeof(float)));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_p), blocks * sizeof(uint32_t)));
    HIPCHECK(hipMemcpyAsync(d_t, table, idx->M * 16 * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    launch_float_top1(idx->M, p.d_codes, p.n, d_t, d_v, d_p, blocks, idx->stream);
    HIPCHECK(hipGetLastError());
    std::vector<float> hv(blocks);
    std::vector<uint32_t> hp(blocks);
    HIPCHECK(hipMemcpyAsync(hv.data(), d_v, blocks * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipMemcpyAsync(hp.data(), d_p, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    int b = 0;
    for (int i = 1; i < blocks; ++i)
        if (hv[i] < hv[b] || (hv[i] == hv[b] && hp[i] < hp[b])) b = i;
    uint32_t key = p.key_base + p.first_pos + hp[b];
    if (p.d_labels) HIPCHECK(hipMemcpy(&key, p.d_labels + hp[b], sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (out_key) *out_key = key;
    if (out_pos) *out_pos = hp[b];
    if (out_dist) *out_dist = hv[b];
    HIPCHECK(hipFree(d_t));
    HIPCHECK(hipFree(d_v));
    HIPCHECK(hipFree(d_p));
    return QADC_OK;
}

int qadc_index_set_pq(qadc_index* idx, int dim, const float* codebooks) {
    if (!idx || !codebooks || dim <= 0 || dim % idx->M != 0) return fail(QADC_E_ARG, "dim must be a positive multiple of M");
    if (int rc = use_device(idx)) return rc;
    const size_t n = (size_t)idx->M * 16 * (dim / idx->M);
    HIPCHECK(idx->d_codebooks.ensure(n));
    HIPCHECK(hipMemcpy(idx->d_codebooks.p, codebooks, n * sizeof(float), hipMemcpyHostToDevice));
    idx->dim = dim;
    return QADC_OK;
}

int qadc_index_set_rotation(qadc_index* idx, const float* rotation) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (idx->dim == 0) return fail(QADC_E_STATE, "call qadc_index_set_pq first");
    if (!rotation) {
        idx->has_rotation = false;
        return QADC_OK;
    }
    if (int rc = use_device(idx)) return rc;
    const size_t n = (size_t)idx->dim * idx->dim;
    HIPCHECK(idx->d_rotation.ensure(n));
    HIPCHECK(hipMemcpy(idx->d_rotation.p, rotation, n * sizeof(float), hipMemcpyHostToDevice));
    idx->has_rotation = true;
    return QADC_OK;
}

int qadc_index_set_coarse(qadc_index* idx, int K, const float* centroids) {
    if (!idx || K <= 0 || !centroids) return fail(QADC_E_ARG, "bad arguments");
    if (idx->dim == 0) return fail(QADC_E_STATE, "call qadc_index_set_pq first");
    if (int rc = use_device(idx)) return rc;
    HIPCHECK(idx->d_coarse.ensure((size_t)K * idx->dim));
    HIPCHECK(hipMemcpy(idx->d_coarse.p, centroids, (size_t)K * idx->dim * sizeof(float), hipMemcpyHostToDevice));
    idx->K = K;
    return QADC_OK;
}

int qadc_search_submit(qadc_index* idx, int slot, int nq, const float* queries, int ma, int R) {
    return search_submit(idx, slot, nq, queries, ma, R);
}

int qadc_search_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                        int32_t* assign_out) {
    if (int rc = collect_common(idx, slot, /*need_stream=*/false)) return rc;
    Slot& s = idx->slot[slot];
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(s.nq);
        status = st_local.data();
    }
    finish_float_outputs(idx, s, status, nullptr, nullptr);
    if (assign_out) std::memcpy(assign_out, s.assign.data(), sizeof(int32_t) * s.assign.size());
    return replay_outputs(idx, s, keys, values, sizes, status);
}

int qadc_search(qadc_index* idx, int nq, const float* queries, int ma, int R, uint32_t* keys, int8_t* values, int32_t* sizes,
                int32_t* status, int32_t* assign_out) {
    if (int rc = search_submit(idx, 0, nq, queries, ma, R)) return rc;
    return qadc_search_collect(idx, 0, keys, values, sizes, status, assign_out);
}

int qadc_pq_encode(int M, int dim, const float* codebooks, const void* d_vectors, uint64_t n, void* d_codes, int device_id) {
    if ((M != 16 && M != 32) || dim <= 0 || dim % M != 0 || !codebooks || (n && (!d_vectors || !d_codes)))
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    const size_t ncb = (size_t)M * 16 * (dim / M);
    float* d_cb = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_cb), ncb * sizeof(float)));
    HIPCHECK(hipMemcpy(d_cb, codebooks, ncb * sizeof(float), hipMemcpyHostToDevice));
    if (n) launch_pq_encode(static_cast<const float*>(d_vectors), n, M, dim, d_cb, static_cast<uint8_t*>(d_codes), nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipFree(d_cb));
    return QADC_OK;
}

int qadc_pq_encode_host(int M, int dim, const float* codebooks, const float* vectors, uint64_t n, uint8_t* codes, int device_id) {
    if (!vectors || !codes) return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    float* d_v = nullptr;
    uint8_t* d_c = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_v), std::max<size_t>(1, n * dim * sizeof(float))));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_c), std::max<size_t>(1, n * (M / 2))));
    HIPCHECK(hipMemcpy(d_v, vectors, n * dim * sizeof(float), hipMemcpyHostToDevice));
    const int rc = qadc_pq_encode(M, dim, codebooks, d_v, n, d_c, device_id);
    if (rc == QADC_OK) HIPCHECK(hipMemcpy(codes, d_c, n * (M / 2), hipMemcpyDeviceToHost));
    HIPCHECK(hipFree(d_v));
    HIPCHECK(hipFree(d_c));
    return rc;
}

/* ---- database build (N4): index_db::add_vectors' compute and the k-means iterations, host buffers in and out ---- */
extern "C++" {
namespace {
struct ScratchFree {                                           // hipFree on every exit path
    std::vector<void*> p;
    ~ScratchFree() { for (void* x : p) if (x) (void)hipFree(x); }
    template <typename T> hipError_t alloc(T** out, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) { p.push_back(q); *out = static_cast<T*>(q); }
        return e;
    }
};
constexpr uint64_t kBuildChunk = 32768;                        // vectors per pass (the distance scratch is chunk x K floats)

// nearest centroid of every vector, chunk by chunk: the coarse kernels of qadc_search with ma = 1
int assign_nearest(const float* d_vectors, uint64_t n, int dim, int K, const float* d_coarse, float* d_dist, int32_t* d_assign) {
    for (uint64_t o = 0; o < n; o += kBuildChunk) {
        const int cnt = (int)std::min<uint64_t>(kBuildChunk, n - o);
        launch_coarse_assign(d_vectors + o * dim, d_coarse, cnt, K, dim, 1, d_dist, d_assign + o, nullptr);
    }
    HIPCHECK(hipGetLastError());
    return QADC_OK;
}
}  // namespace
}  // extern "C++"

int qadc_ivf_encode_host(int M, int dim, const float* codebooks, const float* rotation, int K, const float* coarse,
                         const float* vectors, uint64_t n, int32_t* assign_out, uint8_t* codes, int device_id) {
    if ((M != 16 && M != 32) || dim <= 0 || dim % M != 0 || !codebooks || K < 0 || (K > 0 && !coarse) ||
        (n && (!vectors || !codes)))
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    ScratchFree mem;
    const size_t ncb = (size_t)M * 16 * (dim / M);
    float *d_cb = nullptr, *d_rot = nullptr, *d_coarse = nullptr, *d_v = nullptr, *d_x = nullptr, *d_dist = nullptr;
    int32_t* d_assign = nullptr;
    uint8_t* d_codes = nullptr;
    HIPCHECK(mem.alloc(&d_cb, ncb * sizeof(float)));
    HIPCHECK(hipMemcpy(d_cb, codebooks, ncb * sizeof(float), hipMemcpyHostToDevice));
    if (rotation) {
        HIPCHECK(mem.alloc(&d_rot, sizeof(float) * (size_t)dim * dim));
        HIPCHECK(hipMemcpy(d_rot, rotation, sizeof(float) * (size_t)dim * dim, hipMemcpyHostToDevice));
    }
    if (K > 0) {
        HIPCHECK(mem.alloc(&d_coarse, sizeof(float) * (size_t)K * dim));
        HIPCHECK(hipMemcpy(d_coarse, coarse, sizeof(float) * (size_t)K * dim, hipMemcpyHostToDevice));
        HIPCHECK(mem.alloc(&d_dist, sizeof(float) * (size_t)std::min<uint64_t>(kBuildChunk, std::max<uint64_t>(n, 1)) * K));
        HIPCHECK(mem.alloc(&d_assign, sizeof(int32_t) * n));
    }
    HIPCHECK(mem.alloc(&d_v, sizeof(float) * n * dim));
    HIPCHECK(mem.alloc(&d_codes, n * (size_t)(M / 2)));
    HIPCHECK(hipMemcpy(d_v, vectors, sizeof(float) * n * dim, hipMemcpyHostToDevice));
    const float* d_enc = d_v;
    if (n && K > 0)
        if (int rc = assign_nearest(d_v, n, dim, K, d_coarse, d_dist, d_assign)) return rc;
    if (n && (K > 0 || rotation)) {
        HIPCHECK(mem.alloc(&d_x, sizeof(float) * n * dim));
        launch_residual_rotate(d_v, n, dim, d_coarse, d_assign, d_rot, d_x, nullptr);
        d_enc = d_x;
    }
    if (n) launch_pq_encode(d_enc, n, M, dim, d_cb, d_codes, nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(codes, d_codes, n * (size_t)(M / 2), hipMemcpyDeviceToHost));
    if (assign_out && K > 0) HIPCHECK(hipMemcpy(assign_out, d_assign, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return QADC_OK;
}

int qadc_kmeans_iterations_host(const float* vectors, uint64_t n, int dim, int K, float* centroids, int iters, int32_t* assign_out,
                                int device_id) {
    if (!vectors || !centroids || n == 0 || dim <= 0 || dim > 2048 || K <= 0 || iters < 0) return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    ScratchFree mem;
    float *d_v = nullptr, *d_c = nullptr, *d_dist = nullptr;
    int32_t* d_assign = nullptr;
    HIPCHECK(mem.alloc(&d_v, sizeof(float) * n * dim));
    HIPCHECK(mem.alloc(&d_c, sizeof(float) * (size_t)K * dim));
    HIPCHECK(mem.alloc(&d_dist, sizeof(float) * (size_t)std::min<uint64_t>(kBuildChunk, n) * K));
    HIPCHECK(mem.alloc(&d_assign, sizeof(int32_t) * n));
    HIPCHECK(hipMemcpy(d_v, vectors, sizeof(float) * n * dim, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_c, centroids, sizeof(float) * (size_t)K * dim, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(d_assign, 0, sizeof(int32_t) * n));
    for (int it = 0; it < iters; ++it) {                       // databases.cpp:57-89
        if (int rc = assign_nearest(d_v, n, dim, K, d_c, d_dist, d_assign)) return rc;
        launch_kmeans_update(d_v, n, dim, K, d_assign, d_c, nullptr);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(centroids, d_c, sizeof(float) * (size_t)K * dim, hipMemcpyDeviceToHost));
    if (assign_out) HIPCHECK(hipMemcpy(assign_out, d_assign, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return QADC_OK;
}

extern "C++" {
namespace {
// The host half of qadc_dist_collect for few-query batches: rank `rank` replays the queries q = rank, rank + world, ...
// of the gathered blocks ([nq x {offset, count, flags, -} as u32][entries ...] per rank, `bw` words each) in GLOBAL scan
// order — assign slot, then rank (= ascending code range), then position — through the reference's heap, sentinel first
// (db_query_4.cpp:276).  myheaps: [ceil(nq / world)][R + 1] words, heap entries key | value << 32, then the size.
void replay_my_share(const uint64_t* gathered, size_t bw, int world, int rank, int nq, int ma, int R, const int32_t* status,
                     uint64_t* myheaps, WorkerPool* pool) {
    const int per = (nq + world - 1) / world;
    const size_t hw = (size_t)R + 1;
    auto work = [&](int j0, int j1) {
        kv_heap<uint32_t, int8_t> bh(R);
        std::vector<uint32_t> cur(world), end(world);
        for (int j = j0; j < j1; ++j) {
            const int q = j * world + rank;
            if (q >= nq || (status && status[q])) continue;
            bh.reset();
            bh.push(0, 127);
            for (int g = 0; g < world; ++g) {
                const uint32_t* h = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * bw) + 4 * (size_t)q;
                cur[g] = h[0];
                end[g] = h[0] + h[1];
            }
            for (int slot = 0; slot < ma; ++slot)
                for (int g = 0; g < world; ++g) {
                    const uint64_t* ent = gathered + (size_t)g * bw + 2 * (size_t)nq;
                    while (cur[g] < end[g]) {                    // a rank scans its partitions in assign order: slots ascend
                        const uint64_t e = ent[cur[g]];
                        if (ma > 1 && (int)((e >> 40) & 0x3fffu) != slot) break;
                        bh.push((uint32_t)e, (int8_t)(e >> 32));
                        ++cur[g];
                    }
                }
            uint64_t* o = myheaps + (size_t)j * hw;
            for (int i = 0; i < bh.size(); ++i) o[i] = (uint64_t)bh.keys()[i] | ((uint64_t)(uint8_t)bh.values()[i] << 32);
            o[R] = (uint64_t)bh.size();
        }
    };
    const int nt = std::max(1, std::min<int>(std::min(per, 8), (int)std::thread::hardware_concurrency()));
    if (nt == 1 || !pool) {
        work(0, per);
    } else {
        pool->run(per, nt, [&](int j) { work(j, j + 1); });     // one query per task: their stream lengths differ
    }
}
}  // namespace
}  // extern "C++"

/* ---- native multi-GPU merge: one ncclAllGather per batch, device memory to device memory ---- */
int qadc_dist_unique_id(uint8_t* id128) {
    if (!id128) return fail(QADC_E_ARG, "id is null");
    DistState tmp;
    std::string err;
    if (load_rccl(tmp, err)) return fail(QADC_E_HIP, err);
    QadcNcclId id;
    const int rc = tmp.GetUniqueId(&id);
    if (rc != 0) return fail(QADC_E_HIP, std::string("ncclGetUniqueId: ") + (tmp.GetErrorString ? tmp.GetErrorString(rc) : "error"));
    std::memcpy(id128, id.internal, 128);
    return QADC_OK;                                          // (the library handle stays loaded for the process)
}

extern "C++" {
namespace {
// Releases a half-built DistState on every exit path of the init calls (communicator, stream).
struct DistGuard {
    std::unique_ptr<DistState> d;
    ~DistGuard() {
        if (!d) return;
        if (d->stream) { (void)hipStreamSynchronize(d->stream); }
        for (hipStream_t m : d->merge_stream) if (m) { (void)hipStreamSynchronize(m); }
        if (d->comm && d->CommDestroy) (void)d->CommDestroy(d->comm);
        if (d->stream) (void)hipStreamDestroy(d->stream);
        for (hipStream_t m : d->merge_stream) if (m) (void)hipStreamDestroy(m);
    }
};
int dist_init_checks(qadc_index* idx, int rank, int world) {
    if (!idx || world < 1 || world > 16 || rank < 0 || rank >= world) return fail(QADC_E_ARG, "need 0 <= rank < world <= 16");
    if (idx->dist) return fail(QADC_E_STATE, "qadc_dist_init was already called");
    for (auto& sl : idx->slot)
        if (sl.busy) return fail(QADC_E_STATE, "collect every batch before qadc_dist_init");
    return use_device(idx);
}
}  // namespace
}  // extern "C++"

int qadc_dist_init(qadc_index* idx, int rank, int world, const uint8_t* id128) {
    if (!id128) return fail(QADC_E_ARG, "id is null");
    if (int rc = dist_init_checks(idx, rank, world)) return rc;
    DistGuard g;
    g.d.reset(new DistState());
    DistState* d = g.d.get();
    std::string err;
    if (load_rccl(*d, err)) return fail(QADC_E_HIP, err);
    QadcNcclId id;
    std::memcpy(id.internal, id128, 128);
    const int rc = d->CommInitRank(&d->comm, world, id, rank);
    if (rc != 0) {
        d->comm = nullptr;
        return fail(QADC_E_HIP, std::string("ncclCommInitRank: ") + (d->GetErrorString ? d->GetErrorString(rc) : "error"));
    }
    d->rank = rank;
    d->world = world;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIPCHECK(hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, prio_greatest));
    // (NOT the collectives' priority: HIP multiplexes the streams of one priority over a few hardware queues, and on the same
    // queue as the collectives' stream a replay sat in front of the next batch's gather — seen in the trace.  And not ABOVE
    // the scans either: at normal priority the interleave / replay waves held up the short launches between two batches'
    // scans on the lowest-priority scan stream — one of 8 ranks, 1024-query batches, C5 shape 1.18 -> 1.10 ms, C3 0.70 ->
    // 0.60 with the merges at the scans' own, lowest priority.)
    for (int i = 0; i < kSlots; ++i)
        HIPCHECK(hipStreamCreateWithPriority(&d->merge_stream[i], hipStreamNonBlocking, prio_least));
    // RCCL finishes setting up its channels lazily, inside the first collectives of a communicator (the very first
    // all-gather takes ~8 ms); a few throw-away gathers here keep that out of the first batches' collect calls.
    {
        constexpr size_t kWords = 1 << 16;
        DevBuf<uint64_t> src, dst;
        hipError_t he = src.ensure(kWords);
        if (he == hipSuccess) he = dst.ensure(kWords * world);
        if (he == hipSuccess) he = hipMemsetAsync(src.p, 0, sizeof(uint64_t) * kWords, d->stream);
        int rc2 = 0;
        for (int i = 0; i < 16 && he == hipSuccess && rc2 == 0; ++i)
            rc2 = d->AllGather(src.p, dst.p, kWords, /*ncclUint64*/ 5, d->comm, d->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(d->stream);
        src.release(); dst.release();
        if (rc2 != 0) return fail(QADC_E_HIP, std::string("ncclAllGather: ") + (d->GetErrorString ? d->GetErrorString(rc2) : "error"));
        HIPCHECK(he);
    }
    idx->dist = g.d.release();
    return QADC_OK;
}

extern "C++" {
namespace {
// Measurement aid (qadc_dist_init_loopback): ONE rank stands in for a whole world — its block fills every slot of the
// gather, so the merge replays `world` ranks' worth of entries while only this rank's shard is scanned.
int loopback_allgather(void* ctx, const void* d_send, void* d_recv, uint64_t bytes, void* st) {
    const int world = (int)reinterpret_cast<intptr_t>(ctx);
    return launch_replicate_block(d_send, d_recv, (size_t)(bytes / 8), world, static_cast<hipStream_t>(st)) == hipSuccess ? 0 : 1;
}
}  // namespace
}  // extern "C++"

int qadc_dist_init_loopback(qadc_index* idx, int rank, int world) {
    return qadc_dist_init_transport(idx, rank, world, loopback_allgather, reinterpret_cast<void*>((intptr_t)world));
}

int qadc_dist_init_transport(qadc_index* idx, int rank, int world, qadc_allgather_fn fn, void* ctx) {
    if (!fn) return fail(QADC_E_ARG, "the all-gather callback is null");
    if (int rc = dist_init_checks(idx, rank, world)) return rc;
    DistGuard g;
    g.d.reset(new DistState());
    DistState* d = g.d.get();
    d->user_fn = fn;
    d->user_ctx = ctx;
    d->rank = rank;
    d->world = world;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIPCHECK(hipStreamCreateWithPriority(&d->stream, hipStreamNonBlocking, prio_greatest));
    // (NOT the collectives' priority: HIP multiplexes the streams of one priority over a few hardware queues, and on the same
    // queue as the collectives' stream a replay sat in front of the next batch's gather — seen in the trace.  And not ABOVE
    // the scans either: at normal priority the interleave / replay waves held up the short launches between two batches'
    // scans on the lowest-priority scan stream — one of 8 ranks, 1024-query batches, C5 shape 1.18 -> 1.10 ms, C3 0.70 ->
    // 0.60 with the merges at the scans' own, lowest priority.)
    for (int i = 0; i < kSlots; ++i)
        HIPCHECK(hipStreamCreateWithPriority(&d->merge_stream[i], hipStreamNonBlocking, prio_least));
    idx->dist = g.d.release();
    return QADC_OK;
}

int qadc_dist_merge_blocks(int device_id, int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                           uint32_t* keys, int8_t* values, int32_t* sizes) {
    if (world < 1 || world > 16 || nq <= 0 || ma <= 0 || R <= 0 || (uint32_t)R > replay_wave_max_R() ||
        (size_t)ma * world > dist_interleave_max_cells() || !gathered || !sizes)
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    DevBuf<uint64_t> d_g, d_h, d_off, d_m;
    DevBuf<uint32_t> d_s, d_c;
    HIPCHECK(d_g.ensure((size_t)block_words * world));
    HIPCHECK(d_h.ensure((size_t)nq * R));
    HIPCHECK(d_s.ensure(nq));
    HIPCHECK(d_off.ensure(nq));
    HIPCHECK(d_c.ensure(2 * (size_t)nq));
    HIPCHECK(d_m.ensure((size_t)block_words * world));
    HIPCHECK(hipMemcpy(d_g.p, gathered, sizeof(uint64_t) * (size_t)block_words * world, hipMemcpyHostToDevice));
    HIPCHECK(launch_dist_merge(d_g.p, (size_t)block_words, world, nq, ma, (uint32_t)R, d_off.p, d_c.p, d_c.p + nq, d_m.p, d_h.p, d_s.p,
                               nullptr));
    std::vector<uint64_t> hv((size_t)nq * R);
    std::vector<uint32_t> hs(nq);
    HIPCHECK(hipMemcpy(hv.data(), d_h.p, sizeof(uint64_t) * hv.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hs.data(), d_s.p, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost));
    d_g.release(); d_h.release(); d_s.release(); d_off.release(); d_c.release(); d_m.release();
    for (int q = 0; q < nq; ++q) {
        sizes[q] = hs[q] == 0xffffffffu ? -1 : (int32_t)hs[q];
        for (uint32_t i = 0; hs[q] != 0xffffffffu && i < hs[q]; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[(size_t)q * R + i];
            if (values) values[(size_t)q * R + i] = (int8_t)(hv[(size_t)q * R + i] >> 32);
        }
    }
    return QADC_OK;
}

int qadc_dist_merge_blocks_host(int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                                uint32_t* keys, int8_t* values, int32_t* sizes) {
    if (world < 1 || world > 16 || nq <= 0 || ma <= 0 || R <= 0 || !gathered || !sizes) return fail(QADC_E_ARG, "bad arguments");
    // what the ranks of qadc_dist_collect do between their two all-gathers, rank by rank: every rank replays its share,
    // the second gather concatenates the shares ([rank][ceil(nq / world)][R + 1]), every rank reads all heaps back
    const int per = (nq + world - 1) / world;
    const size_t hw = (size_t)R + 1;
    std::vector<uint64_t> all((size_t)world * per * hw, 0);
    for (int r = 0; r < world; ++r)
        replay_my_share(gathered, (size_t)block_words, world, r, nq, ma, R, nullptr, all.data() + (size_t)r * per * hw, nullptr);
    for (int q = 0; q < nq; ++q) {
        const uint64_t* o = all.data() + ((size_t)(q % world) * per + q / world) * hw;
        sizes[q] = (int32_t)o[R];
        for (int i = 0; i < sizes[q]; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)o[i];
            if (values) values[(size_t)q * R + i] = (int8_t)(o[i] >> 32);
        }
    }
    return QADC_OK;
}

int qadc_dist_shutdown(qadc_index* idx) {
    if (!idx || !idx->dist) return QADC_OK;
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    DistState* d = idx->dist;
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (hipStream_t m : d->merge_stream) if (m) (void)hipStreamSynchronize(m);
    if (d->comm && d->CommDestroy) (void)d->CommDestroy(d->comm);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    for (hipStream_t m : d->merge_stream) if (m) (void)hipStreamDestroy(m);
    d->d_fix.release(); d->h_fix.release(); d->d_moff.release(); d->d_merged.release(); d->d_mcnt.release();
    for (auto& ds : d->slot) ds.release();
    d->d_block.release(); d->d_gathered.release(); d->d_src.release(); d->h_src.release(); d->d_extra.release(); d->d_extra_all.release();
    d->h_extra.release(); d->h_out.release(); d->h_hdr.release(); d->h_extra_all.release();
    d->h_gathered.release(); d->h_myheaps.release(); d->d_myheaps.release(); d->d_allheaps.release(); d->h_allheaps.release();
    delete d;
    idx->dist = nullptr;
    return QADC_OK;
}

int qadc_dist_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                      const float* extra, int extra_n, float* extra_out) {
    if (!idx || !idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
    if (slot < 0 || slot >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (extra_n < 0 || (extra_n && (!extra || !extra_out))) return fail(QADC_E_ARG, "extra payload buffers missing");
    DistState& d = *idx->dist;
    Slot& s = idx->slot[slot];
    // caller errors — the same on every rank of a well-formed program — return before any rank enters the collective
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no batch");
    if (!s.dist_batch) return fail(QADC_E_STATE, "the batch was submitted before qadc_dist_init");
    if (int rc = use_device(idx)) return rc;
    const int nq = s.nq, R = s.R, world = d.world;
    // A failure of THIS rank's batch (candidate buffers that keep overflowing, a HIP error while re-running it) must not
    // leave the other ranks blocked in the gather: the rank still contributes a block, with bit7 set in every header, and
    // all ranks return the error after the gather.
    DistSlot& ds = d.slot[slot];
    if (ds.pending)
        if (int rc = flush_merges(idx, ds.seq)) return rc;
    const bool was_enqueued = ds.enqueued;
    if (was_enqueued) {                                       // the merge ran behind the scan: wait for all of it
        HIPCHECK(hipEventSynchronize(ds.ev_done));
        ds.enqueued = false;
    }
    int local_rc = collect_common(idx, slot, /*need_stream=*/false, /*from_dist=*/true);
    std::string local_err = local_rc ? g_err : std::string();
    if (was_enqueued) {
        const uint32_t* h_sz = reinterpret_cast<const uint32_t*>(ds.h_out.p + sizeof(uint64_t) * (size_t)R * nq);
        const uint32_t bad = h_sz[nq], need = h_sz[nq + 1];
        if (!bad) {
            // every rank saw clean headers: the heaps are final (a local failure of collect_common concerns this rank only)
            if (local_rc) return fail(local_rc, local_err);
            idx->prof.dist_async_collects++;
            std::vector<int32_t> st_async;
            int32_t* stp = status;
            if (!stp) { st_async.resize(nq); stp = st_async.data(); }
            finish_float_outputs(idx, s, stp, nullptr, nullptr);
            const uint64_t* hh = reinterpret_cast<const uint64_t*>(ds.h_out.p);
            for (int q = 0; q < nq; ++q) {
                uint32_t sz = h_sz[q];
                if (stp[q] || sz == 0xffffffffu) sz = 0;
                if (sizes) sizes[q] = (int32_t)sz;
                const uint64_t* hv = hh + (size_t)q * R;
                for (uint32_t i = 0; i < sz; ++i) {
                    if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[i];
                    if (values) values[(size_t)q * R + i] = (int8_t)(hv[i] >> 32);
                }
            }
            if (extra_n) {                                    // (a payload on such a batch travels by a small gather of its own)
                const size_t w = ((size_t)extra_n + 1) / 2;
                HIPCHECK(d.h_extra.ensure(2 * w));
                HIPCHECK(d.d_extra.ensure(2 * w));
                HIPCHECK(d.h_extra_all.ensure(2 * w * world));
                HIPCHECK(d.d_extra_all.ensure(w * world));       // (not d_block: a later batch's merge may be using that)
                std::memcpy(d.h_extra.p, extra, sizeof(float) * extra_n);
                HIPCHECK(hipMemcpyAsync(d.d_extra.p, d.h_extra.p, sizeof(uint64_t) * w, hipMemcpyHostToDevice, d.stream));
                std::string gerr2;
                if (d.gather(d.d_extra.p, d.d_extra_all.p, w, d.stream, gerr2)) return fail(QADC_E_HIP, gerr2);
                HIPCHECK(hipMemcpy2DAsync(d.h_extra_all.p, sizeof(float) * extra_n, d.d_extra_all.p, sizeof(uint64_t) * w,
                                          sizeof(float) * extra_n, world, hipMemcpyDeviceToHost, d.stream));
                HIPCHECK(hipStreamSynchronize(d.stream));
                std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
            }
            return QADC_OK;
        }
        // Some rank's stream overflowed its region or the gather block (identical verdict on every rank: they read the same
        // headers): the merge is redone below, at collect time, after collect_common re-ran what had to be re-run.
        if ((bad & 64u) && need + need / 8 > d.cap_entries && need < (1ull << 31))
            d.cap_entries = (uint32_t)(((uint64_t)need + need / 8 + 4095) / 4096 * 4096);
        idx->prof.regrows++;
    }
    if (d.inject_failure && !local_rc) {                      // test hook (option "dist_inject_failure")
        d.inject_failure = 0;
        local_rc = QADC_E_STATE;
        local_err = "injected failure (test hook)";
    }
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(nq);
        status = st_local.data();
    }
    if (!local_rc) finish_float_outputs(idx, s, status, nullptr, nullptr);
    // where this rank's ordered streams lie in device memory
    HIPCHECK(d.h_src.ensure(3 * (size_t)nq));
    HIPCHECK(d.d_src.ensure(3 * (size_t)nq));
    uint64_t fix_total = 0;
    for (int q = 0; q < nq && !local_rc; ++q) {
        const uint32_t fl = s.h_qout[q].flags;
        if (!(fl & 4u) && !(fl & 1u)) fix_total += s.out_off[q + 1] - s.out_off[q];   // ordered by collect_common on the host
    }
    if (fix_total >= (1ull << 31)) return fail(QADC_E_CAPACITY, "host-ordered streams exceed 2^31 entries");
    if (fix_total) {
        HIPCHECK(d.h_fix.ensure(fix_total));
        HIPCHECK(d.d_fix.ensure(fix_total));
    }
    uint64_t fix_off = 0;
    for (int q = 0; q < nq; ++q) {
        if (local_rc) {
            d.h_src.p[q] = 0;
            d.h_src.p[nq + q] = 0;
            d.h_src.p[2 * nq + q] = 128u;
            continue;
        }
        const QueryOut& qs = s.h_qout[q];
        const bool ordered = (qs.flags & 4u) != 0;
        d.h_src.p[q] = qs.out_off;
        d.h_src.p[nq + q] = ordered ? qs.count + qs.reps : 0u;
        d.h_src.p[2 * nq + q] = qs.flags & 0x3fu;
        if (!ordered && !(qs.flags & 1u)) {
            // more candidates than the device sort takes (> 16384): collect_common sorted the query's raw region on the host;
            // its stream travels in the same gather from a side buffer
            const uint64_t n = s.out_off[q + 1] - s.out_off[q];
            std::memcpy(d.h_fix.p + fix_off, s.out_entries.data() + s.out_off[q], sizeof(uint64_t) * n);
            d.h_src.p[q] = (uint32_t)fix_off;
            d.h_src.p[nq + q] = (uint32_t)n;
            d.h_src.p[2 * nq + q] = (qs.flags & 0x3fu) | 4u | 256u;
            fix_off += n;
        }
    }
    // Buffers are sized on the FIRST call for a payload of nq x R floats per rank (the sharded pre-scan's) whether or not
    // this call carries one: the pinned allocations and the larger gather blocks a first payload would otherwise need
    // cost milliseconds, and a pipeline's first batches typically come without payload.
    const size_t extra_room = std::max<size_t>((size_t)extra_n, (size_t)nq * (size_t)R);
    HIPCHECK(d.h_extra.ensure(extra_room));
    HIPCHECK(d.d_extra.ensure(extra_room));
    HIPCHECK(d.h_extra_all.ensure((size_t)world * extra_room));
    if (extra_n) std::memcpy(d.h_extra.p, extra, sizeof(float) * extra_n);
    const size_t heaps_bytes = (sizeof(uint64_t) * (size_t)R + sizeof(uint32_t)) * (size_t)nq;
    HIPCHECK(d.h_out.ensure(heaps_bytes + 16, hipHostMallocMapped | hipHostMallocCoherent));
    if (d.h_out.p != d.h_out_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d.d_out), d.h_out.p, 0));
        d.h_out_mapped = d.h_out.p;
    }
    HIPCHECK(d.h_hdr.ensure((size_t)world * nq * 4));
    hipStream_t st = d.stream;                               // the batch itself is complete (collect_common waited for it)
    uint64_t* h_heaps = reinterpret_cast<uint64_t*>(d.h_out.p);
    uint32_t* h_sizes = reinterpret_cast<uint32_t*>(d.h_out.p + sizeof(uint64_t) * (size_t)R * nq);
    // the device merge keeps a query's heap in the registers of one wave (R <= 320) and interleaves the ranks' streams with
    // ma x world counters in LDS; anything larger takes the host share
    const bool on_device = nq >= d.device_nq && (uint32_t)R <= replay_wave_max_R() &&
                           (size_t)s.ma * world <= dist_interleave_max_cells();
    std::string gerr;
    for (int attempt = 0;; ++attempt) {
        const size_t bw = dist_block_words(nq, d.cap_entries, (uint32_t)extra_n);
        const size_t bw_room = dist_block_words(nq, d.cap_entries, (uint32_t)extra_room);
        HIPCHECK(d.d_block.ensure(bw_room));
        HIPCHECK(d.d_gathered.ensure(bw_room * world));
        HIPCHECK(hipMemcpyAsync(d.d_src.p, d.h_src.p, sizeof(uint32_t) * 3 * nq, hipMemcpyHostToDevice, st));
        if (extra_n) HIPCHECK(hipMemcpyAsync(d.d_extra.p, d.h_extra.p, sizeof(float) * extra_n, hipMemcpyHostToDevice, st));
        if (fix_total) HIPCHECK(hipMemcpyAsync(d.d_fix.p, d.h_fix.p, sizeof(uint64_t) * fix_total, hipMemcpyHostToDevice, st));
        HIPCHECK(launch_dist_pack(d.d_src.p, d.d_src.p + nq, d.d_src.p + 2 * nq, nq, s.d_stream.p, d.d_fix.p, d.cap_entries,
                                  extra_n ? d.d_extra.p : nullptr, (uint32_t)extra_n, d.d_block.p, st));
        if (d.gather(d.d_block.p, d.d_gathered.p, bw, st, gerr)) return fail(QADC_E_HIP, gerr);
        if (on_device) {
            HIPCHECK(d.d_moff.ensure(nq));
            HIPCHECK(d.d_mcnt.ensure(2 * (size_t)nq));
            HIPCHECK(d.d_merged.ensure((size_t)d.cap_entries * world));
            HIPCHECK(launch_dist_merge(d.d_gathered.p, bw, world, nq, s.ma, (uint32_t)R, d.d_moff.p, d.d_mcnt.p, d.d_mcnt.p + nq,
                                       d.d_merged.p, reinterpret_cast<uint64_t*>(d.d_out),
                                       reinterpret_cast<uint32_t*>(d.d_out + sizeof(uint64_t) * (size_t)R * nq), st));
        } else {
            HIPCHECK(d.h_gathered.ensure(bw_room * world));
            HIPCHECK(hipMemcpyAsync(d.h_gathered.p, d.d_gathered.p, sizeof(uint64_t) * bw * world, hipMemcpyDeviceToHost, st));
        }
        // every rank's header (to size a retry identically everywhere) and extra payload come back with the heaps
        HIPCHECK(hipMemcpy2DAsync(d.h_hdr.p, sizeof(uint32_t) * 4 * nq, d.d_gathered.p, sizeof(uint64_t) * bw,
                                  sizeof(uint32_t) * 4 * nq, world, hipMemcpyDeviceToHost, st));
        if (extra_n)
            HIPCHECK(hipMemcpy2DAsync(d.h_extra_all.p, sizeof(float) * extra_n, d.d_gathered.p + 2 * (size_t)nq + d.cap_entries,
                                      sizeof(uint64_t) * bw, sizeof(float) * extra_n, world, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        uint64_t need = 0;
        bool overflow = false, unordered = false;
        int failed_rank = -1;
        for (int g = 0; g < world; ++g) {
            uint64_t tot = 0;
            for (int q = 0; q < nq; ++q) {
                const uint32_t* h = d.h_hdr.p + ((size_t)g * nq + q) * 4;
                tot += h[1];
                overflow |= (h[2] & 64u) != 0;
                if ((h[2] & 128u) && failed_rank < 0) failed_rank = g;
                unordered |= !(h[2] & 4u) && !(h[2] & 1u) && !(h[2] & 128u);
            }
            need = std::max(need, tot);
        }
        // (every rank reads the same headers: the same branch is taken everywhere, no rank stays behind in a collective)
        if (failed_rank >= 0) {
            if (extra_n) std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
            return fail(local_rc ? local_rc : QADC_E_STATE,
                        local_rc ? local_err : "rank " + std::to_string(failed_rank) + " failed before the gather (see its own error)");
        }
        if (unordered) return fail(QADC_E_STATE, "a rank shipped a query without ordering it");
        if (overflow) {
            if (attempt >= 2 || need >= (1ull << 31)) return fail(QADC_E_CAPACITY, "gather block overflow persists");
            d.cap_entries = (uint32_t)((need + need / 8 + 4095) / 4096 * 4096);   // the same on every rank: they all saw the same headers
            idx->prof.regrows++;
            continue;
        }
        if (on_device) break;
        // ---- few queries (or R > 288): replay my share on the host (global scan order: assign slot, rank, position), share the heaps ----
        const int per = (nq + world - 1) / world;
        const size_t hw = (size_t)R + 1;                       // words per query in the heap exchange
        HIPCHECK(d.h_myheaps.ensure((size_t)per * hw));
        HIPCHECK(d.d_myheaps.ensure((size_t)per * hw));
        HIPCHECK(d.d_allheaps.ensure((size_t)per * hw * world));
        HIPCHECK(d.h_allheaps.ensure((size_t)per * hw * world));
        std::memset(d.h_myheaps.p, 0, sizeof(uint64_t) * (size_t)per * hw);
        {
            ScopedMs timer(idx->prof.host_heap_ms);
            replay_my_share(d.h_gathered.p, bw, world, d.rank, nq, s.ma, R, status, d.h_myheaps.p, &idx->pool);
        }
        HIPCHECK(hipMemcpyAsync(d.d_myheaps.p, d.h_myheaps.p, sizeof(uint64_t) * (size_t)per * hw, hipMemcpyHostToDevice, st));
        if (d.gather(d.d_myheaps.p, d.d_allheaps.p, (size_t)per * hw, st, gerr)) return fail(QADC_E_HIP, gerr);
        HIPCHECK(hipMemcpyAsync(d.h_allheaps.p, d.d_allheaps.p, sizeof(uint64_t) * (size_t)per * hw * world, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        for (int q = 0; q < nq; ++q) {
            const uint64_t* o = d.h_allheaps.p + ((size_t)(q % world) * per + q / world) * hw;
            h_sizes[q] = (uint32_t)o[R];
            std::memcpy(h_heaps + (size_t)q * R, o, sizeof(uint64_t) * (size_t)R);
        }
        break;
    }
    for (int q = 0; q < nq; ++q) {
        uint32_t sz = h_sizes[q];
        if (status[q] || sz == 0xffffffffu) sz = 0;
        if (sizes) sizes[q] = (int32_t)sz;
        const uint64_t* hv = h_heaps + (size_t)q * R;
        for (uint32_t i = 0; i < sz; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[i];
            if (values) values[(size_t)q * R + i] = (int8_t)(hv[i] >> 32);
        }
    }
    if (extra_n) std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
    return QADC_OK;
}

int qadc_slot_assign(qadc_index* idx, int slot, int32_t* assign_out) {
    if (!idx || slot < 0 || slot >= kSlots || !assign_out) return fail(QADC_E_ARG, "bad arguments");
    const Slot& s = idx->slot[slot];
    if (s.busy) return fail(QADC_E_STATE, "collect the batch first");
    if (s.assign.size() != (size_t)s.nq * s.ma) return fail(QADC_E_STATE, "slot has held no batch");
    std::memcpy(assign_out, s.assign.data(), sizeof(int32_t) * s.assign.size());
    return QADC_OK;
}

int qadc_slot_qtables(qadc_index* idx, int slot, int q_first, int q_count, int8_t* out) {
    if (!idx || slot < 0 || slot >= kSlots || !out || q_first < 0 || q_count < 0) return fail(QADC_E_ARG, "bad arguments");
    const Slot& s = idx->slot[slot];
    if (s.busy) return fail(QADC_E_STATE, "collect the batch first");
    if (!s.d_qt || s.nq <= 0) return fail(QADC_E_STATE, "slot has held no batch");
    if ((int64_t)q_first + q_count > s.nq) return fail(QADC_E_ARG, "query range outside the batch");
    if (int rc = use_device(idx)) return rc;
    const size_t per_q = (size_t)s.ma * idx->M * 16;
    HIPCHECK(hipMemcpyAsync(out, s.d_qt + (size_t)q_first * per_q, (size_t)q_count * per_q, hipMemcpyDeviceToHost, idx->copy_stream));
    HIPCHECK(hipStreamSynchronize(idx->copy_stream));
    return QADC_OK;
}

int qadc_place_partitions(int part_count, const uint32_t* sizes, int world, int32_t* owner_out) {
    if (part_count < 0 || world < 1 || (part_count && (!sizes || !owner_out))) return fail(QADC_E_ARG, "bad arguments");
    std::vector<int> order(part_count);
    for (int p = 0; p < part_count; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] > sizes[b]; });
    std::vector<uint64_t> load(world, 0);
    for (int p : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        owner_out[p] = best;
        load[best] += sizes[p];
    }
    return QADC_OK;
}

int qadc_profile_read(qadc_index* idx, qadc_profile* out) {
    if (!idx || !out) return fail(QADC_E_ARG, "bad arguments");
    *out = idx->prof;
    return QADC_OK;
}

int qadc_profile_reset(qadc_index* idx) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    idx->prof = qadc_profile{};
    return QADC_OK;
}

}  // extern "C"
