// Internal host-side header of libqadc_hip.so (not part of the C-ABI; see include/qadc.h for that): the state
// structs and helpers shared by the translation units of the host side —
//   qadc_capi.cpp   index / database side, level-path planner, submit / collect, the plain query entry points
//   qadc_ivf.cpp    one-workgroup-per-query batches, partition-major second phase, device-side feeders (qadc_search)
//   qadc_dist.cpp   native multi-GPU merge (qadc_dist_*: RCCL by dlopen, or a caller-supplied all-gather)
//   qadc_build.cpp  database build entry points (PQ / IVF encode, k-means iterations)
// The last three keep their state in structs of their own hung off qadc_index (FeederState, GroupState, DistState).
#pragma once
#include "../../include/qadc.h"

#include <hip/hip_runtime.h>

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

namespace qadc {
namespace host {

extern thread_local std::string g_err;                       // message of the last failure on this thread (qadc_last_error)

inline int fail(int code, const std::string& msg) {
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

constexpr int kMergeStreams = 3;   // streams the enqueued merges MAY rotate over (measurement hook); the default creates ONE:
                                   // more merge streams than pipes to put them on only move a merge onto the front's or the
                                   // scan's pipe (qadc_index_create) — one of 8 ranks, C5 shape: 1 stream 0.89 ms, 2: 0.93, 3: 1.06
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
    bool group_fell_back = false;       // ... and overflowed its candidate regions (redone on the level path): under the multi-GPU merge the
                                        // strike is counted from the GATHERED headers, on every rank alike (qadc_dist_collect)
    int group_head_slots = 0;           // ... after a head of this many local probes per query
    DevBuf<float> d_fc;

    // one-workgroup-per-query path (qadc_query_kernel.hip): no planner, no levels, no sort
    bool wgq = false;
    bool dist_batch = false;            // launched with the native multi-GPU merge active: streams kept in device memory
    uint32_t wgq_cap = 0;               // stream entries per query workgroup (regrown on overflow)
    bool poll = false;                  // collect watches the workgroups' done bits instead of the completion event
    bool appended = false;              // ... and appended the sub-streams to out_entries as they finished (no copy left to do)
    bool replayed = false;              // ... and pushed a lone query's entries through early_heap as they arrived (replay_outputs takes it)
    kv_heap<uint32_t, int8_t> early_heap;
    bool ev_valid = false;              // ev_done was recorded behind this batch's launches (not for a polled batch)
    int wgq_G = 1;                      // workgroups per query (small batches: the scan order of a query is split)
    uint64_t wgq_codes = 0;             // codes a query probes (exact maximum, or an estimate) — sizes wgq_G
    uint64_t head_codes = 0;            // level path: codes of every query's scan order covered by the head launch (0 = none)
    uint64_t wgq_fcap = 0;              // pre-scan values per query in the global scratch (0 = they fit LDS)
    DevBuf<uint32_t> d_qflags;          // [nq][4]: {flags, entries} for replay_heap_wave_kernel
    DevBuf<float> d_fvals;
    DevBuf<QCand> d_qcands;             // unordered candidates of the query workgroups (scratch)
    DevBuf<uint32_t> d_lfstate;         // sliced front of a lone query (lone_front_kernel): counter + the slices' smallest keys
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
    hipEvent_t ev_pre = nullptr;        // tables + state clear + partition-major plan done (enqueued off the scan stream: launch_wgq_batch)

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
    // replay sharded by query (option "dist_shard_replay"): this rank replayed queries rank, rank + world, ... into d_share; the shares'
    // all-gather + unpack (`pending_share`) is issued behind a LATER merge's first gather, when the replay's latency chain is over
    DevBuf<uint64_t> d_share, d_shares;
    hipEvent_t ev_replayed = nullptr;
    bool pending_share = false;
    int share_nq = 0, share_R = 0;
    void release() {
        d_block.release(); d_gathered.release(); d_merged.release(); d_moff.release(); d_mcnt.release(); h_out.release();
        d_src.release(); d_share.release(); d_shares.release();
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (ev_gathered) (void)hipEventDestroy(ev_gathered);
        if (ev_replayed) (void)hipEventDestroy(ev_replayed);
        ev_ready = ev_done = ev_gathered = ev_replayed = nullptr;
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
    // (both BORROWED from the index, which creates every stream it will ever use in one fixed order — see qadc_index_create)
    hipStream_t stream = nullptr;                            // the collectives' own high-priority stream: the merge of batch s must not
                                                             // queue behind the scan kernels of batches s+1.. on the scan stream
    hipStream_t merge_stream[kMergeStreams] = {};            // interleave + replay of a merge enqueued with its batch: a millisecond of
                                                             // latency that must neither sit in front of the NEXT batch's collectives
                                                             // nor behind the PREVIOUS batch's replay (rotating, by merge sequence)
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
    int shard_replay = 1;                                    // an enqueued merge replays only this rank's share of the queries; a second,
                                                             // small all-gather shares the heaps (option "dist_shard_replay")
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

// N1 — the host feeders of the path on the device (qadc_index_set_pq / _set_rotation / _set_coarse, qadc_search): qadc_ivf.cpp
struct FeederState {
    int dim = 0;                 // vector dimension (0 = qadc_index_set_pq not called)
    DevBuf<float> d_codebooks;   // [M][16][dim/M]
    DevBuf<float> d_rotation;    // [dim][dim] OPQ rotation (empty = plain PQ)
    bool has_rotation = false;
    int K = 0;                   // coarse centroids (0 = flat)
    DevBuf<float> d_coarse;      // [K][dim]
    DevBuf<float> d_cnorm;       // [K] ||centroid||^2 as compute_cross_dists_blas adds it (launch_row_sqnorm), for sum_mode cnorm_mode
    int cnorm_mode = -1;         // (-1: not computed yet / the centroids changed)
    int table_form = 2;          // float tables of qadc_search: 0 direct, 1 BLAS expansion, 2 the reference's nns_engine rule
};

// Fixed tuning constants (each was an option while it was being measured; the sweeps are in profiles/, see profiles/README.md)
constexpr uint32_t kShareCodesPerWg = 1u << 20;    // codes per workgroup of a sibling-major shared launch
constexpr uint32_t kMqCodesPerWg = 1u << 16;       // ... of a multi-query launch (8 queries per pass)
constexpr uint32_t kMqMinWgs = 4096;               // workgroups a multi-query launch should have at least (2 rounds of the chip)
constexpr uint32_t kMqMinTiles = 4;                // ... but never fewer than this many 4 KiB tiles per workgroup
constexpr uint32_t kSmallVecPerWg = 512;           // 16-byte vectors one small-run workgroup covers
constexpr uint64_t kFrontMinBatch = 3000000000ull; // leading levels join the front stream only in batches of at least this many (code, query)
                                                   // pairs: under a shorter last level they only make the front stream the step's longest chain
constexpr int kWgqMinNq = 128;                     // query-kernel path, auto: batches of at least this many queries ...
constexpr uint64_t kWgqMaxCodes = 1ull << 24;      //   ... probing at most this many codes per query, or
constexpr uint64_t kWgqSmallCodes = 1ull << 18;    //   any batch probing at most this many codes per query, or
constexpr uint64_t kWgqLoneCodes = 6ull << 20;     //   a call of one or two queries probing at most this many each, or
constexpr uint64_t kWgqLoneSlicedCodes = 24ull << 20;   //   ONE query on one partition of at most this many codes whose front can be sliced
constexpr uint32_t kGroupBytesPerWg = 131072;      // partition-major phase: bytes of the longest partition's codes per workgroup of a group
constexpr int kMaxSplit = 64;                      // most workgroups a query is spread over (option "wgq_split" is clamped to it)
constexpr int kSplitBatch = 12;                    // workgroups per query of a small batch of three or more queries (one or two: option "wgq_split")
constexpr int kShareLag = 1;                       // multi-GPU: a merge's heap-share gather is issued behind the first gather of the next merge

// Partition-major second phase of large IVF batches (launch_wgq_batch, qadc_ivf.cpp)
struct GroupState {
    int strikes = 0;       // grouped batches whose candidate regions overflowed (data whose later probes fall below the head's bound)
    int mode = 1;          // option "wgq_group": 0 never, 1 auto, 2 whenever possible
    int head = 2;          // ... after a head of this many probes per query (one workgroup per query).  Round 6, same-box A/B: 2 instead of 3
                           // -1 % at C3 with the 8-wave head, -1.7 % at C5 (-2.8 % at 2048-query batches); 1 leaves the bound too loose
    int head_dist = 4;     // ... under the multi-GPU merge (probes with codes on this rank; option "wgq_group_head_dist")
    uint32_t cand_cap = kOrderCandCap;   // candidates per query of such a batch before it falls back (option "wgq_group_cand_cap")
};

}  // namespace host
}  // namespace qadc

using namespace qadc;
using namespace qadc::host;   // (internal header: only the host-side translation units include it)

struct qadc_index {
    int M = 16, cs = 8, device = 0;
    // The streams belong to the PROCESS: one set per device, created in one fixed order by the first index on the device and
    // shared by every later one (attach_streams in qadc_capi.cpp has the measurements: which compute pipe a stream's queue
    // lands on decides what it can be blocked behind, and only the first set a process creates lands predictably).
    hipStream_t stream = nullptr;       // scan stream of the level path (lowest priority)
    hipStream_t wgq_stream = nullptr;   // scan stream of the one-workgroup-per-query batches when option "wgq_stream" says so (normal priority)
    hipStream_t coll_stream = nullptr;  // the multi-GPU merge's collectives (highest priority; unused until qadc_dist_init)
    hipStream_t merge_streams[kMergeStreams] = {};   // the merges' interleave + replay (lowest priority; unused until qadc_dist_init)
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
    int sum_mode = 1;                   // grouping of the float pre-scan's adds (qadc_float_sum.h): 1 = as the reference is compiled
    uint32_t cand_capacity = kSortCap;  // candidate region entries per query
    uint64_t level_base = 512;
    uint64_t level_growth = 4;
    int wgs_per_item = 0;  // 0 = auto
    int share_variant = 0x41;            // streaming-kernel variant for shared launches: sibling-major, U=2, cached loads
    int mq = 1;                          // shared launches use the 8-queries-per-pass kernel
    int device_replay_nq = 64;           // batches of at least this many queries replay their streams on the device (0 = never)
    int device_replay_alone_nq = 512;    // ... a batch with nothing else in flight (a synchronous call): from this many
    uint64_t front_run_max = 2u << 20;   // leading levels whose runs are at most this long join the front (0 = none); they are
                                         // counted with the small launches, not event-timed.  125M x 32: 2 Mi -4 %, 8 Mi +1 %
    uint32_t wgq_split_codes = 2048;    // a query is split over several workgroups only down to this many codes each (round 6, with the
                                        // one-step first block: lone query, 10^5 codes, same box: 12 workgroups 36.1 us, 16: 33.3, 32: 32.6, 48: 32.6)
    int head_wg = 0;       // 512: the IVF head launch runs in 512-thread workgroups (8 waves per query); 0 / 1024: 16 waves
    uint32_t prescan_sample = 1u << 16;  // starts pre-scanned unfiltered before the survivor filter kicks in
    WorkerPool pool;                   // host replay workers (started on first use)
    uint32_t small_run = 1u << 17;  // runs shorter than this use the small-run kernel
    int variant = 0x0d;    // kernel tuning variant (see launch_scan_i8): U=2, non-temporal loads, chunked tiles
    // one workgroup per query (IVF batches, small lists): 0 = never, 1 = auto, 2 = whenever structurally possible
    int wgq = 1;
    uint32_t wgq_capacity = 4096;        // stream entries per query to start with
    int wgq_split = 32;                 // workgroups a small batch may spread one query's scan order over
    int head_level = 5;                  // level path: bound levels 0..head_level-1 (the first 512 Ki codes of every query) are
                                         // scanned by ONE launch of the query kernel in head mode instead of head_level dependent
                                         // level launches (0 = off): -5 % per step on a 125M-code shard, neutral at 1B
    uint32_t wgq_cand_cap = kQueryCandCap;   // candidates per query before the batch falls back to the level path (test knob)
    DevBuf<PartDesc> d_partdesc;         // device partition table (qadc_index_finalize)
    std::vector<PartDesc> h_partdesc;    // its host copy (a lone small query carries the descriptors it needs in its launch)
    uint32_t max_start_n = 0;
    uint64_t total_codes = 0;            // codes held HERE (differs from rank to rank under the multi-GPU merge)
    uint64_t total_global_codes = 0;     // codes of the whole partitions: the same on every rank — what any decision that adds,
                                         // removes or resizes a collective must be derived from
    uint32_t max_part_n = 0;
    bool profile = false;
    Slot slot[kSlots];
    Slot pre_slot[2];                   // sharded pre-scan passes (mode 1): own buffers, so that one can run
                                        // while slot[i] still holds an uncollected batch
    qadc_profile prof{};
    FeederState feed;                   // qadc_ivf.cpp: N1, the feeders on the device
    GroupState group;                   // qadc_ivf.cpp: partition-major second phase of large IVF batches
    DistState* dist = nullptr;          // qadc_dist.cpp: qadc_dist_init
};

namespace qadc {
namespace host {

// std::thread::hardware_concurrency() asks the kernel on every call (sched_getaffinity: ~1 us — a thirtieth of a lone query's call,
// twice per call); the host's thread count does not change under the library
inline unsigned host_threads() {
    static const unsigned n = std::max(1u, std::thread::hardware_concurrency());
    return n;
}

struct ScopedMs {
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedMs(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedMs() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// qadc_capi.cpp
int use_device(const qadc_index* idx);
hipError_t prof_event(Slot& s, hipStream_t st);
int plan_and_launch(qadc_index* idx, Slot& s);               // plans the batch in slot s and enqueues all of its GPU work
int collect_common(qadc_index* idx, int slot_i, bool need_stream = true, bool from_dist = false);
void finish_float_outputs(qadc_index* idx, Slot& s, int32_t* status, float* qmin, float* qmax);
int replay_outputs(qadc_index* idx, Slot& s, uint32_t* keys, int8_t* values, int32_t* sizes, const int32_t* status);
// qadc_ivf.cpp
int table_expansion(const qadc_index* idx, int ma);
bool wgq_eligible(const qadc_index* idx, int nq, int ma, int R, int mode, uint64_t codes_per_query, bool host_float_tables = false,
                  bool alone = false);
uint32_t lone_front_slices(uint32_t starts, int R);
bool will_group(const qadc_index* idx, int nq, int ma, bool dev_replay);
int launch_wgq_batch(qadc_index* idx, Slot& s);
int search_submit(qadc_index* idx, int slot_i, int nq, const float* queries, int ma, int R);
// qadc_dist.cpp
int load_rccl(DistState& d, std::string& err);
int enqueue_merge(qadc_index* idx, Slot& s, hipStream_t scan_stream);
int flush_merges(qadc_index* idx, uint64_t upto);
int flush_shares(qadc_index* idx, uint64_t upto);            // heap-share gathers of sharded replays (qadc_dist.cpp)

}  // namespace host
}  // namespace qadc
