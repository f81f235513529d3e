// Device-side data structures and kernel launchers of the MI355X Quick-ADC scan engine.
// Internal header (not part of the C-ABI; see include/qadc.h for that).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>

namespace qadc {

// One contiguous run of codes of one probed partition, scanned with one int8 table.
// (A probed partition is cut into one item per bound level, see DESIGN.md "Exact filtering".)
struct ScanItem {
    const uint8_t* codes;    // first code of the run (16-byte aligned)
    const uint32_t* labels;  // partition labels (indexed by position in partition) or nullptr
    uint32_t n;              // codes in the run
    uint32_t pos0;           // position of the first code inside its partition
    uint32_t key_base;       // added to the position when labels == nullptr (shard offset)
    uint32_t table;          // table index: qtables + table * M * 16
    uint32_t query;          // per-query state index
    uint32_t order;          // (level << 16) | assign slot (< 2^14): scan-order major key of emitted entries
    uint32_t dup_pos;        // position (in the partition) of the code the reference replays in its padding
                             // lanes, if this run's partition end is held here; else 0xffffffff
    uint32_t dup_reps;       // number of extra replays of that code: (16 - n % 16) % 16
};

// Candidate emitted by the scan: value < bound derived from a strict prefix of the scan order.
struct Cand {
    uint32_t order;  // ScanItem::order | (extra replays << 20)
    uint32_t pos;    // position inside the partition
    uint32_t key;    // label, or key_base + pos
    uint32_t val;    // int8 distance sum (0..126)
};

constexpr uint32_t kSortCap = 16384;   // candidates per query the device sort handles (LDS: 16384 x 8 B)

// Batch-wide flags (one per in-flight batch).
struct CandHeader {
    uint32_t overflow;      // candidates dropped because some query's region was full
    uint32_t out_overflow;  // sorted entries that did not fit the sorted output buffer
    uint32_t pad[2];
};

constexpr int kMaxLevels = 16;  // bound levels per query

// Per-query device state.  hist[l][v] counts the candidates of value v emitted by level l.
struct QueryState {
    uint32_t hist[kMaxLevels * 128];
    uint32_t count;     // candidates this query emitted (slots requested in its region)
    uint32_t reps;      // extra padding-lane replays among them
    uint32_t flags;     // bit0: qmax > 1e30 (reference would exit), bit1: negative table entries clamped,
                        // bit2: candidates sorted into scan order on the device,
                        // bit3: pre-scan survivor buffer overflowed (host re-runs the batch unfiltered)
    float qmin;
    float qmax;
    uint32_t sel_prefix;  // radix-select running prefix (in key - sel_min space)
    uint32_t sel_k;       // radix-select remaining rank (1-based)
    uint32_t sel_hi;      // bits [sel_hi, 32) of (key - sel_min) are already fixed to sel_prefix
    uint32_t sel_nmin;    // ~min(key) over the query's pre-scan values (atomicMax of ~key; 0 = none)
    uint32_t sel_max;     // max(key)
    uint32_t out_off;     // first entry of this query in the sorted output (valid with flags bit2)
    uint32_t fc_n;        // pre-scan survivors appended behind the query's sample values
    uint32_t pad;
};

// What the host reads back per query (32 bytes instead of the 8 KiB QueryState).
struct QueryOut {
    uint32_t count;    // candidates emitted (region slots requested)
    uint32_t reps;     // extra padding-lane replays among them
    uint32_t flags;    // QueryState::flags (bit2 = ordered on the device, entries valid)
    uint32_t out_off;  // first entry in the sorted output
    float qmin, qmax;
    uint32_t pad[2];
};

// Float ADC item for the "starts" pre-scan (scanner_4::query_scan_start).
struct StartItem {
    const uint8_t* codes;  // first code of the partition
    uint32_t n;            // starts size of that partition
    uint32_t table;        // float table index: ftables + table * M * 16
    uint32_t query;
    uint32_t out_off;      // filter == 0: offset inside the query's float value buffer
    uint32_t filter;       // 1: append only values <= QueryState::qmax (the sample's R-th smallest)
};

// ---- one workgroup per query (qadc_query_kernel.hip): IVF batches and small lists ---------------------
// Device-resident partition table, one entry per database partition (built by qadc_index_finalize).
struct PartDesc {
    const uint8_t* codes;    // row-major codes of the local range
    const uint32_t* labels;  // labels of the local range, or nullptr
    const uint8_t* starts;   // replica of the partition's first codes (sharded lists), or nullptr: codes
    uint32_t n;              // codes held here
    uint32_t global_n;       // codes of the whole partition
    uint32_t first_pos;      // global position of local code 0
    uint32_t key_base;
    uint32_t start_n;        // max(1, unsigned(global_n * keep)), 0 for an empty partition
    uint32_t pad;
};

// Unordered candidate of the one-workgroup-per-query scan (sorted by (slot, pos) inside the workgroup afterwards).
struct QCand {
    uint32_t key;       // label, or key_base + first_pos + pos
    uint32_t val_reps;  // value | extra padding-lane replays << 8
    uint32_t pos;       // position inside the partition's local range
    uint32_t slot;      // assign slot
};
constexpr uint32_t kQueryCandCap = 4096;   // candidates per query the in-workgroup sort takes (64 KiB of LDS)
constexpr uint32_t kOrderCandCap = 8192;   // ... and order_cands_kernel, the ordering pass of the partition-major second phase (64 KiB of keys)

struct QueryKernelArgs {
    const PartDesc* parts;
    const int32_t* assign;   // [nq][ma] probed partitions in scan order
    int ma;
    float* ftables;          // [nq][ma][M*16] float tables (negatives clamped in place), or nullptr: int8 path
    int8_t* qtables;         // [nq][ma][M*16]: written by the quantizer (float path) / read as given (int8 path)
    float* fvals;            // [nq][fcap] scratch for a query's pre-scan values when they exceed the LDS budget
    uint32_t fcap;
    uint64_t* stream;        // [nq][cap] ordered push stream: key | value << 32 | assign slot << 40
    uint32_t cap;
    QCand* cands;            // [nq][ccap] unordered candidates (scratch)
    uint32_t ccap;           // <= kQueryCandCap
    QueryOut* qout;          // [nq]
    uint32_t* qstate_flags;  // optional [nq][4]: {flags, entries} for replay_heap_wave_kernel, or nullptr
    uint32_t R;
    int quant_mode;
    int sum_mode;            // grouping of the float pre-scan's adds: 1 = as the reference is compiled, 0 = source order (qadc_float_sum.h)
    int nontemporal;         // non-temporal code loads (database larger than the Infinity Cache)
    // head mode (level-structured path): scan only the first head_codes codes of every query's scan order with the int8
    // tables in `qtables`, emit Cand records / level-0 histogram counts like emit_candidate does (0 = normal mode)
    uint64_t head_codes;
    uint32_t head_slots;     // head mode, other cut: walk the first head_slots probed partitions in full (head_codes = ~0)
    QueryState* qstates;
    Cand* cand_regions;      // [nq][cand_cap]
    uint32_t cand_cap;
    CandHeader* hdr;
    int G;                   // workgroups per query (>= 1): the grid is nq * G, every output array is indexed by q * G + g;
                             // workgroup g scans the first block (bound only, g > 0) and the g-th chunk of the rest
    // inline input (G > 1 only): assign / parts / ftables are NOT pointers the host filled in but live in the kernel-argument
    // segment itself, behind this struct (QueryKernelInline::payload) — a single query's 1-2 KiB of input ride in the
    // dispatch packet and no host-to-device copy precedes the launch.  payload = [i32 assign[nq*ma] = 0,1,2...]
    // [PartDesc[nq*ma] of the probed partitions, at inline_off_parts][float tables, at inline_off_tables].
    uint32_t inline_input, inline_off_parts, inline_off_tables;
    // Front sharded over the ranks of a multi-GPU merge (HEAD instantiation only): rank r runs the front — pre-scan,
    // select, qmin / clamp, quantizer — for ITS share of the queries with front_only set (results: qtables + front_out[q] =
    // {flags & 3, qmin, qmax, 0}; no walk), the shares are all-gathered, and the head launch of the whole batch takes the
    // gathered int8 tables with front_in seeding every query's flags / qmin / qmax.
    uint32_t front_only;
    uint32_t pos_bits;       // the ordering pass's bucket sort: bits of the longest partition's length << 16 | largest bucket it ranks
    uint32_t head_wg;        // host side only: 512 = the head launch in 512-thread workgroups (8 waves per query; option "head_wg")
    uint32_t* front_out;
    const uint32_t* front_in;
};

constexpr size_t kInlineBytes = 3072;            // kernel arguments are limited to 4 KiB
struct QueryKernelInline {
    QueryKernelArgs a;
    alignas(16) unsigned char payload[kInlineBytes];
};

// ---- the sliced front of a lone query on a LONG flat list (lone_front_kernel, qadc_query_kernel.hip) ----
// Every workgroup of a split query repeats the front, and the front grows with the list (keep x codes starts: 40 K on 4 x 10^6
// codes at 1 %): on long lists it, not the walk and not the host's replay, sets the call's time (profiles/
// r06_lone_query_latency_ab.txt).  Here S workgroups pre-scan one slice of the starts each (scan_4<M>, query_common.hpp:59-90,
// same grouping of the adds) and leave the slice's R smallest values behind; the workgroup that finishes LAST (a counter —
// nobody waits) takes the R-th smallest of their union — which IS the R-th smallest of all starts, db_query_4.cpp:259 —,
// qmin, the clamp and QuantizerMAX (db_query_4.cpp:258-284, 37-71) and writes the int8 table and {flags, qmin, qmax} for the walk
// launch behind it (QueryKernelArgs::front_in).  One probed partition (a flat list).
constexpr int kLoneFrontSlice = 4096;           // start values per slice (one bitonic sort in LDS)
constexpr int kLoneFrontMaxKeys = 8192;         // S x R keys the last workgroup sorts
struct LoneFrontArgs {
    const PartDesc* part;                       // the probed partition (device partition table entry)
    uint32_t R, S;                              // top-R; slices = workgroups
    int quant_mode, sum_mode;
    uint32_t* state;                            // [0] counter (left at 0), [4 ...] keys[S][R]
    int8_t* qtables;                            // out: [M*16]
    uint32_t* front_out;                        // out: {flags & 3, qmin, qmax, 0}
    alignas(16) float table[32 * 16];           // the query's float table, in the kernel-argument segment
};
hipError_t launch_lone_front(int M, const LoneFrontArgs& args, hipStream_t stream);

uint32_t query_kernel_lds_values(int M);       // pre-scan values a query may have before fvals is needed
hipError_t launch_scan_query(int M, int nq, const QueryKernelArgs& args, hipStream_t stream,
                             const void* inline_payload = nullptr, size_t inline_bytes = 0);
// Large IVF batches, partition-major second phase (qadc_query_kernel.hip): regroup the (query, probe) pairs of the
// probes s0 .. ma-1 by partition into ScanItem groups of 8 for scan_i8_mq_kernel (d_cnt, d_fill: K zeroed counters;
// d_goff: K + 1; d_items: zeroed, ivf_max_groups() * 8 entries), and order a query's Cand region into its push stream.
inline size_t ivf_max_groups(size_t pairs, size_t K) { return pairs / 8 + std::min(pairs, K); }
void launch_ivf_plan(const int32_t* d_assign, const PartDesc* d_parts, int nq, int ma, int s0, int K, uint32_t* d_cnt,
                     uint32_t* d_goff, uint32_t* d_fill, ScanItem* d_items, hipStream_t stream);
hipError_t launch_order_cands(const QueryState* d_qs, const Cand* d_regions, uint32_t cand_cap, uint32_t ccap, int nq,
                              uint64_t* d_stream, uint32_t cap, QueryOut* d_qout, uint32_t* d_qflags, int ma, uint32_t pos_bits, hipStream_t stream);
// Front shares of the ranks (gathered blocks of block_bytes each: [qtables per x tab][assign per x ma i32][front per x 4 u32])
// -> the batch's arrays in query order; assign and front records also into host-mapped memory for the collect call.
hipError_t launch_front_unpack(const unsigned char* d_gathered, size_t block_bytes, int world, int per, int nq, int ma, size_t tab,
                               int8_t* d_qt, int32_t* d_assign, uint32_t* d_front, int32_t* h_assign, uint32_t* h_front,
                               hipStream_t stream);
// kv_binheap push replay of the ordered streams, 64 queries per wave (one lane each); R <= replay_lanes_max_R().
// ---- multi-GPU merge around one ncclAllGather (qadc_dist_collect) ----
// Per-rank block (u64 words): [nq x {offset, count, flags, 0} as u32][cap_entries entries][extra floats, padded to u64].
inline size_t dist_block_words(int nq, uint32_t cap_entries, uint32_t extra_n) {
    return 2 * (size_t)nq + cap_entries + (extra_n + 1) / 2;
}
// d_src_off/cnt/flags[q]: where query q's ordered stream starts inside d_stream, its entries, its QueryState flags.
// flags bit8: the query's stream lies in d_fix (ordered on the host), not in d_stream.
hipError_t launch_dist_pack(const uint32_t* d_src_off, const uint32_t* d_src_cnt, const uint32_t* d_src_flags, int nq,
                            const uint64_t* d_stream, const uint64_t* d_fix, uint32_t cap_entries, const float* d_extra,
                            uint32_t extra_n, uint64_t* d_block, hipStream_t stream);
// world blocks -> heaps [nq][R] (key | value << 32) + sizes (0xffffffff: some rank's block overflowed / a query was not
// ordered: the caller regrows and repeats, or falls back).  Three launches: per-query totals + prefix, interleave into
// ONE stream per query in global scan order (assign slot, rank, position), wave-per-query replay with the heap in
// registers.  world <= 16, ma * world <= dist_interleave_max_cells(), R <= replay_wave_max_R().
// Scratch of the caller: d_moff [nq] u64, d_mcnt / d_info [nq] u32, d_merged [world x entries of a block] u64.
uint32_t replay_wave_max_R();
size_t dist_interleave_max_cells();
// d_status (optional, 3 words, may be host-mapped): [0] = OR of the header bits that void the merge (bit6 block too
// small, bit7 a rank must re-run / failed, bit8 a query shipped unordered), [1] = entries the fullest rank block needs,
// [2] = 1 when a rank's stream of one of the queries interleaved HERE was not grouped by ascending assign slot (an invariant
// of the ordering passes; the heaps of such a batch are not to be used).
// q0 / qstep: only queries q0, q0 + qstep, ... are interleaved and replayed, and their heaps / sizes are written DENSELY
// (heap w = query q0 + w * qstep) — the share of a rank that splits the replay with its peers (q0 = rank, qstep = world; the
// shares travel by a second, small all-gather and launch_dist_heaps_unpack puts them in query order).  Totals, prefix and the
// status words always cover all nq queries (every rank must reach the same verdict).
hipError_t launch_dist_merge(const uint64_t* d_gathered, size_t block_words, int world, int nq, int ma, uint32_t R,
                             uint64_t* d_moff, uint32_t* d_mcnt, uint32_t* d_info, uint64_t* d_merged, uint64_t* d_heaps,
                             uint32_t* d_heap_sizes, hipStream_t stream, uint32_t* d_status = nullptr, int q0 = 0, int qstep = 1);
// d_all = the gathered shares, `share_words` u64 each: heaps u64[per][R], sizes u32[per] (padded to u64), one "broken stream" word;
// rank r's j-th heap is query j * world + r.  A set word in ANY share raises d_sizes[nq + 2] (the batch's status word 2) on this rank.
hipError_t launch_dist_heaps_unpack(const uint64_t* d_all, size_t share_words, int world, int per, int nq, uint32_t R, uint64_t* d_heaps,
                                    uint32_t* d_sizes, hipStream_t stream);
// The loopback stand-in for an all-gather (qadc_dist_init_loopback): the block into all `world` slots, one kernel.
hipError_t launch_replicate_block(const void* d_src, void* d_dst, size_t words, int world, hipStream_t stream);
// Level-path batches: {offset, count, flags}[nq] (the three arrays launch_dist_pack takes) from the query states the
// ordering pass left; bit7 of the flags = the collect call must redo the merge.
hipError_t launch_dist_src_from_states(const QueryState* d_qs, int nq, uint32_t out_cap, uint32_t* d_src, hipStream_t stream);
// The pack step of a merge enqueued together with its batch: the streams are described by the query kernels' own
// {flags, entries} records in device memory (stream q at q * qcap), no host-provided offsets.
hipError_t launch_dist_pack_qflags(const uint32_t* d_qflags, int nq, const uint64_t* d_stream, uint32_t qcap, uint32_t cap_entries,
                                   uint64_t* d_block, hipStream_t stream);
// kv_binheap push replay, ONE WAVE per query, heap in registers (the lanes sift all levels of a push at once and
// pre-filter 64 stream entries per ballot): stream[off[q] .. off[q] + cnt[q]); info[q] bit0 = skip (size 0),
// bit1 = leave to the host (size 0xffffffff).
// (q0 / qstep: wave w replays query q0 + w * qstep and writes heap w — see launch_dist_merge)
hipError_t launch_replay_heap_wave(const uint64_t* d_stream, const uint64_t* d_off, const uint32_t* d_cnt, const uint32_t* d_info,
                                   int nq, uint32_t R, uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream, int q0 = 0, int qstep = 1);
// ... on the single-GPU layout of the query kernel's streams: query q's stream at q * cap, {flags, entries} in d_qflags[4q..].
// ... on the level path's layout: the compact ordered output of sort_cands_kernel, described by the query states.
hipError_t launch_replay_heap_wave_states(const QueryState* d_qs, const uint64_t* d_stream, uint32_t out_cap, int nq, uint32_t R,
                                          uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream);
hipError_t launch_replay_heap_wave_qflags(const uint32_t* d_qflags, const uint64_t* d_stream, uint32_t cap, int nq, uint32_t R,
                                          uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream);

// A failed per-device setup step of a launcher (dynamic-LDS opt-in) since the last call, or hipSuccess.
hipError_t take_launch_error();

void launch_scan_i8(int M, int variant, const ScanItem* d_items, int nitems, int wgs_per_item,
                    const int8_t* d_qtables, QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands,
                    uint32_t cap_per_query, uint32_t R, hipStream_t stream);

// Multi-query streaming scan: groups of up to 8 consecutive runs (all over the same codes, one per query) share
// ONE pass; wgs_per_group workgroups of 256 threads per group, sibling-major over the groups.
// narrow != 0: the build in which a group whose seats 4..7 are empty takes the 4-seat form (8-byte rows, half the LDS cycles
// per lookup) — for the IVF second phase; the flat list's bound levels take the 8-seat-only build (narrow = 0), which the
// two-body build would slow by 5-8 %.
void launch_scan_i8_mq(int M, const ScanItem* d_items, int nitems, int wgs_per_group, const int8_t* d_qtables,
                       QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cap_per_query, uint32_t R,
                       hipStream_t stream, int narrow = 1);

// Small-run variant of the scan (256-thread workgroups, unreplicated tables): IVF partitions, early levels.
void launch_scan_i8_small(int M, const ScanItem* d_items, int nitems, int wgs_per_item, const int8_t* d_qtables,
                          QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cap_per_query, uint32_t R,
                          hipStream_t stream);

// One workgroup per query: bitonic sort of the query's candidates into scan order (level, assign slot,
// position), padding-lane replays expanded, written compactly to d_out_keys / d_out_vals at the prefix
// offset of the query.  Queries with more than kSortCap candidates (or an overflowed region) are left
// to the host (flags bit2 stays clear).
// Sorted entries are written as u64 = key | value << 32 | assign slot << 40; every query gets its QueryOut record.
void launch_sort_cands(QueryState* d_qs, const Cand* d_cands, uint32_t cap_per_query, int nq, QueryOut* d_qout,
                       uint64_t* d_entries, uint32_t out_cap, CandHeader* d_hdr, hipStream_t stream,
                       uint64_t* d_dev_entries = nullptr);

// kv_binheap<unsigned,int8_t> on the device: one wave per query replays the query's ordered stream (the copy
// launch_sort_cands left in d_dev_entries) through a max-heap of capacity R in LDS, sentinel (0,127) first.
// d_heaps[q][R] = key | value << 32 in heap-array order; d_heap_sizes[q] = heap size, or 0xffffffff when the
// query was not ordered on the device (the host replays it).
void launch_replay_heap(const QueryState* d_qs, const uint64_t* d_entries, uint32_t out_cap, int nq, uint32_t R,
                        uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream);

// d_fc_init[2q] = values of query q written unfiltered (phase A), d_fc_init[2q+1] = capacity of its buffer.
// sum_mode: grouping of the float adds (qadc_float_sum.h): 1 = as the reference is compiled, 0 = source order.
void launch_start_scan_f32(int M, int sum_mode, const StartItem* d_items, int nitems, int wgs_per_item,
                           const float* d_ftables, float* d_fc, uint64_t fc_stride, const uint32_t* d_fc_init,
                           QueryState* d_qs, hipStream_t stream);

// Multi-query form: groups of up to 8 consecutive items that share codes / n / out_off / filter (one per query)
// are evaluated in one pass; values bit-identical to launch_start_scan_f32.
void launch_start_scan_mq(int M, int sum_mode, const StartItem* d_items, int nitems, int wgs_per_group, const float* d_ftables, float* d_fc,
                          uint64_t fc_stride, const uint32_t* d_fc_init, QueryState* d_qs, hipStream_t stream);

// Per-block (min float ADC distance, lowest position) over a whole partition; host reduces the blocks.
void launch_float_top1(int M, int sum_mode, const uint8_t* d_codes, uint32_t n, const float* d_ftable, float* d_val, uint32_t* d_pos,
                       int blocks, hipStream_t stream);

// ---- host feeders on the device --------------------------------------------------
// Coarse assignment of index_db::assign_compute_residuals (databases.hpp:201-211): the `ma` nearest of K
// centroids per query, ascending by squared L2 distance (lower index first on ties).  d_dist = scratch [nq][K].
// database build (N4): residual + optional OPQ rotation of vectors already assigned; k-means centroid update
void launch_residual_rotate(const float* d_vectors, uint64_t n, int dim, const float* d_coarse, const int32_t* d_assign,
                            const float* d_rotation, float* d_out, hipStream_t stream);
void launch_kmeans_update(const float* d_vectors, uint64_t n, int dim, int K, const int32_t* d_assign, float* d_centroids,
                          int div_mode, hipStream_t stream);
// Coarse assignment of nq queries: the reference's find_k_neighbors (neighbors.cpp:30-76) — expansion distances
// (||q||^2 + ||c||^2) - 2 q.c (compute_cross_dists_blas, distances.hpp:151-183: norms as compiled under sum_mode 1, the product one
// sequential dot) into d_dist [nq][K], then the ma nearest per query as its heaps select them (exact ties included) into
// d_assign [nq][ma].  d_cnorm [K] = launch_row_sqnorm of the centroids (the caller's, once per centroid set); d_qnorm [nq] scratch.
void launch_row_sqnorm(const float* d_rows, int n, int dim, int sum_mode, float* d_out, hipStream_t stream);
void launch_coarse_assign(const float* d_queries, const float* d_coarse, int nq, int K, int dim, int ma, float* d_qnorm,
                          const float* d_cnorm, int sum_mode, float* d_dist, int32_t* d_assign, hipStream_t stream);
// Residual + per-query distance tables (compute_dists_single_simd_cg's result, distances.hpp:294-311):
// tables[q][a][m][c] = sum_d ((x - centroid[assign])[m*ds+d] - codebook[m][c][d])^2; sum_mode 1: added like the
// reference's fmanorm as compiled (AVX lanes + fused multiply-add + reduceadd tree: direct_sqdist in qadc_kernels.hip),
// sum_mode 0: one sequential sum in ascending d.
// d_coarse == nullptr (flat DB): residual = query.  d_rotation != nullptr (OPQ): the residual is rotated first,
// rotated[r] = sum_c x[c] * rotation[r][c]  (opq::rotate_multiple_vectors, quantizers.hpp:289-301).
// expansion != 0: the BLAS-expansion form (||v||^2 + ||c||^2) - 2 v.c of compute_cross_dists_blas
// (distances.hpp:151-183, 277-292) — what the reference evaluates for ma > 1 and in batch mode; may yield negatives.
void launch_build_tables(const float* d_queries, const float* d_coarse, const int32_t* d_assign, const float* d_codebooks,
                         const float* d_rotation, int nq, int ma, int M, int dim, int expansion, int sum_mode,
                         float* d_ftables, hipStream_t stream);

// R-th smallest of each query's stored float values (one workgroup per query) -> QueryState::qmax
// (FLT_MAX if fewer than R values).  max_passes < 4: upper bound only (survivor filter of the pre-scan).
// d_qtables != nullptr: the same workgroup then runs QuantizerMAX<int8> for the query (qmin, in-place negative
// clamp of d_ftables, int8 tables) — the last step of the float chain, saving a launch.
// export_vals != nullptr (needs max_passes = 4): also writes the query's R smallest values to export_vals[q][R]
// (padded with FLT_MAX when there are fewer) and QueryState::flags to export_flags[q] — the sharded pre-scan.
void launch_select_kth(const float* d_fc, uint64_t fc_stride, const uint32_t* d_fc_init, int nq, uint32_t R,
                       QueryState* d_qs, int max_passes, float* d_ftables, int8_t* d_qtables, int table_dim_all,
                       int quant_mode, hipStream_t stream, float* export_vals = nullptr,
                       uint32_t* export_flags = nullptr, uint32_t* d_front_out = nullptr, int small_wg = 0);
// d_front_out (optional): {flags & 3, qmin, qmax, 0} per query = the front_in record of scan_query_kernel's HEAD; small_wg:
// 256-thread workgroups (a batch of many queries beside running scans) instead of 1024.

// Stream-layout probe: a launch of spin_wgs two-per-CU workgroups spinning spin_ticks (100 MHz wall clock) each on stream a, then a
// one-wave marker on stream b; d_t[0] = first spin workgroup's start (initialise to ~0), [1] = last one's end (0), [2] = marker start.
hipError_t launch_stream_probe(unsigned long long* d_t, int spin_wgs, uint32_t spin_ticks, hipStream_t a, hipStream_t b);

// Key range of d_vals[q][nvals] into QueryState::sel_nmin / sel_max (injected pre-scan values).
void launch_prescan_minmax(const float* d_vals, uint32_t nvals, int nq, QueryState* d_qs, hipStream_t stream);

// PQ encode of device-resident vectors [n][dim] with codebooks [M][16][dim/M] -> codes [n][M/2].  form 1 = the reference's
// (find_k_neighbors with k = 1 on the BLAS-expansion distances), 0 = direct sum (x - c)^2; sum_mode: the norms' grouping.
void launch_pq_encode(const float* d_vectors, uint64_t n, int M, int dim, const float* d_codebooks, int form, int sum_mode,
                      uint8_t* d_codes, hipStream_t stream);

void launch_fill_codes(uint8_t* d_dst, uint64_t first_word, uint64_t nwords, uint64_t seed, hipStream_t stream);

void launch_deinterleave(uint8_t* d_rowmajor, const uint8_t* d_inter, uint32_t n, int cs, hipStream_t stream);

// All-code candidate values min(127, sum) (diagnostic / parity helper; not on the query path).
void launch_candidates_i8(int M, const uint8_t* d_codes, uint64_t n, const int8_t* d_qtable, int8_t* d_out,
                          hipStream_t stream);

}  // namespace qadc
