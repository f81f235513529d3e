/*
 * qadc.h — C-ABI of the MI355X-native Quick-ADC scan engine (libqadc_hip.so).
 *
 * Drop-in boundary for ONE path of technicolor-research/quick-adc: the 4-bit PQ scan + top-R
 * that lives behind the reference's duck-typed ScannerType.  Every entry point
 * names the reference interface it replaces (paths relative to the reference tree).  Plain
 * pointers and sizes only; no C++ types, no exceptions cross this boundary.  All functions
 * return QADC_OK (0) or a negative QADC_E_* code; qadc_last_error() gives the message of the
 * last failure on the calling thread.
 *
 * Threading: one host thread drives one index (as the reference's query loop,
 * query_common.hpp:351-365).  One index = one GPU (one process per GPU for multi-GPU).
 */
#ifndef QADC_H_
#define QADC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QADC_OK 0
#define QADC_E_ARG (-1)       /* bad argument (unsupported M, mixed labels, not finalized, ...) */
#define QADC_E_HIP (-2)       /* HIP runtime failure */
#define QADC_E_CAPACITY (-3)  /* candidate buffer too small even after regrowth / caller buffer too small */
#define QADC_E_STATE (-4)     /* call out of order */

typedef struct qadc_index qadc_index;

const char* qadc_last_error(void);
const char* qadc_version(void);

/* ---------------------------------------------------------------------------------------------
 * Database side — replaces scanner_4::prepare_database (db_query_4.cpp:210-228) and the layout
 * step interleave_partition_4 (simd_layout.hpp:55-65).  The device layout is plain row-major
 * [n][M/2] (one coalesced 16-byte load per lane); the 16-code block transpose is an AVX2 need.
 * ------------------------------------------------------------------------------------------- */

/* Optional, idempotent: creates the library's per-device set of HIP streams NOW (the first qadc_index_create on the device does
 * it otherwise).  The set — scan, copy, ordering, front, collectives, merge — is created once per process and device, back to
 * back, because which hardware queue and compute pipe a stream lands on depends on the streams that exist already (DESIGN.md
 * section 5).  Four of the seven are highest-priority streams and the runtime keeps at most four queues per priority: a process that
 * creates a communicator (RCCL / torch.distributed "nccl": one more highest-priority stream) BEFORE the set measured a 14-30 %
 * slower multi-GPU IVF batch (40-60 % with the priorities of rounds 1-4: profiles/r05_queue_map_rccl.txt, r05_stream_priorities.txt);
 * created AFTER the set it costs nothing.  So: call this (or create the
 * first index) right after the process selected its GPU and before it initialises any communicator.  No reference counterpart
 * (the reference is single-process CPU code).  After hipDeviceReset() the set is rebuilt by the next call / index. */
int qadc_device_prepare(int device_id);

/* Diagnostic of the stream layout (no reference counterpart): launches a kernel whose workgroups wait for CUs for several rounds on
 * stream `a` of the device's set and a one-wave marker on stream `b` right behind it (streams: 0 scan, 1 copy, 2 ordering, 3 front,
 * 4 alternative scan, 5 collectives, 6 merge).  *wait_us = when the marker started, counted from the first workgroup of the long
 * launch; *spin_us (optional) = how long that launch lasted.  A marker that starts within a few microseconds sits on another
 * compute pipe; one that waits most of spin_us shares the long launch's pipe (or its hardware queue): what a scan does to the
 * collectives or to the next batch's front when the process created other queues before the set. */
int qadc_stream_probe(int device_id, int a, int b, double* wait_us, double* spin_us);
/* The check a deployment runs once: "<creation order of the set> | ok" or "... | <pairs still obstructed, with the marker's
 * wait>", from ten probes of the pairs that matter (nothing may hold up the scan stream but its idle alternative; copy, ordering
 * and collectives must not hold up the front stream; ordering and front not the collectives').  ~3 ms, on demand only.  Not "ok"
 * means the process created queues before the set: call qadc_device_prepare earlier (a search over pad streams that would repair
 * the layout from inside was built and does not converge: DESIGN.md section 5). */
const char* qadc_stream_layout(int device_id);

/* M = 16 or 32 sub-quantizers of 4 bits (get_simd_scan_func_epi8, db_query_4.cpp:22-35). */
int qadc_index_create(qadc_index** out, int M, int device_id);
int qadc_index_destroy(qadc_index* idx);

/* Append partitions given as base_db::get_partition() yields them (databases.hpp:50-55):
 * row-major codes [sizes[p]][M/2], labels[p] = u32[sizes[p]] or labels == NULL for a flat DB
 * (key = position).  Host buffers are copied to the GPU; the caller may free them afterwards
 * (the reference does, db_query_4.cpp:190).  All-or-none labels (db_query_4.cpp:118-124). */
int qadc_index_add_partitions(qadc_index* idx, int part_count, const uint8_t* const* codes,
                              const uint32_t* const* labels, const uint32_t* sizes);

/* Append one partition already laid out in the reference's block layout [ceil(n/16)][M/2][16]
 * (what scanner_4::parts holds, db_query_4.cpp:171-177); converted on the GPU. */
int qadc_index_add_partition_interleaved(qadc_index* idx, const uint8_t* interleaved,
                                         const uint32_t* labels, uint32_t size);

/* Append one partition whose row-major codes (and labels) already live in device memory of this
 * GPU (borrowed, not freed; must stay valid; 16-byte aligned, readable up to size*M/2 rounded up
 * to 16 bytes). */
int qadc_index_add_partition_device(qadc_index* idx, const void* d_codes, const void* d_labels, uint32_t size);

/* Append one synthetic flat partition generated on the GPU: 8-byte word w of the code stream =
 * splitmix64(seed ^ splitmix64(first_word + w)) (reproducible on the CPU). */
int qadc_index_add_partition_synthetic(qadc_index* idx, uint32_t size, uint64_t seed, uint64_t first_word);

/* Multi-GPU: append the local range [first_pos, first_pos + local_n) of a partition of global_n
 * codes (one process per GPU, contiguous ranges in rank order; every range but the last a multiple
 * of 16 codes).  Keys of an unlabeled shard are first_pos + local position.  `starts` = the first
 * starts_count codes of the WHOLE partition (>= max(1, unsigned(global_n * keep))), needed by every
 * rank whose range does not begin the partition: the pre-scan and therefore qmax / the int8 tables
 * are then identical on all ranks without any exchange (the reference keeps the starts in a
 * separate buffer too, db_query_4.cpp:137-145).  The padding-lane replay of the partition's last
 * code happens only on the rank that holds it. */
int qadc_index_add_partition_shard(qadc_index* idx, const uint8_t* codes, const uint32_t* labels, uint32_t local_n,
                                   uint32_t global_n, uint32_t first_pos, const uint8_t* starts, uint32_t starts_count);
/* (synthetic form: the partition's generator stream is `seed`; local_n == 0 keeps only the starts replica) */
int qadc_index_add_partition_synthetic_shard(qadc_index* idx, uint32_t global_n, uint32_t first_pos, uint32_t local_n,
                                             uint64_t seed, uint32_t starts_count);

/* Keys reported for an unlabeled partition are key_base + position (shard offset for multi-GPU). */
int qadc_index_set_key_base(qadc_index* idx, int part, uint32_t key_base);

/* Fix the "starts" sizes: max(1, unsigned(size * keep)) with the product in float
 * (db_query_4.cpp:125-126).  keep is a fraction (the CLI's -k percent * 0.01). */
int qadc_index_finalize(qadc_index* idx, float keep);

int qadc_index_partition_count(const qadc_index* idx);
uint32_t qadc_index_partition_size(const qadc_index* idx, int part);
uint32_t qadc_index_start_size(const qadc_index* idx, int part);

/* Options (30; qadc_option_names() returns the list, comma separated).  None changes WHAT is computed except the three parity
 * switches of the float half; the rest choose paths and sizes, and tests/test_gpu_fuzz.py draws them at random against the oracle.
 *  parity      "quant_mode"  1 = QuantizerMAX as the reference is compiled (one reciprocal, multiply), 0 = its source's division
 *              "sum_mode"    grouping of the float sums of the pre-scan (scan_4, query_common.hpp:72-80), of the direct table form
 *                            (fmanorm, distances.hpp:60-76) and of the expansion form's norms: 1 = as the reference binary adds them —
 *                            it is built with -ffast-math, CMakeLists.txt:7 —, 0 = source order
 *              "table_form"  float tables of qadc_search: 0 direct, 1 BLAS expansion, 2 (default) nns_engine's rule: expansion iff ma > 1
 *  diagnostics "profile"     0/1: HIP-event timing of the scan launches (qadc_profile_read)
 *  path        "wgq"         the one-workgroup-per-query path: 0 never, 1 auto (default), 2 whenever structurally possible
 *              "wgq_group"   large IVF batches take a partition-major second phase: 0 never, 1 auto, 2 whenever possible
 *              "wgq_group_head"  probes per query the head walks before it (default 2; 4 under the multi-GPU merge)
 *              "head_level"  level path: bound levels 0 .. head_level-1 are scanned by ONE head launch (default 5; 0 = off)
 *              "head_wg"     512 = the IVF head in 512-thread workgroups (8 waves per query; default at 16x4), 0 = 1024
 *  level path  "mq"          queries that share a run are scanned 8 per pass (default 1)
 *              "share_variant"  kernel form of such shared launches (bit 6 = sibling-major, bit 3 = chunked; 0 = never share)
 *              "variant"     kernel form of the other launches: bit 2 non-temporal loads, bit 3 chunked tiles, bit 4 the PROBE
 *                            diagnostic (lookups replaced by an XOR: bench.py's measured streaming ceiling; results meaningless)
 *              "front_run_max"  leading levels whose runs are at most this long join the front stream (0 = none)
 *              "wgs_per_item"   workgroups per (query, level) run (0 = auto)
 *              "cand_capacity", "level_base", "level_growth", "small_run", "prescan_sample"   sizes of the candidate regions, the
 *                            bound levels, the small-run kernel's limit, the unfiltered part of the pre-scan
 *  replay      "device_replay_nq" / "device_replay_alone_nq"  batches of at least this many queries replay their streams on the
 *                            device (0 = never) / the same for a batch with nothing else in flight (a synchronous call)
 *  query path  "wgq_split" / "wgq_split_codes"  workgroups a call of one or two queries spreads each over (default 32) / codes each
 *                            keeps at least (2048); small batches of three or more queries: at most 12, four times the codes
 *              "wgq_capacity" / "wgq_cand_cap"  stream entries per query to start with / candidates per query before a batch falls
 *                            back to the level path
 *  multi-GPU   "dist_cap_entries", "dist_device_nq", "dist_shard_replay" (an enqueued merge replays only this rank's share of the
 *              queries and a second, small all-gather shares the heaps; default 1), "dist_shard_front" (feeders + pre-scan +
 *              quantizer of a qadc_search batch are split over the ranks; default 1), "dist_inject_failure" (tests: this rank's
 *              next collect fails before the gather) — after qadc_dist_init only.
 * Streams: the library keeps ONE set of HIP streams per process and device, created by the first index on the device and shared
 * by every later one. */
const char* qadc_option_names(void);
int qadc_set_option(qadc_index* idx, const char* name, double value);

/* Copy codes back (tests / checksums): partition `part`, codes [first, first+count). */
int qadc_index_read_codes(qadc_index* idx, int part, uint32_t first, uint32_t count, uint8_t* out);

/* ---------------------------------------------------------------------------------------------
 * Query side.
 * ------------------------------------------------------------------------------------------- */

/* scanner_4::query_scan (db_query_4.cpp:245-309) for nq queries at once.
 *   assign  [nq][ma]          probed partitions in scan order (index_db::assign_compute_residuals)
 *   tables  [nq][ma][M*16]    float distance tables; MUTATED like the reference (negatives -> 0)
 *   R                         heap capacity (-r)
 * Outputs, per query q (any may be NULL):
 *   keys[q][R], values[q][R], sizes[q]   the heap ARRAYS the reference's kv_binheap would hold
 *                                        after the query (incl. the (0,127) sentinel if it survived)
 *   status[q]   0 ok; 1 = qmax > 1e30: the reference prints "Max quantization bound too high" and
 *               exit(1)s (db_query_4.cpp:271-274); here the query is skipped (sizes[q] = 0)
 *   qmin[q], qmax[q], qtables[q][ma][M][16]   the quantizer inputs/outputs (diagnostics, parity) */
int qadc_query_scan(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R,
                    uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                    float* qmin, float* qmax, int8_t* qtables);

/* Same work, but returns the ordered candidate stream instead of replaying it: pushing
 * (cand_keys[i], cand_vals[i]) for i in [offsets[q], offsets[q+1]) in order into the reference's
 * own kv_binheap<unsigned,int8_t>(R), after bh.push(0,127), leaves it in exactly the state the
 * reference scan would (a superset of its successful pushes, in scan order, padding-lane
 * duplicates included).  This is what a ScannerType::query_scan wrapper calls.
 * cand_capacity = entries available in cand_keys/cand_vals; QADC_E_CAPACITY if too small
 * (offsets[nq] then holds the required count). */
int qadc_query_scan_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R,
                               uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals,
                               uint64_t* offsets, int32_t* status, float* qmin, float* qmax);

/* Integer half only — replaces the scan_avx_4<M> calls (simd_scan.hpp:125-187; call sites
 * db_query_4.cpp:287-308): the caller supplies int8 tables qtables[nq][ma][M][16] with entries
 * in [0,127] (what QuantizerMAX<int8_t> produces) and gets the heap arrays. */
int qadc_scan_i8(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R,
                 uint32_t* keys, int8_t* values, int32_t* sizes);

int qadc_scan_i8_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R,
                            uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets);

/* Float pre-scan only — replaces scanner_4::query_scan_start + tmp_bh.max()
 * (db_query_4.cpp:230-242, 259): qmax[q] = R-th smallest float ADC distance over the starts of
 * the probed partitions (FLT_MAX when fewer than R starts). */
int qadc_scan_start(qadc_index* idx, int nq, int ma, const int32_t* assign, const float* tables, int R, float* qmax);

/* Asynchronous form of qadc_query_scan for throughput: submit enqueues all GPU work of a batch on the
 * index's streams and returns; collect waits for that batch, replays and fills the outputs.  slot is
 * 0..7 (up to eight batches in flight: one being collected, one scanning, the others queued with their
 * pre-scan fronts running ahead); a slot must be collected before it is submitted again.  `tables` must stay valid until
 * collect (it is mutated then). */
int qadc_query_scan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R);
int qadc_query_scan_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes,
                            int32_t* status, float* qmin, float* qmax, int8_t* qtables);

/* Sharded pre-scan for multi-GPU (no counterpart in the reference, whose scan is one process).  With every rank
 * holding a replica of the starts, the float pre-scan (query_scan_start, db_query_4.cpp:230-242) would be repeated
 * on every rank; instead rank r of w pre-scans slice r of w of every probed partition's starts:
 *   qadc_prescan_submit   enqueues that pass (own buffers and stream: it may overlap an uncollected batch);
 *   qadc_prescan_collect  returns vals[nq][R] = the R smallest distances of the slice per query, FLT_MAX-padded;
 *   the caller gathers vals over the ranks (one small all-gather) into gathered[nq][w*R] and submits the batch with
 *   qadc_query_scan_submit_prescanned, which selects qmax from the gathered values instead of pre-scanning.
 * The R-th smallest of the union of the slices' R smallest values is the R-th smallest of all starts, so qmax, the
 * int8 tables and the heaps are bit-identical to qadc_query_scan_submit.  Collect with the usual collect calls. */
int qadc_prescan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R,
                        int slice, int nslices);
int qadc_prescan_collect(qadc_index* idx, int slot, float* vals);
int qadc_query_scan_submit_prescanned(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables,
                                      int R, const float* prescan_vals, int nvals);

/* collect variant returning the ordered candidate stream (see qadc_query_scan_candidates): what a
 * rank hands to the cross-GPU gather.  cand_slots[i] (nullable) = position in assign[] of the probed
 * partition entry i comes from: with every partition range-sharded over the ranks, the global scan order is
 * (assign slot, rank, position), so the merge needs the slot boundaries of each rank's stream.
 * On QADC_E_CAPACITY the result is kept: call again with buffers of offsets[nq] entries. */
int qadc_query_scan_collect_candidates(qadc_index* idx, int slot, uint64_t cand_capacity, uint32_t* cand_keys,
                                       int8_t* cand_vals, uint16_t* cand_slots, uint64_t* offsets, int32_t* status,
                                       float* qmin, float* qmax);

/* ---------------------------------------------------------------------------------------------
 * "Next" row N1 of SURVEY.md §8(f): the host feeders of the path, on the device.  Queries in, heaps out;
 * assignment, residuals and float tables never cross PCIe.  Replaces:
 *   index_db::assign_compute_residuals  (databases.hpp:201-211; find_k_neighbors, neighbors.cpp:30-76)
 *   flat_db::assign_compute_residuals   (databases.hpp:93-101)
 *   opq::rotate_multiple_vectors        (quantizers.hpp:289-301)
 *   compute_dists_single_simd_cg        (distances.hpp:294-311)
 * followed by the same chain as qadc_query_scan.  Pinned to the reference's own code: the direct
 * table form (fmanorm as compiled), the residuals, the SELECTION of the ma nearest — find_k_neighbors' heaps, exact
 * distance ties included (what the heap's history and kv_binheap::sort leave, neighbors.cpp:18-28, 47-71) — and the NORM half
 * of the BLAS-expansion form (||v||^2 + ||c||^2 as compute_cross_dists_blas hands it to sgemm, distances.hpp:151-215): the
 * coarse distances under the selection and the ma > 1 tables are in that form.  Restated, unpinned: the products of cblas_sgemm
 * (OpenBLAS is not in the image: one sequential dot each); bit-exact against quick-adc_amd/host/query_driver.hpp and the
 * oracle, which evaluate the same sums on the host.
 * ------------------------------------------------------------------------------------------- */
/* codebooks [M][16][dim/M] (base_pq::centroids_flat order). */
int qadc_index_set_pq(qadc_index* idx, int dim, const float* codebooks);
/* OPQ rotation [dim][dim] (opq::rotation): residuals are rotated before the tables are built,
 * rotated[r] = sum_c x[c] * rotation[r][c] (opq::rotate_multiple_vectors, quantizers.hpp:289-301).  NULL = plain PQ. */
int qadc_index_set_rotation(qadc_index* idx, const float* rotation);
/* coarse centroids [K][dim], K == partition count (partition p belongs to centroid p).  Not called = flat. */
int qadc_index_set_coarse(qadc_index* idx, int K, const float* centroids);
/* queries [nq][dim]; outputs as qadc_query_scan; assign_out [nq][ma] (nullable) = probed partitions. */
int qadc_search(qadc_index* idx, int nq, const float* queries, int ma, int R, uint32_t* keys, int8_t* values,
                int32_t* sizes, int32_t* status, int32_t* assign_out);
int qadc_search_submit(qadc_index* idx, int slot, int nq, const float* queries, int ma, int R);
int qadc_search_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                        int32_t* assign_out);

/* "Next" row N4 (database build): PQ encode on the device — base_pq::encode_multiple_vectors for plain PQ
 * (quantizers.hpp:222-245): per sub-quantizer find_k_neighbors(count, 16, sq_dim, k = 1, ...) (neighbors.cpp:30-76) — the
 * BLAS-expansion distances (||v||^2 + ||c||^2) - 2 v.c of compute_cross_dists_blas (distances.hpp:151-215) pushed in
 * centroid order into a capacity-1 kv_binheap: the first strict minimum — packed by multiple_set_bits_4
 * (quantizers.hpp:49-68).  The norms add as the reference is compiled (pinned to its own text), the
 * product is one sequential dot (the reference's is OpenBLAS's sgemm: restated).  d_vectors [n][dim] float and d_codes
 * [n][M/2] are device pointers of `device_id` (the _host form stages host buffers).
 * _mode: encode_form 1 = that (what the plain entry points do), 0 = the direct form sum (x - c)^2, first minimum (this
 * library's encoder before round 6: vectors nearly equidistant from two centroids can get another code than the
 * reference writes); sum_mode 1 = norms as compiled, 0 = sequential. */
int qadc_pq_encode(int M, int dim, const float* codebooks, const void* d_vectors, uint64_t n, void* d_codes, int device_id);
int qadc_pq_encode_host(int M, int dim, const float* codebooks, const float* vectors, uint64_t n, uint8_t* codes,
                        int device_id);
int qadc_pq_encode_mode(int M, int dim, const float* codebooks, const void* d_vectors, uint64_t n, void* d_codes,
                        int encode_form, int sum_mode, int device_id);
int qadc_pq_encode_host_mode(int M, int dim, const float* codebooks, const float* vectors, uint64_t n, uint8_t* codes,
                             int encode_form, int sum_mode, int device_id);

/* N4, the rest of the build path (host buffers in and out, any device).
 * qadc_ivf_encode_host = the compute of index_db::add_vectors (databases.hpp:270-298) / flat_db::add_vectors (136-156):
 *   nearest coarse centroid per vector (K > 0; find_k_neighbors with k = 1 on the expansion distances: first strict
 *   minimum), residual, optional OPQ rotation rotated[r] = sum_c x[c] * rotation[r][c] (quantizers.hpp:289-301; rotation
 *   [dim][dim] or NULL), PQ encode (quantizers.hpp:222-245; qadc_pq_encode above, _mode likewise).  assign_out [n]
 *   (nullable; untouched when K == 0), codes [n][M/2].  The caller dispatches (assign, code, label = index + offset) to its partitions in vector order
 *   like databases.hpp:291-297 (host/db_build.hpp does).
 * qadc_kmeans_iterations_host = kmeans_fast_iterations_thread (databases.cpp:50-90): `iters` rounds of assign-to-nearest
 *   + centroid = (sum of the members in ascending vector order) * (1.0f / count) — AS THE REFERENCE IS COMPILED: under its
 *   -ffast-math g++ replaces the source's division (databases.cpp:83-88) by a reciprocal and a multiplication, pinned to those
 *   loops compiled here with the reference's flags (tests/test_oracle_float_ref.py); an empty cluster becomes NaN either way;
 *   centroids [K][dim] in and out.  qadc_kmeans_iterations_host_mode: div_mode 1 = that, 0 = the source's division.  The reference seeds these iterations with two OpenCV
 *   k-means++ iterations (databases.cpp:96-113) — third-party, not restated: the caller provides the seed. */
int qadc_ivf_encode_host(int M, int dim, const float* codebooks, const float* rotation, int K, const float* coarse,
                         const float* vectors, uint64_t n, int32_t* assign_out, uint8_t* codes, int device_id);
int qadc_ivf_encode_host_mode(int M, int dim, const float* codebooks, const float* rotation, int K, const float* coarse,
                              const float* vectors, uint64_t n, int32_t* assign_out, uint8_t* codes, int encode_form,
                              int sum_mode, int device_id);
int qadc_kmeans_iterations_host(const float* vectors, uint64_t n, int dim, int K, float* centroids, int iters,
                                int32_t* assign_out, int device_id);
int qadc_kmeans_iterations_host_mode(const float* vectors, uint64_t n, int dim, int K, float* centroids, int iters,
                                     int32_t* assign_out, int div_mode, int device_id);

/* Host-only helper (no GPU involved): push (keys[i], vals[i]), i = 0..n-1, in order into an empty
 * heap of capacity R with kv_binheap<unsigned,int8_t>::push semantics (binheap.hpp:75-116), after
 * an optional (0,127) sentinel (db_query_4.cpp:276), and return the heap arrays.  This is the
 * replay the other entry points apply to the device's candidate stream. */
int qadc_replay_i8(uint64_t n, const uint32_t* keys, const int8_t* vals, int R, int push_sentinel,
                   uint32_t* out_keys, int8_t* out_vals, int32_t* out_size);

/* Host-only helper: kv_binheap<unsigned,int8_t>::sort_keys (binheap.hpp:129-137) of a heap whose arrays are
 * heap_keys / heap_vals[size] (as returned by the entry points above): the keys ascending by value, tied values in
 * the order std::sort leaves them on that array — what a caller of the reference gets from bh.sort_keys(). */
int qadc_sort_keys_i8(int size, const uint32_t* heap_keys, const int8_t* heap_vals, uint32_t* out_keys);

/* Host half of the multi-GPU merge: `gathered` = the world int32 buffers of buflen words each that the ranks
 * contributed to ONE all-gather, each laid out as [nq counts][cap keys][ceil(cap/4) words of int8 values]
 * [only when ma > 1: ceil(cap/2) words of u16 assign slots][anything else].  Replays queries q_first, q_first + q_step, ... in
 * global scan order (assign slot, rank, position) after the (0,127) sentinel; keys/vals are [nq][R], rows of other
 * queries are left untouched.  No GPU needed. */
int qadc_merge_streams_i8(int world, int nq, int R, uint64_t cap, int ma, const int32_t* gathered, uint64_t buflen,
                          int q_first, int q_step, const int32_t* status, uint32_t* keys, int8_t* vals, int32_t* sizes);

/* Diagnostic: all candidate values min(127, sum) of one partition for one int8 table [M][16]. */
int qadc_candidates_i8(qadc_index* idx, int part, const int8_t* qtable, int8_t* out);

/* Diagnostic / recall ground truth: the exact float-ADC nearest code of one
 * partition for one float table [M*16]: smallest distance (scan_4<M> summation order), lowest
 * position on ties.  key = label, or key_base + position. */
int qadc_float_top1(qadc_index* idx, int part, const float* table, uint32_t* out_key, uint32_t* out_pos, float* out_dist);

/* ---------------------------------------------------------------------------------------------
 * Measurement (bench.py): HIP-event timing of the int8 scan kernel on the index's stream.
 * Enabled with qadc_set_option(idx, "profile", 1).  Totals since the last reset.
 * ------------------------------------------------------------------------------------------- */
/* ---- Multi-GPU (north star: "the code list shards across the 8 GPUs of one node with a final RCCL
 * allgather of per-shard top-k over xGMI").  One process per GPU; every rank holds a contiguous range of every
 * partition (qadc_index_add_partition_shard) and submits the same batches.  What is gathered is each shard's ordered
 * PUSH STREAM, not its final top-R: re-pushing final heaps is not exact under ties (binheap.hpp:75-116 resolves ties by
 * push order).  qadc_dist_collect replaces qadc_query_scan_collect: it packs this rank's streams (already in device
 * memory), runs ONE ncclAllGather (device to device, no host staging), and replays the world's streams in global
 * scan order (assign slot, rank, position) on the GPU, one wave per query — at collect time every rank replays every query
 * (no second collective); a merge enqueued with its batch (below) shards the replay by query and shares the heaps with a
 * second, small all-gather.  `extra` (optional, extra_n floats per rank) rides in the same all-gather and comes
 * back as extra_out[world][extra_n]: the multi-rank loop of bench.py ships the next batch's sharded pre-scan values
 * this way (qadc_prescan_submit).  RCCL is loaded with dlopen by qadc_dist_unique_id / qadc_dist_init; a single-GPU
 * user never loads it.  world <= 16.  Any R: the device merge (one wave per query, heap in registers) holds R <= 320,
 * larger heaps take the host-share replay.  A query some rank could not order on the device (> 16384 candidates) is
 * sorted on that rank's host and shipped in the same gather.  A rank whose batch failed locally still takes part in the
 * gather with a failure flag in its header, so every rank returns an error instead of one rank leaving the others
 * blocked.  A batch submitted after qadc_dist_init may still be collected with the plain collect calls (host replay
 * of this rank's streams only).  Large batches (>= "dist_device_nq" queries: IVF; on either scan path) have their merge —
 * pack, all-gather, interleave, replay — ENQUEUED WITH THE BATCH, behind its scan, so qadc_dist_collect only waits for it
 * (option "dist_async", default 1): after qadc_dist_init every rank must therefore SUBMIT the same batches in the same
 * order, not just collect them (the all-gather of such a batch is issued by its submit call).  Such merges SHARD THEIR REPLAY
 * (option "dist_shard_replay", default 1: rank r replays queries q = r (mod world); the heap shares' all-gather is issued by a later
 * submit call, or by collect).  qadc_search batches of that kind also SHARD THEIR FRONT (option "dist_shard_front", default 1): what is per query rather than per code — coarse
 * assignment, residual tables, pre-scan, select, quantizer — runs on rank r for queries [r * ceil(nq / world), ...) only, and
 * one more all-gather (issued by the submit call as well) ships assign[], the int8 tables and (flags, qmin, qmax) of every
 * query to every rank before the sharded scan.
 *   rank 0:      qadc_dist_unique_id(id)  ... ship the 128 bytes to the other ranks by any means ...
 *   every rank:  qadc_dist_init(idx, rank, world, id);  then per batch  qadc_query_scan_submit(...); qadc_dist_collect(...) */
#define QADC_DIST_ID_BYTES 128
int qadc_dist_unique_id(uint8_t* id128);
int qadc_dist_init(qadc_index* idx, int rank, int world, const uint8_t* id128);
int qadc_dist_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                      const float* extra, int extra_n, float* extra_out);
int qadc_dist_shutdown(qadc_index* idx);
/* The same merge over a caller-supplied all-gather instead of RCCL (no counterpart in the reference): `fn` gathers
 * bytes_per_rank bytes of DEVICE memory from every rank into d_recv[world][bytes_per_rank] (rank order), is called with
 * the producing work already enqueued on hip_stream, and must have completed (or be ordered on hip_stream) when it
 * returns 0; any other return value fails the collect on this rank.  Every rank must call it the same number of times
 * with the same size — qadc_dist_collect guarantees that, including on its retry and failure paths.  Uses: ranks that
 * share one GPU (a single-GPU box exercising world > 1), hosts without librccl, MPI or other fabrics. */
typedef int (*qadc_allgather_fn)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
int qadc_dist_init_transport(qadc_index* idx, int rank, int world, qadc_allgather_fn fn, void* ctx);
/* Measurement aid: ONE process stands in for rank `rank` of `world` — the transport copies this rank's block into every
 * slot of the gather, so the merge replays a world's worth of entries while only this rank's shard is scanned.  Results
 * are NOT the database's answers (every stream counts `world` times); tools/ivf_shard_sizes.py times one of 8 ranks' step
 * on one GPU this way. */
int qadc_dist_init_loopback(qadc_index* idx, int rank, int world);
/* Built-in transport for qadc_dist_init_transport: host-staged all-gather through a POSIX shared-memory segment `name`
 * ("/something", unique per run; rank 0 creates it, the others wait up to timeout_s seconds — 0 = 120 s — for it).
 * slot_bytes = largest block a rank may contribute (a gather beyond it fails on every rank alike).  Barriers time out
 * after timeout_s and poison the segment, so a missing rank turns into an error instead of a hang.
 *   qadc_shm_transport_open(name, rank, world, slot_bytes, timeout_s, &ctx);
 *   qadc_dist_init_transport(idx, rank, world, qadc_shm_transport_allgather, ctx);  ...  qadc_shm_transport_close(ctx);
 * _allgather_host exchanges host buffers (no GPU): the CPU-side test of the protocol. */
int qadc_shm_transport_open(const char* name, int rank, int world, uint64_t slot_bytes, double timeout_s, void** out_ctx);
int qadc_shm_transport_allgather(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
int qadc_shm_transport_allgather_host(void* ctx, const void* send, void* recv, uint64_t bytes_per_rank);
int qadc_shm_transport_close(void* ctx);
const char* qadc_shm_transport_error(void);
/* The probed partitions of the batch last collected from `slot` (qadc_search_submit computes assign[] on the GPU;
 * qadc_dist_collect has no assign_out): assign_out [nq][ma]. */
int qadc_slot_assign(qadc_index* idx, int slot, int32_t* assign_out);
/* The int8 tables (QuantizerMAX<int8_t> output, db_query_4.cpp:277-284) of queries [q_first, q_first + q_count) of the
 * batch last collected from `slot`, [q_count][ma][M][16] — still resident on the GPU until the slot is submitted again.
 * qadc_search builds its float tables on the device and returns no tables; this is how a caller (the parity tests: the
 * reference's own scan_avx_4 on the same int8 tables) gets at them. */
int qadc_slot_qtables(qadc_index* idx, int slot, int q_first, int q_count, int8_t* out);
/* Size-balanced placement of whole partitions on `world` ranks (IVF option 1): partitions in descending
 * size order, each to the currently lightest rank (ties: lowest rank).  owner_out[p] = rank of partition p.  A rank adds
 * the partitions it owns in full and the others with local_n = 0 (starts replica only, qadc_index_add_partition_shard),
 * so partition numbering — and assign[] — is the same on every rank and the merge order (assign slot, rank, position)
 * degenerates to (assign slot, position).  Host-only. */
int qadc_place_partitions(int part_count, const uint32_t* sizes, int world, int32_t* owner_out);
/* The merge half of qadc_dist_collect on a caller-assembled gather result (host memory; `world` blocks of block_words
 * u64 each: [nq x {offset, count, flags, 0} as u32][entries = key | value << 32 | assign slot << 40]...): lets a single
 * GPU check the multi-rank replay order.  sizes[q] = -1 when a block reports an overflow / unordered query. */
int qadc_dist_merge_blocks(int device_id, int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                           uint32_t* keys, int8_t* values, int32_t* sizes);
/* The host half of the same merge (few-query batches: every rank replays its share of the queries between two
 * all-gathers), executed for all `world` ranks in turn on the caller's thread — no GPU, no RCCL: the test hook of the
 * code path 2/4/8-rank runs of 32-query batches take. */
int qadc_dist_merge_blocks_host(int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                                uint32_t* keys, int8_t* values, int32_t* sizes);

typedef struct qadc_profile {
    uint64_t scan_launches;   /* launches of the streaming int8 scan kernels (scan_i8_kernel, scan_i8_mq_kernel) */
    uint64_t scan_codes;      /* codes those launches scanned (algorithmic bytes = codes * M/2) */
    double scan_ms;           /* HIP-event time of those launches (one event pair per run of consecutive launches,
                                 i.e. including the ~2 us hand-over between them) */
    uint64_t small_launches;  /* launches of the small-run int8 scan kernel (scan_i8_small_kernel): counted, */
    uint64_t small_codes;     /* not event-timed (an event pair costs as much stream time as such a launch) */
    double small_ms;          /* always 0; per-kernel times of these launches come from rocprofv3 */
    uint64_t start_codes;     /* codes scanned by the float pre-scan */
    double start_ms;          /* float pre-scan + select + quantize, HIP-event duration */
    uint64_t candidates;      /* candidates returned by the device (before padding duplicates) */
    uint64_t regrows;         /* batches re-run because the candidate buffer overflowed */
    double host_replay_ms;    /* host wall time assembling the ordered candidate streams */
    double host_plan_ms;      /* host wall time planning + enqueueing batches */
    double host_heap_ms;      /* host wall time replaying streams through the heap */
    uint64_t host_sorted_queries; /* queries whose candidates the host had to sort (> 16384 candidates) */
    uint64_t mq_launches;     /* of scan_launches: multi-query launches (scan_i8_mq_kernel, 8 queries per pass) */
    uint64_t pass_codes;      /* codes the streaming launches READ: a run's codes once per query, or once per group
                                 of 8 queries in a multi-query launch (LDS row reads = pass_codes * M there) */
    uint64_t wgq_launches;    /* batches scanned by scan_query_kernel (one workgroup per query) */
    uint64_t wgq_queries;     /* queries in them */
    uint64_t wgq_codes;       /* codes they probed (algorithmic bytes = codes * M/2) */
    double wgq_ms;            /* HIP-event time of those launches */
    uint64_t wgq_front_cycles; /* shader cycles the query workgroups spent in pre-scan + select + quantizer (summed over queries) */
    uint64_t wgq_scan_cycles;  /* ... in the int8 scan */
    uint64_t wgq_sort_cycles;  /* ... and in the final candidate sort + ordered stream write */
    uint64_t head_launches;    /* level path: batches whose first bound levels were scanned by one head launch */
    uint64_t group_launches;   /* large IVF batches that took the partition-major second phase ... */
    uint64_t group_fallbacks;  /* ... and those of them whose candidate regions overflowed (redone on the level path) */
    /* the partition-major second phase in detail (profile on): HIP-event times of its three launches and the work in them */
    double group_head_ms;      /* scan_query_kernel in HEAD mode: front (pre-scan, select, quantizer) + the head probes */
    double group_scan_ms;      /* scan_i8_mq_kernel over the regrouped (query, probe) pairs */
    double group_order_ms;     /* order_cands_kernel */
    uint64_t group_head_codes; /* codes the heads walked (algorithmic HBM bytes = codes * M/2) */
    uint64_t group_pairs;      /* (query, probe) pairs of the second phase */
    uint64_t group_seats;      /* seats of their groups (8 per group, 4 for a narrow remainder group): fill = pairs / seats */
    uint64_t group_pass_codes8; /* codes read by 8-seat passes (LDS cycles = codes * M * 4 / 64) */
    uint64_t group_pass_codes4; /* ... by 4-seat passes (LDS cycles = codes * M * 2 / 64) */
    uint64_t group_batches;    /* batches the above figures cover */
    uint64_t front_sharded_batches; /* multi-GPU: qadc_search batches whose front ran on 1/world of the queries per rank */
    uint64_t dist_async_collects;   /* multi-GPU: qadc_dist_collect calls served by a merge enqueued with the batch (one event wait) */
    uint64_t lone_front_launches;   /* lone queries on one long partition whose front ran sliced over workgroups, in a launch of its own
                                       in front of the walk (counted with or without "profile") */
} qadc_profile;

int qadc_profile_read(qadc_index* idx, qadc_profile* out);
int qadc_profile_reset(qadc_index* idx);

#ifdef __cplusplus
}
#endif
#endif /* QADC_H_ */
