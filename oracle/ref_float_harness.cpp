// TEST INFRASTRUCTURE — not product code.
//
// C-ABI harness around the FLOAT half of the reference's query path and the glue around it,
// compiled from the reference's own source text under /root/reference (never copied into this
// repo).  The files that hold these functions (db_query_4.cpp, query_common.hpp, quantizers.hpp,
// databases.hpp, distances.hpp, databases.cpp, db_query.cpp) include Cereal / cblas / OpenCV
// headers the image lacks and so do not compile whole; the functions on the path use none of
// those libraries.  oracle/ref_extract.sh cuts the line ranges listed there (sha256-checked)
// out of the files where they lie into a temporary directory that is on the include path of
// THIS compile only and is deleted right after it.  The x_*.inc names below are those ranges:
//
//   x_quantizers_a/b   quantizers.hpp:24-169,188-246   multiple_set_bits_4, class base_pq
//                                                       (without its cereal save/load templates)
//   x_base_db          databases.hpp:34-63             struct base_db
//   x_query_metrics    query_common.hpp:21-56          struct query_metrics
//   x_scan_funcs       query_common.hpp:59-143         scan_4<NSQ>, scan_standard<T,NSQ>, get_scan_func
//   x_scanner_4        db_query_4.cpp:22-310           QuantizerMAX<T>, struct scanner_4 (whole)
//   x_scanner_simple   db_query.cpp:17-46              struct scanner_simple (BASELINE configs[0])
//   x_distances_a..d   distances.hpp:21-36,60-92,237-275,294-311
//                                                       reduceadd, fmanorm, centroids getters,
//                                                       compute_dists_single_simd_cg<DSQ>
//   x_substract        databases.cpp:24-48             substract_vectors(_from_unique)
//   x_opq_a/z          quantizers.hpp:248-277,324      struct opq (rotation member, setup / set_rotation; without
//                                                       its cblas rotate_* overrides and cereal templates)
//   x_pq_files_a/b     quantizers.cpp:16-46,48-103     the .pq.data / .opq.data readers, parse_data_filename,
//                                                       the pq_from_data_file factory (N3)
//   x_neighbors_heaps  neighbors.cpp:15-28             add_candidates_heaps: the selection half of
//                                                       find_k_neighbors (its distance half is cblas_sgemm) (N1)
//   x_norm_4, x_cross_norms(_4)  distances.hpp:51-57,151-176,185-208
//                                                       norm_4 and compute_cross_dists_blas<DSQ> / <4> UP TO their cblas_sgemm call:
//                                                       the norms and the ||v||^2 + ||c||^2 matrix sgemm receives as C (N4, round 6);
//                                                       this file closes the two function bodies
//   x_kmeans_update    databases.cpp:70-88             the centroid-update loops of kmeans_fast_iterations_thread, included
//                                                       inside a harness function that declares the variables they use
//                                                       under the reference's names (its assignment half is find_k_neighbors) (N4)
// binheap.hpp, simd_layout.hpp, simd_scan.hpp, neighbors.hpp and config.h are included whole.
//
// What the harness itself adds: `mem_db`, an in-memory base_db (the reference's flat_db /
// index_db read cereal archives; their get_partition / free_partition / partition_count are the
// only members scanner_4 and scanner_simple touch, databases.hpp:118-128,237-243), and the C
// entry points.  Every number these entry points return is computed by the reference's text as
// g++ compiles it here with the reference's flags (CMakeLists.txt:7,19; -march=native replaced
// by the explicit ISA list of oracle/Makefile so the .so runs on the GPU box's host).
//
// Output: oracle/_ref/libqadc_ref_float.so (git-ignored, travels with gpurun).
#include <immintrin.h>
#include <x86intrin.h>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>
#define _mm256_set_m128i qadc_ref_mm256_set_m128i      // simd_scan.hpp:120 vs GCC >= 8's own intrinsic (see ref_harness.cpp)
#include "config.h"
#include "binheap.hpp"
#include "neighbors.hpp"
#include "simd_layout.hpp"
#include "simd_scan.hpp"
#include "vector_io.hpp"
#undef _mm256_set_m128i
#include "x_quantizers_a.inc"
#include "x_quantizers_b.inc"
#include "x_distances_a.inc"
#include "x_distances_b.inc"
#include "x_distances_c.inc"
#include "x_distances_d.inc"
#include "x_base_db.inc"
#include "x_query_metrics.inc"
#include "x_scan_funcs.inc"
#include "x_scanner_4.inc"
#include "x_scanner_simple.inc"
#include "x_substract.inc"
#include <fstream>
#include "x_opq_a.inc"
#include "x_opq_z.inc"
#include "x_pq_files_a.inc"
#include "x_pq_files_b.inc"
#include "x_neighbors_heaps.inc"
#include "x_norm_4.inc"
#include "x_cross_norms.inc"
}   // closes compute_cross_dists_blas<DSQ>: the range ends before its "BLAS Call" block (distances.hpp:178-183)
#include "x_cross_norms_4.inc"
}   // closes compute_cross_dists_blas<4> likewise (210-215)

namespace {

// In-memory base_db: partitions are owned copies, handed out row-major as flat_db / index_db do.
struct mem_db : base_db {
    std::vector<std::vector<std::uint8_t>> codes;
    std::vector<std::vector<unsigned>> labels;
    std::vector<unsigned> sizes;
    bool labelled;

    mem_db(int M, int bits, int nparts, const std::uint8_t* const* parts, const std::uint32_t* const* labs,
           const std::uint32_t* szs)
        : base_db(std::unique_ptr<base_pq>(new base_pq(M, bits, M))), labelled(labs != nullptr) {
        const int cs = pq->code_size();
        codes.resize(nparts); labels.resize(nparts); sizes.assign(szs, szs + nparts);
        for (int p = 0; p < nparts; ++p) {
            codes[p].assign(parts[p], parts[p] + static_cast<long>(szs[p]) * cs);
            if (labs && labs[p]) labels[p].assign(labs[p], labs[p] + szs[p]);
        }
    }
    void assign_compute_residuals(const float*, int, int*, float*) override {}
    void assign_compute_residuals_mutiple(const float*, const int, const int, int*, float*) override {}
    int partition_count() const override { return static_cast<int>(sizes.size()); }
    void get_partition(int part_i, const std::uint8_t*& c, unsigned*& l, unsigned& size) const override {
        c = codes[part_i].data();
        // per-partition presence, so that the mixed-labels exit of compute_sizes (db_query_4.cpp:118-124) is reachable
        l = (labelled && !labels[part_i].empty()) ? const_cast<unsigned*>(labels[part_i].data()) : nullptr;
        size = sizes[part_i];
    }
    void free_partition(int part_i) override {
        std::vector<std::uint8_t>().swap(codes[part_i]);
        std::vector<unsigned>().swap(labels[part_i]);
    }
    void add_vectors(float*, unsigned, unsigned, int) override {}
    void print(std::ostream&) const override {}
};

struct s4_handle {
    std::unique_ptr<mem_db> db;
    scanner_4 sc;
    int M;
    s4_handle(float keep) : sc(keep), M(0) {}
};

template <typename V>
void heap_out(kv_binheap<unsigned, V>& bh, std::uint32_t* ok, V* ov, int* osz) {
    *osz = bh.size();
    std::memcpy(ok, bh.keys(), sizeof(unsigned) * bh.size());
    std::memcpy(ov, bh.values(), sizeof(V) * bh.size());
}

// Runs fn() in a child process and returns its exit status (the reference reports errors by message + std::exit(1)).
template <typename F>
int run_forked(F fn) {
    std::cerr.flush();
    const pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        const int devnull = ::open("/dev/null", 1);
        if (devnull >= 0) { dup2(devnull, 2); }
        fn();
        _exit(0);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    return WIFEXITED(st) ? WEXITSTATUS(st) : 128;
}

}  // namespace

extern "C" {

// ---- A5: QuantizerMAX<int8_t>::quantize_tables (db_query_4.cpp:37-71) on `sq_count` tables of 16 floats.
void qadc_reff_quantize_tables(const float* tables, int sq_count, float qmin, float qmax, std::int8_t* out) {
    QuantizerMAX<std::int8_t> q127(qmin, qmax);
    std::unique_ptr<__m128i[]> qt(new __m128i[sq_count]);
    q127.quantize_tables(tables, qt.get(), sq_count);
    std::memcpy(out, qt.get(), static_cast<size_t>(sq_count) * 16);
}

// ---- A7: the float start scan exactly as query_scan_start composes it (db_query_4.cpp:230-242): push (0, FLT_MAX),
// then get_scan_func(pq)'s scan_4<M> over each given code run with its own table [M*16].  Heap arrays out; qmax = vals[0].
int qadc_reff_scan4_start(int M, int nparts, const std::uint8_t* const* parts, const std::uint32_t* const* labels,
                          const std::uint32_t* sizes, const float* tables, int R,
                          std::uint32_t* out_keys, float* out_vals, int* out_size) {
    if (M != 16 && M != 32) return -1;
    base_pq pq(M, 4, M);
    scan_func volatile f = get_scan_func(pq);          // called through the pointer, as scanner_4 does (db_query_4.cpp:213,238)
    kv_binheap<unsigned, float> bh(R);
    bh.push(0, std::numeric_limits<float>::max());
    for (int p = 0; p < nparts; ++p)
        f(parts[p], labels ? labels[p] : nullptr, sizes[p], tables + static_cast<long>(p) * M * 16, bh);
    heap_out(bh, out_keys, out_vals, out_size);
    return 0;
}

// ---- scanner_simple::query_scan (db_query.cpp:26-45) with scan_standard<uint8_t,NSQ> over an in-memory db: BASELINE configs[0].
int qadc_reff_scan_standard_u8(int NSQ, int nparts, const std::uint8_t* const* parts, const std::uint32_t* const* labels,
                               const std::uint32_t* sizes, const float* tables, int R,
                               std::uint32_t* out_keys, float* out_vals, int* out_size) {
    if (NSQ != 4 && NSQ != 8 && NSQ != 16) return -1;
    mem_db db(NSQ, 8, nparts, parts, labels, sizes);
    scanner_simple sc;
    sc.prepare_database(db);
    std::vector<int> assign(nparts);
    for (int p = 0; p < nparts; ++p) assign[p] = p;
    std::vector<float> tb(tables, tables + static_cast<long>(nparts) * NSQ * 256);
    kv_binheap<unsigned, float> bh(R);
    query_metrics m;
    sc.query_scan(nullptr, assign.data(), nparts, tb.data(), NSQ * 256, bh, m);
    heap_out(bh, out_keys, out_vals, out_size);
    return 0;
}

// ---- multiple_set_bits_4 (quantizers.hpp:49-68) driven as encode_multiple_vectors drives it (quantizers.hpp:232-244):
// one call per sub-quantizer with that sub-quantizer's assignment column.  assign is [n][M].
void qadc_reff_pack4(const std::int32_t* assign, long n, int M, std::uint8_t* codes) {
    std::vector<int> col(n);
    auto set_bits = prepare_multiple_set_bits(4, codes, static_cast<int>(n * (M / 2)));
    for (int m = 0; m < M; ++m) {
        for (long i = 0; i < n; ++i) col[i] = assign[i * M + m];
        set_bits(col.data(), static_cast<int>(n), M, 4, m, codes);
    }
}

// ---- direct table form: compute_dists_single_simd_cg<DSQ> through base_centroids_getter (distances.hpp:294-311, 250-275).
// centroids_flat is [M][16][DSQ]; vector is [M*DSQ]; dists [M*16].
int qadc_reff_tables_direct(int DSQ, int M, const float* centroids_flat, const float* vector, float* dists) {
    base_pq pq(M, 4, M * DSQ, const_cast<float*>(centroids_flat));
    base_centroids_getter cg(&pq);
    // called through a pointer to the stand-alone instance, as nns_engine does with get_dists_function's result
    // (query_common.hpp:268,296; the reference's dispatch table is distances.cpp:50-84, which needs cblas to compile)
    typedef decltype(&compute_dists_single_simd_cg<128>) dists_func;
    dists_func volatile f = nullptr;
    switch (DSQ) {
#define QADC_REFF_CASE(D) case D: f = compute_dists_single_simd_cg<D>; break;
        QADC_REFF_CASE(4) QADC_REFF_CASE(8) QADC_REFF_CASE(16) QADC_REFF_CASE(30) QADC_REFF_CASE(32) QADC_REFF_CASE(48)
        QADC_REFF_CASE(60) QADC_REFF_CASE(64) QADC_REFF_CASE(96) QADC_REFF_CASE(120) QADC_REFF_CASE(128)
        QADC_REFF_CASE(192) QADC_REFF_CASE(240) QADC_REFF_CASE(256)       // the sq_dim list of distances.cpp:50-84
#undef QADC_REFF_CASE
        default: return -1;
    }
    f(dists, cg, vector);
    return 0;
}

// ---- residuals of one query against its `ma` coarse centroids (databases.cpp:37-48).
void qadc_reff_substract_from_unique(const float* vector, int dim, const float* base_vectors, const int* assign, int ma,
                                     float* out) {
    substract_vectors_from_unique(vector, dim, base_vectors, const_cast<int*>(assign), ma, out);
}

// ---- scanner_4, whole (db_query_4.cpp:73-310) over an in-memory database of row-major partitions.
void* qadc_reff_scanner4_create(int M, float keep, int nparts, const std::uint8_t* const* parts,
                                const std::uint32_t* const* labels, const std::uint32_t* sizes) {
    if (M != 16 && M != 32) return nullptr;
    s4_handle* h = new s4_handle(keep);
    h->M = M;
    h->db.reset(new mem_db(M, 4, nparts, parts, labels, sizes));
    h->sc.prepare_database(*h->db);                     // A8
    return h;
}

void qadc_reff_scanner4_destroy(void* hv) { delete static_cast<s4_handle*>(hv); }

// exit status of prepare_database in a child process (1 = "Some partitions have labels and some have not").
int qadc_reff_scanner4_try_prepare(int M, float keep, int nparts, const std::uint8_t* const* parts,
                                   const std::uint32_t* const* labels, const std::uint32_t* sizes) {
    return run_forked([&] {
        void* h = qadc_reff_scanner4_create(M, keep, nparts, parts, labels, sizes);
        qadc_reff_scanner4_destroy(h);
    });
}

// what prepare_database derived: starts_sizes[], parts_sizes[], has_labels
void qadc_reff_scanner4_sizes(void* hv, std::uint32_t* starts_sizes, std::uint32_t* parts_sizes, int* has_labels) {
    s4_handle* h = static_cast<s4_handle*>(hv);
    for (int p = 0; p < h->sc.part_count; ++p) {
        starts_sizes[p] = h->sc.starts_sizes[p];
        parts_sizes[p] = h->sc.parts_sizes[p];
    }
    *has_labels = h->sc.has_labels ? 1 : 0;
}

// scanner_4::query_scan_start (230-242): float heap after the start scan; qmax = vals[0].
void qadc_reff_scanner4_query_start(void* hv, const int* assign, int ma, const float* tables, int R,
                                    std::uint32_t* out_keys, float* out_vals, int* out_size) {
    s4_handle* h = static_cast<s4_handle*>(hv);
    kv_binheap<unsigned, float> bh(R);
    h->sc.query_scan_start(const_cast<int*>(assign), ma, const_cast<float*>(tables), h->M * 16, bh);
    heap_out(bh, out_keys, out_vals, out_size);
}

// scanner_4::query_scan (245-309), whole: `tables` [ma][M*16] is MUTABLE (negative clamp, 262-269).
// Exits the process when qmax > 1e30 (271-274): probe with qadc_reff_scanner4_try_query first when that can happen.
void qadc_reff_scanner4_query_scan(void* hv, const int* assign, int ma, float* tables, int R,
                                   std::uint32_t* out_keys, std::int8_t* out_vals, int* out_size,
                                   std::uint32_t* out_sorted_keys) {
    s4_handle* h = static_cast<s4_handle*>(hv);
    kv_binheap<unsigned, std::int8_t> bh(R);
    query_metrics m;
    h->sc.query_scan(nullptr, const_cast<int*>(assign), ma, tables, h->M * 16, bh, m);
    heap_out(bh, out_keys, out_vals, out_size);
    if (out_sorted_keys) bh.sort_keys(out_sorted_keys);
}

// exit status of query_scan in a child process (0 = returned, 1 = the reference's "Max quantization bound too high" exit).
int qadc_reff_scanner4_try_query(void* hv, const int* assign, int ma, const float* tables, int R) {
    s4_handle* h = static_cast<s4_handle*>(hv);
    return run_forked([&] {
        std::vector<float> tb(tables, tables + static_cast<long>(ma) * h->M * 16);
        kv_binheap<unsigned, std::int8_t> bh(R);
        query_metrics m;
        h->sc.query_scan(nullptr, const_cast<int*>(assign), ma, tb.data(), h->M * 16, bh, m);
    });
}

// ---- N3: .pq.data / .opq.data through the reference's own readers (quantizers.cpp:27-46, 58-103) ----
// parse_data_filename in a child process: 0 = ".pq.data", 1 = ".opq.data", 101 = the reference's "Invalid data filename" exit(1)
int qadc_reff_parse_data_filename(const char* filename) {
    const int rc = run_forked([&] { _exit(parse_data_filename(filename) == type_opq ? 11 : 10); });
    return rc == 10 ? 0 : rc == 11 ? 1 : 101;
}

// The factory pq_from_data_file(name) (89-103): header fields, codebooks (cap floats; returns the count needed) and, for an
// .opq.data file, the rotation (rcap floats).  The name must be valid (probe with qadc_reff_parse_data_filename).
long qadc_reff_pq_from_data_file(const char* filename, int* dim, int* sq_count, int* sq_bits, int* is_opq, float* centroids,
                                 long cap, float* rotation, long rcap) {
    std::unique_ptr<base_pq> pq = pq_from_data_file(filename);
    *dim = pq->dim;
    *sq_count = pq->sq_count;
    *sq_bits = pq->sq_bits;
    opq* o = dynamic_cast<opq*>(pq.get());
    *is_opq = o != nullptr;
    const long need = pq->all_centroids_dim();
    if (centroids && cap >= need) std::memcpy(centroids, pq->centroids_flat.get(), sizeof(float) * need);
    if (o && rotation && rcap >= static_cast<long>(pq->dim) * pq->dim)
        std::memcpy(rotation, o->rotation.get(), sizeof(float) * pq->dim * pq->dim);
    return need;
}

// ---- N1: the selection half of find_k_neighbors (neighbors.cpp:30-76) — given the distances of `count` vectors to
// `neighbor_count` neighbours (row-major), the k nearest per vector in the order find_k_neighbors writes them: its own
// add_candidates_heaps (18-28) fed block by block as its loop does (BLOCK_VECS x BLOCK_NEIGHS, 47-65), then
// kv_binheap::sort (67-71).  The distance half (dists_func = cblas_sgemm expansion, 42, 58-59) is not here: no cblas.
void qadc_reff_select_k_neighbors(const float* dists, int count, int neighbor_count, int k, int* assign, float* sorted) {
    std::unique_ptr<kv_binheap<int, float>[]> heaps(new kv_binheap<int, float>[BLOCK_VECS]);
    for (int h = 0; h < BLOCK_VECS; ++h) heaps[h].reset_capacity(k);
    std::vector<float> block(static_cast<size_t>(BLOCK_VECS) * BLOCK_NEIGHS);
    for (int v0 = 0; v0 < count; v0 += BLOCK_VECS) {
        const int bv = std::min(BLOCK_VECS, count - v0);
        for (int v = 0; v < bv; ++v) heaps[v].reset();
        for (int n0 = 0; n0 < neighbor_count; n0 += BLOCK_NEIGHS) {
            const int bn = std::min(BLOCK_NEIGHS, neighbor_count - n0);
            for (int v = 0; v < bv; ++v)
                std::memcpy(block.data() + static_cast<size_t>(v) * bn, dists + static_cast<size_t>(v0 + v) * neighbor_count + n0,
                            sizeof(float) * bn);
            add_candidates_heaps(block.data(), heaps.get(), bv, bn, n0);
        }
        for (int v = 0; v < bv; ++v)
            heaps[v].sort(assign + static_cast<size_t>(v0 + v) * k, sorted + static_cast<size_t>(v0 + v) * k);
    }
}

// ---- N4: the encoder's distances up to the BLAS call.  find_k_neighbors (neighbors.cpp:30-76), which base_pq::encode_multiple_vectors
// calls per sub-quantizer with k = 1 (quantizers.hpp:222-245), gets its distances from get_cross_dists_func(dim) =
// compute_cross_dists_blas<dim> (distances.cpp:87-121): dists[v][c] = ||v||^2 + ||c||^2, then cblas_sgemm(alpha = -2, beta = 1)
// adds -2 v.c.  This returns the matrix as it stands BEFORE the sgemm, computed by the reference's own text with its flags
// (dists_dim = cent_count).  Returns 0, or -1 for a dimension outside the reference's dispatch.
int qadc_reff_cross_norms(int DSQ, const float* centroids, int cent_count, const float* vectors, int vec_count, float* dists) {
    switch (DSQ) {
#define QADC_CN(D) case D: compute_cross_dists_blas<D>(dists, centroids, cent_count, vectors, vec_count, cent_count); return 0;
        QADC_CN(4) QADC_CN(8) QADC_CN(16) QADC_CN(30) QADC_CN(32) QADC_CN(48) QADC_CN(60) QADC_CN(64) QADC_CN(96) QADC_CN(120)
        QADC_CN(128) QADC_CN(192) QADC_CN(240) QADC_CN(256)
#undef QADC_CN
        default: return -1;
    }
}

// extract_subvectors (quantizers.hpp:70-77 region): sub-vector sq_i of every vector, contiguous.
void qadc_reff_extract_subvectors(const float* vectors, int dim, int count, int subq_dim, int sq_i, float* out) {
    extract_subvectors(vectors, dim, count, subq_dim, sq_i, out);
}

// ---- N4: kmeans_fast_iterations_thread's centroid update (databases.cpp:70-88) as g++ compiles it with the reference's flags
// (-ffast-math turns the division by the member count into a multiplication by its reciprocal).  The loops are included as
// they stand; the function around them declares what they use under the reference's own names (databases.cpp:50-54).
void qadc_reff_kmeans_update(const float* vectors, long n, int dimension, int centroid_count, const int* assign, float* centroids) {
    vectors_owner<float> learn_vectors;
    learn_vectors.data.reset(new float[static_cast<size_t>(n) * dimension]);
    std::memcpy(learn_vectors.data.get(), vectors, sizeof(float) * static_cast<size_t>(n) * dimension);
    learn_vectors.dimension = dimension;
    learn_vectors.count = n;
    const int* assignements = assign;
    const int dim = learn_vectors.dimension;
    std::unique_ptr<int[]> assign_count(new int[centroid_count]);
#include "x_kmeans_update.inc"
}

}  // extern "C"
