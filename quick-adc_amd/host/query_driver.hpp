// Host-side query driver around a ScannerType — C++14, header only, no GPU code.
//
// The pieces of the reference that stay on the host next to the accelerated scan (SURVEY.md §8 rows
// A9/A10), restated for this repo's stand-alone driver and examples:
//   pq4            base_pq for 4-bit sub-quantizers: encode + per-query float distance tables
//                  (quantizers.hpp:96-246; distances.hpp:294-311 "single" form ||x_m - c||^2 and the
//                  BLAS-expansion form of distances.hpp:151-183, 277-292)
//   flat_database  flat_db  (databases.hpp:77-167): one partition, key = position
//   ivf_database   index_db (databases.hpp:176-331): coarse centroids, per-partition codes + labels,
//                  assign = the ma nearest centroids in ascending distance, residual = x - centroid
//   nns_engine     per-query sequence assign -> (rotate: none for plain PQ) -> tables -> scanner.query_scan
//                  with the four phase timers (query_common.hpp:245-309)
//   process_queries  fresh heap per query, recall = true nearest neighbour among the R returned keys,
//                  averaged metrics (query_common.hpp:330-368) and the CSV line of db_query_4.cpp:387-390
// The direct table form (ma == 1: pq4::tables_direct) adds like the reference's AVX/FMA kernel as compiled
// (host/float_sum.hpp, pinned to the reference build by the oracle's tests).  The BLAS-expansion form's norm half is pinned
// the same way; its product, the OPQ rotation and the coarse distances go through OpenBLAS in the reference and are NOT
// pinned (sequential sums here).
#pragma once
#include <sys/time.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <vector>

#include "float_sum.hpp"
#include "qadc_heap.hpp"

namespace qadc {

static inline std::uint64_t ustime() {  // common.hpp:17-21
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return (std::uint64_t)tv.tv_sec * 1000000 + tv.tv_usec;
}

struct pq4 {
    int sq_count;                  // M
    int sq_bits;                   // 4
    int dim;
    std::vector<float> centroids;  // [M][16][sq_dim]
    std::vector<float> rotation;   // OPQ: [dim][dim], empty for plain PQ (quantizers.hpp:248-324)

    pq4(int m, int d) : sq_count(m), sq_bits(4), dim(d), centroids((size_t)m * 16 * (d / m)) {}
    // opq::rotate_multiple_vectors (quantizers.hpp:289-301): sgemm(NoTrans, Trans) = every vector times the
    // TRANSPOSED rotation, rotated[r] = sum_c x[c] * rotation[r][c]; plain PQ: no-op (189-195).  Sequential float
    // sums in ascending c (what the device feeder does too; a BLAS may associate differently).
    void rotate_multiple_vectors(float* vecs, int count) const {
        if (rotation.empty()) return;
        std::vector<float> out((size_t)dim);
        for (int v = 0; v < count; ++v) {
            float* x = vecs + (size_t)v * dim;
            for (int r = 0; r < dim; ++r) {
                float acc = 0;
                for (int c = 0; c < dim; ++c) acc += x[c] * rotation[(size_t)r * dim + c];
                out[r] = acc;
            }
            std::copy(out.begin(), out.end(), x);
        }
    }
    int sq_dim() const { return dim / sq_count; }
    int code_size() const { return sq_count / 2; }
    int table_dim() const { return sq_count * 16; }
    const float* centroid(int m, int c) const { return centroids.data() + ((size_t)m * 16 + c) * sq_dim(); }

    // tables[m][c] = ||x_m - centroid(m,c)||^2, one sequential sum in ascending d: the direct-form encoder's distances
    // (encode_form 0; this repository's encoder before round 6)
    void tables(const float* x, float* out) const {
        const int ds = sq_dim();
        for (int m = 0; m < sq_count; ++m)
            for (int c = 0; c < 16; ++c) {
                const float* ce = centroid(m, c);
                float s = 0;
                for (int d = 0; d < ds; ++d) {
                    const float t = x[m * ds + d] - ce[d];
                    s += t * t;
                }
                out[m * 16 + c] = s;
            }
    }

    // The direct ("single") form the query path uses for ma == 1: compute_dists_single_simd_cg -> fmanorm
    // (distances.hpp:294-311, 60-76), in the reference's as-compiled grouping (float_sum.hpp).
    void tables_direct(const float* x, float* out) const {
        const int ds = sq_dim();
        for (int m = 0; m < sq_count; ++m)
            for (int c = 0; c < 16; ++c) out[m * 16 + c] = sqdist(x + m * ds, centroid(m, c), ds);
    }

    // The BLAS-expansion form of `count` vectors' tables (compute_dists_multiple_blas_cg -> compute_cross_dists_blas,
    // distances.hpp:151-183, 277-292): per sub-quantizer ||v||^2 + ||c||^2 first, then sgemm(alpha = -2, beta = 1) adds
    // -2 v.c.  What nns_engine evaluates for ma > 1 and nns_engine_batch always (query_common.hpp:194-213, 292-297).
    // Cancellation makes entries slightly NEGATIVE when v ~ c: the case scanner_4::query_scan clamps in place
    // (db_query_4.cpp:258-269).  The norms add as the reference is compiled (float_sum.hpp sqnorm, pinned); the product is
    // one sequential dot (OpenBLAS's sgemm in the reference: restated).  The device twin, build_tables_kernel, adds in
    // the same order: bit-compatible with the device and the oracle's orc_tables_expansion.
    void tables_blas(const float* vecs, int count, float* out) const {
        const int ds = sq_dim();
        std::vector<float> cn((size_t)sq_count * 16);                        // (||c||^2: per call — the codebooks are a public member)
        for (int e = 0; e < sq_count * 16; ++e) cn[e] = sqnorm(centroids.data() + (size_t)e * ds, ds);
        tables_blas(vecs, count, cn.data(), out);
    }
    void tables_blas(const float* vecs, int count, const float* cn, float* out) const {
        const int ds = sq_dim();
        for (int v = 0; v < count; ++v) {
            const float* x = vecs + (size_t)v * dim;
            float* o = out + (size_t)v * sq_count * 16;
            for (int m = 0; m < sq_count; ++m) {
                const float vn = sqnorm(x + m * ds, ds);
                for (int c = 0; c < 16; ++c) o[m * 16 + c] = expansion_dist(x + m * ds, centroid(m, c), ds, vn, cn[m * 16 + c]);
            }
        }
    }

    // encode_multiple_vectors (quantizers.hpp:222-245): rotate (opq), then per sub-quantizer find_k_neighbors with k = 1
    // (neighbors.cpp:30-76) on the BLAS-expansion distances — a capacity-1 kv_binheap fed in centroid order keeps the
    // first strict minimum (centroid 0 when its distance is NaN) — packed two per byte: even sub-quantizer in the low
    // nibble, odd one in the high nibble of byte m/2 (multiple_set_bits_4, quantizers.hpp:49-68).
    // encode_form 1 = that; 0 = the direct form of tables().  Device twin: pq_encode_kernel; oracle: orc_pq_encode.
    int encode_form = 1;
    void encode(const float* vecs, size_t n, std::uint8_t* codes) const {
        const int cs = code_size(), ds = sq_dim();
        std::vector<float> t((size_t)sq_count * 16), rot, cn((size_t)sq_count * 16);
        for (int e = 0; e < sq_count * 16; ++e) cn[e] = sqnorm(centroids.data() + (size_t)e * ds, ds);
        for (size_t i = 0; i < n; ++i) {
            const float* x = vecs + i * dim;
            if (!rotation.empty()) {                       // encode_multiple_vectors rotates first (quantizers.hpp:224)
                rot.assign(x, x + dim);
                rotate_multiple_vectors(rot.data(), 1);
                x = rot.data();
            }
            if (encode_form) tables_blas(x, 1, cn.data(), t.data());
            else tables(x, t.data());
            std::uint8_t* code = codes + i * cs;
            for (int m = 0; m < sq_count; ++m) {
                int best = 0;
                for (int c = 1; c < 16; ++c)
                    if (t[m * 16 + c] < t[m * 16 + best]) best = c;
                if (m % 2 == 0) code[m / 2] = (std::uint8_t)best;
                else code[m / 2] = (std::uint8_t)(code[m / 2] | (best << 4));
            }
        }
    }
};

template <typename Pq>
struct flat_database_t {                 // flat_db over any quantizer: pq4 here, pq_bytes (host/scanner_simple.hpp) for PQ 8x8
    std::unique_ptr<Pq> pq;
    std::vector<std::uint8_t> codes;
    unsigned count = 0;

    void add_vectors(const float* vecs, unsigned n) {
        codes.resize((size_t)(count + n) * pq->code_size());
        pq->encode(vecs, n, codes.data() + (size_t)count * pq->code_size());
        count += n;
    }
    int partition_count() const { return 1; }
    void get_partition(int, const std::uint8_t*& c, unsigned*& labels, unsigned& size) {
        c = codes.data();
        labels = nullptr;
        size = count;
    }
    void free_partition(int) { std::vector<std::uint8_t>().swap(codes); }
    void assign_compute_residuals(const float* x, int ma, int* assign, float* residuals) const {
        for (int a = 0; a < ma; ++a) {  // databases.hpp:93-101
            assign[a] = 0;
            std::memcpy(residuals + (size_t)a * pq->dim, x, sizeof(float) * pq->dim);
        }
    }
};
typedef flat_database_t<pq4> flat_database;

struct ivf_database {
    std::unique_ptr<pq4> pq;
    int part_count;
    std::vector<float> coarse;  // [K][dim]
    std::vector<std::vector<std::uint8_t>> partitions;
    std::vector<std::vector<unsigned>> labels;

    ivf_database(std::unique_ptr<pq4> p, int k, std::vector<float> c)
        : pq(std::move(p)), part_count(k), coarse(std::move(c)), partitions(k), labels(k) {}

    // the coarse distance as find_k_neighbors gets it from compute_cross_dists_blas (distances.hpp:151-183): (||x||^2 + ||c||^2) with the
    // norms as compiled (float_sum.hpp sqnorm), then -2 x.c — the sgemm, restated as one sequential dot (device twin: coarse_dist_kernel)
    float dist2(const float* x, float xn, int k) const {
        const float* c = coarse.data() + (size_t)k * pq->dim;
        return expansion_dist(x, c, pq->dim, xn, sqnorm(c, pq->dim));
    }
    // the ma nearest coarse centroids, ascending by distance, as find_k_neighbors selects them (neighbors.cpp:18-28, 47-71): the
    // distances go through a kv_binheap<int, float> of capacity ma in index order, then kv_binheap::sort — which is "the ma smallest
    // by (distance, index)" unless distances tie exactly, and then whatever the heap's history and std::sort leave (pinned to the
    // reference's own heaps: tests/test_oracle_float_ref.py; the device kernels do the same, coarse_exact_select)
    void nearest(const float* x, int ma, int* out) const {
        kv_heap<int, float> h(ma);
        const float xn = sqnorm(x, pq->dim);
        for (int k = 0; k < part_count; ++k) h.push(k, dist2(x, xn, k));
        h.sort_keys(out);
    }
    void add_vectors(const float* vecs, unsigned n, unsigned labels_offset) {  // databases.hpp:270-298
        std::vector<float> res(pq->dim);
        std::vector<std::uint8_t> code(pq->code_size());
        for (unsigned i = 0; i < n; ++i) {
            const float* x = vecs + (size_t)i * pq->dim;
            int p;
            nearest(x, 1, &p);
            for (int d = 0; d < pq->dim; ++d) res[d] = x[d] - coarse[(size_t)p * pq->dim + d];
            pq->encode(res.data(), 1, code.data());
            partitions[p].insert(partitions[p].end(), code.begin(), code.end());
            labels[p].push_back(i + labels_offset);
        }
    }
    int partition_count() const { return part_count; }
    void get_partition(int i, const std::uint8_t*& c, unsigned*& l, unsigned& size) {
        c = partitions[i].data();
        l = labels[i].data();
        size = (unsigned)labels[i].size();
    }
    void free_partition(int i) {
        std::vector<std::uint8_t>().swap(partitions[i]);
        std::vector<unsigned>().swap(labels[i]);
    }
    void assign_compute_residuals(const float* x, int ma, int* assign, float* residuals) const {  // databases.hpp:201-211
        nearest(x, ma, assign);
        for (int a = 0; a < ma; ++a)
            for (int d = 0; d < pq->dim; ++d)
                residuals[(size_t)a * pq->dim + d] = x[d] - coarse[(size_t)assign[a] * pq->dim + d];
    }
};

struct query_metrics {  // query_common.hpp:21-56
    std::uint64_t index_us = 0, rotate_us = 0, table_us = 0, scan_us = 0;
    query_metrics& operator+=(const query_metrics& o) {
        index_us += o.index_us; rotate_us += o.rotate_us; table_us += o.table_us; scan_us += o.scan_us;
        return *this;
    }
    query_metrics& operator/=(int f) {
        index_us /= f; rotate_us /= f; table_us /= f; scan_us /= f;
        return *this;
    }
};
inline std::ostream& operator<<(std::ostream& os, const query_metrics& m) {
    return os << m.index_us << "," << m.rotate_us << "," << m.table_us << "," << m.scan_us;
}

template <typename Db, typename Scanner>
struct nns_engine {  // query_common.hpp:245-309
    Db& db;
    Scanner& scanner;
    int ma, table_dim;
    std::vector<float> residuals, dists;
    std::vector<int> assign;
    nns_engine(Scanner& s, Db& d, int ma_)
        : db(d), scanner(s), ma(ma_), table_dim(d.pq->table_dim()), residuals((size_t)ma_ * d.pq->dim),
          dists((size_t)ma_ * d.pq->table_dim()), assign(ma_) {}
    void prepare_database() { scanner.prepare_database(db); }
    template <typename Heap>
    void process_query(const float* query, Heap& bh, query_metrics& metrics) {
        const std::uint64_t t0 = ustime();
        db.assign_compute_residuals(query, ma, assign.data(), residuals.data());
        const std::uint64_t t1 = ustime();
        db.pq->rotate_multiple_vectors(residuals.data(), ma);   // query_common.hpp:206-207 (no-op for plain PQ)
        const std::uint64_t t2 = ustime();
        if (ma == 1) db.pq->tables_direct(residuals.data(), dists.data());   // "Optimized" single form (query_common.hpp:292-294)
        else db.pq->tables_blas(residuals.data(), ma, dists.data());       // dist_mult_func_ (295-297)
        const std::uint64_t t3 = ustime();
        scanner.query_scan(residuals.data(), assign.data(), ma, dists.data(), table_dim, bh, metrics);
        metrics.scan_us = ustime() - t3;
        metrics.table_us = t3 - t2;
        metrics.rotate_us = t2 - t1;
        metrics.index_us = t1 - t0;
    }
};

// nns_engine_batch (query_common.hpp:149-243): tables of `batch` queries are prepared together when the first
// query of a batch is asked for; here the scanner also scans the whole batch in one GPU call at that point
// (scanner.batch_scan) and the per-query calls only replay the cached candidate stream, so the reference's
// per-query process_queries<> loop drives a batched GPU scan unchanged.
template <typename Db, typename Scanner>
struct nns_engine_batch {
    Db& db;
    Scanner& scanner;
    int ma, batch, table_dim, r;
    std::vector<float> residuals, dists;
    std::vector<int> assign;
    nns_engine_batch(Scanner& s, Db& d, int ma_, int batch_, int r_)
        : db(d), scanner(s), ma(ma_), batch(batch_), table_dim(d.pq->table_dim()), r(r_),
          residuals((size_t)batch_ * ma_ * d.pq->dim), dists((size_t)batch_ * ma_ * d.pq->table_dim()),
          assign((size_t)batch_ * ma_) {}
    void prepare_database() { scanner.prepare_database(db); }
    template <typename Heap>
    void process_query(int query_i, const float* queries, int count, Heap& bh, query_metrics& metrics) {
        const int dim = db.pq->dim;
        const int b = query_i % batch;
        metrics = query_metrics();
        if (b == 0) {
            const int nb = std::min(batch, count - query_i);
            const std::uint64_t t0 = ustime();
            for (int i = 0; i < nb; ++i)
                db.assign_compute_residuals(queries + (size_t)(query_i + i) * dim, ma, assign.data() + (size_t)i * ma,
                                            residuals.data() + (size_t)i * ma * dim);
            const std::uint64_t t1 = ustime();
            db.pq->rotate_multiple_vectors(residuals.data(), nb * ma);
            const std::uint64_t tr = ustime();
            metrics.rotate_us = tr - t1;
            db.pq->tables_blas(residuals.data(), nb * ma, dists.data());       // dist_func (query_common.hpp:209-213)
            const std::uint64_t t2 = ustime();
            scanner.batch_scan(nb, assign.data(), ma, dists.data(), table_dim, r);
            metrics.scan_us = ustime() - t2;
            metrics.table_us = t2 - tr;
            metrics.index_us = t1 - t0;
        }
        const std::uint64_t t3 = ustime();
        scanner.batch_replay(b, bh);
        metrics.scan_us += ustime() - t3;
    }
};

// query_common.hpp:330-368.  groundtruth[q] = id of the true nearest neighbour; recall@R counts the queries
// whose true neighbour is among the keys — read unsorted.  The reference reads keys()[0..R) even when the heap is not
// full (uninitialised tail, warning at query_common.hpp:357-361); this driver deliberately reads only keys()[0..size()).
template <typename Db, typename Scanner, typename Heap>
inline void call_engine(nns_engine<Db, Scanner>& e, int q, const float* queries, int, int dim, Heap& bh, query_metrics& m) {
    e.process_query(queries + (size_t)q * dim, bh, m);
}
template <typename Db, typename Scanner, typename Heap>
inline void call_engine(nns_engine_batch<Db, Scanner>& e, int q, const float* queries, int count, int, Heap& bh,
                        query_metrics& m) {
    e.process_query(q, queries, count, bh, m);
}

// recall_file::check_labels / all_in (recall.hpp:21-32, 46-54): 1 iff every one of the query's first t ground-truth ids is
// among [first, last).  The ids are int (the .ivecs ground truth, recall.hpp:34-40), the keys unsigned: std::find compares
// them as the reference's does (the id converted to unsigned).  Pinned to the reference's own recall.hpp compiled into
// oracle/_ref (tests/test_io_formats.py).
template <typename It>
inline int check_labels(const int* groundtruth_query, int t, It first, It last) {
    for (int i = 0; i < t; ++i)
        if (std::find(first, last, groundtruth_query[i]) == last) return 0;
    return 1;
}

template <typename Engine, typename Heap>
void process_queries(Engine& engine, const float* queries, int count, int dim, int r, const unsigned* groundtruth,
                     query_metrics& total_metrics, double& total_recall) {
    engine.prepare_database();
    total_metrics = query_metrics();
    total_recall = 0;
    for (int q = 0; q < count; ++q) {
        Heap bh(r);
        query_metrics metrics;
        call_engine(engine, q, queries, count, dim, bh, metrics);
        if (bh.size() != r) std::cerr << " WARNING: Binheap not full" << std::endl;
        const unsigned* k = bh.keys();
        const int truth = (int)groundtruth[q];                   // t = 1 (query_common.hpp:347): the true nearest neighbour
        total_recall += check_labels(&truth, 1, k, k + bh.size());
        total_metrics += metrics;
    }
    total_metrics /= count;
    total_recall /= count;
}

inline void print_csv(std::ostream& os, int r, double recall, int ma, float keep, const query_metrics& m) {
    os << "r,recall,ma,adc_type,keep,index_us,rotate_us,table_us,scan_us" << std::endl;  // db_query_4.cpp:387-390
    os << r << "," << recall << "," << ma << ",qadc," << keep << "," << m << std::endl;
}
inline void print_csv_adc(std::ostream& os, int r, double recall, int ma, const query_metrics& m) {
    os << "r,recall,ma,adc_type,index_us,rotate_us,table_us,scan_us" << std::endl;       // db_query.cpp:112-115
    os << r << "," << recall << "," << ma << ",adc," << m << std::endl;
}

}  // namespace qadc
