// Host path of BASELINE configs[0] — the reference's plain ADC scanner (db_query.cpp:17-46) and its scan kernels
// (query_common.hpp:59-146), C++14, header only, NO GPU: this is the CPU plumbing every other path is measured against
// ("Flat DB, PQ 8x8 scalar ADC on CPU, db_query reference path"), kept next to the accelerated scanner so that the
// stand-alone driver (tests/cpp/db_query_simple.cpp) offers both of the reference's query front ends.
//
//   pq_bytes            base_pq with whole-byte codes (sq_bits 8 or 16; quantizers.hpp:96-246): encode, the two table forms
//   scan_standard<T,N>  query_common.hpp:92-118: candidate = sum of NSQ table entries in sub-quantizer order
//   scan_4f<N>          query_common.hpp:59-90: the same on row-major 4-bit codes (low nibble = even sub-quantizer)
//   get_scan_func       query_common.hpp:120-146: the (sq_count, sq_bits) dispatch and its error text
//   scanner_simple      db_query.cpp:17-46: R sentinel pushes FLT_MAX - t, then every probed partition in assign[] order
// Float sums take the grouping of the reference AS COMPILED with its -ffast-math (host/float_sum.hpp; float_sum_mode() = 0
// gives the source order); this file itself is built without -ffast-math.  The oracle's orc_scan_standard_u8 /
// orc_candidates_f32 / orc_tables_direct, pinned to the reference build, are the checkers (tests/test_scanner_hip_cpp.py).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <vector>

#include "float_sum.hpp"
#include "qadc_heap.hpp"

namespace qadc {

typedef kv_heap<unsigned, float> float_heap;

template <typename T, int NSQ>
void scan_standard(const std::uint8_t* pqcodes_, const unsigned* labels, const unsigned pqcodes_count, const float* dists,
                   float_heap& bh) {
    const int NCENT = 1 << (sizeof(T) * 8);
    const T* const pqcodes = reinterpret_cast<const T*>(pqcodes_);
    float min = bh.max();
    for (unsigned i = 0; i < pqcodes_count; ++i) {
        const T* const code = pqcodes + (std::size_t)i * NSQ;
        float t[NSQ];
        for (int sq = 0; sq < NSQ; ++sq) t[sq] = dists[sq * NCENT + code[sq]];
        const float candidate = adc_sum<NSQ, sizeof(T) == 1>(t);   // (the as-compiled grouping was read off the uint8_t instances only)
        if (candidate < min) {
            bh.push(labels != nullptr ? labels[i] : i, candidate);
            min = bh.max();
        }
    }
}

template <int NSQ>
void scan_4f(const std::uint8_t* pqcodes, const unsigned* labels, const unsigned pqcodes_count, const float* dists,
             float_heap& bh) {
    float min = bh.max();
    for (unsigned i = 0; i < pqcodes_count; ++i) {
        const std::uint8_t* const code = pqcodes + (std::size_t)i * (NSQ / 2);
        float t[NSQ];
        for (int b = 0; b < NSQ / 2; ++b) {                      // byte b: low nibble = sub-quantizer 2b, high nibble = 2b + 1
            t[2 * b] = dists[(2 * b) * 16 + (code[b] & 0xf)];
            t[2 * b + 1] = dists[(2 * b + 1) * 16 + (code[b] >> 4)];
        }
        const float candidate = adc_sum<NSQ>(t);
        if (candidate < min) {
            bh.push(labels != nullptr ? labels[i] : i, candidate);
            min = bh.max();
        }
    }
}

typedef void (*scan_func)(const std::uint8_t*, const unsigned*, unsigned, const float*, float_heap&);

template <typename Pq>
scan_func get_scan_func(const Pq& pq) {
    if (pq.sq_count == 16 && pq.sq_bits == 4) return scan_4f<16>;
    if (pq.sq_count == 32 && pq.sq_bits == 4) return scan_4f<32>;
    if (pq.sq_count == 4 && pq.sq_bits == 8) return scan_standard<std::uint8_t, 4>;
    if (pq.sq_count == 8 && pq.sq_bits == 8) return scan_standard<std::uint8_t, 8>;
    if (pq.sq_count == 16 && pq.sq_bits == 8) return scan_standard<std::uint8_t, 16>;
    if (pq.sq_count == 2 && pq.sq_bits == 16) return scan_standard<std::uint16_t, 2>;
    if (pq.sq_count == 4 && pq.sq_bits == 16) return scan_standard<std::uint16_t, 4>;
    if (pq.sq_count == 8 && pq.sq_bits == 16) return scan_standard<std::uint16_t, 8>;
    std::cerr << "Unsupported (nsq,nsq_bits) configuration." << std::endl;
    std::cerr << "Supported configurations are: (16,4) (4,8) (8,8) (16,8) (2,16) (4,16) (8,16)." << std::endl;
    std::exit(1);
}

// base_pq with whole-byte codes: sq_bits 8 (one byte per sub-quantizer) or 16 (two, little endian).
struct pq_bytes {
    int sq_count, sq_bits, dim;
    std::vector<float> centroids;                                // [sq_count][ncent][sq_dim]
    pq_bytes(int m, int bits, int d) : sq_count(m), sq_bits(bits), dim(d), centroids((std::size_t)m * (1u << bits) * (d / m)) {}
    int ncent() const { return 1 << sq_bits; }
    int sq_dim() const { return dim / sq_count; }
    int code_size() const { return sq_count * sq_bits / 8; }
    int table_dim() const { return sq_count * ncent(); }
    const float* centroid(int m, int c) const { return centroids.data() + ((std::size_t)m * ncent() + c) * sq_dim(); }
    void rotate_multiple_vectors(float*, int) const {}           // plain PQ (quantizers.hpp:189-195)
    void tables(const float* x, float* out) const {              // ||x_m - c||^2 as fmanorm adds it (float_sum.hpp)
        const int ds = sq_dim(), nc = ncent();
        for (int m = 0; m < sq_count; ++m)
            for (int c = 0; c < nc; ++c) out[m * nc + c] = sqdist(x + m * ds, centroid(m, c), ds);
    }
    void tables_direct(const float* x, float* out) const { tables(x, out); }   // the engine's name for the ma == 1 form
    void tables_blas(const float* vecs, int count, float* out) const {   // (||v||^2 + ||c||^2) - 2 v.c, distances.hpp:151-183
        const int ds = sq_dim(), nc = ncent();                           // (norms as compiled, one sequential dot: float_sum.hpp)
        std::vector<float> cn((std::size_t)sq_count * nc);
        for (int e = 0; e < sq_count * nc; ++e) cn[e] = sqnorm(centroids.data() + (std::size_t)e * ds, ds);
        for (int v = 0; v < count; ++v)
            for (int m = 0; m < sq_count; ++m) {
                const float* x = vecs + (std::size_t)v * dim + m * ds;
                const float vn = sqnorm(x, ds);
                for (int c = 0; c < nc; ++c)
                    out[((std::size_t)v * sq_count + m) * nc + c] = expansion_dist(x, centroid(m, c), ds, vn, cn[m * nc + c]);
            }
    }
    // encode_multiple_vectors (quantizers.hpp:222-245): find_k_neighbors with k = 1 on the expansion distances = their first
    // strict minimum in centroid order
    void encode(const float* vecs, std::size_t n, std::uint8_t* codes) const {
        const int nc = ncent(), cs = code_size();
        std::vector<float> t((std::size_t)table_dim());
        for (std::size_t i = 0; i < n; ++i) {
            tables_blas(vecs + i * dim, 1, t.data());
            for (int m = 0; m < sq_count; ++m) {
                int best = 0;
                for (int c = 1; c < nc; ++c)
                    if (t[m * nc + c] < t[m * nc + best]) best = c;
                if (sq_bits == 8) codes[i * cs + m] = (std::uint8_t)best;
                else { codes[i * cs + 2 * m] = (std::uint8_t)best; codes[i * cs + 2 * m + 1] = (std::uint8_t)(best >> 8); }
            }
        }
    }
};

// db_query.cpp:17-46.  Db offers base_db's get_partition (databases.hpp:50-55) and a `pq` with sq_count / sq_bits.
template <typename Db>
struct scanner_simple {
    typedef float_heap BhType;
    Db* db = nullptr;
    scan_func scan = nullptr;
    void prepare_database(Db& database) {
        db = &database;
        scan = get_scan_func(*database.pq);
    }
    template <typename Metrics>
    void query_scan(const float*, int* assign, int ma, float* tables, int table_dim, BhType& bh, Metrics&) {
        for (int t = 0; t < bh.capacity(); ++t) bh.push(0, std::numeric_limits<float>::max() - t);   // "Fill binary heap"
        const std::uint8_t* codes;
        unsigned* labels;
        unsigned count;
        for (int a = 0; a < ma; ++a) {
            db->get_partition(assign[a], codes, labels, count);
            scan(codes, labels, count, tables, bh);
            tables += table_dim;
        }
    }
};

}  // namespace qadc
