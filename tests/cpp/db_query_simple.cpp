// BASELINE configs[0] end to end on the CPU, C++14, no GPU: the reference's `db_query` front end (db_query.cpp:48-140 —
// same options -r -m -b, same CSV line) over a flat database with whole-byte or 4-bit PQ codes, the plain ADC scanner
// (host/scanner_simple.hpp = db_query.cpp:17-46 + query_common.hpp:59-146) under the engines of host/query_driver.hpp.
// Instead of the reference's database / vecs files the data is synthetic and seeded (tests/test_scanner_hip_cpp.py
// rebuilds nothing: the driver dumps codes, per-query tables and heaps, and the test replays them through the oracle).
//   usage: db_query_simple [-r R] [-b BATCH] SQ_COUNT SQ_BITS DIM N NQ SEED DUMP_FILE
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <vector>

#include "../../quick-adc_amd/host/query_driver.hpp"
#include "../../quick-adc_amd/host/scanner_simple.hpp"

using namespace qadc;

static std::uint64_t splitmix64(std::uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static float unit(std::uint64_t seed, std::uint64_t i) { return (float)(splitmix64(seed ^ splitmix64(i)) >> 40) * (1.0f / 16777216.0f); }

struct no_batch_metrics {};

// a scanner_simple that also records what it was handed (tables) and what it produced (heap arrays)
template <typename Db>
struct recording_scanner : scanner_simple<Db> {
    std::vector<float> tables_seen;
    void prepare_database(Db& d) { scanner_simple<Db>::prepare_database(d); }
    template <typename Metrics>
    void query_scan(const float* q, int* assign, int ma, float* tables, int table_dim, float_heap& bh, Metrics& m) {
        tables_seen.insert(tables_seen.end(), tables, tables + (std::size_t)ma * table_dim);
        scanner_simple<Db>::query_scan(q, assign, ma, tables, table_dim, bh, m);
    }
    // nns_engine_batch drives a scanner through batch_scan / batch_replay (host/query_driver.hpp); for the CPU scanner the
    // batch call only remembers the inputs and every query is scanned at its replay, as db_query's batch engine does
    // (query_common.hpp:215-243: tables of the batch first, then one query_scan per query)
    int b_ma = 0, b_dim = 0;
    const int* b_assign = nullptr;
    float* b_tables = nullptr;
    void batch_scan(int, const int* assign, int ma, float* tables, int table_dim, int) {
        b_assign = assign; b_ma = ma; b_tables = tables; b_dim = table_dim;
    }
    void batch_replay(int b, float_heap& bh) {
        query_metrics m;
        std::vector<int> a(b_assign + (std::size_t)b * b_ma, b_assign + (std::size_t)(b + 1) * b_ma);
        query_scan(nullptr, a.data(), b_ma, b_tables + (std::size_t)b * b_ma * b_dim, b_dim, bh, m);
    }
};

int main(int argc, char** argv) {
    int r = 100, batch = 1, opt;
    while ((opt = getopt(argc, argv, "r:m:b:")) != -1) {
        if (opt == 'r') r = std::atoi(optarg);
        else if (opt == 'b') batch = std::atoi(optarg);
        else if (opt == 'm') { if (std::atoi(optarg) != 1) { std::cerr << "a flat database has ma = 1" << std::endl; return 1; } }
        else { std::cerr << "Usage: db_query_simple [-r R] [-b BATCH_SIZE] SQ_COUNT SQ_BITS DIM N NQ SEED DUMP" << std::endl; return 1; }
    }
    if (argc - optind < 7) { std::cerr << "Usage: db_query_simple [-r R] [-b BATCH_SIZE] SQ_COUNT SQ_BITS DIM N NQ SEED DUMP" << std::endl; return 1; }
    const int M = std::atoi(argv[optind]), bits = std::atoi(argv[optind + 1]), dim = std::atoi(argv[optind + 2]);
    const unsigned n = (unsigned)std::atol(argv[optind + 3]);
    const int nq = std::atoi(argv[optind + 4]);
    const std::uint64_t seed = std::strtoull(argv[optind + 5], nullptr, 10);
    const char* dump = argv[optind + 6];

    std::vector<float> base((std::size_t)n * dim), queries((std::size_t)nq * dim);
    for (std::size_t i = 0; i < base.size(); ++i) base[i] = unit(seed, i) * 4.0f - 2.0f;
    for (std::size_t i = 0; i < queries.size(); ++i) queries[i] = unit(seed + 1, i) * 4.0f - 2.0f;
    // exact nearest neighbour of every query (the groundtruth file's first column)
    std::vector<unsigned> gt(nq);
    for (int q = 0; q < nq; ++q) {
        float best = std::numeric_limits<float>::max();
        for (unsigned i = 0; i < n; ++i) {
            float s = 0;
            for (int d = 0; d < dim; ++d) { const float t = queries[(std::size_t)q * dim + d] - base[(std::size_t)i * dim + d]; s += t * t; }
            if (s < best) { best = s; gt[q] = i; }
        }
    }
    query_metrics metrics;
    double recall = 0;
    std::vector<std::uint8_t> codes;
    std::vector<float> tables;
    std::vector<float_heap> heaps;
    auto run = [&](auto& db) {
        typedef typename std::remove_reference<decltype(db)>::type Db;
        const int ds = dim / M, nc = db.pq->table_dim() / M;
        for (int m = 0; m < M; ++m)                              // codebooks = sub-vectors of the first vectors
            for (int c = 0; c < nc; ++c)
                for (int d = 0; d < ds; ++d) db.pq->centroids[((std::size_t)m * nc + c) * ds + d] = base[(std::size_t)(c % n) * dim + m * ds + d];
        db.add_vectors(base.data(), n);
        codes = db.codes;
        recording_scanner<Db> scanner;
        // process_queries<> with the heaps kept (query_common.hpp:330-368)
        auto loop = [&](auto& engine) {
            engine.prepare_database();
            for (int q = 0; q < nq; ++q) {
                float_heap bh(r);
                query_metrics m;
                call_engine(engine, q, queries.data(), nq, dim, bh, m);
                if (bh.size() != r) std::cerr << " WARNING: Binheap not full" << std::endl;
                const unsigned* k = bh.keys();
                recall += std::find(k, k + bh.size(), gt[q]) != k + bh.size() ? 1 : 0;
                metrics += m;
                heaps.push_back(bh);
            }
        };
        if (batch != 1) { nns_engine_batch<Db, recording_scanner<Db>> e(scanner, db, 1, batch, r); loop(e); }
        else { nns_engine<Db, recording_scanner<Db>> e(scanner, db, 1); loop(e); }
        tables = scanner.tables_seen;
    };
    if (bits == 4) {
        flat_database db;
        db.pq.reset(new pq4(M, dim));
        run(db);
    } else {
        flat_database_t<pq_bytes> db;
        db.pq.reset(new pq_bytes(M, bits, dim));
        run(db);
    }
    metrics /= nq;
    recall /= nq;
    print_csv_adc(std::cout, r, recall, 1, metrics);
    std::ofstream f(dump, std::ios::binary);
    const std::int32_t hdr[6] = {M, bits, dim, (std::int32_t)n, nq, r};
    f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    f.write(reinterpret_cast<const char*>(codes.data()), (std::streamsize)codes.size());
    f.write(reinterpret_cast<const char*>(tables.data()), (std::streamsize)(tables.size() * sizeof(float)));
    for (auto& h : heaps) {
        const std::int32_t sz = h.size();
        f.write(reinterpret_cast<const char*>(&sz), 4);
        f.write(reinterpret_cast<const char*>(h.keys()), (std::streamsize)(sizeof(unsigned) * sz));
        f.write(reinterpret_cast<const char*>(h.values()), (std::streamsize)(sizeof(float) * sz));
    }
    return 0;
}
