// End-to-end C++14 driver shaped like the reference's db_query_4 (db_query_4.cpp:358-414), with the GPU
// scanner behind the ScannerType seam: synthetic clustered vectors -> PQ 16x4/32x4 encode (flat or IVF)
// -> per query: assign, tables (host) -> scanner_hip::query_scan (MI355X) -> recall@R against the exact
// float nearest neighbour -> the reference's CSV line.  With --dump every query_scan call's inputs and
// the resulting heap are written to a file so the Python test can replay them through the CPU oracle.
//   usage: db_query_4_hip flat|ivf M N nq R keep_percent ma K seed [dumpfile|-] [batch]
// File mode = the reference's own command line (db_query_4.cpp:316-356, README.md:327-330) on the reference's own
// file formats (host/qadc_io.hpp): database archive, .fvecs/.bvecs/.ivecs queries, .ivecs ground truth:
//   usage: db_query_4_hip [-r R] [-m ma] [-b batch] [-k keep_percent] DB QUERIES GROUNDTRUTH [dumpfile]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>

#include "../../quick-adc_amd/host/qadc_io.hpp"
#include "../../quick-adc_amd/host/query_driver.hpp"
#include "../../quick-adc_amd/host/scanner_hip.hpp"

using namespace qadc;

template <typename Db>
struct recording_scanner {
    typedef kv_heap<unsigned, std::int8_t> BhType;
    scanner_hip<Db, BhType, query_metrics> inner;
    std::ofstream* dump;
    recording_scanner(float keep, std::ofstream* d) : inner(keep, 0, /*free_host_partitions=*/false), dump(d) {}
    void prepare_database(Db& db) { inner.prepare_database(db); }
    // batched form: remember the inputs of the batch, dump them query by query at replay time
    std::vector<int> b_assign;
    std::vector<float> b_tables;
    int b_ma = 0, b_dim = 0;
    void batch_scan(int nq, int* assign, int ma, float* tables, int table_dim, int R) {
        b_assign.assign(assign, assign + (size_t)nq * ma);
        b_tables.assign(tables, tables + (size_t)nq * ma * table_dim);
        b_ma = ma;
        b_dim = table_dim;
        inner.batch_scan(nq, assign, ma, tables, table_dim, R);
    }
    void batch_replay(int q, BhType& bh) {
        inner.batch_replay(q, bh);
        if (!dump) return;
        const int sz = bh.size();
        dump->write(reinterpret_cast<const char*>(b_assign.data() + (size_t)q * b_ma), sizeof(int) * b_ma);
        dump->write(reinterpret_cast<const char*>(b_tables.data() + (size_t)q * b_ma * b_dim), sizeof(float) * b_ma * b_dim);
        dump->write(reinterpret_cast<const char*>(&sz), sizeof(int));
        dump->write(reinterpret_cast<const char*>(bh.keys()), sizeof(unsigned) * sz);
        dump->write(reinterpret_cast<const char*>(bh.values()), sz);
    }
    void query_scan(const float* q, int* assign, int ma, float* tables, int table_dim, BhType& bh, query_metrics& m) {
        std::vector<float> before(tables, tables + (size_t)ma * table_dim);
        inner.query_scan(q, assign, ma, tables, table_dim, bh, m);
        if (!dump) return;
        const int sz = bh.size();
        dump->write(reinterpret_cast<const char*>(assign), sizeof(int) * ma);
        dump->write(reinterpret_cast<const char*>(before.data()), sizeof(float) * before.size());
        dump->write(reinterpret_cast<const char*>(&sz), sizeof(int));
        dump->write(reinterpret_cast<const char*>(bh.keys()), sizeof(unsigned) * sz);
        dump->write(reinterpret_cast<const char*>(bh.values()), sz);
    }
};

template <typename Db>
static void dump_db(std::ofstream& f, Db& db, int M) {
    const int parts = db.partition_count();
    f.write(reinterpret_cast<const char*>(&parts), sizeof(int));
    for (int p = 0; p < parts; ++p) {
        const std::uint8_t* c;
        unsigned* l;
        unsigned n;
        db.get_partition(p, c, l, n);
        const int labeled = l != nullptr;
        f.write(reinterpret_cast<const char*>(&n), sizeof(unsigned));
        f.write(reinterpret_cast<const char*>(&labeled), sizeof(int));
        f.write(reinterpret_cast<const char*>(c), (size_t)n * (M / 2));
        if (labeled) f.write(reinterpret_cast<const char*>(l), sizeof(unsigned) * n);
    }
}

static int g_batch = 1;  // the CLI's -b (db_query_4.cpp:343)

template <typename Db>
static void run(Db& db, const std::vector<float>& queries, int nq, int dim, int R, float keep, int ma,
                const std::vector<unsigned>& gt, std::ofstream* dump, int M) {
    if (dump) dump_db(*dump, db, M);
    recording_scanner<Db> scanner(keep, dump);
    query_metrics metrics;
    double recall = 0;
    if (g_batch != 1) {  // db_query_4.cpp:368-376
        nns_engine_batch<Db, recording_scanner<Db>> engine(scanner, db, ma, g_batch, R);
        process_queries<nns_engine_batch<Db, recording_scanner<Db>>, typename recording_scanner<Db>::BhType>(
            engine, queries.data(), nq, dim, R, gt.data(), metrics, recall);
    } else {
        nns_engine<Db, recording_scanner<Db>> engine(scanner, db, ma);
        process_queries<nns_engine<Db, recording_scanner<Db>>, typename recording_scanner<Db>::BhType>(
            engine, queries.data(), nq, dim, R, gt.data(), metrics, recall);
    }
    print_csv(std::cout, R, recall, ma, keep, metrics);
}

static std::unique_ptr<pq4> pq_from_archive(const io::pq_data& d) {
    if (d.sq_bits != 4 || (d.sq_count != 16 && d.sq_count != 32)) {  // get_simd_scan_func_epi8, db_query_4.cpp:22-35
        std::cerr << "Unsupported (nsq,nsq_bits) configuration. Supported configurations are: (16,4) (32,4)." << std::endl;
        std::exit(1);
    }
    std::unique_ptr<pq4> pq(new pq4(d.sq_count, d.dim));
    pq->centroids = d.centroids;
    pq->rotation = d.rotation;
    return pq;
}

// the reference's db_query_4 main (db_query_4.cpp:316-414) on files
static int main_files(int argc, char** argv) {
    int R = 100, ma = 1;
    float keep = 0.005f;  // db_query_4.cpp:324
    std::vector<const char*> pos;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (a[0] == '-' && a[1] && std::strchr("rmbk", a[1])) {
            const char* v = a[2] ? a + 2 : (i + 1 < argc ? argv[++i] : "");
            if (a[1] == 'r') R = std::atoi(v);
            if (a[1] == 'm') ma = std::atoi(v);
            if (a[1] == 'b') g_batch = std::atoi(v);
            if (a[1] == 'k') keep = (float)std::atof(v) * 0.01f;  // a percentage (db_query_4.cpp:345)
        } else {
            pos.push_back(a);
        }
    }
    if (pos.size() < 3) {
        std::cerr << "Usage: " << argv[0] << " [-r R] [-m ma] [-b batch] [-k keep%] [db_file] [query_file] [groundtruth_file]" << std::endl;
        return 1;
    }
    try {
        std::cerr << "Database file: " << pos[0] << std::endl;
        io::db_archive ar = io::load_database(pos[0]);
        io::vectors_owner<float> queries = io::load_vectors_by_extension(pos[1]);
        io::vectors_owner<int> gtv = io::load_ivecs(pos[2]);
        if (queries.dimension != ar.pq.dim) {
            std::cerr << "Query vectors have " << queries.dimension << " dimensions, the database " << ar.pq.dim << std::endl;
            return 1;
        }
        const int nq = (int)std::min<long>(queries.count, gtv.count);
        std::vector<unsigned> gt(nq);
        for (int q = 0; q < nq; ++q) gt[q] = (unsigned)gtv.get(q)[0];
        std::ofstream dumpf;
        if (pos.size() > 3) dumpf.open(pos[3], std::ios::binary);
        const int M = ar.pq.sq_count;
        if (!ar.indexed) {
            flat_database db;
            db.pq = pq_from_archive(ar.pq);
            db.codes.swap(ar.codes);
            db.count = ar.codes_count;
            run(db, queries.data, nq, queries.dimension, R, keep, 1, gt, dumpf.is_open() ? &dumpf : nullptr, M);
        } else {
            ivf_database db(pq_from_archive(ar.pq), ar.part_count, ar.centroids);
            db.partitions.swap(ar.partitions);
            db.labels.swap(ar.labels);
            run(db, queries.data, nq, queries.dimension, R, keep, ma, gt, dumpf.is_open() ? &dumpf : nullptr, M);
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;   // the reference prints and exits 1
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) != "flat" && std::string(argv[1]) != "ivf") return main_files(argc, argv);
    if (argc < 10) {
        std::fprintf(stderr, "usage: %s flat|ivf M N nq R keep_percent ma K seed [dumpfile]\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1];
    const int M = std::atoi(argv[2]);
    const unsigned N = (unsigned)std::atol(argv[3]);
    const int nq = std::atoi(argv[4]), R = std::atoi(argv[5]);
    const float keep = (float)std::atof(argv[6]) * 0.01f;  // the CLI's -k is a percentage (db_query_4.cpp:324,345)
    const int ma = std::atoi(argv[7]), K = std::atoi(argv[8]);
    const unsigned seed = (unsigned)std::atol(argv[9]);
    const int dim = 128;
    std::ofstream dumpf;
    if (argc > 10 && std::string(argv[10]) != "-") dumpf.open(argv[10], std::ios::binary);
    if (argc > 11) g_batch = std::atoi(argv[11]);

    // clustered synthetic vectors: 2000 centres ~ 3*N(0,1), points = centre + N(0,1)
    std::mt19937 rng(seed);
    std::normal_distribution<float> g(0.0f, 1.0f);
    const int C = 2000;
    std::vector<float> centres((size_t)C * dim);
    for (auto& v : centres) v = 3.0f * g(rng);
    auto draw = [&](std::vector<float>& out, size_t n) {
        out.resize(n * dim);
        for (size_t i = 0; i < n; ++i) {
            const float* c = centres.data() + (size_t)(rng() % C) * dim;
            for (int d = 0; d < dim; ++d) out[i * dim + d] = c[d] + g(rng);
        }
    };
    std::vector<float> base, queries;
    draw(base, N);
    draw(queries, nq);

    // exact float nearest neighbour of every query (ground truth rank 0)
    std::vector<unsigned> gt(nq);
    for (int q = 0; q < nq; ++q) {
        float best = std::numeric_limits<float>::max();
        for (unsigned i = 0; i < N; ++i) {
            float s = 0;
            for (int d = 0; d < dim; ++d) {
                const float t = queries[(size_t)q * dim + d] - base[(size_t)i * dim + d];
                s += t * t;
            }
            if (s < best) { best = s; gt[q] = i; }
        }
    }

    std::unique_ptr<pq4> pq(new pq4(M, dim));
    const int ds = dim / M;
    if (mode == "flat") {
        // codebooks = sub-vectors of sampled base vectors
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < 16; ++c) {
                const float* v = base.data() + (size_t)(rng() % N) * dim + m * ds;
                std::copy(v, v + ds, pq->centroids.begin() + ((size_t)m * 16 + c) * ds);
            }
        flat_database db;
        db.pq = std::move(pq);
        db.add_vectors(base.data(), N);
        run(db, queries, nq, dim, R, keep, 1, gt, dumpf.is_open() ? &dumpf : nullptr, M);
    } else {
        std::vector<float> coarse((size_t)K * dim);
        for (int k = 0; k < K; ++k) {
            const float* v = base.data() + (size_t)(rng() % N) * dim;
            std::copy(v, v + dim, coarse.begin() + (size_t)k * dim);
        }
        // codebooks = sub-vectors of sampled residuals (vector minus a coarse centroid)
        for (int m = 0; m < M; ++m)
            for (int c = 0; c < 16; ++c) {
                const float* v = base.data() + (size_t)(rng() % N) * dim + m * ds;
                const float* cc = coarse.data() + (size_t)(rng() % K) * dim + m * ds;
                for (int d = 0; d < ds; ++d) pq->centroids[((size_t)m * 16 + c) * ds + d] = 0.5f * (v[d] - cc[d]);
            }
        ivf_database db(std::move(pq), K, coarse);
        db.add_vectors(base.data(), N, 0);
        run(db, queries, nq, dim, R, keep, ma, gt, dumpf.is_open() ? &dumpf : nullptr, M);
    }
    return 0;
}
