// Stand-alone C++14 driver of scanner_hip with a minimal in-memory database that offers the
// members of base_db the scanner uses (databases.hpp:34-63).  Mirrors the inner part of
// process_queries<> (query_common.hpp:351-365): fresh heap per query, query_scan, dump keys/values.
// Data: counter-based synthetic codes (same splitmix64 stream as the oracle / HIP generator) so the
// Python test can rebuild the inputs and compare against the oracle.
//   usage: scanner_hip_demo M nparts size0 size1 ... labeled(0/1) keep R nq ma seed
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../quick-adc_amd/host/scanner_hip.hpp"
#ifdef QADC_USE_REFERENCE_HEAP
// Built by oracle/Makefile with -I/root/reference: BhType is then the reference's OWN kv_binheap<unsigned, int8_t>
// (binheap.hpp, compiled from where it lies) — the type nns_engine<scanner_4> hands to query_scan
// (db_query_4.cpp:244).  The output of this flavour must equal the kv_heap flavour's line for line.
#include "binheap.hpp"
typedef kv_binheap<unsigned, std::int8_t> demo_heap;
#else
typedef qadc::kv_heap<unsigned, std::int8_t> demo_heap;
#endif

static std::uint64_t splitmix64(std::uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct mini_pq { int sq_count; int sq_bits; };

struct mini_db {
    std::unique_ptr<mini_pq> pq;
    std::vector<std::vector<std::uint8_t>> codes;
    std::vector<std::vector<unsigned>> labels;
    bool labeled;
    int partition_count() { return (int)codes.size(); }
    void get_partition(int i, const std::uint8_t*& c, unsigned*& l, unsigned& size) {
        c = codes[i].data();
        l = labeled ? labels[i].data() : nullptr;
        size = (unsigned)(codes[i].size() / (pq->sq_count / 2));
    }
    void free_partition(int i) { std::vector<std::uint8_t>().swap(codes[i]); }
};

int main(int argc, char** argv) {
    int a = 1;
    const int M = std::atoi(argv[a++]);
    const int nparts = std::atoi(argv[a++]);
    std::vector<unsigned> sizes(nparts);
    for (int p = 0; p < nparts; ++p) sizes[p] = (unsigned)std::atol(argv[a++]);
    const bool labeled = std::atoi(argv[a++]) != 0;
    const float keep = (float)std::atof(argv[a++]);
    const int R = std::atoi(argv[a++]), nq = std::atoi(argv[a++]), ma = std::atoi(argv[a++]);
    const std::uint64_t seed = std::strtoull(argv[a++], nullptr, 10);
    const int cs = M / 2;

    mini_db db;
    db.pq.reset(new mini_pq{M, 4});
    db.labeled = labeled;
    for (int p = 0; p < nparts; ++p) {
        std::vector<std::uint8_t> c(((size_t)sizes[p] * cs + 7) / 8 * 8);
        for (size_t w = 0; w < c.size() / 8; ++w) {
            const std::uint64_t v = splitmix64((seed + p) ^ splitmix64(w));
            for (int k = 0; k < 8; ++k) c[8 * w + k] = (std::uint8_t)(v >> (8 * k));
        }
        c.resize((size_t)sizes[p] * cs);
        db.codes.push_back(c);
        std::vector<unsigned> l(sizes[p]);
        for (unsigned i = 0; i < sizes[p]; ++i) l[i] = 1000000u * (p + 1) + 7u * i;
        db.labels.push_back(l);
    }
    typedef qadc::scanner_hip<mini_db, demo_heap> Scanner;
    Scanner scanner(keep);
    scanner.prepare_database(db);
    qadc::no_metrics metrics;
    for (int q = 0; q < nq; ++q) {
        std::vector<int> assign(ma);
        for (int i = 0; i < ma; ++i) assign[i] = (q + i * 3) % nparts;
        std::vector<float> tables((size_t)ma * M * 16);
        for (size_t i = 0; i < tables.size(); ++i)
            tables[i] = (float)(splitmix64(seed * 31 + q * 1000003ull + i) >> 40) * (1.0f / 16777216.0f) * 4.0f;
        Scanner::BhType bh(R);
        scanner.query_scan(nullptr, assign.data(), ma, tables.data(), M * 16, bh, metrics);
        std::printf("q %d size %d\n", q, bh.size());
        for (int i = 0; i < bh.size(); ++i) std::printf("%u %d\n", bh.keys()[i], (int)bh.values()[i]);
        std::vector<unsigned> sorted(bh.size() > 0 ? bh.size() : 1);
        bh.sort_keys(sorted.data());                       // binheap.hpp:129-137
        std::printf("sorted");
        for (int i = 0; i < bh.size(); ++i) std::printf(" %u", sorted[i]);
        std::printf("\n");
    }
    return 0;
}
