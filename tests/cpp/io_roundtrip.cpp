// CPU-only exerciser of host/qadc_io.hpp for tests/test_io_formats.py: reads a file in one of the reference's
// formats, writes it back, prints a one-line summary.  usage: io_roundtrip vecs|pq|db IN OUT
#include <cstdio>
#include <iostream>

#include "../../quick-adc_amd/host/qadc_io.hpp"

using namespace qadc::io;

int main(int argc, char** argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s vecs|pq|db IN OUT\n", argv[0]);
        return 2;
    }
    const std::string mode = argv[1];
    try {
        if (mode == "vecs") {
            vectors_owner<float> v = load_vectors_by_extension(argv[2]);
            save_vectors(v.data.data(), v.dimension, v.count, argv[3]);
            std::cout << "vecs dim=" << v.dimension << " count=" << v.count << std::endl;
        } else if (mode == "pq") {
            pq_data pq = pq_from_data_file(argv[2]);
            pq_to_data_file(pq, argv[3]);
            std::cout << "pq dim=" << pq.dim << " m=" << pq.sq_count << " b=" << pq.sq_bits << " opq=" << pq.is_opq << std::endl;
        } else if (mode == "db") {
            db_archive db = load_database(argv[2]);
            save_database(db, argv[3]);
            size_t codes = db.codes.size(), labels = 0;
            for (auto& p : db.partitions) codes += p.size();
            for (auto& l : db.labels) labels += l.size();
            std::cout << "db indexed=" << db.indexed << " dim=" << db.pq.dim << " m=" << db.pq.sq_count << " b=" << db.pq.sq_bits
                      << " opq=" << db.pq.is_opq << " parts=" << db.part_count << " codes_count=" << db.codes_count
                      << " code_bytes=" << codes << " labels=" << labels << std::endl;
        } else {
            return 2;
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
