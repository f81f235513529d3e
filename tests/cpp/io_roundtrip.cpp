// CPU-only exerciser of host/qadc_io.hpp for tests/test_io_formats.py: reads a file in one of the reference's
// formats, writes it back, prints a one-line summary.  usage: io_roundtrip vecs|pq|db IN OUT
//   io_roundtrip save f32|u8|i32 RAW DIM OUT     save_vectors<T> of raw row-major data (the writer, all element types)
//   io_roundtrip chunks IN CHUNK OUT             vectors_reader on its thread, consumed like db_add_hip: prints "offset count"
//                                                per chunk, writes every vector (float) to OUT
//   io_roundtrip recall GT.ivecs KEYS R T        check_labels (host/query_driver.hpp) of raw u32 keys [nq][R]: prints 0/1 per query
#include <cstdio>
#include <iostream>
#include <thread>

#include "../../quick-adc_amd/host/qadc_io.hpp"
#include "../../quick-adc_amd/host/query_driver.hpp"

using namespace qadc::io;

static std::vector<unsigned char> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("Could not open ") + path);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int extra_modes(int argc, char** argv) {
    const std::string mode = argv[1];
    if (mode == "save" && argc == 6) {
        const std::string kind = argv[2];
        const std::vector<unsigned char> raw = slurp(argv[3]);
        const int dim = std::atoi(argv[4]);
        const size_t esz = kind == "u8" ? 1 : 4;
        const long count = (long)(raw.size() / (esz * (size_t)dim));
        if (kind == "f32") save_vectors(reinterpret_cast<const float*>(raw.data()), dim, count, argv[5]);
        else if (kind == "u8") save_vectors(reinterpret_cast<const std::uint8_t*>(raw.data()), dim, count, argv[5]);
        else if (kind == "i32") save_vectors(reinterpret_cast<const std::int32_t*>(raw.data()), dim, count, argv[5]);
        else return 2;
        std::cout << "saved dim=" << dim << " count=" << count << std::endl;
        return 0;
    }
    if (mode == "chunks" && argc == 5) {
        vectors_reader reader(argv[2], (unsigned)std::atoi(argv[3]));
        std::thread th([&reader] { reader.run(); });
        std::ofstream out(argv[4], std::ios::binary);
        std::string error;
        while (!reader.done()) {
            vectors_chunk c = reader.get_chunk();
            if (c.failed) {
                error = c.error;
                break;
            }
            std::cout << c.offset << " " << c.count << std::endl;
            out.write(reinterpret_cast<const char*>(c.data.data()), (std::streamsize)(sizeof(float) * c.data.size()));
        }
        th.join();
        if (!error.empty()) throw std::runtime_error(error);
        std::cout << "total dim=" << reader.dim() << " count=" << reader.count() << std::endl;
        return 0;
    }
    if (mode == "recall" && argc == 6) {
        vectors_owner<int> gt = load_ivecs(argv[2]);
        const std::vector<unsigned char> raw = slurp(argv[3]);
        const int R = std::atoi(argv[4]), t = std::atoi(argv[5]);
        const unsigned* keys = reinterpret_cast<const unsigned*>(raw.data());
        const long nq = (long)(raw.size() / (sizeof(unsigned) * (size_t)R));
        if (t > gt.dimension || nq > gt.count) return 2;
        for (long q = 0; q < nq; ++q) std::cout << qadc::check_labels(gt.get((int)q), t, keys + q * R, keys + (q + 1) * R);
        std::cout << std::endl;
        return 0;
    }
    return -1;
}

int main(int argc, char** argv) {
    if (argc >= 2) {
        try {
            const int rc = extra_modes(argc, argv);
            if (rc >= 0) return rc;
        } catch (const std::exception& e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
    }
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
