// scanner_hip — C++14 host mirror of the reference's ScannerType for the Quick-ADC path.
//
// nns_engine<ScannerType> / nns_engine_batch<ScannerType> (query_common.hpp:149-309) consume a
// duck-typed scanner: `typedef BhType`, `prepare_database(base_db&)`, `query_scan(query, assign,
// ma, tables, table_dim, bh, metrics)`.  scanner_4 (db_query_4.cpp:73-310) is the AVX2
// implementation; this type has the same three members, the same argument meaning and the same
// error behaviour (message on std::cerr + std::exit(1)), and forwards to the C-ABI in
// include/qadc.h.  Template parameters let it compile both inside the reference tree
// (Db = base_db, Heap = kv_binheap<unsigned, std::int8_t>, Metrics = query_metrics) and stand-alone
// (tests/cpp) with any types that offer the same members:
//   Db:   int partition_count(); void get_partition(int, const std::uint8_t*&, unsigned*&, unsigned&);
//         void free_partition(int); pq->sq_count, pq->sq_bits          (databases.hpp:34-63)
//   Heap: Heap(int capacity); int capacity(); void push(unsigned, std::int8_t)   (binheap.hpp)
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../../include/qadc.h"
#include "qadc_heap.hpp"

namespace qadc {

struct no_metrics {};

template <typename Db, typename Heap = kv_heap<unsigned, std::int8_t>, typename Metrics = no_metrics>
struct scanner_hip {
    typedef Heap BhType;

    float keep;
    int device;
    bool free_host_partitions;  // scanner_4 frees the originals after its own copy (db_query_4.cpp:190)
    qadc_index* index;
    std::vector<std::uint32_t> cand_keys;
    std::vector<std::int8_t> cand_vals;

    explicit scanner_hip(float keep_, int device_ = 0, bool free_host_partitions_ = true)
        : keep(keep_), device(device_), free_host_partitions(free_host_partitions_), index(nullptr) {}
    scanner_hip(const scanner_hip&) = delete;
    scanner_hip& operator=(const scanner_hip&) = delete;
    ~scanner_hip() { qadc_index_destroy(index); }

    static void die(const char* what) {
        std::cerr << what << ": " << qadc_last_error() << std::endl;
        std::exit(1);
    }

    // scanner_4::prepare_database (db_query_4.cpp:210-228)
    void prepare_database(Db& db) {
        if (db.pq->sq_bits != 4 || qadc_index_create(&index, db.pq->sq_count, device) != QADC_OK) {
            std::cerr << "Unsupported (nsq,nsq_bits) configuration." << std::endl;
            std::cerr << "Supported configurations are: (16,4) (32,4)." << std::endl;
            std::cerr << qadc_last_error() << std::endl;
            std::exit(1);
        }
        const int part_count = db.partition_count();
        for (int part_i = 0; part_i < part_count; ++part_i) {
            const std::uint8_t* codes;
            unsigned* labels;
            unsigned size;
            db.get_partition(part_i, codes, labels, size);
            if (size == 0) std::cerr << "Warning: Partition " << part_i << " is empty" << std::endl;
            const std::uint32_t* lab = labels;
            const std::uint32_t sz = size;
            if (qadc_index_add_partitions(index, 1, &codes, labels ? &lab : nullptr, &sz) != QADC_OK)
                die("Cannot prepare database");  // incl. "Some partitions have labels and some have not"
            if (size != 0 && free_host_partitions) db.free_partition(part_i);
        }
        if (qadc_index_finalize(index, keep) != QADC_OK) die("Cannot prepare database");
    }

    // scanner_4::query_scan (db_query_4.cpp:245-309).  `query` is unused there too.
    void query_scan(const float* /*query*/, int* assign, int ma, float* tables, int /*table_dim*/, BhType& bh,
                    Metrics& /*metrics*/) {
        std::uint64_t offsets[2] = {0, 0};
        std::int32_t status = 0;
        if (cand_keys.empty()) {
            cand_keys.resize(1 << 16);
            cand_vals.resize(1 << 16);
        }
        int rc = qadc_query_scan_candidates(index, 1, ma, assign, tables, bh.capacity(), cand_keys.size(), cand_keys.data(),
                                            cand_vals.data(), offsets, &status, nullptr, nullptr);
        if (rc == QADC_E_CAPACITY) {  // offsets[1] holds the required size: grow and ask again
            cand_keys.resize(offsets[1]);
            cand_vals.resize(offsets[1]);
            rc = qadc_query_scan_candidates(index, 1, ma, assign, tables, bh.capacity(), cand_keys.size(), cand_keys.data(),
                                            cand_vals.data(), offsets, &status, nullptr, nullptr);
        }
        if (rc != QADC_OK) die("query_scan");
        if (status != 0) {  // db_query_4.cpp:271-274
            std::cerr << "Warning: Max quantization bound too high. Try larger keep value." << std::endl;
            std::exit(1);
        }
        bh.push(0, 127);  // db_query_4.cpp:276
        for (std::uint64_t i = offsets[0]; i < offsets[1]; ++i) bh.push(cand_keys[i], cand_vals[i]);
    }

    // Batched form for nns_engine_batch-style callers (query_common.hpp:149-243): the whole batch is scanned by
    // ONE call (tables [nq][ma][table_dim], assign [nq][ma]); batch_replay(q, bh) then fills query q's heap
    // exactly as query_scan would have.  R = heap capacity the caller will use.
    std::vector<std::uint64_t> batch_offsets;
    std::vector<std::int32_t> batch_status;

    void batch_scan(int nq, int* assign, int ma, float* tables, int /*table_dim*/, int R) {
        batch_offsets.assign((std::size_t)nq + 1, 0);
        batch_status.assign(nq, 0);
        if (cand_keys.empty()) {
            cand_keys.resize(1 << 16);
            cand_vals.resize(1 << 16);
        }
        int rc = qadc_query_scan_submit(index, 0, nq, ma, assign, tables, R);
        if (rc == QADC_OK)
            rc = qadc_query_scan_collect_candidates(index, 0, cand_keys.size(), cand_keys.data(), cand_vals.data(), nullptr,
                                                    batch_offsets.data(), batch_status.data(), nullptr, nullptr);
        if (rc == QADC_E_CAPACITY) {  // the result is kept: fetch it again with buffers of the required size
            cand_keys.resize(batch_offsets[nq]);
            cand_vals.resize(batch_offsets[nq]);
            rc = qadc_query_scan_collect_candidates(index, 0, cand_keys.size(), cand_keys.data(), cand_vals.data(), nullptr,
                                                    batch_offsets.data(), batch_status.data(), nullptr, nullptr);
        }
        if (rc != QADC_OK) die("batch_scan");
    }

    void batch_replay(int q, BhType& bh) {
        if (batch_status[q] != 0) {  // db_query_4.cpp:271-274
            std::cerr << "Warning: Max quantization bound too high. Try larger keep value." << std::endl;
            std::exit(1);
        }
        bh.push(0, 127);  // db_query_4.cpp:276
        for (std::uint64_t i = batch_offsets[q]; i < batch_offsets[q + 1]; ++i) bh.push(cand_keys[i], cand_vals[i]);
    }
};

}  // namespace qadc
