// Database build on the GPU — C++14, header only, calls the C-ABI (include/qadc.h).
//
//   add_vectors_hip(flat_database&, ...)   flat_db::add_vectors  (databases.hpp:136-156)
//   add_vectors_hip(ivf_database&, ...)    index_db::add_vectors (databases.hpp:270-298): the device does the compute
//                                          (nearest centroid, residual, OPQ rotation, PQ encode: qadc_ivf_encode_host),
//                                          the host dispatches codes and labels to the partitions in vector order
//                                          exactly like lines 291-297
//   kmeans_fast_iterations(...)            CPU restatement of kmeans_fast_iterations_thread (databases.cpp:50-90), the
//                                          definition the device version is tested against bit for bit
//   learn_coarse_quantizer_hip(...)        learn_coarse_quantizer (databases.cpp:94-118) from a caller-provided seed:
//                                          the reference seeds with two OpenCV k-means++ iterations (third-party, absent
//                                          here), then runs kmeans_iter_max - 2 = 48 fast iterations — those run on the GPU
//   db_add_hip(db, base_file, chunk_count) db_add's add_vectors (db_add.cpp:52-82): a reader thread (io::vectors_reader,
//                                          vector_io.hpp:231-288) fills a two-chunk queue from the .fvecs/.bvecs file while
//                                          this thread encodes the previous chunk on the GPU; labels = index in chunk + the
//                                          chunk's offset, as there.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include <thread>

#include "../../include/qadc.h"
#include "qadc_io.hpp"
#include "query_driver.hpp"

namespace qadc {

inline void add_vectors_hip(flat_database& db, const float* vecs, unsigned n, int device = 0) {
    const pq4& pq = *db.pq;
    db.codes.resize((size_t)(db.count + n) * pq.code_size());
    if (qadc_ivf_encode_host(pq.sq_count, pq.dim, pq.centroids.data(), pq.rotation.empty() ? nullptr : pq.rotation.data(), 0,
                             nullptr, vecs, n, nullptr, db.codes.data() + (size_t)db.count * pq.code_size(), device) != QADC_OK)
        throw std::runtime_error(std::string("qadc_ivf_encode_host: ") + qadc_last_error());
    db.count += n;
}

inline void add_vectors_hip(ivf_database& db, const float* vecs, unsigned n, unsigned labels_offset, int device = 0) {
    const pq4& pq = *db.pq;
    const size_t cs = (size_t)pq.code_size();
    std::vector<std::int32_t> assign(n);
    std::vector<std::uint8_t> codes((size_t)n * cs);
    if (qadc_ivf_encode_host(pq.sq_count, pq.dim, pq.centroids.data(), pq.rotation.empty() ? nullptr : pq.rotation.data(),
                             db.part_count, db.coarse.data(), vecs, n, assign.data(), codes.data(), device) != QADC_OK)
        throw std::runtime_error(std::string("qadc_ivf_encode_host: ") + qadc_last_error());
    for (unsigned i = 0; i < n; ++i) {                         // databases.hpp:291-297
        const int p = assign[i];
        db.partitions[p].insert(db.partitions[p].end(), codes.begin() + (size_t)i * cs, codes.begin() + (size_t)(i + 1) * cs);
        db.labels[p].push_back(i + labels_offset);
    }
}

inline void add_chunk_hip(flat_database& db, const io::vectors_chunk& c, int device) { add_vectors_hip(db, c.data.data(), c.count, device); }
inline void add_chunk_hip(ivf_database& db, const io::vectors_chunk& c, int device) {
    add_vectors_hip(db, c.data.data(), c.count, c.offset, device);
}
inline void add_chunk_cpu(flat_database& db, const io::vectors_chunk& c) { db.add_vectors(c.data.data(), c.count); }
inline void add_chunk_cpu(ivf_database& db, const io::vectors_chunk& c) { db.add_vectors(c.data.data(), c.count, c.offset); }

// db_add.cpp:52-82.  on_gpu = false runs the host loops instead (the definition the GPU build is compared with).
// Returns the vectors added; throws with the reference's message if the reader fails (wrong dimension, unknown extension).
template <typename Db>
unsigned db_add_hip(Db& db, const char* base_filename, unsigned chunk_count = 1000000, int device = 0, bool on_gpu = true) {
    io::vectors_reader reader(base_filename, chunk_count);
    if (reader.dim() != db.pq->dim) throw std::runtime_error("base vectors and quantizer disagree on the dimension");
    std::thread read_thread([&reader] { reader.run(); });
    unsigned added = 0;
    std::string error;
    while (!reader.done()) {
        io::vectors_chunk chunk = reader.get_chunk();
        if (chunk.failed) {
            error = chunk.error;
            break;
        }
        if (error.empty()) {
            try {
                if (on_gpu) add_chunk_hip(db, chunk, device);
                else add_chunk_cpu(db, chunk);
                added += chunk.count;
            } catch (const std::exception& e) {
                error = e.what();                                // keep draining: the reader must not stay blocked on a full queue
            }
        }
    }
    read_thread.join();
    if (!error.empty()) throw std::runtime_error(error);
    return added;
}

// databases.cpp:50-90 on the host: assign every vector to its closest centroid (find_k_neighbors with k = 1: the expansion
// distances, first strict minimum), then centroid = (sum of its members in vector order) times the reciprocal of the count — what the reference
// binary does under -ffast-math (div_mode 1, pinned to its loops as compiled: oracle/_ref) — or divided by it as the source
// reads (div_mode 0).  An empty cluster becomes NaN either way.
inline void kmeans_fast_iterations(const float* vecs, size_t n, int dim, int K, float* centroids, int iters, int* assign,
                                   int div_mode = 1) {
    std::vector<int> cnt((size_t)K);
    std::vector<float> cn((size_t)K);
    for (int it = 0; it < iters; ++it) {
        // find_k_neighbors with k = 1 (databases.cpp:60-66 -> neighbors.cpp:30-76): expansion distances (float_sum.hpp), the first
        // strict minimum in centroid order (a capacity-1 kv_binheap: centroid 0 stays when its distance is NaN)
        for (int k = 0; k < K; ++k) cn[k] = sqnorm(centroids + (size_t)k * dim, dim);
        for (size_t i = 0; i < n; ++i) {
            const float* x = vecs + i * dim;
            const float xn = sqnorm(x, dim);
            int best = 0;
            float bestd = expansion_dist(x, centroids, dim, xn, cn[0]);
            for (int k = 1; k < K; ++k) {
                const float s = expansion_dist(x, centroids + (size_t)k * dim, dim, xn, cn[k]);
                if (s < bestd) { bestd = s; best = k; }
            }
            assign[i] = best;
        }
        std::fill(centroids, centroids + (size_t)K * dim, 0.0f);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = 0; i < n; ++i) {
            cnt[assign[i]]++;
            float* c = centroids + (size_t)assign[i] * dim;
            for (int d = 0; d < dim; ++d) c[d] += vecs[i * dim + d];
        }
        for (int k = 0; k < K; ++k) {
            const volatile float r = 1.0f / (float)cnt[k];          // (volatile: the host compiler must not fold the two forms)
            for (int d = 0; d < dim; ++d) {
                float& c = centroids[(size_t)k * dim + d];
                c = div_mode ? c * r : c / (float)cnt[k];
            }
        }
    }
}

constexpr int kmeans_iter_max = 50;                            // databases.cpp:92

// `seed` [K][dim]: what the reference gets from cv::kmeans(KMEANS_PP_CENTERS, 2 iterations)
inline std::vector<float> learn_coarse_quantizer_hip(const float* vecs, size_t n, int dim, int K, const float* seed,
                                                     int device = 0) {
    std::vector<float> centroids(seed, seed + (size_t)K * dim);
    if (qadc_kmeans_iterations_host(vecs, n, dim, K, centroids.data(), kmeans_iter_max - 2, nullptr, device) != QADC_OK)
        throw std::runtime_error(std::string("qadc_kmeans_iterations_host: ") + qadc_last_error());
    return centroids;
}

}  // namespace qadc
