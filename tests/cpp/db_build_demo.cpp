// C++14 check of host/db_build.hpp against the host loops of host/query_driver.hpp (run by tests/test_scanner_hip_cpp.py):
// the GPU-built IVF / flat databases must hold exactly the codes, labels and partitions the CPU build produces, and the
// GPU k-means iterations exactly the centroids of the sequential restatement.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "../../quick-adc_amd/host/db_build.hpp"

using namespace qadc;

int main(int argc, char** argv) {
    const int M = argc > 1 ? std::atoi(argv[1]) : 16;
    const int dim = argc > 2 ? std::atoi(argv[2]) : 32;
    const int K = argc > 3 ? std::atoi(argv[3]) : 37;
    const unsigned n = argc > 4 ? (unsigned)std::atoi(argv[4]) : 5000;
    const bool opq = argc > 5 && std::atoi(argv[5]) != 0;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto make_pq = [&]() {
        std::unique_ptr<pq4> p(new pq4(M, dim));
        std::mt19937 r2(77);
        for (auto& x : p->centroids) x = nd(r2);
        if (opq) {
            p->rotation.resize((size_t)dim * dim);
            for (auto& x : p->rotation) x = nd(r2) * 0.3f;
        }
        return p;
    };
    std::vector<float> vecs((size_t)n * dim), coarse((size_t)K * dim);
    for (auto& x : vecs) x = nd(rng);
    for (auto& x : coarse) x = nd(rng);
    for (int d = 0; d < dim; ++d) coarse[(size_t)3 * dim + d] = coarse[(size_t)5 * dim + d];   // an exact tie between two centroids

    // IVF: two chunks with their offsets, like db_add.cpp:52-82
    ivf_database cpu(make_pq(), K, coarse), gpu(make_pq(), K, coarse);
    const unsigned half = n / 2 + 3;
    cpu.add_vectors(vecs.data(), half, 0);
    cpu.add_vectors(vecs.data() + (size_t)half * dim, n - half, half);
    add_vectors_hip(gpu, vecs.data(), half, 0);
    add_vectors_hip(gpu, vecs.data() + (size_t)half * dim, n - half, half);
    int bad = 0;
    for (int p = 0; p < K; ++p) bad += cpu.partitions[p] != gpu.partitions[p] || cpu.labels[p] != gpu.labels[p];
    // flat
    flat_database fc, fg;
    fc.pq = make_pq();
    fg.pq = make_pq();
    fc.add_vectors(vecs.data(), n);
    add_vectors_hip(fg, vecs.data(), n);
    const int flat_bad = fc.codes != fg.codes || fc.count != fg.count;
    // k-means: 3 iterations from the first K vectors
    std::vector<float> c1(vecs.begin(), vecs.begin() + (size_t)K * dim), c2 = c1;
    std::vector<int> a1(n), a2(n);
    kmeans_fast_iterations(vecs.data(), n, dim, K, c1.data(), 3, a1.data());
    if (qadc_kmeans_iterations_host(vecs.data(), n, dim, K, c2.data(), 3, a2.data(), 0) != QADC_OK) {
        std::fprintf(stderr, "%s\n", qadc_last_error());
        return 2;
    }
    const int km_bad = std::memcmp(c1.data(), c2.data(), sizeof(float) * c1.size()) != 0 || a1 != a2;
    // db_add's loop: the base vectors come from an .fvecs file through the reader thread + two-chunk queue, in chunks that
    // do not divide n (db_add.cpp:52-82, vector_io.hpp:231-288); GPU build == host build == the in-memory build above
    int stream_bad = 0;
    if (argc > 6) {
        const std::string file = std::string(argv[6]) + ".fvecs";
        io::save_vectors(vecs.data(), dim, (long)n, file.c_str());
        ivf_database sg(make_pq(), K, coarse), sc(make_pq(), K, coarse);
        const unsigned chunk = n / 7 + 5;
        const unsigned got_g = db_add_hip(sg, file.c_str(), chunk, 0, true), got_c = db_add_hip(sc, file.c_str(), chunk, 0, false);
        stream_bad += got_g != n || got_c != n;
        for (int p = 0; p < K; ++p)
            stream_bad += sg.partitions[p] != cpu.partitions[p] || sg.labels[p] != cpu.labels[p] || sc.partitions[p] != cpu.partitions[p] ||
                          sc.labels[p] != cpu.labels[p];
        flat_database fs;
        fs.pq = make_pq();
        stream_bad += db_add_hip(fs, file.c_str(), chunk) != n || fs.codes != fc.codes;
        bool threw = false;
        try { db_add_hip(fs, (std::string(argv[6]) + ".nope").c_str()); } catch (const std::exception&) { threw = true; }
        stream_bad += !threw;
        std::remove(file.c_str());
    }
    std::printf("ivf_partitions_differing %d flat_differs %d kmeans_differs %d streamed_add_differs %d\n", bad, flat_bad, km_bad, stream_bad);
    return (bad || flat_bad || km_bad || stream_bad) ? 1 : 0;
}
