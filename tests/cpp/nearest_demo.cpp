// TEST DRIVER (CPU only): ivf_database::nearest of host/query_driver.hpp — the host twin of the device's coarse selection —
// on distances that tie exactly.  in: a binary file {int32 K, dim, ma, nq; float coarse[K][dim]; float queries[nq][dim]};
// out: nq lines of ma indices.  tests/test_scanner_hip_cpp.py compares them with the reference's own heaps
// (find_k_neighbors' selection half, oracle/_ref) on the same sequential distances.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../quick-adc_amd/host/query_driver.hpp"

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[4];
    if (std::fread(hdr, sizeof(int), 4, f) != 4) return 2;
    const int K = hdr[0], dim = hdr[1], ma = hdr[2], nq = hdr[3];
    std::vector<float> coarse((size_t)K * dim), queries((size_t)nq * dim);
    if (std::fread(coarse.data(), sizeof(float), coarse.size(), f) != coarse.size()) return 2;
    if (std::fread(queries.data(), sizeof(float), queries.size(), f) != queries.size()) return 2;
    std::fclose(f);
    qadc::ivf_database db(std::unique_ptr<qadc::pq4>(new qadc::pq4(16, dim)), K, coarse);
    std::vector<int> out(ma);
    for (int q = 0; q < nq; ++q) {
        db.nearest(queries.data() + (size_t)q * dim, ma, out.data());
        for (int a = 0; a < ma; ++a) std::printf("%d%c", out[a], a + 1 == ma ? '\n' : ' ');
    }
    return 0;
}
