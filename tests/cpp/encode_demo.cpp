// TEST DRIVER (CPU only): the host twins of the device encoder and of the BLAS-expansion table form —
// pq4::encode / pq4::tables_blas (host/query_driver.hpp) and pq_bytes::encode (host/scanner_simple.hpp).
// in: a binary file {int32 M, bits, dim, n, form, has_rotation; float codebooks[M][2^bits][dim/M];
//     float rotation[dim][dim] (if has_rotation); float vectors[n][dim]};
// out: a binary file {uint8 codes[n][code_size]; float tables_blas[n][M * 2^bits]}.
// tests/test_scanner_hip_cpp.py compares both with the oracle (orc_pq_encode, orc_tables_expansion).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../quick-adc_amd/host/query_driver.hpp"
#include "../../quick-adc_amd/host/scanner_simple.hpp"

template <typename Pq>
static int run(Pq& pq, const std::vector<float>& vecs, int n, const char* out) {
    std::vector<std::uint8_t> codes((size_t)n * pq.code_size());
    std::vector<float> tables((size_t)n * pq.table_dim());
    pq.encode(vecs.data(), (size_t)n, codes.data());
    std::vector<float> rot(vecs);
    pq.rotate_multiple_vectors(rot.data(), n);
    pq.tables_blas(rot.data(), n, tables.data());
    FILE* f = std::fopen(out, "wb");
    if (!f) return 2;
    std::fwrite(codes.data(), 1, codes.size(), f);
    std::fwrite(tables.data(), sizeof(float), tables.size(), f);
    std::fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[6];
    if (std::fread(hdr, sizeof(int), 6, f) != 6) return 2;
    const int M = hdr[0], bits = hdr[1], dim = hdr[2], n = hdr[3], form = hdr[4], has_rot = hdr[5];
    std::vector<float> cb((size_t)M * (1u << bits) * (dim / M)), rot(has_rot ? (size_t)dim * dim : 0), vecs((size_t)n * dim);
    if (std::fread(cb.data(), sizeof(float), cb.size(), f) != cb.size()) return 2;
    if (has_rot && std::fread(rot.data(), sizeof(float), rot.size(), f) != rot.size()) return 2;
    if (std::fread(vecs.data(), sizeof(float), vecs.size(), f) != vecs.size()) return 2;
    std::fclose(f);
    if (bits == 4) {
        qadc::pq4 pq(M, dim);
        pq.centroids = cb;
        pq.rotation = rot;
        pq.encode_form = form;
        return run(pq, vecs, n, argv[2]);
    }
    qadc::pq_bytes pq(M, bits, dim);
    pq.centroids = cb;
    return run(pq, vecs, n, argv[2]);
}
