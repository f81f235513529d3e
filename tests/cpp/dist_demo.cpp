// C++14 multi-process driver of the native multi-GPU merge (include/qadc.h, qadc_dist_*): one process per GPU,
// no Python, no torch.  Every rank holds a contiguous range of ONE synthetic flat list (the same counter-based
// generator as bench.py), submits the same query batch, and qadc_dist_collect gathers the ranks' push streams
// with one ncclAllGather and replays them on the GPU — every rank prints the same heap checksum, which equals
// the checksum of an unsharded scan of the list (world = 1).
//   usage:  RANK=r WORLD_SIZE=w [LOCAL_RANK=d] dist_demo <id_file | shm:/name> <codes> <nq> [R]
// Rank 0 writes the 128-byte RCCL id to <id_file>.tmp and renames it; the others wait for the file.
// With "shm:/name" the ranks use the library's shared-memory transport instead of RCCL (qadc_dist_init_transport +
// qadc_shm_transport_*): several ranks may then share one GPU (LOCAL_RANK=0 for all) — how a single-GPU box runs
// world > 1 through the unmodified qadc_dist_collect.
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qadc.h"

static std::uint64_t splitmix64(std::uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        if ((call) != QADC_OK) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #call, qadc_last_error());              \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: RANK=r WORLD_SIZE=w dist_demo <id_file> <codes> <nq> [R]\n");
        return 2;
    }
    const char* env_rank = std::getenv("RANK");
    const char* env_world = std::getenv("WORLD_SIZE");
    const char* env_local = std::getenv("LOCAL_RANK");
    const int rank = env_rank ? std::atoi(env_rank) : 0, world = env_world ? std::atoi(env_world) : 1;
    const int device = env_local ? std::atoi(env_local) : rank;
    const std::string id_file = argv[1];
    const std::uint32_t N = (std::uint32_t)std::strtoull(argv[2], nullptr, 10);
    const int nq = std::atoi(argv[3]), R = argc > 4 ? std::atoi(argv[4]) : 100, M = 16;
    const float keep = 0.01f;
    const std::uint64_t seed = 0x5EED0001ull;

    const bool use_shm = id_file.compare(0, 4, "shm:") == 0;
    std::uint8_t id[QADC_DIST_ID_BYTES];
    void* shm = nullptr;
    if (use_shm) {
        if (qadc_shm_transport_open(id_file.c_str() + 4, rank, world, 32u << 20, 90.0, &shm) != QADC_OK) {
            std::fprintf(stderr, "qadc_shm_transport_open: %s\n", qadc_shm_transport_error());
            return 3;
        }
    } else if (rank == 0) {
        CHECK(qadc_dist_unique_id(id));
        const std::string tmp = id_file + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 3;
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file.c_str()) != 0) return 3;
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(id_file.c_str(), "rb")); ++tries) usleep(100000);
        if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) return 3;
        std::fclose(f);
    }

    // contiguous ranges in rank order, multiples of 16 codes (pyqadc/sharded.py: shard_ranges)
    const std::uint32_t per = (N / world) / 16 * 16, first = rank * per, local_n = rank == world - 1 ? N - first : per;
    const std::uint32_t starts = (std::uint32_t)((float)N * keep) > 0 ? (std::uint32_t)((float)N * keep) : 1;
    qadc_index* idx = nullptr;
    CHECK(qadc_index_create(&idx, M, device));
    CHECK(qadc_index_add_partition_synthetic_shard(idx, N, first, local_n, seed, starts));
    CHECK(qadc_index_finalize(idx, keep));
    if (use_shm) CHECK(qadc_dist_init_transport(idx, rank, world, qadc_shm_transport_allgather, shm));
    else CHECK(qadc_dist_init(idx, rank, world, id));

    std::vector<float> tables((std::size_t)nq * M * 16);
    for (std::size_t i = 0; i < tables.size(); ++i) tables[i] = (float)(splitmix64(977 + i) >> 40) * (1.0f / 16777216.0f) * 4.0f;
    std::vector<std::int32_t> assign(nq, 0), sizes(nq), status(nq);
    std::vector<std::uint32_t> keys((std::size_t)nq * R);
    std::vector<std::int8_t> vals((std::size_t)nq * R);
    const float extra_in = 100.0f + rank;
    std::vector<float> extra_out(world);
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<float> t = tables;                                   // query_scan clamps its tables in place
        CHECK(qadc_query_scan_submit(idx, rep % 2, nq, 1, assign.data(), t.data(), R));
        CHECK(qadc_dist_collect(idx, rep % 2, keys.data(), vals.data(), sizes.data(), status.data(), &extra_in, 1, extra_out.data()));
    }
    std::uint64_t sum = 0;
    for (int q = 0; q < nq; ++q)
        for (int i = 0; i < sizes[q]; ++i) sum = splitmix64(sum ^ ((std::uint64_t)keys[(std::size_t)q * R + i] << 8 | (std::uint8_t)vals[(std::size_t)q * R + i]));
    bool extra_ok = true;
    for (int g = 0; g < world; ++g) extra_ok = extra_ok && extra_out[g] == 100.0f + g;
    std::printf("rank %d of %d: %d queries, heap checksum %016llx, extra payload %s\n", rank, world, nq, (unsigned long long)sum,
                extra_ok ? "ok" : "BAD");
    CHECK(qadc_index_destroy(idx));
    if (shm) qadc_shm_transport_close(shm);
    return extra_ok ? 0 : 4;
}
