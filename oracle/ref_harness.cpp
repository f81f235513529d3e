// TEST INFRASTRUCTURE — not product code.
//
// C-ABI harness around the reference's OWN header-only hot-path kernels, compiled
// from where they lie under /root/reference (never copied into this repo):
//   binheap.hpp      kv_binheap<K,V>                       (binheap.hpp:18-142)
//   simd_layout.hpp  interleave_partition_4 & helpers      (simd_layout.hpp:16-65)
//   simd_scan.hpp    scan_avx_4<16>, scan_avx_4<32>        (simd_scan.hpp:125-187)
// Output goes to oracle/_ref/libqadc_ref.so only (git-ignored, travels with gpurun).
//
// Only accommodation: simd_scan.hpp:120 defines its own _mm256_set_m128i, which
// GCC >= 8 already provides in <immintrin.h>; the macro below renames the
// reference's helper (same semantics) so the header compiles unmodified.
// No stand-in headers or libraries are involved: these three headers need only
// <immintrin.h> and the STL.  db_query_4.cpp (scanner_4, QuantizerMAX) and
// query_common.hpp (scan_4) pull in Cereal/cblas/OpenCV through databases.hpp and
// do not compile as whole files; the functions of theirs that are on the path are
// compiled from line ranges of those files by ref_float_harness.cpp (libqadc_ref_float.so).
//
// What the harness itself adds (and nothing more): the call sequence of
// scanner_4::query_scan's integer half (db_query_4.cpp:276, 287-308):
// push the (0,127) sentinel, then scan each probed partition in order into one heap.
#include <immintrin.h>
#include <x86intrin.h>
#include <sched.h>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#define _mm256_set_m128i qadc_ref_mm256_set_m128i
#include "binheap.hpp"
#include "simd_layout.hpp"
#include "simd_scan.hpp"
#undef _mm256_set_m128i

extern "C" {

long qadc_ref_interleaved_size(unsigned n, int code_size) {
    return compute_interleaved_size_4(n, code_size, 16);
}

// Row-major [n][code_size] -> reference block layout (simd_layout.hpp:55-65).
void qadc_ref_interleave(std::uint8_t* dst, const std::uint8_t* rowmajor, unsigned n, int code_size) {
    source_partition src{rowmajor, code_size, n};
    interleave_partition_4(dst, src, 16);
}

// Integer half of scanner_4::query_scan over `nparts` probed partitions sharing one heap.
//   parts[p]   : interleaved partition (qadc_ref_interleave output)
//   labels[p]  : u32[size] or NULL  (labels == NULL => all partitions unlabeled)
//   qtables    : int8 [nparts][M][16], centroid c at byte c (db_query_4.cpp:57-70)
// Outputs the raw heap arrays (binheap.hpp keys()/values()), its size and sort_keys().
int qadc_ref_scan(int M, int nparts, const std::uint8_t* const* parts,
                  const std::uint32_t* const* labels, const std::uint32_t* sizes,
                  const std::int8_t* qtables, int R, int push_sentinel,
                  std::uint32_t* out_keys, std::int8_t* out_values, int* out_size,
                  std::uint32_t* out_sorted_keys) {
    if (M != 16 && M != 32) return -1;
    kv_binheap<unsigned, std::int8_t> bh(R);
    if (push_sentinel) bh.push(0, 127);                       // db_query_4.cpp:276
    std::unique_ptr<__m128i[]> qt(new __m128i[M]);
    for (int p = 0; p < nparts; ++p) {
        if (sizes[p] == 0) continue;                          // db_query_4.cpp:291-293
        for (int m = 0; m < M; ++m)
            qt[m] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(qtables + (static_cast<long>(p) * M + m) * 16));
        const unsigned* lab = labels ? labels[p] : nullptr;
        if (M == 16) scan_avx_4<16>(parts[p], lab, 0, sizes[p], qt.get(), bh);
        else         scan_avx_4<32>(parts[p], lab, 0, sizes[p], qt.get(), bh);
    }
    *out_size = bh.size();
    std::memcpy(out_keys, bh.keys(), sizeof(unsigned) * bh.size());
    std::memcpy(out_values, bh.values(), bh.size());
    if (out_sorted_keys) bh.sort_keys(out_sorted_keys);
    return 0;
}

// Plain push replay through the reference heap (int8 values).
void qadc_ref_heap_replay_i8(long n, const std::uint32_t* keys, const std::int8_t* vals, int R,
                             std::uint32_t* out_keys, std::int8_t* out_values, int* out_size,
                             std::uint32_t* out_sorted_keys) {
    kv_binheap<unsigned, std::int8_t> bh(R);
    for (long i = 0; i < n; ++i) bh.push(keys[i], vals[i]);
    *out_size = bh.size();
    std::memcpy(out_keys, bh.keys(), sizeof(unsigned) * bh.size());
    std::memcpy(out_values, bh.values(), bh.size());
    if (out_sorted_keys) bh.sort_keys(out_sorted_keys);
}

// All-cores leg of bench.py's CPU baseline: `nthreads` host threads (one per PHYSICAL core when the caller passes
// cpus[]; thread t is pinned to cpus[t]), each scanning its OWN first-touched copy of the interleaved partition
// with scan_avx_4<M>, one whole query after the other with a fresh heap (what the reference's single-threaded
// query loop does, query_common.hpp:351-365, replicated per core), until `seconds` have passed.
// Returns the number of whole queries finished by all threads and the wall time they took.
int qadc_ref_scan_mt(int M, const std::uint8_t* part, std::uint32_t size, const std::int8_t* qtables, int nqt, int R,
                     int nthreads, const int* cpus, double seconds, long* out_queries, double* out_elapsed) {
    if ((M != 16 && M != 32) || size == 0 || nqt <= 0 || nthreads <= 0) return -1;
    const long bytes = compute_interleaved_size_4(size, M / 2, 16);
    std::vector<long> done(nthreads, 0);
    std::vector<double> elapsed(nthreads, 0.0);
    auto work = [&](int t) {
        if (cpus) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[t], &set);
            (void)sched_setaffinity(0, sizeof(set), &set);
        }
        void* mem = nullptr;
        if (posix_memalign(&mem, 64, (bytes + 63) / 64 * 64) != 0) return;
        std::uint8_t* mine = static_cast<std::uint8_t*>(mem);
        std::memcpy(mine, part, bytes);                            // first touch on this thread's node
        std::unique_ptr<__m128i[]> qt(new __m128i[M]);
        const auto t0 = std::chrono::steady_clock::now();
        long n = 0;
        double dt = 0;
        do {
            const std::int8_t* q = qtables + static_cast<long>((n + t) % nqt) * M * 16;
            for (int m = 0; m < M; ++m) qt[m] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(q + m * 16));
            kv_binheap<unsigned, std::int8_t> bh(R);
            bh.push(0, 127);
            if (M == 16) scan_avx_4<16>(mine, nullptr, 0, size, qt.get(), bh);
            else         scan_avx_4<32>(mine, nullptr, 0, size, qt.get(), bh);
            ++n;
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (dt < seconds);
        done[t] = n;
        elapsed[t] = dt;
        std::free(mine);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    long total = 0;
    double longest = 0;
    for (int t = 0; t < nthreads; ++t) { total += done[t]; longest = elapsed[t] > longest ? elapsed[t] : longest; }
    *out_queries = total;
    *out_elapsed = longest;
    return 0;
}

// Plain push replay through the reference heap (float values).
void qadc_ref_heap_replay_f32(long n, const std::uint32_t* keys, const float* vals, int R,
                              std::uint32_t* out_keys, float* out_values, int* out_size) {
    kv_binheap<unsigned, float> bh(R);
    for (long i = 0; i < n; ++i) bh.push(keys[i], vals[i]);
    *out_size = bh.size();
    std::memcpy(out_keys, bh.keys(), sizeof(unsigned) * bh.size());
    std::memcpy(out_values, bh.values(), sizeof(float) * bh.size());
}

}  // extern "C"
