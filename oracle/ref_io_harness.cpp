// TEST INFRASTRUCTURE — not product code.
//
// C-ABI harness around the reference's OWN vecs I/O and recall rule, compiled from where they lie under
// /root/reference (never copied into this repo):
//   vector_io.hpp / vector_io.cpp   load_vectors_by_extension, load_vectors<T>, save_vectors<T>, vectors_reader +
//                                   vectors_reader_by_extension                      (vector_io.hpp:69-290, vector_io.cpp:40-91)
//   recall.hpp                      recall_file::check_labels / all_in               (recall.hpp:21-61)
// Both compile as they are with plain g++ -std=c++14 (STL only; recall.hpp pulls in binheap.hpp, also STL only): no
// stand-in headers, no macro.  Output goes to oracle/_ref/libqadc_ref_io.so only (git-ignored, travels with gpurun).
//
// What pins what: quick-adc_amd/host/qadc_io.hpp's vecs readers / writer / chunked reader and
// host/query_driver.hpp's check_labels — the recall column of process_queries<> (8 A9) — are compared with these entry
// points by tests/test_io_formats.py.
//
// What the harness itself adds: copies out of the reference's owning containers into malloc'ed buffers, the consumer loop
// of db_add.cpp:52-82 around the reference's reader thread, and — because the reference reports errors by message +
// std::exit(1) (vector_io.cpp:20-38, vector_io.hpp:58-66) — a fork()ed probe that returns the exit code and the message.
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "recall.hpp"
#include "vector_io.hpp"

namespace {
template <typename T>
int copy_out(const vectors_owner<T>& v, T** data, int* dim, long* count) {
    *dim = v.dimension;
    *count = v.count;
    const size_t n = (size_t)v.count * (size_t)v.dimension;
    *data = static_cast<T*>(std::malloc(n ? n * sizeof(T) : 1));
    if (!*data) return -1;
    if (n) std::memcpy(*data, v.data.get(), n * sizeof(T));
    return 0;
}
template <typename T>
void save_as(const char* filename, const T* data, int dim, long count) {
    vectors_owner<T> v;
    v.dimension = dim;
    v.count = count;
    v.data.reset(new T[(size_t)count * dim + 1]);
    std::memcpy(v.data.get(), data, sizeof(T) * (size_t)count * dim);
    save_vectors(v, filename);                                   // vector_io.hpp:153-166
}
}  // namespace

extern "C" {

void qadc_ref_io_free(void* p) { std::free(p); }

// load_vectors_by_extension (vector_io.cpp:40-58): .bvecs / .fvecs / .ivecs, everything becomes float.
// EXITS THE PROCESS on an unknown extension or a dimension mismatch (probe with qadc_ref_io_try_load first).
int qadc_ref_io_load(const char* filename, float** data, int* dim, long* count) {
    vectors_owner<float> v = load_vectors_by_extension(filename);
    return copy_out(v, data, dim, count);
}

// load_vectors<int> as recall_file's constructor reads the ground truth (recall.hpp:37-39).
int qadc_ref_io_load_ivecs(const char* filename, int** data, int* dim, long* count) {
    vectors_owner<int> v = load_vectors<int>(filename);
    return copy_out(v, data, dim, count);
}

// save_vectors<T> (vector_io.hpp:153-166) for the three element types of the formats.
void qadc_ref_io_save_f32(const char* filename, const float* data, int dim, long count) { save_as<float>(filename, data, dim, count); }
void qadc_ref_io_save_u8(const char* filename, const std::uint8_t* data, int dim, long count) { save_as<std::uint8_t>(filename, data, dim, count); }
void qadc_ref_io_save_i32(const char* filename, const int* data, int dim, long count) { save_as<int>(filename, data, dim, count); }

// The reference's error behaviour, observed from outside: a child process calls load_vectors_by_extension; returns its
// exit code (0 = loaded) and what it wrote to stderr.
int qadc_ref_io_try_load(const char* filename, char* err, int err_cap) {
    int fds[2];
    if (pipe(fds) != 0) return -1;
    const pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        close(fds[0]);
        dup2(fds[1], 2);
        (void)load_vectors_by_extension(filename);
        std::cerr.flush();
        _exit(0);
    }
    close(fds[1]);
    int used = 0;
    for (;;) {
        char buf[256];
        const ssize_t n = read(fds[0], buf, sizeof(buf));
        if (n <= 0) break;
        for (ssize_t i = 0; i < n && err && used + 1 < err_cap; ++i) err[used++] = buf[i];
    }
    if (err && err_cap > 0) err[used] = 0;
    close(fds[0]);
    int status = 0;
    waitpid(pid, &status, 0);
    return WIFEXITED(status) ? WEXITSTATUS(status) : -2;
}

// The chunked reader the way db_add.cpp:52-82 drives it: vectors_reader_by_extension (vector_io.cpp:60-91), run() on a
// thread of its own, `while (!reader->done()) chunk = reader->get_chunk()`.  chunk_count replaces the constructor's
// default of 1 000 000 (the factory passes none; the member is public).  out: all vectors in arrival order; offsets /
// counts: per chunk.
// The reference publishes its read count BEFORE the push (vector_io.hpp:256-258), so its done() can be seen true with the
// last chunk still on its way into the queue and the loop above then ends a chunk early — under CPU load, with small
// chunks, almost always.  What is pinned here is the READER (which chunks it cuts, what they hold), not that window: after
// the loop the harness joins the reader thread and takes what is still in the (public) queue, in order; *early_exit says
// whether the reference's own loop had ended early.
int qadc_ref_io_read_chunked(const char* filename, unsigned chunk_count, float* out, long out_cap, unsigned* offsets,
                             unsigned* counts, int max_chunks, int* nchunks, int* dim, unsigned* total, int* early_exit) {
    std::unique_ptr<vectors_reader> reader = vectors_reader_by_extension(filename);
    reader->wanted_chunk_count_ = chunk_count;
    vectors_reader* rp = reader.get();
    *dim = reader->dim();
    *total = reader->count();
    std::thread th([rp] { rp->run(); });
    long got = 0;
    int nc = 0;
    int rc = 0;
    auto take = [&](vectors_chunk<float>& chunk) {
        if (nc < max_chunks) {
            offsets[nc] = chunk.offset;
            counts[nc] = chunk.count;
        }
        ++nc;
        const long n = (long)chunk.count * reader->dim();
        if (got + n <= out_cap) std::memcpy(out + got, chunk.data.get(), sizeof(float) * (size_t)n);
        else rc = -1;
        got += n;
    };
    while (!reader->done()) {
        vectors_chunk<float> chunk = reader->get_chunk();
        take(chunk);
    }
    th.join();
    *early_exit = 0;
    while (!reader->queue_.empty()) {                            // (the reader has exited: nothing is pushed any more)
        vectors_chunk<float> chunk = reader->get_chunk();
        take(chunk);
        *early_exit = 1;
    }
    *nchunks = nc;
    if (rc == 0 && got != (long)reader->count() * reader->dim()) rc = 2;
    return rc;
}

// recall_file(gt).check_labels(query_i, keys, keys + n, t) (recall.hpp:46-54) — called with unsigned keys exactly as
// process_queries does (query_common.hpp:360-361: bh.keys(), bh.keys() + r, t = 1).
int qadc_ref_io_check_labels(const char* gt_filename, int nq, const unsigned* keys, int n, int t, int* out) {
    recall_file rec(gt_filename);
    if (t > rec.max_t()) return -1;
    for (int q = 0; q < nq; ++q) out[q] = rec.check_labels(q, keys + (long)q * n, keys + (long)q * n + n, t);
    return 0;
}

int qadc_ref_io_max_t(const char* gt_filename) {
    recall_file rec(gt_filename);
    return rec.max_t();
}

}  // extern "C"
