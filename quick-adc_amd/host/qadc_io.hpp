// On-disk formats of the reference, restated for this repo's stand-alone driver:
//
//   .fvecs / .bvecs / .ivecs   per vector: int32 dimension, then `dimension` float32 / uint8 / int32 values
//                              (vector_io.hpp:69-149, vector_io.cpp:40-58; count = file size / record size)
//   .pq.data / .opq.data       int32 dim, m, b; float32 codebooks [m][2^b][dim/m]; .opq.data: + float32
//                              rotation [dim][dim]  (quantizers.cpp:27-103, convert-quantizer.py:8-40)
//   database archives          what `ar(std::unique_ptr<base_db>)` of cereal 1.2.2's BinaryOutputArchive writes
//                              for flat_db / index_db (databases.hpp:158-166, 300-330; quantizers.hpp:171-187,
//                              303-323; query_common.hpp:321-328).
//
// PARITY STATUS.  The vecs and .data layouts are pinned by the reference's own reader code cited above (plain
// fread-style records; the tests write them with numpy exactly as convert-quantizer.py does).  The archive layout
// is PARITY UNPINNED: Cereal is a third-party dependency of the reference (getdeps.sh pins cereal v1.2.2) that is
// absent from /root/reference and from this image, so the layout below is restated from cereal 1.2.2's published
// binary-archive rules and could not be checked against a file written by the reference:
//   * arithmetic values raw little-endian; std::vector<arithmetic> = uint64 size + raw elements;
//     std::string = uint64 length + bytes; cereal::binary_data = raw bytes, no size;
//   * std::unique_ptr<T> of a polymorphic T = uint32 polymorphic_id [+ std::string type name the first time an
//     id is used, when the id's top bit is set] + uint8 "valid" + the object:
//       id 0                  null pointer
//       id 0x40000000         dynamic type == static type and not abstract (a plain base_pq behind base_pq)
//       id 0x80000000 | n     first object of the n-th registered type name in this archive (n from 1), name follows
//       id n                  later objects of that type
//   * no class-version fields (the reference declares none).
// Errors throw std::runtime_error carrying the reference's message text; a CLI prints it and exits 1 like the
// reference does.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <queue>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace qadc {
namespace io {

template <typename T>
struct vectors_owner {  // vector_io.hpp:38-55
    std::vector<T> data;
    int dimension = 0;
    long count = 0;
    const T* get(long i) const { return data.data() + (size_t)i * dimension; }
};

inline void open_or_throw(const char* filename, std::ifstream& f) {
    f.open(filename, std::ifstream::in | std::ifstream::binary);
    if (!f) throw std::runtime_error(std::string("Could not open ") + filename);
}

// InType on disk -> OutType in memory (implicit cast, load_vectors_convert vector_io.hpp:132-149)
template <typename InType, typename OutType>
vectors_owner<OutType> load_vectors_convert(const char* filename) {
    std::ifstream f;
    open_or_throw(filename, f);
    vectors_owner<OutType> v;
    std::int32_t dim = 0;
    f.read(reinterpret_cast<char*>(&dim), sizeof(dim));
    if (!f || dim <= 0) throw std::runtime_error(std::string("Could not load vectors from ") + filename);
    f.seekg(0, std::ifstream::end);
    const long bytes = (long)f.tellg();
    f.seekg(0, std::ifstream::beg);
    v.dimension = dim;
    v.count = bytes / ((long)dim * (long)sizeof(InType) + (long)sizeof(dim));   // count_vectors, vector_io.hpp:69-76
    v.data.resize((size_t)v.count * dim);
    std::vector<InType> tmp((size_t)dim);
    for (long i = 0; i < v.count; ++i) {
        std::int32_t d = 0;
        f.read(reinterpret_cast<char*>(&d), sizeof(d));
        if (d != dim) {  // check_dimension, vector_io.cpp:20-32
            std::ostringstream os;
            os << "Error while reading vectors.\nVector " << i << " has " << d << " dimensions while other vectors have "
               << dim << " dimensions\nAll vectors must have the same number of dimensions";
            throw std::runtime_error(os.str());
        }
        f.read(reinterpret_cast<char*>(tmp.data()), sizeof(InType) * (size_t)dim);
        for (int k = 0; k < dim; ++k) v.data[(size_t)i * dim + k] = (OutType)tmp[k];
    }
    if (!f) throw std::runtime_error(std::string("Could not load vectors from ") + filename);
    return v;
}

// load_vectors_by_extension (vector_io.cpp:40-58): everything becomes float
inline vectors_owner<float> load_vectors_by_extension(const char* filename) {
    const char* ext = std::strrchr(filename, '.');
    if (ext && !std::strcmp(ext, ".bvecs")) return load_vectors_convert<std::uint8_t, float>(filename);
    if (ext && !std::strcmp(ext, ".fvecs")) return load_vectors_convert<float, float>(filename);
    if (ext && !std::strcmp(ext, ".ivecs")) return load_vectors_convert<std::int32_t, float>(filename);
    throw std::runtime_error(std::string("Could not load vectors from ") + filename +
                             "\nUnknown extension\nKnown extensions: .bvecs, .ivecs, .fvecs");
}

// ground truth: .ivecs kept as int (recall_file, recall.hpp:33-40)
inline vectors_owner<int> load_ivecs(const char* filename) { return load_vectors_convert<std::int32_t, int>(filename); }

template <typename T>
void save_vectors(const T* data, int dim, long count, const char* filename) {  // vector_io.hpp:153-166
    std::ofstream f(filename, std::ios_base::out | std::ios_base::binary);
    if (!f) throw std::runtime_error(std::string("Could not open ") + filename);
    const std::int32_t d = dim;
    for (long i = 0; i < count; ++i) {
        f.write(reinterpret_cast<const char*>(&d), sizeof(d));
        f.write(reinterpret_cast<const char*>(data + (size_t)i * dim), sizeof(T) * (size_t)dim);
    }
}

// ---- chunked reading on a thread of its own (vector_io.hpp:180-290) ----------------------------------------------
// db_add (db_add.cpp:52-82) does not load the base file whole: a reader thread fills a bounded queue (two chunks of
// wanted_chunk_count vectors) while the main thread encodes the previous chunk.  Same here; the chunk carries its offset
// (= label of its first vector).
template <typename T>
class safe_bounded_queue {  // vector_io.hpp:189-229
    std::mutex mutex_;
    std::condition_variable not_empty_, not_full_;
    std::queue<T> queue_;
    std::size_t max_size_;

public:
    explicit safe_bounded_queue(int max_size) : max_size_((std::size_t)max_size) {}
    bool empty() {
        std::lock_guard<std::mutex> lock(mutex_);
        return queue_.empty();
    }
    void push(T&& item) {
        std::unique_lock<std::mutex> lock(mutex_);
        not_full_.wait(lock, [this] { return queue_.size() < max_size_; });
        queue_.push(std::move(item));
        lock.unlock();
        not_empty_.notify_one();
    }
    void pop(T& item) {
        std::unique_lock<std::mutex> lock(mutex_);
        not_empty_.wait(lock, [this] { return !queue_.empty(); });
        item = std::move(queue_.front());
        queue_.pop();
        lock.unlock();
        not_full_.notify_one();
    }
};

struct vectors_chunk {  // vector_io.hpp:168-185
    std::vector<float> data;
    int dim = 0;
    unsigned count = 0, offset = 0;
    bool failed = false;         // the reader hit an error: `error` holds the reference's message
    std::string error;
};

class vectors_reader {  // vector_io.hpp:231-288 + vectors_reader_by_extension (vector_io.cpp:60-91)
public:
    static constexpr int MAX_QUEUE_SIZE = 2;
    vectors_reader(const char* filename, unsigned chunk_count = 1000000)
        : queue_(MAX_QUEUE_SIZE), wanted_chunk_count_(chunk_count), filename_(filename) {
        const char* ext = std::strrchr(filename, '.');
        if (ext && !std::strcmp(ext, ".bvecs")) elem_ = 1;
        else if (ext && !std::strcmp(ext, ".fvecs")) elem_ = 4;
        else if (ext && !std::strcmp(ext, ".ivecs")) elem_ = -4;
        else throw std::runtime_error(std::string("Could not load vectors from ") + filename +
                                      "\nUnknown extension\nKnown extensions: .bvecs, .ivecs, .fvecs");
        std::ifstream f;
        open_or_throw(filename, f);
        std::int32_t dim = 0;
        f.read(reinterpret_cast<char*>(&dim), sizeof(dim));
        if (!f || dim <= 0) throw std::runtime_error(std::string("Could not load vectors from ") + filename);
        f.seekg(0, std::ifstream::end);
        dim_ = dim;
        count_ = (unsigned)((long)f.tellg() / ((long)dim * (elem_ < 0 ? 4 : elem_) + 4));
    }
    // the reader thread's body
    void run() {
        std::ifstream f;
        try {
            open_or_throw(filename_.c_str(), f);
            const std::size_t esz = (std::size_t)(elem_ < 0 ? 4 : elem_);
            std::vector<unsigned char> rec((std::size_t)dim_ * esz);
            // (the reference publishes its read count BEFORE the push, vector_io.hpp:231-288, and its done() reads that count:
            // a consumer asking in between sees the count complete and the queue empty and leaves without the last chunk — or
            // without the reader's error.  Here done() counts what the CONSUMER has taken out: no window on either side.)
            unsigned read = 0;
            while (read != count_) {
                vectors_chunk chunk;
                chunk.dim = dim_;
                chunk.count = std::min(wanted_chunk_count_, count_ - read);
                chunk.offset = read;
                chunk.data.resize((std::size_t)chunk.count * dim_);
                for (unsigned i = 0; i < chunk.count; ++i) {
                    std::int32_t d = 0;
                    f.read(reinterpret_cast<char*>(&d), sizeof(d));
                    if (!f || d != dim_) {
                        std::ostringstream os;
                        os << "Error while reading vectors.\nVector " << (chunk.offset + i) << " has " << d
                           << " dimensions while other vectors have " << dim_ << " dimensions\nAll vectors must have the same number of dimensions";
                        throw std::runtime_error(os.str());
                    }
                    f.read(reinterpret_cast<char*>(rec.data()), (std::streamsize)rec.size());
                    float* o = chunk.data.data() + (std::size_t)i * dim_;
                    if (elem_ == 1) for (int k = 0; k < dim_; ++k) o[k] = (float)rec[k];
                    else if (elem_ == 4) std::memcpy(o, rec.data(), rec.size());
                    else for (int k = 0; k < dim_; ++k) { std::int32_t v; std::memcpy(&v, rec.data() + 4 * k, 4); o[k] = (float)v; }
                }
                read += chunk.count;
                queue_.push(std::move(chunk));
            }
        } catch (const std::exception& e) {
            vectors_chunk bad;
            bad.failed = true;
            bad.error = e.what();
            queue_.push(std::move(bad));
        }
    }
    unsigned count() const { return count_; }
    int dim() const { return dim_; }
    // (consumer side, one thread: every vector was handed out, or the reader's error was)
    bool done() const { return consumed_ == count_ || failed_; }
    vectors_chunk get_chunk() {
        vectors_chunk c;
        queue_.pop(c);
        if (c.failed) failed_ = true;
        else consumed_ += c.count;
        return c;
    }

private:
    safe_bounded_queue<vectors_chunk> queue_;
    unsigned wanted_chunk_count_;
    int dim_ = 0, elem_ = 4;
    unsigned count_ = 0;
    unsigned consumed_ = 0;              // vectors get_chunk() has handed out
    bool failed_ = false;                // ... or the chunk carrying the reader's error
    std::string filename_;
};

// ---- product quantizer files ------------------------------------------------------------------
struct pq_data {
    int dim = 0, sq_count = 0, sq_bits = 0;
    bool is_opq = false;
    std::vector<float> centroids;  // [sq_count][2^sq_bits][dim / sq_count]
    std::vector<float> rotation;   // [dim][dim] (opq only)
    size_t all_centroids_dim() const { return ((size_t)1 << sq_bits) * (size_t)dim; }
};

// parse_data_filename (quantizers.cpp:58-87): "....pq.data" or "....opq.data"
inline bool data_filename_is_opq(const char* filename) {
    const std::string fn(filename);
    const size_t ext = fn.rfind('.');
    const std::string bad = std::string("Invalid data filename: ") + filename + "\nFilename must end with: .pq.data or .opq.data";
    if (ext == std::string::npos || fn.substr(ext) != ".data") throw std::runtime_error(bad);
    const std::string before = fn.substr(0, ext);
    const size_t p = before.rfind('.');
    if (p == std::string::npos) throw std::runtime_error(bad);
    const std::string t = before.substr(p);
    if (t == ".pq") return false;
    if (t == ".opq") return true;
    throw std::runtime_error(bad);
}

inline pq_data pq_from_data_file(const char* filename) {  // quantizers.cpp:27-46, 89-103
    pq_data pq;
    pq.is_opq = data_filename_is_opq(filename);
    std::ifstream f;
    open_or_throw(filename, f);
    std::int32_t hdr[3];
    f.read(reinterpret_cast<char*>(hdr), sizeof(hdr));
    pq.dim = hdr[0];
    pq.sq_count = hdr[1];
    pq.sq_bits = hdr[2];
    if (!f || pq.dim <= 0 || pq.sq_count <= 0 || pq.sq_bits <= 0 || pq.sq_bits > 16 || pq.dim % pq.sq_count)
        throw std::runtime_error(std::string("Invalid quantizer file: ") + filename);
    pq.centroids.resize(pq.all_centroids_dim());
    f.read(reinterpret_cast<char*>(pq.centroids.data()), sizeof(float) * pq.centroids.size());
    if (pq.is_opq) {
        pq.rotation.resize((size_t)pq.dim * pq.dim);
        f.read(reinterpret_cast<char*>(pq.rotation.data()), sizeof(float) * pq.rotation.size());
    }
    if (!f) throw std::runtime_error(std::string("Invalid quantizer file: ") + filename);
    return pq;
}

inline void pq_to_data_file(const pq_data& pq, const char* filename) {  // convert-quantizer.py:8-40
    std::ofstream f(filename, std::ios_base::out | std::ios_base::binary);
    if (!f) throw std::runtime_error(std::string("Could not open ") + filename);
    const std::int32_t hdr[3] = {pq.dim, pq.sq_count, pq.sq_bits};
    f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    f.write(reinterpret_cast<const char*>(pq.centroids.data()), sizeof(float) * pq.centroids.size());
    if (pq.is_opq) f.write(reinterpret_cast<const char*>(pq.rotation.data()), sizeof(float) * pq.rotation.size());
}

// ---- database archives (cereal 1.2.2 binary layout, see the header comment) --------------------------
struct db_archive {
    bool indexed = false;  // flat_db / index_db
    pq_data pq;
    // flat_db (databases.hpp:38-40, 158-166)
    unsigned codes_count = 0;
    std::vector<std::uint8_t> codes;
    // index_db (databases.hpp:168-172, 300-330)
    int part_count = 0;
    std::vector<float> centroids;                       // [part_count][dim]
    std::vector<std::vector<std::uint8_t>> partitions;  // codes per partition
    std::vector<std::vector<unsigned>> labels;
};

namespace detail {

constexpr std::uint32_t kMsb = 0x80000000u, kMsb2 = 0x40000000u;

struct reader {
    std::ifstream f;
    std::map<std::uint32_t, std::string> names;  // polymorphic ids seen so far
    template <typename T>
    void raw(T* p, size_t n) {
        f.read(reinterpret_cast<char*>(p), sizeof(T) * n);
        if (!f) throw std::runtime_error("Database file is truncated");
    }
    template <typename T>
    T one() {
        T v;
        raw(&v, 1);
        return v;
    }
    std::string str() {
        const std::uint64_t n = one<std::uint64_t>();
        if (n > (1u << 20)) throw std::runtime_error("Database file is corrupt (type name length)");
        std::string s((size_t)n, '\0');
        if (n) raw(&s[0], (size_t)n);
        return s;
    }
    template <typename T>
    void vec(std::vector<T>& v) {
        const std::uint64_t n = one<std::uint64_t>();
        v.resize((size_t)n);
        if (n) raw(v.data(), (size_t)n);
    }
    // polymorphic pointer header -> dynamic type name ("" = the static type itself)
    std::string poly(bool& null) {
        const std::uint32_t id = one<std::uint32_t>();
        null = id == 0;
        std::string name;
        if (null) return name;
        if (id & kMsb2) {
            name = "";
        } else if (id & kMsb) {
            name = str();
            names[id & ~kMsb] = name;
        } else {
            auto it = names.find(id);
            if (it == names.end()) throw std::runtime_error("Database file is corrupt (unknown polymorphic id)");
            name = it->second;
        }
        if (one<std::uint8_t>() == 0) null = true;  // ptr_wrapper "valid"
        return name;
    }
};

inline void read_base_pq(reader& r, pq_data& pq) {  // quantizers.hpp:180-187
    pq.sq_count = r.one<std::int32_t>();
    pq.sq_bits = r.one<std::int32_t>();
    pq.dim = r.one<std::int32_t>();
    if (pq.dim <= 0 || pq.sq_count <= 0 || pq.sq_bits <= 0 || pq.sq_bits > 16)
        throw std::runtime_error("Database file is corrupt (quantizer header)");
    pq.centroids.resize(pq.all_centroids_dim());
    r.raw(pq.centroids.data(), pq.centroids.size());
}

inline void read_pq_ptr(reader& r, pq_data& pq) {
    bool null = false;
    const std::string type = r.poly(null);
    if (null) throw std::runtime_error("Database file holds no quantizer");
    if (type == "") {
        pq.is_opq = false;
        read_base_pq(r, pq);
    } else if (type == "opq") {  // quantizers.hpp:313-323: base class first, then the rotation
        pq.is_opq = true;
        read_base_pq(r, pq);
        pq.rotation.resize((size_t)pq.dim * pq.dim);
        r.raw(pq.rotation.data(), pq.rotation.size());
    } else {
        throw std::runtime_error("Database file holds an unknown quantizer type: " + type);
    }
}

struct writer {
    std::ofstream f;
    std::map<std::string, std::uint32_t> ids;
    template <typename T>
    void raw(const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }
    template <typename T>
    void one(T v) { raw(&v, 1); }
    void str(const std::string& s) {
        one<std::uint64_t>(s.size());
        raw(s.data(), s.size());
    }
    template <typename T>
    void vec(const std::vector<T>& v) {
        one<std::uint64_t>(v.size());
        raw(v.data(), v.size());
    }
    void poly(const std::string& type) {  // "" = static type
        if (type.empty()) {
            one<std::uint32_t>(kMsb2);
        } else {
            auto it = ids.find(type);
            if (it == ids.end()) {
                const std::uint32_t id = (std::uint32_t)ids.size() + 1;
                ids[type] = id;
                one<std::uint32_t>(id | kMsb);
                str(type);
            } else {
                one<std::uint32_t>(it->second);
            }
        }
        one<std::uint8_t>(1);
    }
};

inline void write_pq_ptr(writer& w, const pq_data& pq) {
    w.poly(pq.is_opq ? "opq" : "");
    w.one<std::int32_t>(pq.sq_count);
    w.one<std::int32_t>(pq.sq_bits);
    w.one<std::int32_t>(pq.dim);
    w.raw(pq.centroids.data(), pq.centroids.size());
    if (pq.is_opq) w.raw(pq.rotation.data(), pq.rotation.size());
}

}  // namespace detail

// load_database (query_common.hpp:321-328)
inline db_archive load_database(const char* filename) {
    detail::reader r;
    open_or_throw(filename, r.f);
    db_archive db;
    bool null = false;
    const std::string type = r.poly(null);
    if (null) throw std::runtime_error(std::string("Database file holds no database: ") + filename);
    if (type == "flat_db") {  // ar(pq, codes_count, codes)
        detail::read_pq_ptr(r, db.pq);
        db.codes_count = r.one<std::uint32_t>();
        r.vec(db.codes);
    } else if (type == "index_db") {  // ar(part_count, pq); centroids; partitions...; labels...
        db.indexed = true;
        db.part_count = r.one<std::int32_t>();
        detail::read_pq_ptr(r, db.pq);
        if (db.part_count <= 0) throw std::runtime_error("Database file is corrupt (partition count)");
        db.centroids.resize((size_t)db.part_count * db.pq.dim);
        r.raw(db.centroids.data(), db.centroids.size());
        db.partitions.resize(db.part_count);
        db.labels.resize(db.part_count);
        for (auto& p : db.partitions) r.vec(p);
        for (auto& l : db.labels) r.vec(l);
    } else {
        throw std::runtime_error("Database file holds an unknown database type: " + type);
    }
    return db;
}

// save_database (flatdb_create.cpp:49-53, same archive call)
inline void save_database(const db_archive& db, const char* filename) {
    detail::writer w;
    w.f.open(filename, std::ios_base::out | std::ios_base::binary);
    if (!w.f) throw std::runtime_error(std::string("Could not open ") + filename);
    if (!db.indexed) {
        w.poly("flat_db");
        detail::write_pq_ptr(w, db.pq);
        w.one<std::uint32_t>(db.codes_count);
        w.vec(db.codes);
    } else {
        w.poly("index_db");
        w.one<std::int32_t>(db.part_count);
        detail::write_pq_ptr(w, db.pq);
        w.raw(db.centroids.data(), db.centroids.size());
        for (auto& p : db.partitions) w.vec(p);
        for (auto& l : db.labels) w.vec(l);
    }
}

}  // namespace io
}  // namespace qadc
