// Host-staged all-gather over POSIX shared memory: the built-in alternative to RCCL for qadc_dist_init_transport
// (include/qadc.h).  For ranks that cannot form an RCCL communicator — several processes sharing ONE GPU (how the
// multi-rank merge is exercised on a single-GPU box), or a host without librccl.  Every rank owns one slot of the
// segment: device -> own slot, barrier, all slots -> device, barrier.  The payload of the merge is a few hundred KB per
// rank and batch (latency-bound), so a staged copy costs microseconds next to the scan.  The segment is ordinary
// pageable memory to HIP (not registered: pinning would touch every page of every slot in every process).
// Every rank publishes the size of its block before the barrier and all of them must agree: a collective mismatch between
// the ranks (different gathers meeting each other) is an error on every rank, not silent garbage.  A segment left behind by a
// crashed run under the same name is recognised by its dead creator / replaced inode and waited out, not joined.
//
// No counterpart in the reference (one process, query_common.hpp:351-365).
#include "../../include/qadc.h"

#include <hip/hip_runtime.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>

namespace {

constexpr uint32_t kMagic = 0x51414443u;   // "QADC"
constexpr size_t kHeaderBytes = 4096;

struct ShmHeader {
    std::atomic<uint32_t> magic;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> abort_flag;       // a rank timed out or failed: every later barrier fails at once
    uint32_t world;
    uint32_t creator_pid;                   // rank 0's process: a segment whose creator is gone is a stale one
    uint64_t slot_bytes;
    std::atomic<uint64_t> bytes[16];        // size of the block every rank contributes to the gather in progress
};
static_assert(sizeof(ShmHeader) <= kHeaderBytes, "header must fit its page");

struct ShmTransport {
    std::string name;
    int fd = -1;
    unsigned char* base = nullptr;
    size_t map_bytes = 0;
    int rank = 0, world = 1;
    uint64_t slot_bytes = 0;
    double timeout_s = 120.0;
    std::string err;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    unsigned char* slot(int r) const { return base + kHeaderBytes + (size_t)r * slot_bytes; }
};

thread_local std::string g_shm_err;

int shm_fail(int code, const std::string& msg) {
    g_shm_err = msg;
    return code;
}

// Does `name` still lead to the segment this rank has mapped?  (rank 0 of a NEW run unlinks a stale segment and creates its own)
bool shm_same_segment(const ShmTransport* t) {
    struct stat a, b;
    const std::string path = "/dev/shm" + t->name;
    if (fstat(t->fd, &a) != 0 || stat(path.c_str(), &b) != 0) return false;
    return a.st_ino == b.st_ino && a.st_dev == b.st_dev;
}

// Sense-reversing barrier over the segment's counters; fails (and poisons the segment) after timeout_s.
int shm_barrier(ShmTransport* t) {
    ShmHeader* h = t->hdr();
    if (h->abort_flag.load(std::memory_order_acquire)) return shm_fail(QADC_E_STATE, "shm transport: another rank aborted");
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)t->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_acq_rel);
        return QADC_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (h->generation.load(std::memory_order_acquire) != gen) return QADC_OK;
        if (h->abort_flag.load(std::memory_order_acquire)) return shm_fail(QADC_E_STATE, "shm transport: another rank aborted");
        if ((spins & 255u) == 255u) {
            if ((spins & 0xffffu) == 0xffffu && t->rank != 0 && !shm_same_segment(t)) {
                // The name no longer leads here.  That is also what a FINISHED run looks like: rank 0 may have left this very
                // barrier and unlinked the name (qadc_shm_transport_close) between this rank's generation load and the stat.
                // So look at the barrier once more before calling it fatal: released or aborted wins over "replaced".
                if (h->generation.load(std::memory_order_acquire) != gen) return QADC_OK;
                if (h->abort_flag.load(std::memory_order_acquire)) return shm_fail(QADC_E_STATE, "shm transport: another rank aborted");
                return shm_fail(QADC_E_STATE, "shm transport: the segment was replaced under this rank (it had joined a stale segment of an earlier run)");
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > t->timeout_s) {
                h->abort_flag.store(1, std::memory_order_release);
                return shm_fail(QADC_E_STATE, "shm transport: barrier timed out (a rank is missing or failed)");
            }
            sched_yield();
        }
    }
}

// After the first barrier of a gather: every rank must have come with a block of the same size.  Ranks that disagree are in
// DIFFERENT collectives (one issued a gather the others did not): an error everywhere, and the segment is poisoned.
int shm_check_sizes(ShmTransport* t, uint64_t bytes_per_rank) {
    for (int r = 0; r < t->world; ++r) {
        const uint64_t b = t->hdr()->bytes[r].load(std::memory_order_acquire);
        if (b != bytes_per_rank) {
            t->hdr()->abort_flag.store(1, std::memory_order_release);
            return shm_fail(QADC_E_STATE, "shm transport: collective mismatch: rank " + std::to_string(r) + " came with " + std::to_string(b) +
                                              " bytes, this rank (" + std::to_string(t->rank) + ") with " + std::to_string(bytes_per_rank));
        }
    }
    return QADC_OK;
}

}  // namespace

extern "C" {

const char* qadc_shm_transport_error(void) { return g_shm_err.c_str(); }

int qadc_shm_transport_open(const char* name, int rank, int world, uint64_t slot_bytes, double timeout_s, void** out_ctx) {
    if (!name || name[0] != '/' || !out_ctx || world < 1 || world > 16 || rank < 0 || rank >= world || slot_bytes == 0)
        return shm_fail(QADC_E_ARG, "shm transport: need a name starting with '/', 0 <= rank < world <= 16, slot_bytes > 0");
    slot_bytes = (slot_bytes + 4095) & ~(uint64_t)4095;
    ShmTransport* t = new (std::nothrow) ShmTransport();
    if (!t) return shm_fail(QADC_E_HIP, "out of memory");
    t->name = name;
    t->rank = rank;
    t->world = world;
    t->slot_bytes = slot_bytes;
    if (timeout_s > 0) t->timeout_s = timeout_s;
    t->map_bytes = kHeaderBytes + (size_t)world * slot_bytes;
    const auto t0 = std::chrono::steady_clock::now();
    auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > t->timeout_s; };
    if (rank == 0) {
        (void)shm_unlink(name);                                   // a stale segment of a crashed run with the same name
        t->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (t->fd < 0 || ftruncate(t->fd, (off_t)t->map_bytes) != 0) {
            if (t->fd >= 0) { close(t->fd); (void)shm_unlink(name); }
            delete t;
            return shm_fail(QADC_E_HIP, std::string("shm transport: cannot create ") + name);
        }
    } else {
        // rank 0 creates the segment; wait until it exists at full size and carries this run's geometry
        for (;;) {
            t->fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (t->fd >= 0 && fstat(t->fd, &sb) == 0 && (size_t)sb.st_size >= t->map_bytes) break;
            if (t->fd >= 0) close(t->fd);
            t->fd = -1;
            if (expired()) {
                delete t;
                return shm_fail(QADC_E_STATE, std::string("shm transport: rank 0 never created ") + name);
            }
            usleep(1000);
        }
    }
    for (;;) {                                                    // (ranks other than 0 come back here after meeting a stale segment)
        void* p = mmap(nullptr, t->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
        if (p == MAP_FAILED) {
            close(t->fd);
            if (rank == 0) (void)shm_unlink(name);
            delete t;
            return shm_fail(QADC_E_HIP, "shm transport: mmap failed");
        }
        t->base = static_cast<unsigned char*>(p);
        ShmHeader* h = t->hdr();
        if (rank == 0) {
            new (h) ShmHeader();
            h->arrived.store(0);
            h->generation.store(0);
            h->abort_flag.store(0);
            h->world = (uint32_t)world;
            h->creator_pid = (uint32_t)getpid();
            h->slot_bytes = slot_bytes;
            for (auto& b : h->bytes) b.store(0);
            h->magic.store(kMagic, std::memory_order_release);
            break;
        }
        // A segment of this name may be the leftover of a crashed run (full size, valid magic, possibly poisoned): this
        // run's rank 0 will unlink it and create a new one.  Join only a segment whose creator is alive and that the name
        // still leads to; otherwise let go and look again.
        bool ok = false;
        while (!expired()) {
            if (h->magic.load(std::memory_order_acquire) == kMagic) { ok = true; break; }
            if (!shm_same_segment(t)) break;
            usleep(200);
        }
        if (ok) {
            const pid_t cp = (pid_t)h->creator_pid;
            const bool alive = cp > 0 && (kill(cp, 0) == 0 || errno == EPERM);
            ok = alive && !h->abort_flag.load(std::memory_order_acquire) && shm_same_segment(t);
        }
        if (ok) {
            if (h->world != (uint32_t)world || h->slot_bytes != slot_bytes) {
                munmap(t->base, t->map_bytes);
                close(t->fd);
                delete t;
                return shm_fail(QADC_E_ARG, "shm transport: the ranks disagree on world / slot_bytes");
            }
            break;
        }
        munmap(t->base, t->map_bytes);
        t->base = nullptr;
        close(t->fd);
        t->fd = -1;
        for (;;) {                                                // wait for a (new) segment of full size under the name
            if (expired()) {
                delete t;
                return shm_fail(QADC_E_STATE, std::string("shm transport: no live segment ") + name + " (rank 0 never created it, or only a stale one exists)");
            }
            usleep(1000);
            t->fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (t->fd >= 0 && fstat(t->fd, &sb) == 0 && (size_t)sb.st_size >= t->map_bytes) break;
            if (t->fd >= 0) close(t->fd);
            t->fd = -1;
        }
    }
    *out_ctx = t;
    return QADC_OK;
}

// qadc_allgather_fn: device buffers, stream-ordered on entry, complete on return.
int qadc_shm_transport_allgather(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream) {
    ShmTransport* t = static_cast<ShmTransport*>(ctx);
    if (!t || !d_send || !d_recv) return shm_fail(QADC_E_ARG, "shm transport: null argument");
    if (bytes_per_rank > t->slot_bytes) {
        t->hdr()->abort_flag.store(1, std::memory_order_release);
        return shm_fail(QADC_E_CAPACITY, "shm transport: a rank's block exceeds slot_bytes");
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipError_t e = hipMemcpyAsync(t->slot(t->rank), d_send, bytes_per_rank, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        t->hdr()->abort_flag.store(1, std::memory_order_release);
        return shm_fail(QADC_E_HIP, std::string("shm transport: device-to-host copy: ") + hipGetErrorString(e));
    }
    t->hdr()->bytes[t->rank].store(bytes_per_rank, std::memory_order_release);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (int rc = shm_barrier(t)) return rc;                      // every slot is written
    if (int rc = shm_check_sizes(t, bytes_per_rank)) return rc;
    e = hipMemcpy2DAsync(d_recv, bytes_per_rank, t->slot(0), t->slot_bytes, bytes_per_rank, (size_t)t->world,
                         hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        t->hdr()->abort_flag.store(1, std::memory_order_release);
        return shm_fail(QADC_E_HIP, std::string("shm transport: host-to-device copy: ") + hipGetErrorString(e));
    }
    return shm_barrier(t);                                       // every slot is read: it may be overwritten
}

// The same exchange between host buffers (no GPU involved): how the CPU test suite drives the barrier protocol.
int qadc_shm_transport_allgather_host(void* ctx, const void* send, void* recv, uint64_t bytes_per_rank) {
    ShmTransport* t = static_cast<ShmTransport*>(ctx);
    if (!t || !send || !recv) return shm_fail(QADC_E_ARG, "shm transport: null argument");
    if (bytes_per_rank > t->slot_bytes) {
        t->hdr()->abort_flag.store(1, std::memory_order_release);
        return shm_fail(QADC_E_CAPACITY, "shm transport: a rank's block exceeds slot_bytes");
    }
    std::memcpy(t->slot(t->rank), send, bytes_per_rank);
    t->hdr()->bytes[t->rank].store(bytes_per_rank, std::memory_order_release);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (int rc = shm_barrier(t)) return rc;
    if (int rc = shm_check_sizes(t, bytes_per_rank)) return rc;
    for (int r = 0; r < t->world; ++r) std::memcpy(static_cast<unsigned char*>(recv) + (size_t)r * bytes_per_rank, t->slot(r), bytes_per_rank);
    return shm_barrier(t);
}

int qadc_shm_transport_close(void* ctx) {
    ShmTransport* t = static_cast<ShmTransport*>(ctx);
    if (!t) return QADC_OK;
    if (t->base) munmap(t->base, t->map_bytes);
    if (t->fd >= 0) close(t->fd);
    if (t->rank == 0) (void)shm_unlink(t->name.c_str());          // (mappings of the other ranks stay valid)
    delete t;
    return QADC_OK;
}

}  // extern "C"
