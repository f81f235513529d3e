// Native multi-GPU merge (include/qadc.h "Multi-GPU"): one all-gather of device-resident push streams per
// batch — RCCL through dlopen (no link-time dependency; a single-GPU user never loads it) or a caller-supplied transport —
// then the world's streams replayed in global scan order (assign slot, rank, position).  State: qadc_index::dist (DistState).
#include "qadc_host.h"

#include <dlfcn.h>

using namespace qadc;
using namespace qadc::host;

namespace qadc {
namespace host {

int load_rccl(DistState& d, std::string& err) {
    if (d.lib) return 0;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        d.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (d.lib) break;
    }
    if (!d.lib) { err = std::string("cannot load RCCL: ") + dlerror(); return -1; }
    d.GetUniqueId = reinterpret_cast<int (*)(QadcNcclId*)>(dlsym(d.lib, "ncclGetUniqueId"));
    d.CommInitRank = reinterpret_cast<int (*)(void**, int, QadcNcclId, int)>(dlsym(d.lib, "ncclCommInitRank"));
    d.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(d.lib, "ncclAllGather"));
    d.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(d.lib, "ncclCommDestroy"));
    d.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(d.lib, "ncclGetErrorString"));
    if (!d.GetUniqueId || !d.CommInitRank || !d.AllGather || !d.CommDestroy) { err = "RCCL lacks an expected symbol"; return -1; }
    return 0;
}

// The merge of a one-workgroup-per-query batch, enqueued behind its scan: pack (from the kernels' own records in
// device memory) -> all-gather -> interleave -> replay; the collectives on the merge's stream, the compute on a stream
// of its own; the heaps and a status word land in pinned host memory.  Every rank enqueues the same collectives in the
// same order (the ranks submit and collect the same batches in the same order).
// Not taken (the collect-time merge runs instead): few-query batches (host-share replay), R or ma x world beyond the device
// merge, option dist_async = 0.
// The replay is sharded by query (option dist_shard_replay, world > 1): rank r interleaves and replays queries r, r + world, ...
// only — 1/world of the waves, 1/world of the scattered bytes — and a second all-gather of the heap shares ((R + 1/2) words
// per query) + dist_heaps_unpack_kernel put every rank's d_out in the state the unsharded replay leaves.  The replay is a
// latency chain of ~0.4 ms; its share gather is therefore NOT issued right behind it on the collectives' stream (everything
// later on that stream — the next batches' front and merge gathers — would wait for it) but behind the first gather of a
// LATER merge (one batch later; or at collect), when the chain is long over.  All of it in host program order, which
// is the same on every rank, from rank-invariant state (sequence numbers, option values): the ranks issue the same collectives
// in the same order.
// Two steps.  enqueue_merge (at the end of the batch's launch) records where the scan ends and marks the merge PENDING;
// flush_merges issues the pending merges in submission order.  A batch with a sharded front flushes the OLDER merges right
// after issuing its own front gather: on the collectives' stream that gather then lies in front of the previous batch's
// merge gather (which waits for that batch's scan), so a front that runs under the previous scan is not held up by it.
// heap share of a rank: heaps u64[per][R], then sizes u32[per]
static size_t share_words(int per, int R) { return (size_t)per * R + ((size_t)per + 1) / 2 + 1; }   // heaps, sizes, the "broken stream" word

int enqueue_merge_now(qadc_index* idx, Slot& s) {
    DistState& d = *idx->dist;
    const int slot_i = (int)(&s - idx->slot);
    DistSlot& ds = d.slot[slot_i];
    ds.pending = false;
    const int nq = s.nq, R = s.R, world = d.world;
    const size_t bw = dist_block_words(nq, d.cap_entries, 0);
    HIPCHECK(ds.d_block.ensure(bw));
    HIPCHECK(ds.d_gathered.ensure(bw * world));
    HIPCHECK(ds.d_merged.ensure((size_t)d.cap_entries * world));
    HIPCHECK(ds.d_moff.ensure(nq));
    HIPCHECK(ds.d_mcnt.ensure(2 * (size_t)nq));
    const size_t heaps_bytes = (sizeof(uint64_t) * (size_t)R + sizeof(uint32_t)) * (size_t)nq;
    HIPCHECK(ds.h_out.ensure(heaps_bytes + 32, hipHostMallocMapped | hipHostMallocCoherent));
    if (ds.h_out.p != ds.h_out_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ds.d_out), ds.h_out.p, 0));
        ds.h_out_mapped = ds.h_out.p;
    }
    if (!ds.ev_done) HIPCHECK(hipEventCreateWithFlags(&ds.ev_done, hipEventDisableTiming));
    if (!ds.ev_gathered) HIPCHECK(hipEventCreateWithFlags(&ds.ev_gathered, hipEventDisableTiming));
    const bool sharded = d.shard_replay && world > 1;
    const int per = (nq + world - 1) / world;
    const size_t sw = share_words(per, R);
    if (sharded) {
        // everything that can fail for lack of memory is done BEFORE the batch's first collective: between the stream gather and
        // the share gather only launches and event records remain, so a rank cannot drop out between two collectives its peers enter
        HIPCHECK(ds.d_share.ensure(sw));
        HIPCHECK(ds.d_shares.ensure(sw * world));
        if (!ds.ev_replayed) HIPCHECK(hipEventCreateWithFlags(&ds.ev_replayed, hipEventDisableTiming));
    }
    hipStream_t st = d.stream;
    HIPCHECK(hipStreamWaitEvent(st, ds.ev_ready, 0));
    if (s.wgq) {
        HIPCHECK(launch_dist_pack_qflags(s.d_qflags.p, nq, s.d_stream.p, s.wgq_cap, d.cap_entries, ds.d_block.p, st));
    } else {
        // level path: the ordering pass left {out_off, count, reps, flags} in the query states and the compact ordered
        // streams in d_stream; whatever the collect call would have to redo first (a query the device did not order, an
        // overflowed region / output / pre-scan buffer) raises bit7 and the merge is redone at collect time
        HIPCHECK(ds.d_src.ensure(3 * (size_t)nq));
        HIPCHECK(launch_dist_src_from_states(s.d_qs, nq, s.out_cap, ds.d_src.p, st));
        HIPCHECK(launch_dist_pack(ds.d_src.p, ds.d_src.p + nq, ds.d_src.p + 2 * (size_t)nq, nq, s.d_stream.p, nullptr, d.cap_entries,
                                  nullptr, 0, ds.d_block.p, st));
    }
    std::string gerr;
    if (d.gather(ds.d_block.p, ds.d_gathered.p, bw, st, gerr)) return fail(QADC_E_HIP, gerr);
    // the collectives keep `st` to themselves (the next batch's front gather is issued right behind this one); the merge's
    // compute — a millisecond of replay latency — goes to a stream of its own
    HIPCHECK(hipEventRecord(ds.ev_gathered, st));
    hipStream_t ms = d.merge_stream[ds.seq % kMergeStreams] ? d.merge_stream[ds.seq % kMergeStreams] : st;
    if (ms != st) HIPCHECK(hipStreamWaitEvent(ms, ds.ev_gathered, 0));
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(ds.d_out + sizeof(uint64_t) * (size_t)R * nq);
    if (sharded) {
        // (a rank past the last query of a ragged batch replays fewer; the unpack never reads those heaps)
        HIPCHECK(launch_dist_merge(ds.d_gathered.p, bw, world, nq, s.ma, (uint32_t)R, ds.d_moff.p, ds.d_mcnt.p, ds.d_mcnt.p + nq,
                                   ds.d_merged.p, ds.d_share.p, reinterpret_cast<uint32_t*>(ds.d_share.p + (size_t)per * R), ms,
                                   d_sizes + nq, d.rank, world));
        HIPCHECK(hipEventRecord(ds.ev_replayed, ms));
        ds.pending_share = true;
        ds.share_nq = nq;
        ds.share_R = R;
        ds.enqueued = true;                                      // (ev_done is recorded by issue_share)
        return QADC_OK;
    }
    HIPCHECK(launch_dist_merge(ds.d_gathered.p, bw, world, nq, s.ma, (uint32_t)R, ds.d_moff.p, ds.d_mcnt.p, ds.d_mcnt.p + nq,
                               ds.d_merged.p, reinterpret_cast<uint64_t*>(ds.d_out), d_sizes, ms, d_sizes + nq));
    HIPCHECK(hipEventRecord(ds.ev_done, ms));
    ds.enqueued = true;
    return QADC_OK;
}

// The second half of a sharded replay: all-gather of the heap shares + unpack into the slot's mapped result block.
int issue_share(qadc_index* idx, DistSlot& ds) {
    DistState& d = *idx->dist;
    ds.pending_share = false;
    const int nq = ds.share_nq, R = ds.share_R, world = d.world;
    const int per = (nq + world - 1) / world;
    const size_t sw = share_words(per, R);
    hipStream_t st = d.stream;
    HIPCHECK(hipStreamWaitEvent(st, ds.ev_replayed, 0));
    std::string gerr;
    if (d.gather(ds.d_share.p, ds.d_shares.p, sw, st, gerr)) return fail(QADC_E_HIP, gerr);
    uint32_t* d_sizes = reinterpret_cast<uint32_t*>(ds.d_out + sizeof(uint64_t) * (size_t)R * nq);
    HIPCHECK(launch_dist_heaps_unpack(ds.d_shares.p, sw, world, per, nq, (uint32_t)R, reinterpret_cast<uint64_t*>(ds.d_out), d_sizes, st));
    HIPCHECK(hipEventRecord(ds.ev_done, st));
    return QADC_OK;
}

// Issues the share gathers of the merges with seq <= upto, oldest first.
int flush_shares(qadc_index* idx, uint64_t upto) {
    DistState& d = *idx->dist;
    for (;;) {
        int best = -1;
        for (int i = 0; i < kSlots; ++i)
            if (d.slot[i].pending_share && d.slot[i].seq <= upto && (best < 0 || d.slot[i].seq < d.slot[best].seq)) best = i;
        if (best < 0) return QADC_OK;
        if (int rc = issue_share(idx, d.slot[best])) return rc;
    }
}

// Issues the pending merges with seq <= upto, oldest first.
int flush_merges(qadc_index* idx, uint64_t upto) {
    if (!idx->dist) return QADC_OK;
    DistState& d = *idx->dist;
    for (;;) {
        int best = -1;
        for (int i = 0; i < kSlots; ++i)
            if (d.slot[i].pending && d.slot[i].seq <= upto && (best < 0 || d.slot[i].seq < d.slot[best].seq)) best = i;
        if (best < 0) return QADC_OK;
        const uint64_t seq = d.slot[best].seq;
        if (int rc = enqueue_merge_now(idx, idx->slot[best])) return rc;
        // the heap shares of the merges kShareLag (1) back: their replays are over by now, the gather does not hold up this stream
        if (seq >= (uint64_t)kShareLag)
            if (int rc = flush_shares(idx, seq - (uint64_t)kShareLag)) return rc;
    }
}

int enqueue_merge(qadc_index* idx, Slot& s, hipStream_t scan_stream) {
    DistState& d = *idx->dist;
    const int slot_i = (int)(&s - idx->slot);
    if (slot_i < 0 || slot_i >= kSlots) return QADC_OK;
    DistSlot& ds = d.slot[slot_i];
    ds.enqueued = false;
    ds.pending = false;
    ds.pending_share = false;                                 // (a resubmitted slot: nothing of an earlier, failed enqueue may survive)
    const int nq = s.nq, R = s.R, world = d.world;
    // (level-path batches as well: the ordering pass leaves what the pack needs in the query states.  Below dist_device_nq
    // queries the merge stays at collect time, where the ranks replay shares on the host's otherwise idle cores: enqueuing
    // such a batch's device merge — or just its pack, gather and copy-out — behind the scan was measured on bench.py's
    // 32-query flat steps, one of 8 ranks: 1.61-1.65 ms per step against 1.22; interleave + replay kernels under the long
    // scan launches take 1.5-1.9 ms per batch, the host 0.07)
    if (nq < d.device_nq || (uint32_t)R > replay_wave_max_R() || (size_t)s.ma * world > dist_interleave_max_cells())
        return QADC_OK;
    if (!ds.ev_ready) HIPCHECK(hipEventCreateWithFlags(&ds.ev_ready, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(ds.ev_ready, scan_stream));
    ds.pending = true;
    ds.seq = d.next_seq++;
    if (!s.front_sharded) return flush_merges(idx, ds.seq);     // no front gather to let pass: issue it (and anything older) now
    return QADC_OK;
}
}  // namespace host
}  // namespace qadc

extern "C" {

// Host half of the multi-GPU merge (pyqadc/sharded.py): `gathered` holds, rank after rank, the int32 buffers the
// ranks contributed to the all-gather: [nq counts][cap keys][ceil(cap/4) words of int8 values]
// [only when ma > 1: ceil(cap/2) words of u16 assign slots][...].  Queries q_first, q_first + q_step, ... are replayed here in
// global scan order (assign slot, rank, position) through the reference heap after the (0,127) sentinel.
int qadc_merge_streams_i8(int world, int nq, int R, uint64_t cap, int ma, const int32_t* gathered, uint64_t buflen,
                          int q_first, int q_step, const int32_t* status, uint32_t* keys, int8_t* vals, int32_t* sizes) {
    if (world <= 0 || nq <= 0 || R <= 0 || ma <= 0 || !gathered || !keys || !vals || !sizes || q_step <= 0 || q_first < 0)
        return fail(QADC_E_ARG, "bad arguments");
    const uint64_t nv = (cap + 3) / 4, ns = ma > 1 ? (cap + 1) / 2 : 0;
    if (buflen < (uint64_t)nq + cap + nv + ns) return fail(QADC_E_ARG, "gathered buffers shorter than their layout");
    // entry offsets of every (rank, query)
    std::vector<uint64_t> offs((size_t)world * (nq + 1), 0);
    for (int g = 0; g < world; ++g) {
        const int32_t* cnt = gathered + (uint64_t)g * buflen;
        uint64_t* o = offs.data() + (size_t)g * (nq + 1);
        for (int q = 0; q < nq; ++q) o[q + 1] = o[q] + (uint32_t)cnt[q];
        if (o[nq] > cap) return fail(QADC_E_CAPACITY, "a rank's stream exceeds the gathered capacity");
    }
    std::vector<int> mine;
    for (int q = q_first; q < nq; q += q_step) mine.push_back(q);
    auto work = [&](size_t i0, size_t i1) {
        kv_heap<uint32_t, int8_t> bh(R);
        std::vector<uint64_t> cur(world);
        for (size_t i = i0; i < i1; ++i) {
            const int q = mine[i];
            sizes[q] = 0;
            if (status && status[q]) continue;
            bh.reset();
            bh.push(0, 127);                                     // db_query_4.cpp:276
            for (int g = 0; g < world; ++g) cur[g] = offs[(size_t)g * (nq + 1) + q];
            for (int slot = 0; slot < ma; ++slot)
                for (int g = 0; g < world; ++g) {
                    const int32_t* base = gathered + (uint64_t)g * buflen;
                    const uint32_t* k = reinterpret_cast<const uint32_t*>(base + nq);
                    const int8_t* v = reinterpret_cast<const int8_t*>(base + nq + cap);
                    const uint16_t* sl = reinterpret_cast<const uint16_t*>(base + nq + cap + nv);
                    const uint64_t end = offs[(size_t)g * (nq + 1) + q + 1];
                    uint64_t& c = cur[g];
                    // a rank scans its partitions in assign order: its slots are ascending
                    while (c < end && (ma == 1 || sl[c] == (uint16_t)slot)) {
                        bh.push(k[c], v[c]);
                        ++c;
                    }
                }
            sizes[q] = bh.size();
            std::memcpy(keys + (size_t)q * R, bh.keys(), sizeof(uint32_t) * bh.size());
            std::memcpy(vals + (size_t)q * R, bh.values(), bh.size());
        }
    };
    const size_t nt = std::min<size_t>(std::min<size_t>(mine.size(), 4), std::max<unsigned>(host_threads(), 1));
    if (nt <= 1) {
        work(0, mine.size());
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; ++t) th.emplace_back(work, mine.size() * t / nt, mine.size() * (t + 1) / nt);
        for (auto& x : th) x.join();
    }
    return QADC_OK;
}

extern "C++" {
namespace {
// The host half of qadc_dist_collect for few-query batches: rank `rank` replays the queries q = rank, rank + world, ...
// of the gathered blocks ([nq x {offset, count, flags, -} as u32][entries ...] per rank, `bw` words each) in GLOBAL scan
// order — assign slot, then rank (= ascending code range), then position — through the reference's heap, sentinel first
// (db_query_4.cpp:276).  myheaps: [ceil(nq / world)][R + 1] words, heap entries key | value << 32, then the size.
void replay_my_share(const uint64_t* gathered, size_t bw, int world, int rank, int nq, int ma, int R, const int32_t* status,
                     uint64_t* myheaps, WorkerPool* pool) {
    const int per = (nq + world - 1) / world;
    const size_t hw = (size_t)R + 1;
    auto work = [&](int j0, int j1) {
        kv_heap<uint32_t, int8_t> bh(R);
        std::vector<uint32_t> cur(world), end(world);
        for (int j = j0; j < j1; ++j) {
            const int q = j * world + rank;
            if (q >= nq || (status && status[q])) continue;
            bh.reset();
            bh.push(0, 127);
            for (int g = 0; g < world; ++g) {
                const uint32_t* h = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * bw) + 4 * (size_t)q;
                cur[g] = h[0];
                end[g] = h[0] + h[1];
            }
            for (int slot = 0; slot < ma; ++slot)
                for (int g = 0; g < world; ++g) {
                    const uint64_t* ent = gathered + (size_t)g * bw + 2 * (size_t)nq;
                    while (cur[g] < end[g]) {                    // a rank scans its partitions in assign order: slots ascend
                        const uint64_t e = ent[cur[g]];
                        if (ma > 1 && (int)((e >> 40) & 0x3fffu) != slot) break;
                        bh.push((uint32_t)e, (int8_t)(e >> 32));
                        ++cur[g];
                    }
                }
            uint64_t* o = myheaps + (size_t)j * hw;
            for (int i = 0; i < bh.size(); ++i) o[i] = (uint64_t)bh.keys()[i] | ((uint64_t)(uint8_t)bh.values()[i] << 32);
            o[R] = (uint64_t)bh.size();
        }
    };
    const int nt = std::max(1, std::min<int>(std::min(per, 8), (int)host_threads()));
    if (nt == 1 || !pool) {
        work(0, per);
    } else {
        pool->run(per, nt, [&](int j) { work(j, j + 1); });     // one query per task: their stream lengths differ
    }
}
}  // namespace
}  // extern "C++"

/* ---- native multi-GPU merge: one ncclAllGather per batch, device memory to device memory ---- */
int qadc_dist_unique_id(uint8_t* id128) {
    if (!id128) return fail(QADC_E_ARG, "id is null");
    DistState tmp;
    std::string err;
    if (load_rccl(tmp, err)) return fail(QADC_E_HIP, err);
    QadcNcclId id;
    const int rc = tmp.GetUniqueId(&id);
    if (rc != 0) return fail(QADC_E_HIP, std::string("ncclGetUniqueId: ") + (tmp.GetErrorString ? tmp.GetErrorString(rc) : "error"));
    std::memcpy(id128, id.internal, 128);
    return QADC_OK;                                          // (the library handle stays loaded for the process)
}

extern "C++" {
namespace {
// Releases a half-built DistState on every exit path of the init calls (communicator, stream).
struct DistGuard {
    std::unique_ptr<DistState> d;
    ~DistGuard() {
        if (!d) return;
        if (d->stream) { (void)hipStreamSynchronize(d->stream); }
        for (hipStream_t m : d->merge_stream) if (m) { (void)hipStreamSynchronize(m); }
        if (d->comm && d->CommDestroy) (void)d->CommDestroy(d->comm);     // (the streams belong to the index)
    }
};
int dist_init_checks(qadc_index* idx, int rank, int world) {
    if (!idx || world < 1 || world > 16 || rank < 0 || rank >= world) return fail(QADC_E_ARG, "need 0 <= rank < world <= 16");
    if (idx->dist) return fail(QADC_E_STATE, "qadc_dist_init was already called");
    for (auto& sl : idx->slot)
        if (sl.busy) return fail(QADC_E_STATE, "collect every batch before qadc_dist_init");
    return use_device(idx);
}
}  // namespace
}  // extern "C++"

int qadc_dist_init(qadc_index* idx, int rank, int world, const uint8_t* id128) {
    if (!id128) return fail(QADC_E_ARG, "id is null");
    if (int rc = dist_init_checks(idx, rank, world)) return rc;
    DistGuard g;
    g.d.reset(new DistState());
    DistState* d = g.d.get();
    std::string err;
    if (load_rccl(*d, err)) return fail(QADC_E_HIP, err);
    QadcNcclId id;
    std::memcpy(id.internal, id128, 128);
    const int rc = d->CommInitRank(&d->comm, world, id, rank);
    if (rc != 0) {
        d->comm = nullptr;
        return fail(QADC_E_HIP, std::string("ncclCommInitRank: ") + (d->GetErrorString ? d->GetErrorString(rc) : "error"));
    }
    d->rank = rank;
    d->world = world;
    // the merge's streams were created with the index (qadc_index_create: one fixed order, a hardware queue each) — the
    // collectives on the highest priority, interleave + replay at the scans' own lowest one: at normal priority their waves held
    // up the short launches between two batches' scans (one of 8 ranks, C5 shape 1.18 -> 1.10 ms per batch, C3 0.70 -> 0.60)
    d->stream = idx->coll_stream;
    for (int i = 0; i < kMergeStreams; ++i) d->merge_stream[i] = idx->merge_streams[i];
    // RCCL finishes setting up its channels lazily, inside the first collectives of a communicator (the very first
    // all-gather takes ~8 ms); a few throw-away gathers here keep that out of the first batches' collect calls.
    {
        constexpr size_t kWords = 1 << 16;
        DevBuf<uint64_t> src, dst;
        hipError_t he = src.ensure(kWords);
        if (he == hipSuccess) he = dst.ensure(kWords * world);
        if (he == hipSuccess) he = hipMemsetAsync(src.p, 0, sizeof(uint64_t) * kWords, d->stream);
        int rc2 = 0;
        for (int i = 0; i < 16 && he == hipSuccess && rc2 == 0; ++i)
            rc2 = d->AllGather(src.p, dst.p, kWords, /*ncclUint64*/ 5, d->comm, d->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(d->stream);
        src.release(); dst.release();
        if (rc2 != 0) return fail(QADC_E_HIP, std::string("ncclAllGather: ") + (d->GetErrorString ? d->GetErrorString(rc2) : "error"));
        HIPCHECK(he);
    }
    idx->dist = g.d.release();
    return QADC_OK;
}

extern "C++" {
namespace {
// Measurement aid (qadc_dist_init_loopback): ONE rank stands in for a whole world — its block fills every slot of the
// gather, so the merge replays `world` ranks' worth of entries while only this rank's shard is scanned.
int loopback_allgather(void* ctx, const void* d_send, void* d_recv, uint64_t bytes, void* st) {
    const int world = (int)reinterpret_cast<intptr_t>(ctx);
    return launch_replicate_block(d_send, d_recv, (size_t)(bytes / 8), world, static_cast<hipStream_t>(st)) == hipSuccess ? 0 : 1;
}
}  // namespace
}  // extern "C++"

int qadc_dist_init_loopback(qadc_index* idx, int rank, int world) {
    return qadc_dist_init_transport(idx, rank, world, loopback_allgather, reinterpret_cast<void*>((intptr_t)world));
}

int qadc_dist_init_transport(qadc_index* idx, int rank, int world, qadc_allgather_fn fn, void* ctx) {
    if (!fn) return fail(QADC_E_ARG, "the all-gather callback is null");
    if (int rc = dist_init_checks(idx, rank, world)) return rc;
    DistGuard g;
    g.d.reset(new DistState());
    DistState* d = g.d.get();
    d->user_fn = fn;
    d->user_ctx = ctx;
    d->rank = rank;
    d->world = world;
    // the merge's streams were created with the index (qadc_index_create: one fixed order, a hardware queue each) — the
    // collectives on the highest priority, interleave + replay at the scans' own lowest one: at normal priority their waves held
    // up the short launches between two batches' scans (one of 8 ranks, C5 shape 1.18 -> 1.10 ms per batch, C3 0.70 -> 0.60)
    d->stream = idx->coll_stream;
    for (int i = 0; i < kMergeStreams; ++i) d->merge_stream[i] = idx->merge_streams[i];
    idx->dist = g.d.release();
    return QADC_OK;
}

int qadc_dist_merge_blocks(int device_id, int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                           uint32_t* keys, int8_t* values, int32_t* sizes) {
    if (world < 1 || world > 16 || nq <= 0 || ma <= 0 || R <= 0 || (uint32_t)R > replay_wave_max_R() ||
        (size_t)ma * world > dist_interleave_max_cells() || !gathered || !sizes)
        return fail(QADC_E_ARG, "bad arguments");
    HIPCHECK(hipSetDevice(device_id));
    DevBuf<uint64_t> d_g, d_h, d_off, d_m;
    DevBuf<uint32_t> d_s, d_c;
    HIPCHECK(d_g.ensure((size_t)block_words * world));
    HIPCHECK(d_h.ensure((size_t)nq * R));
    HIPCHECK(d_s.ensure(nq));
    HIPCHECK(d_off.ensure(nq));
    HIPCHECK(d_c.ensure(2 * (size_t)nq));
    HIPCHECK(d_m.ensure((size_t)block_words * world));
    HIPCHECK(hipMemcpy(d_g.p, gathered, sizeof(uint64_t) * (size_t)block_words * world, hipMemcpyHostToDevice));
    DevBuf<uint32_t> d_st;
    HIPCHECK(d_st.ensure(4));
    HIPCHECK(launch_dist_merge(d_g.p, (size_t)block_words, world, nq, ma, (uint32_t)R, d_off.p, d_c.p, d_c.p + nq, d_m.p, d_h.p, d_s.p,
                               nullptr, d_st.p));
    uint32_t st_words[4] = {0, 0, 0, 0};
    HIPCHECK(hipMemcpy(st_words, d_st.p, sizeof(st_words), hipMemcpyDeviceToHost));
    d_st.release();
    if (st_words[2]) {
        d_g.release(); d_h.release(); d_s.release(); d_off.release(); d_c.release(); d_m.release();
        return fail(QADC_E_STATE, "multi-GPU merge: a rank's push stream is not grouped by assign slot");
    }
    std::vector<uint64_t> hv((size_t)nq * R);
    std::vector<uint32_t> hs(nq);
    HIPCHECK(hipMemcpy(hv.data(), d_h.p, sizeof(uint64_t) * hv.size(), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hs.data(), d_s.p, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost));
    d_g.release(); d_h.release(); d_s.release(); d_off.release(); d_c.release(); d_m.release();
    for (int q = 0; q < nq; ++q) {
        sizes[q] = hs[q] == 0xffffffffu ? -1 : (int32_t)hs[q];
        for (uint32_t i = 0; hs[q] != 0xffffffffu && i < hs[q]; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[(size_t)q * R + i];
            if (values) values[(size_t)q * R + i] = (int8_t)(hv[(size_t)q * R + i] >> 32);
        }
    }
    return QADC_OK;
}

int qadc_dist_merge_blocks_host(int world, int nq, int ma, int R, const uint64_t* gathered, uint64_t block_words,
                                uint32_t* keys, int8_t* values, int32_t* sizes) {
    if (world < 1 || world > 16 || nq <= 0 || ma <= 0 || R <= 0 || !gathered || !sizes) return fail(QADC_E_ARG, "bad arguments");
    // what the ranks of qadc_dist_collect do between their two all-gathers, rank by rank: every rank replays its share,
    // the second gather concatenates the shares ([rank][ceil(nq / world)][R + 1]), every rank reads all heaps back
    const int per = (nq + world - 1) / world;
    const size_t hw = (size_t)R + 1;
    std::vector<uint64_t> all((size_t)world * per * hw, 0);
    for (int r = 0; r < world; ++r)
        replay_my_share(gathered, (size_t)block_words, world, r, nq, ma, R, nullptr, all.data() + (size_t)r * per * hw, nullptr);
    for (int q = 0; q < nq; ++q) {
        const uint64_t* o = all.data() + ((size_t)(q % world) * per + q / world) * hw;
        sizes[q] = (int32_t)o[R];
        for (int i = 0; i < sizes[q]; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)o[i];
            if (values) values[(size_t)q * R + i] = (int8_t)(o[i] >> 32);
        }
    }
    return QADC_OK;
}

int qadc_dist_shutdown(qadc_index* idx) {
    if (!idx || !idx->dist) return QADC_OK;
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    if (idx->wgq_stream) (void)hipStreamSynchronize(idx->wgq_stream);
    DistState* d = idx->dist;
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    for (hipStream_t m : d->merge_stream) if (m) (void)hipStreamSynchronize(m);
    if (d->comm && d->CommDestroy) (void)d->CommDestroy(d->comm);        // (the streams stay: they belong to the index)
    d->d_fix.release(); d->h_fix.release(); d->d_moff.release(); d->d_merged.release(); d->d_mcnt.release();
    for (auto& ds : d->slot) ds.release();
    d->d_block.release(); d->d_gathered.release(); d->d_src.release(); d->h_src.release(); d->d_extra.release(); d->d_extra_all.release();
    d->h_extra.release(); d->h_out.release(); d->h_hdr.release(); d->h_extra_all.release();
    d->h_gathered.release(); d->h_myheaps.release(); d->d_myheaps.release(); d->d_allheaps.release(); d->h_allheaps.release();
    delete d;
    idx->dist = nullptr;
    return QADC_OK;
}

int qadc_dist_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                      const float* extra, int extra_n, float* extra_out) {
    if (!idx || !idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
    if (slot < 0 || slot >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (extra_n < 0 || (extra_n && (!extra || !extra_out))) return fail(QADC_E_ARG, "extra payload buffers missing");
    DistState& d = *idx->dist;
    Slot& s = idx->slot[slot];
    // caller errors — the same on every rank of a well-formed program — return before any rank enters the collective
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no batch");
    if (!s.dist_batch) return fail(QADC_E_STATE, "the batch was submitted before qadc_dist_init");
    if (int rc = use_device(idx)) return rc;
    const int nq = s.nq, R = s.R, world = d.world;
    // A failure of THIS rank's batch (candidate buffers that keep overflowing, a HIP error while re-running it) must not
    // leave the other ranks blocked in the gather: the rank still contributes a block, with bit7 set in every header, and
    // all ranks return the error after the gather.
    DistSlot& ds = d.slot[slot];
    if (ds.pending)
        if (int rc = flush_merges(idx, ds.seq)) return rc;
    if (ds.pending_share)                                     // (the batch is being collected before a later merge issued its share gather)
        if (int rc = flush_shares(idx, ds.seq)) return rc;
    const bool was_enqueued = ds.enqueued;
    if (was_enqueued) {                                       // the merge ran behind the scan: wait for all of it
        HIPCHECK(hipEventSynchronize(ds.ev_done));
        ds.enqueued = false;
    }
    int local_rc = collect_common(idx, slot, /*need_stream=*/false, /*from_dist=*/true);
    std::string local_err = local_rc ? g_err : std::string();
    if (was_enqueued) {
        const uint32_t* h_sz = reinterpret_cast<const uint32_t*>(ds.h_out.p + sizeof(uint64_t) * (size_t)R * nq);
        const uint32_t bad = h_sz[nq], need = h_sz[nq + 1];
        if (!bad && h_sz[nq + 2])                             // (raised by this rank's dist_interleave_kernel; every collective of the batch is behind us)
            return fail(QADC_E_STATE, "multi-GPU merge: a rank's push stream is not grouped by assign slot");
        if (!bad) {
            // every rank saw clean headers: the heaps are final (a local failure of collect_common concerns this rank only — it
            // is returned AFTER the payload gather below, which the other ranks enter as well)
            idx->prof.dist_async_collects++;
            std::vector<int32_t> st_async;
            int32_t* stp = status;
            if (!stp) { st_async.resize(nq); stp = st_async.data(); }
            if (!local_rc) finish_float_outputs(idx, s, stp, nullptr, nullptr);
            const uint64_t* hh = reinterpret_cast<const uint64_t*>(ds.h_out.p);
            for (int q = 0; q < nq && !local_rc; ++q) {
                uint32_t sz = h_sz[q];
                if (stp[q] || sz == 0xffffffffu) sz = 0;
                if (sizes) sizes[q] = (int32_t)sz;
                const uint64_t* hv = hh + (size_t)q * R;
                for (uint32_t i = 0; i < sz; ++i) {
                    if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[i];
                    if (values) values[(size_t)q * R + i] = (int8_t)(hv[i] >> 32);
                }
            }
            if (extra_n) {                                    // (a payload on such a batch travels by a small gather of its own)
                const size_t w = ((size_t)extra_n + 1) / 2;
                HIPCHECK(d.h_extra.ensure(2 * w));
                HIPCHECK(d.d_extra.ensure(2 * w));
                HIPCHECK(d.h_extra_all.ensure(2 * w * world));
                HIPCHECK(d.d_extra_all.ensure(w * world));       // (not d_block: a later batch's merge may be using that)
                std::memcpy(d.h_extra.p, extra, sizeof(float) * extra_n);
                HIPCHECK(hipMemcpyAsync(d.d_extra.p, d.h_extra.p, sizeof(uint64_t) * w, hipMemcpyHostToDevice, d.stream));
                std::string gerr2;
                if (d.gather(d.d_extra.p, d.d_extra_all.p, w, d.stream, gerr2)) return fail(QADC_E_HIP, gerr2);
                HIPCHECK(hipMemcpy2DAsync(d.h_extra_all.p, sizeof(float) * extra_n, d.d_extra_all.p, sizeof(uint64_t) * w,
                                          sizeof(float) * extra_n, world, hipMemcpyDeviceToHost, d.stream));
                HIPCHECK(hipStreamSynchronize(d.stream));
                std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
            }
            if (local_rc) return fail(local_rc, local_err);   // (every collective of this call is behind us: the peers are not left waiting)
            return QADC_OK;
        }
        // Some rank's stream overflowed its region or the gather block (identical verdict on every rank: they read the same
        // headers): the merge is redone below, at collect time, after collect_common re-ran what had to be re-run.
        if ((bad & 64u) && need + need / 8 > d.cap_entries && need < (1ull << 31))
            d.cap_entries = (uint32_t)(((uint64_t)need + need / 8 + 4095) / 4096 * 4096);
        idx->prof.regrows++;
    }
    if (d.inject_failure && !local_rc) {                      // test hook (option "dist_inject_failure")
        d.inject_failure = 0;
        local_rc = QADC_E_STATE;
        local_err = "injected failure (test hook)";
    }
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(nq);
        status = st_local.data();
    }
    if (!local_rc) finish_float_outputs(idx, s, status, nullptr, nullptr);
    // where this rank's ordered streams lie in device memory
    HIPCHECK(d.h_src.ensure(3 * (size_t)nq));
    HIPCHECK(d.d_src.ensure(3 * (size_t)nq));
    uint64_t fix_total = 0;
    for (int q = 0; q < nq && !local_rc; ++q) {
        const uint32_t fl = s.h_qout[q].flags;
        if (!(fl & 4u) && !(fl & 1u)) fix_total += s.out_off[q + 1] - s.out_off[q];   // ordered by collect_common on the host
    }
    if (fix_total >= (1ull << 31)) {                         // (reported through the gathered headers like any local failure: this rank
        local_rc = QADC_E_CAPACITY;                          //  must not leave before the gather the others are entering)
        local_err = "host-ordered streams exceed 2^31 entries";
        fix_total = 0;
    }
    if (fix_total) {
        // (an allocation that fails HERE must not make this rank leave before the gather its peers are entering: it is a local
        // failure like any other — bit7 in every header of this rank's block, the error returned after the gather)
        hipError_t fe = d.h_fix.ensure(fix_total);
        if (fe == hipSuccess) fe = d.d_fix.ensure(fix_total);
        if (fe != hipSuccess) {
            (void)hipGetLastError();
            local_rc = QADC_E_HIP;
            local_err = std::string("side buffer of the host-ordered streams: ") + hipGetErrorString(fe);
            fix_total = 0;
        }
    }
    uint64_t fix_off = 0;
    for (int q = 0; q < nq; ++q) {
        if (local_rc) {
            d.h_src.p[q] = 0;
            d.h_src.p[nq + q] = 0;
            d.h_src.p[2 * nq + q] = 128u;
            continue;
        }
        const QueryOut& qs = s.h_qout[q];
        const bool ordered = (qs.flags & 4u) != 0;
        d.h_src.p[q] = qs.out_off;
        d.h_src.p[nq + q] = ordered ? qs.count + qs.reps : 0u;
        d.h_src.p[2 * nq + q] = (qs.flags & 0x3fu) | (s.group_fell_back ? 512u : 0u);   // bit9: see the strike count below
        if (!ordered && !(qs.flags & 1u)) {
            // more candidates than the device sort takes (> 16384): collect_common sorted the query's raw region on the host;
            // its stream travels in the same gather from a side buffer
            const uint64_t n = s.out_off[q + 1] - s.out_off[q];
            std::memcpy(d.h_fix.p + fix_off, s.out_entries.data() + s.out_off[q], sizeof(uint64_t) * n);
            d.h_src.p[q] = (uint32_t)fix_off;
            d.h_src.p[nq + q] = (uint32_t)n;
            d.h_src.p[2 * nq + q] = (qs.flags & 0x3fu) | 4u | 256u | (s.group_fell_back ? 512u : 0u);
            fix_off += n;
        }
    }
    // Buffers are sized on the FIRST call for a payload of nq x R floats per rank (the sharded pre-scan's) whether or not
    // this call carries one: the pinned allocations and the larger gather blocks a first payload would otherwise need
    // cost milliseconds, and a pipeline's first batches typically come without payload.
    const size_t extra_room = std::max<size_t>((size_t)extra_n, (size_t)nq * (size_t)R);
    HIPCHECK(d.h_extra.ensure(extra_room));
    HIPCHECK(d.d_extra.ensure(extra_room));
    HIPCHECK(d.h_extra_all.ensure((size_t)world * extra_room));
    if (extra_n) std::memcpy(d.h_extra.p, extra, sizeof(float) * extra_n);
    const size_t heaps_bytes = (sizeof(uint64_t) * (size_t)R + sizeof(uint32_t)) * (size_t)nq;
    HIPCHECK(d.h_out.ensure(heaps_bytes + 32, hipHostMallocMapped | hipHostMallocCoherent));   // (+ the merge's status words)
    if (d.h_out.p != d.h_out_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d.d_out), d.h_out.p, 0));
        d.h_out_mapped = d.h_out.p;
    }
    HIPCHECK(d.h_hdr.ensure((size_t)world * nq * 4));
    hipStream_t st = d.stream;                               // the batch itself is complete (collect_common waited for it)
    uint64_t* h_heaps = reinterpret_cast<uint64_t*>(d.h_out.p);
    uint32_t* h_sizes = reinterpret_cast<uint32_t*>(d.h_out.p + sizeof(uint64_t) * (size_t)R * nq);
    // the device merge keeps a query's heap in the registers of one wave (R <= 320) and interleaves the ranks' streams with
    // ma x world counters in LDS; anything larger takes the host share
    const bool on_device = nq >= d.device_nq && (uint32_t)R <= replay_wave_max_R() &&
                           (size_t)s.ma * world <= dist_interleave_max_cells();
    std::string gerr;
    for (int attempt = 0;; ++attempt) {
        const size_t bw = dist_block_words(nq, d.cap_entries, (uint32_t)extra_n);
        const size_t bw_room = dist_block_words(nq, d.cap_entries, (uint32_t)extra_room);
        HIPCHECK(d.d_block.ensure(bw_room));
        HIPCHECK(d.d_gathered.ensure(bw_room * world));
        HIPCHECK(hipMemcpyAsync(d.d_src.p, d.h_src.p, sizeof(uint32_t) * 3 * nq, hipMemcpyHostToDevice, st));
        if (extra_n) HIPCHECK(hipMemcpyAsync(d.d_extra.p, d.h_extra.p, sizeof(float) * extra_n, hipMemcpyHostToDevice, st));
        if (fix_total) HIPCHECK(hipMemcpyAsync(d.d_fix.p, d.h_fix.p, sizeof(uint64_t) * fix_total, hipMemcpyHostToDevice, st));
        HIPCHECK(launch_dist_pack(d.d_src.p, d.d_src.p + nq, d.d_src.p + 2 * nq, nq, s.d_stream.p, d.d_fix.p, d.cap_entries,
                                  extra_n ? d.d_extra.p : nullptr, (uint32_t)extra_n, d.d_block.p, st));
        if (d.gather(d.d_block.p, d.d_gathered.p, bw, st, gerr)) return fail(QADC_E_HIP, gerr);
        if (on_device) {
            HIPCHECK(d.d_moff.ensure(nq));
            HIPCHECK(d.d_mcnt.ensure(2 * (size_t)nq));
            HIPCHECK(d.d_merged.ensure((size_t)d.cap_entries * world));
            HIPCHECK(launch_dist_merge(d.d_gathered.p, bw, world, nq, s.ma, (uint32_t)R, d.d_moff.p, d.d_mcnt.p, d.d_mcnt.p + nq,
                                       d.d_merged.p, reinterpret_cast<uint64_t*>(d.d_out),
                                       reinterpret_cast<uint32_t*>(d.d_out + sizeof(uint64_t) * (size_t)R * nq), st,
                                       reinterpret_cast<uint32_t*>(d.d_out + sizeof(uint64_t) * (size_t)R * nq) + nq));
        } else {
            HIPCHECK(d.h_gathered.ensure(bw_room * world));
            HIPCHECK(hipMemcpyAsync(d.h_gathered.p, d.d_gathered.p, sizeof(uint64_t) * bw * world, hipMemcpyDeviceToHost, st));
        }
        // every rank's header (to size a retry identically everywhere) and extra payload come back with the heaps
        HIPCHECK(hipMemcpy2DAsync(d.h_hdr.p, sizeof(uint32_t) * 4 * nq, d.d_gathered.p, sizeof(uint64_t) * bw,
                                  sizeof(uint32_t) * 4 * nq, world, hipMemcpyDeviceToHost, st));
        if (extra_n)
            HIPCHECK(hipMemcpy2DAsync(d.h_extra_all.p, sizeof(float) * extra_n, d.d_gathered.p + 2 * (size_t)nq + d.cap_entries,
                                      sizeof(uint64_t) * bw, sizeof(float) * extra_n, world, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        uint64_t need = 0;
        bool overflow = false, unordered = false, struck = false;
        int failed_rank = -1;
        for (int g = 0; g < world; ++g) {
            uint64_t tot = 0;
            for (int q = 0; q < nq; ++q) {
                const uint32_t* h = d.h_hdr.p + ((size_t)g * nq + q) * 4;
                tot += h[1];
                struck |= (h[2] & 512u) != 0;
                overflow |= (h[2] & 64u) != 0;
                if ((h[2] & 128u) && failed_rank < 0) failed_rank = g;
                unordered |= !(h[2] & 4u) && !(h[2] & 1u) && !(h[2] & 128u);
            }
            need = std::max(need, tot);
        }
        // (every rank reads the same headers: the same branch is taken everywhere, no rank stays behind in a collective)
        // Some rank's partition-major second phase overflowed on this batch (bit9): the strike is counted HERE, from the gathered
        // verdict, so that group.strikes — which decides whether later batches shard their front, i.e. issue one more
        // all-gather — is the same number on every rank (counted locally, the ranks' collectives diverged)
        if (struck && attempt == 0) idx->group.strikes++;
        if (failed_rank >= 0) {
            if (extra_n) std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
            return fail(local_rc ? local_rc : QADC_E_STATE,
                        local_rc ? local_err : "rank " + std::to_string(failed_rank) + " failed before the gather (see its own error)");
        }
        if (unordered) return fail(QADC_E_STATE, "a rank shipped a query without ordering it");
        if (overflow) {
            if (attempt >= 2 || need >= (1ull << 31)) return fail(QADC_E_CAPACITY, "gather block overflow persists");
            d.cap_entries = (uint32_t)((need + need / 8 + 4095) / 4096 * 4096);   // the same on every rank: they all saw the same headers
            idx->prof.regrows++;
            continue;
        }
        if (on_device) {
            if (h_sizes[nq + 2]) return fail(QADC_E_STATE, "multi-GPU merge: a rank's push stream is not grouped by assign slot");
            break;
        }
        // ---- few queries (or R > 288): replay my share on the host (global scan order: assign slot, rank, position), share the heaps ----
        const int per = (nq + world - 1) / world;
        const size_t hw = (size_t)R + 1;                       // words per query in the heap exchange
        HIPCHECK(d.h_myheaps.ensure((size_t)per * hw));
        HIPCHECK(d.d_myheaps.ensure((size_t)per * hw));
        HIPCHECK(d.d_allheaps.ensure((size_t)per * hw * world));
        HIPCHECK(d.h_allheaps.ensure((size_t)per * hw * world));
        std::memset(d.h_myheaps.p, 0, sizeof(uint64_t) * (size_t)per * hw);
        {
            ScopedMs timer(idx->prof.host_heap_ms);
            replay_my_share(d.h_gathered.p, bw, world, d.rank, nq, s.ma, R, status, d.h_myheaps.p, &idx->pool);
        }
        HIPCHECK(hipMemcpyAsync(d.d_myheaps.p, d.h_myheaps.p, sizeof(uint64_t) * (size_t)per * hw, hipMemcpyHostToDevice, st));
        if (d.gather(d.d_myheaps.p, d.d_allheaps.p, (size_t)per * hw, st, gerr)) return fail(QADC_E_HIP, gerr);
        HIPCHECK(hipMemcpyAsync(d.h_allheaps.p, d.d_allheaps.p, sizeof(uint64_t) * (size_t)per * hw * world, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        for (int q = 0; q < nq; ++q) {
            const uint64_t* o = d.h_allheaps.p + ((size_t)(q % world) * per + q / world) * hw;
            h_sizes[q] = (uint32_t)o[R];
            std::memcpy(h_heaps + (size_t)q * R, o, sizeof(uint64_t) * (size_t)R);
        }
        break;
    }
    for (int q = 0; q < nq; ++q) {
        uint32_t sz = h_sizes[q];
        if (status[q] || sz == 0xffffffffu) sz = 0;
        if (sizes) sizes[q] = (int32_t)sz;
        const uint64_t* hv = h_heaps + (size_t)q * R;
        for (uint32_t i = 0; i < sz; ++i) {
            if (keys) keys[(size_t)q * R + i] = (uint32_t)hv[i];
            if (values) values[(size_t)q * R + i] = (int8_t)(hv[i] >> 32);
        }
    }
    if (extra_n) std::memcpy(extra_out, d.h_extra_all.p, sizeof(float) * (size_t)world * extra_n);
    return QADC_OK;
}

int qadc_place_partitions(int part_count, const uint32_t* sizes, int world, int32_t* owner_out) {
    if (part_count < 0 || world < 1 || (part_count && (!sizes || !owner_out))) return fail(QADC_E_ARG, "bad arguments");
    std::vector<int> order(part_count);
    for (int p = 0; p < part_count; ++p) order[p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] > sizes[b]; });
    std::vector<uint64_t> load(world, 0);
    for (int p : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        owner_out[p] = best;
        load[best] += sizes[p];
    }
    return QADC_OK;
}

}  // extern "C"
