// Host side of libqadc_hip.so: the C-ABI declared in include/qadc.h.
//
// Mirrors scanner_4 (db_query_4.cpp:73-310): prepare (upload partitions, starts sizes) and
// query_scan (float pre-scan -> qmax, qmin/clamp, int8 quantization, scan of every probed
// partition in assign order into one heap).  All per-query arithmetic runs on the GPU in one
// stream-ordered chain (no host round trip between pre-scan, quantizer, scan and the ordering of the
// candidates); the host only plans the work items and replays the returned, already ordered candidate
// streams through a heap with the reference's push semantics (host/qadc_heap.hpp).
//
// The product path never touches oracle/: if the HIP runtime or the GPU is missing every entry
// point fails loudly with QADC_E_HIP.
#include "qadc_host.h"

using namespace qadc;
using namespace qadc::host;

namespace qadc {
namespace host {
thread_local std::string g_err;
}
}

namespace {

// Level boundaries in the concatenated scan position space of one query.
void level_bounds(const qadc_index* idx, uint64_t* L) {
    L[0] = 0;
    uint64_t b = std::max<uint64_t>(idx->level_base, 16);
    for (int k = 1; k < kMaxLevels; ++k) {
        L[k] = b;
        b = (b > (UINT64_MAX >> 8)) ? UINT64_MAX : b * std::max<uint64_t>(idx->level_growth, 2);
    }
    L[kMaxLevels] = UINT64_MAX;
}

// What the planner hands to the launcher: the work items of a batch in upload order.
struct BatchPlan {
    std::vector<ScanItem> all_items;             // runs, grouped by bound level (Slot::launches indexes into it)
    std::vector<StartItem> sitems_a, sitems_b;   // pre-scan: phase A = unfiltered sample, phase B = filtered remainder
    std::vector<uint32_t> fc_init;               // per query: {sample values, capacity} of its pre-scan buffer
    uint64_t fc_stride = 1;
};

// Host planning of one batch: cuts every query's scan order into bound levels, emits the runs (ScanItem) and the
// pre-scan items (StartItem), and decides kernel and grid per level launch (Slot::launches).  No GPU calls.
int plan_batch(qadc_index* idx, Slot& s, BatchPlan& plan) {
    const int M = idx->M, cs = idx->cs, nq = s.nq, ma = s.ma;
    const uint32_t cpl = 16 / cs;
    uint64_t L[kMaxLevels + 1];
    level_bounds(idx, L);
    std::vector<std::vector<ScanItem>> per_level(kMaxLevels);
    const int k0 = (s.mode != 1 && idx->head_level > 0) ? idx->head_level : 0;   // levels < k0 belong to the head launch
    s.head_codes = k0 ? L[k0] : 0;
    std::vector<StartItem>& sitems_a = plan.sitems_a;
    std::vector<StartItem>& sitems_b = plan.sitems_b;
    std::vector<uint32_t>& fc_init = plan.fc_init;
    uint64_t& fc_stride = plan.fc_stride;
    fc_init.assign(2 * (size_t)nq, 0);
    fc_stride = 1;
    s.start_codes = 0;
    for (int q = 0; q < nq; ++q) {
        uint64_t c = 0;
        uint64_t stotal = 0;
        // the starts of a partition this call pre-scans: all of them, or (mode 1) this rank's slice, cut at
        // multiples of 16 codes so that every slice starts on a 16-byte boundary; (mode 2) none
        auto starts_range = [&](const Part& pt, uint64_t& lo, uint64_t& len) {
            lo = 0;
            len = s.mode == 2 ? 0 : pt.start_n;
            if (s.mode == 1 && s.pre_nslices > 1) {
                lo = ((uint64_t)pt.start_n * s.pre_slice / s.pre_nslices) & ~15ull;
                const uint64_t hi = s.pre_slice + 1 == s.pre_nslices
                                        ? pt.start_n : (((uint64_t)pt.start_n * (s.pre_slice + 1) / s.pre_nslices) & ~15ull);
                len = hi > lo ? hi - lo : 0;
            }
        };
        if (s.float_path)
            for (int a = 0; a < ma; ++a) {
                const int p = s.assign[(size_t)q * ma + a];
                if (p >= 0 && p < (int)idx->parts.size()) {
                    uint64_t lo, len;
                    starts_range(idx->parts[p], lo, len);
                    stotal += len;
                }
            }
        // two-phase pre-scan only pays (and is only needed) when the starts are many
        uint64_t sample = (s.full_prescan || stotal <= 2ull * idx->prescan_sample) ? stotal : idx->prescan_sample;
        uint64_t soff = 0;
        for (int a = 0; a < ma; ++a) {
            const int p = s.assign[(size_t)q * ma + a];
            if (p < 0 || p >= (int)idx->parts.size())
                return fail(QADC_E_ARG, "assign[] names a partition that does not exist");
            const Part& pt = idx->parts[p];
            if (pt.global_n == 0) continue;  // empty partition: db_query_4.cpp:291-293
            uint64_t slo = 0, slen = 0;
            if (s.float_path) starts_range(pt, slo, slen);
            if (slen) {
                const uint8_t* sc = (pt.d_starts ? pt.d_starts : pt.d_codes) + slo * cs;
                const uint64_t in_a = soff < sample ? std::min<uint64_t>(slen, sample - soff) : 0;
                StartItem si;
                si.table = (uint32_t)((size_t)q * ma + a);
                si.query = (uint32_t)q;
                if (in_a) {
                    si.codes = sc;
                    si.n = (uint32_t)in_a;
                    si.out_off = (uint32_t)soff;
                    si.filter = 0;
                    sitems_a.push_back(si);
                }
                if (in_a < slen) {
                    si.codes = sc + in_a * cs;
                    si.n = (uint32_t)(slen - in_a);
                    si.out_off = 0;
                    si.filter = 1;
                    sitems_b.push_back(si);
                }
                soff += slen;
                s.start_codes += slen;
            }
            if (pt.n == 0 || s.mode == 1) continue;   // no codes of the partition here (only its starts replica) / pre-scan only
            uint64_t prev = 0;
            for (int k = 0; k < kMaxLevels && prev < pt.n; ++k) {
                uint64_t cut = pt.n;
                if (L[k + 1] < c + pt.n) {
                    cut = L[k + 1] > c ? L[k + 1] - c : 0;
                    cut -= cut % cpl;  // keep every run 16-byte aligned
                }
                if (cut <= prev) continue;
                if (k < k0) {                                 // scanned by the head launch (same cut: scan_query_kernel, HEAD)
                    prev = cut;
                    continue;
                }
                // runs longer than 2^31 codes are cut so that 32-bit vector indices cannot wrap
                for (uint64_t b0 = prev; b0 < cut;) {
                    const uint64_t len = std::min<uint64_t>(cut - b0, 1ull << 31);
                    ScanItem it;
                    it.codes = pt.d_codes + b0 * cs;
                    it.labels = pt.d_labels;
                    it.n = (uint32_t)len;
                    it.pos0 = (uint32_t)b0;
                    it.key_base = pt.key_base + pt.first_pos;
                    it.table = (uint32_t)((size_t)q * ma + a);
                    it.query = (uint32_t)q;
                    it.order = ((uint32_t)k << 16) | (uint32_t)a;
                    // padding-lane replay of the partition's last code (simd_layout.hpp:46-50, simd_scan.hpp:67)
                    it.dup_pos = (pt.first_pos + pt.n == pt.global_n) ? pt.n - 1u : 0xffffffffu;
                    it.dup_reps = (16u - pt.global_n % 16u) % 16u;
                    per_level[k].push_back(it);
                    b0 += len;
                }
                prev = cut;
            }
            c += pt.n;
        }
        // survivors of the filter: expected R * stotal / sample; 16x head-room, the overflow flag catches the rest
        uint64_t cap = sample;
        if (sample < stotal)
            cap += std::min<uint64_t>(stotal - sample, std::max<uint64_t>(16ull * s.R * ((stotal + sample - 1) / sample), 4096));
        if (s.mode == 2) sample = cap = s.inj_n;              // the gathered values are the whole "pre-scan output"
        fc_init[2 * q] = (uint32_t)sample;
        fc_init[2 * q + 1] = (uint32_t)cap;
        fc_stride = std::max<uint64_t>(fc_stride, cap);
    }
    size_t nitems = 0;
    for (auto& v : per_level) nitems += v.size();
    std::vector<ScanItem>& all_items = plan.all_items;
    all_items.assign(nitems, ScanItem());
    s.launches.clear();
    size_t off = 0;
    const int wgs_cap = idx->wgs_per_item > 0 ? idx->wgs_per_item : (M == 16 ? 1024 : 512);   // (r02 sweep: 1024 reaches the streaming ceiling of the "probe" variant, 512 is 1.6 % below)
    for (int k = 0; k < kMaxLevels; ++k) {
        if (per_level[k].empty()) continue;
        // one launch for the short runs of the level, one for the long ones
        for (int small = 1; small >= 0; --small) {
            uint64_t maxn = 0, codes = 0;
            size_t cnt = 0;
            bool same = true;
            for (auto& it : per_level[k]) {
                if ((it.n < idx->small_run) != (small == 1)) continue;
                if (cnt) {
                    const ScanItem& f = all_items[off];
                    same = same && it.codes == f.codes && it.n == f.n && it.pos0 == f.pos0 && it.labels == f.labels &&
                           it.key_base == f.key_base && it.dup_pos == f.dup_pos && it.dup_reps == f.dup_reps;
                }
                all_items[off + cnt++] = it;
                maxn = std::max<uint64_t>(maxn, it.n);
                codes += it.n;
            }
            if (!cnt) continue;
            const uint64_t nvec = (maxn + cpl - 1) / cpl;
            LevelLaunch ll;
            ll.first = off;
            ll.nitems = (int)cnt;
            ll.small = small == 1;
            ll.maxn = maxn;
            ll.early = false;
            ll.shared = !ll.small && same && cnt >= 2 && idx->share_variant != 0;
            ll.mq = ll.shared && idx->mq;
            if (ll.mq) {
                // 8 queries per pass (scan_i8_mq_kernel): 256-thread workgroups, ~64 Ki codes each, groups of 8
                // queries as L2-sharing siblings
                const uint64_t tiles = std::max<uint64_t>((nvec + 255) / 256, 1);
                uint64_t w = idx->wgs_per_item > 0 ? (uint64_t)idx->wgs_per_item
                                                   : (maxn + kMqCodesPerWg - 1) / kMqCodesPerWg;
                const uint64_t ngroups = (cnt + 7) / 8;
                w = std::max<uint64_t>(w, (kMqMinWgs + ngroups - 1) / ngroups);   // >= 2 rounds of the 2048 resident workgroups
                w = std::min<uint64_t>(std::min<uint64_t>(w, 65536), std::max<uint64_t>(tiles / kMqMinTiles, 1));
                if (w >= 8) w &= ~7ull;
                ll.wgs = (int)w;
            } else if (ll.shared) {
                // Queries of a batch over the same codes (flat database; IVF queries probing the same cell): the
                // sibling-major launch makes them share every tile through one XCD's L2, so the codes cross the
                // HBM interface about once per LAUNCH, not once per query, and the launch is bound by the LDS
                // lookup rate instead.  Workgroups per run: ~2M codes each (amortises the table build, leaves the
                // dispatcher room to balance), a multiple of 8 so that the XCD decode applies.
                uint64_t w = idx->wgs_per_item > 0 ? (uint64_t)idx->wgs_per_item
                                                   : (maxn + kShareCodesPerWg - 1) / kShareCodesPerWg;
                w = std::min<uint64_t>(std::max<uint64_t>(w, 64), 512);
                w = std::min<uint64_t>(w, std::max<uint64_t>((nvec + 4095) / 4096, 1));
                if (w >= 8) w &= ~7ull;
                ll.wgs = (int)w;
            } else if (ll.small) {
                // enough workgroups to fill the chip, but no more: each one pays a table build + bound fetch
                const uint64_t want = std::max<uint64_t>(1, 4096 / cnt);
                ll.wgs = (int)std::min<uint64_t>(std::max<uint64_t>((nvec + kSmallVecPerWg - 1) / kSmallVecPerWg, 1), want);
            }
            else {
                // each streaming workgroup builds a 64-128 KiB table: with many runs in the launch, give every
                // workgroup more tiles instead of more workgroups per run
                const uint64_t want = std::max<uint64_t>(1, 8192 / cnt);
                ll.wgs = (int)std::min<uint64_t>(std::max<uint64_t>((nvec + 4095) / 4096, 1), std::min<uint64_t>(wgs_cap, want));
            }
            ll.codes = codes;
            s.launches.push_back(ll);
            off += cnt;
        }
    }

    return QADC_OK;
}

}  // namespace

namespace qadc {
namespace host {

int use_device(const qadc_index* idx) {
    HIPCHECK(hipSetDevice(idx->device));
    return QADC_OK;
}

hipError_t prof_event(Slot& s, hipStream_t st) {
    if (s.prof_used == s.prof_ev.size()) {
        hipEvent_t e;
        hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess) return r;
        s.prof_ev.push_back(e);
    }
    return hipEventRecord(s.prof_ev[s.prof_used++], st);
}

// Plans the batch in slot s and enqueues all of its GPU work (front, levels, ordering, optional device replay).
int plan_and_launch(qadc_index* idx, Slot& s) {
    if (s.wgq) return launch_wgq_batch(idx, s);
    ScopedMs timer(idx->prof.host_plan_ms);
    const int M = idx->M, cs = idx->cs, nq = s.nq, ma = s.ma;
    // (tried, round 4: consecutive level-path batches alternating between two scan streams on the scan pipe, so that the first
    // workgroups of batch s+1 fill the tail of batch s's last level — 125M-code shard 1.10-1.15 -> 1.15-1.35 ms per step, 1B
    // 7.66 -> 7.72-7.76: the next batch's workgroups do not fill a tail, they compete with the current level for CUs)
    hipStream_t st = idx->stream;
    const size_t table_dim = (size_t)M * 16;
    BatchPlan plan;
    if (int rc = plan_batch(idx, s, plan)) return rc;
    const std::vector<ScanItem>& all_items = plan.all_items;
    const std::vector<StartItem>&sitems_a = plan.sitems_a, &sitems_b = plan.sitems_b;
    const std::vector<uint32_t>& fc_init = plan.fc_init;
    uint64_t fc_stride = plan.fc_stride;
    const size_t nitems = all_items.size();

    // ---- upload: ONE block, ONE copy ----------------------------------------------------------
    const size_t nt = (size_t)nq * ma * table_dim;
    const size_t na = sitems_a.size(), nb = sitems_b.size();
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t off_items = 0;
    const size_t off_sitems = align16(off_items + nitems * sizeof(ScanItem));
    const size_t off_init = align16(off_sitems + (na + nb) * sizeof(StartItem));
    const size_t off_tables = align16(off_init + fc_init.size() * sizeof(uint32_t));
    const size_t tables_bytes = s.float_path ? (s.device_tables ? 0 : nt * sizeof(float)) : (s.front_sharded ? 0 : nt);
    const size_t off_inj = align16(off_tables + tables_bytes);
    const size_t inj_bytes = s.mode == 2 ? sizeof(float) * (size_t)nq * s.inj_n : 0;
    const size_t off_hassign = align16(off_inj + inj_bytes);  // assign[] for the head launch (it walks the partition table itself)
    const size_t hassign_bytes = s.head_codes ? sizeof(int32_t) * (size_t)nq * ma : 0;
    const size_t in_bytes = align16(off_hassign + hassign_bytes);
    HIPCHECK(s.h_in.ensure(in_bytes));
    HIPCHECK(s.d_in.ensure(in_bytes));
    if (nitems) std::memcpy(s.h_in.p + off_items, all_items.data(), nitems * sizeof(ScanItem));
    if (na) std::memcpy(s.h_in.p + off_sitems, sitems_a.data(), na * sizeof(StartItem));
    if (nb) std::memcpy(s.h_in.p + off_sitems + na * sizeof(StartItem), sitems_b.data(), nb * sizeof(StartItem));
    std::memcpy(s.h_in.p + off_init, fc_init.data(), fc_init.size() * sizeof(uint32_t));
    if (s.float_path && !s.device_tables) std::memcpy(s.h_in.p + off_tables, s.tables, nt * sizeof(float));
    if (!s.float_path && !s.front_sharded) std::memcpy(s.h_in.p + off_tables, s.qtables_in.data(), nt);
    if (inj_bytes) std::memcpy(s.h_in.p + off_inj, s.inj_vals.data(), inj_bytes);
    if (hassign_bytes) std::memcpy(s.h_in.p + off_hassign, s.assign.data(), hassign_bytes);
    s.d_items = reinterpret_cast<ScanItem*>(s.d_in.p + off_items);
    s.d_sitems = reinterpret_cast<StartItem*>(s.d_in.p + off_sitems);
    s.d_fc_init = reinterpret_cast<uint32_t*>(s.d_in.p + off_init);
    s.d_ftables_in = reinterpret_cast<float*>(s.d_in.p + off_tables);

    // state block: [CandHeader, 64 B][QueryState[nq]]; result block: [QueryOut[nq]][u64 entries[out_cap]]
    const size_t state_bytes = 64 + sizeof(QueryState) * (size_t)nq;
    bool lone = s.mode != 1;                                    // nothing else in flight: a synchronous call (see launch_wgq_batch)
    for (int i = 0; i < kSlots; ++i) lone = lone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    lone = lone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    const int replay_from = lone ? std::max(idx->device_replay_nq, idx->device_replay_alone_nq) : idx->device_replay_nq;
    s.dev_replay = s.mode != 1 && idx->device_replay_nq > 0 && nq >= replay_from && s.R <= 4096;
    const size_t off_heaps = sizeof(QueryOut) * (size_t)nq + sizeof(uint64_t) * (size_t)s.out_cap;
    const size_t heaps_bytes = s.dev_replay ? (sizeof(uint64_t) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq : 0;
    const size_t result_bytes = std::max(off_heaps + heaps_bytes, (sizeof(float) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq);
    HIPCHECK(s.d_state.ensure(state_bytes));
    HIPCHECK(s.h_result.ensure(result_bytes, hipHostMallocMapped | hipHostMallocCoherent));
    if (s.h_result.p != s.h_result_mapped) {               // (looked up once per allocation: the call costs ~0.1 ms)
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_result_mapped), s.h_result.p, 0));
        s.h_result_mapped = s.h_result.p;
    }
    unsigned char* d_result = s.d_result_mapped;
    s.d_hdr = reinterpret_cast<CandHeader*>(s.d_state.p);
    s.d_qs = reinterpret_cast<QueryState*>(s.d_state.p + 64);
    s.d_qout = reinterpret_cast<QueryOut*>(d_result);
    s.d_entries = reinterpret_cast<uint64_t*>(d_result + sizeof(QueryOut) * (size_t)nq);
    s.h_qout = reinterpret_cast<QueryOut*>(s.h_result.p);
    s.h_entries = reinterpret_cast<uint64_t*>(s.h_result.p + sizeof(QueryOut) * (size_t)nq);
    s.h_heaps = reinterpret_cast<uint64_t*>(s.h_result.p + off_heaps);
    s.h_heap_sizes = reinterpret_cast<uint32_t*>(s.h_result.p + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
    s.dist_batch = idx->dist != nullptr;
    const bool dev_stream = s.dev_replay || s.dist_batch;          // the native multi-GPU merge gathers from device memory
    if (dev_stream) HIPCHECK(s.d_stream.ensure(s.out_cap));
    s.h_export = reinterpret_cast<float*>(s.h_result.p);
    s.h_export_flags = reinterpret_cast<uint32_t*>(s.h_result.p + sizeof(float) * (size_t)s.R * nq);
    HIPCHECK(s.d_cands.ensure((size_t)nq * s.cap_q));
    HIPCHECK(s.d_qtables.ensure(nt));
    // The front of the batch (state clear, table build, float pre-scan, selects, quantizer) depends on nothing the
    // previous batch produces: it runs on its own high-priority stream, under that batch's streaming launches, and
    // the first scan level waits for it.  Its short single-workgroup selects are pure latency; hidden this way they
    // stop being a fixed cost per batch (which is what limits strong scaling when the per-GPU shard gets small).
    // A batch submitted while nothing else is in flight (a synchronous call, the first batch of a pipeline) has
    // nothing to overlap with: it runs front, levels and ordering on ONE stream, which spares it three cross-stream
    // event hops (~15 us of a ~130 us single query).
    bool alone = s.mode != 1;
    for (int i = 0; i < kSlots; ++i) alone = alone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    alone = alone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    hipStream_t main_stream = st;
    if (!alone || s.mode == 1) st = idx->front_stream;
    HIPCHECK(hipMemsetAsync(s.d_state.p, 0, state_bytes, st));
    // The upload goes on the copy stream, where it depends on nothing (the slot's previous batch was collected),
    // and the main stream waits for it.  Issued on the main stream it would sit in the DMA engine's queue until the
    // PREVIOUS batch's kernels finish, and every other copy of the process (the caller's streams, the RCCL gather of
    // the multi-GPU merge) would queue behind it: copies wait in engine order, not stream order.
    if (alone) {
        HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, st));
    } else {
        HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, idx->copy_stream));
        if (!s.ev_up) HIPCHECK(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_up, idx->copy_stream));
        HIPCHECK(hipStreamWaitEvent(st, s.ev_up, 0));
    }
    s.prof_used = 0;

    s.d_qt = s.d_qtables.p;
    if (s.float_path) {
        float* d_ft = s.d_ftables_in;
        if (s.device_tables) {
            // residuals + float tables built on the GPU from the queries uploaded by search_submit
            HIPCHECK(s.d_ftables.ensure(nt));
            d_ft = s.d_ftables.p;
            HIPCHECK(hipStreamWaitEvent(st, s.ev_feed, 0));
            launch_build_tables(s.d_queries.p, idx->feed.K ? idx->feed.d_coarse.p : nullptr, s.d_assign.p, idx->feed.d_codebooks.p,
                                idx->feed.has_rotation ? idx->feed.d_rotation.p : nullptr, nq, ma, M, idx->feed.dim, table_expansion(idx, ma), idx->sum_mode, d_ft, st);
        }
        HIPCHECK(s.d_fc.ensure((size_t)nq * fc_stride));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        auto wgs_for = [](const std::vector<StartItem>& v) {
            uint32_t maxs = 0;
            for (auto& si : v) maxs = std::max(maxs, si.n);
            return (int)std::min<uint32_t>(std::max<uint32_t>((maxs + 4095) / 4096, 1), 512);
        };
        const int tda = (int)(ma * table_dim);
        // phase A: the sample, unfiltered -> its R-th smallest; phase B: the rest, keeping only values <= that.
        // The LAST select of the chain also quantizes the query's tables (QuantizerMAX) in the same workgroup.
        // every query pre-scans the same starts (flat database, or one shared probe): 8 queries per pass
        auto shared_items = [&](const std::vector<StartItem>& v) {
            if (v.size() < 2) return false;
            for (auto& si : v)
                if (si.codes != v[0].codes || si.n != v[0].n || si.out_off != v[0].out_off || si.filter != v[0].filter) return false;
            return true;
        };
        auto start_scan = [&](const std::vector<StartItem>& v, const StartItem* d_v) {
            if (shared_items(v))
                launch_start_scan_mq(M, idx->sum_mode, d_v, (int)v.size(), std::min(2 * wgs_for(v), 1024), d_ft, s.d_fc.p, fc_stride, s.d_fc_init,
                                     s.d_qs, st);
            else
                launch_start_scan_f32(M, idx->sum_mode, d_v, (int)v.size(), wgs_for(v), d_ft, s.d_fc.p, fc_stride, s.d_fc_init, s.d_qs, st);
        };
        if (na) start_scan(sitems_a, s.d_sitems);
        if (nb) {
            // (with a phase B the first select only has to bound the R-th smallest from above: 2 digit passes)
            launch_select_kth(s.d_fc.p, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 2, nullptr, nullptr, tda, 0, st);
            start_scan(sitems_b, s.d_sitems + na);
        }
        if (s.mode == 1) {
            // pre-scan only: the exact R-th smallest of this rank's slice, and the R smallest values themselves,
            // stored straight into the pinned result block; nothing else runs
            float* d_exp = reinterpret_cast<float*>(d_result);
            uint32_t* d_expf = reinterpret_cast<uint32_t*>(d_result + sizeof(float) * (size_t)s.R * nq);
            launch_select_kth(s.d_fc.p, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 4, nullptr, nullptr, tda, 0, st,
                              d_exp, d_expf);
            HIPCHECK(hipGetLastError());
            if (idx->profile) HIPCHECK(prof_event(s, st));
            if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_done, st));
            return QADC_OK;
        }
        const float* d_sel = s.d_fc.p;
        if (s.mode == 2) {
            // the ranks' gathered smallest values stand in for the pre-scan output
            d_sel = reinterpret_cast<const float*>(s.d_in.p + off_inj);
            fc_stride = s.inj_n;
            launch_prescan_minmax(d_sel, s.inj_n, nq, s.d_qs, st);
        }
        launch_select_kth(d_sel, fc_stride, s.d_fc_init, nq, (uint32_t)s.R, s.d_qs, 4, d_ft, s.d_qtables.p, tda,
                          idx->quant_mode, st);
        if (idx->profile) HIPCHECK(prof_event(s, st));
    } else {
        s.d_qt = s.front_sharded ? s.d_qtables.p                             // (a sharded-front batch redone here: the gathered tables)
                                 : reinterpret_cast<const int8_t*>(s.d_in.p + off_tables);     // caller's int8 tables, as uploaded
        if (idx->profile) { HIPCHECK(prof_event(s, st)); HIPCHECK(prof_event(s, st)); }
    }

    // ---- scan levels ------------------------------------------------------------------------
    // a database that fits the 256 MiB Infinity Cache is re-read from it by every query: keep the default
    // cache policy there; non-temporal loads only pay for lists that stream from HBM anyway
    uint64_t db_bytes = 0;
    for (auto& p : idx->parts) db_bytes += (uint64_t)p.n * cs;
    const int variant = db_bytes <= (200ull << 20) ? (idx->variant & ~4) : idx->variant;
    auto launch_level = [&](LevelLaunch& ll, hipStream_t str) {
        if (ll.small)
            launch_scan_i8_small(M, s.d_items + ll.first, ll.nitems, ll.wgs, s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p,
                                 s.cap_q, (uint32_t)s.R, str);
        else if (ll.mq)
            launch_scan_i8_mq(M, s.d_items + ll.first, ll.nitems, ll.wgs, s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p, s.cap_q,
                              (uint32_t)s.R, str, /*narrow=*/ll.nitems <= 4 ? 1 : 0);   // (8 queries per pass: the 8-seat
                                                                   // build, see scan_i8_mq_kernel; <= 4 runs: the build with the 4-seat body)
        else
            launch_scan_i8(M, ll.shared ? idx->share_variant : (variant & ~64), s.d_items + ll.first, ll.nitems, ll.wgs,
                           s.d_qt, s.d_qs, s.d_hdr, s.d_cands.p, s.cap_q, (uint32_t)s.R, str);
    };
    // The first levels of a batch are short launches in a dependent chain (each level's bound needs the previous
    // levels' candidates): latency, not work.  The head launch and the levels with short runs join the front — same
    // stream, after the quantizer — and so run under the previous batch's long levels instead of in front of this
    // batch's; only the long levels stay on the main stream.  Front-stream kernels only find room as workgroups of
    // the long launches retire, so a level with real work is slower there than on the main stream: runs of <= 2 Mi
    // codes go early (125M x 32 queries: 1.143 -> 1.10 ms per step; 1B x 32: -1 %), 8 Mi already costs more than it hides.
    auto launch_head = [&](hipStream_t str) -> int {
        // ONE launch scans the first head_codes codes of every query (bound levels 0..k0-1): every query's scan order
        // split over G workgroups that refresh their bound in LDS, instead of k0 dependent launches of a few
        // microseconds of work each.  It emits into the same candidate regions / level-0 histogram the levels use.
        QueryKernelArgs H{};
        H.parts = idx->d_partdesc.p;
        H.assign = reinterpret_cast<const int32_t*>(s.d_in.p + off_hassign);
        H.ma = ma;
        H.qtables = const_cast<int8_t*>(s.d_qt);
        H.R = (uint32_t)s.R;
        H.head_codes = s.head_codes;
        H.qstates = s.d_qs;
        H.cand_regions = s.d_cands.p;
        H.cand_cap = s.cap_q;
        H.hdr = s.d_hdr;
        H.nontemporal = 0;                                   // the queries of a batch share the head of a flat list through L2
        int G = std::min<int>(nq <= 2 ? idx->wgq_split : std::min(idx->wgq_split, kSplitBatch), std::max(1, 256 / nq));   // (as launch_wgq_batch)
        G = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)G, s.head_codes / 16384));
        H.G = G;
        HIPCHECK(launch_scan_query(M, nq, H, str));
        idx->prof.head_launches++;
        return QADC_OK;
    };
    size_t n_early = 0;
    uint64_t batch_codes = 0;
    for (auto& ll : s.launches) batch_codes += ll.codes;
    if (st != main_stream && (s.mode == 0 || s.mode == 2) && batch_codes >= kFrontMinBatch)
        while (n_early < s.launches.size() && s.launches[n_early].maxn <= idx->front_run_max) ++n_early;
    if (n_early == s.launches.size() && n_early) --n_early;        // the last level closes the batch on the main stream
    // the head precedes every level: with the early levels (or on request) it joins the front as well
    bool head_pending = s.head_codes != 0;
    if (head_pending && st != main_stream) {
        if (int rc = launch_head(st)) return rc;
        head_pending = false;
    }
    for (size_t li = 0; li < n_early; ++li) {
        s.launches[li].early = true;
        s.launches[li].ev = -1;
        launch_level(s.launches[li], st);
    }
    if (st != main_stream) {
        if (!s.ev_front) HIPCHECK(hipEventCreateWithFlags(&s.ev_front, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_front, st));
        st = main_stream;
        HIPCHECK(hipStreamWaitEvent(st, s.ev_front, 0));
    }
    if (head_pending)
        if (int rc = launch_head(st)) return rc;
    // HIP events cost ~10 us of stream time each: with profiling on, every run of consecutive streaming-kernel
    // launches (the roofline figure) shares ONE event pair; small-run launches are counted, not timed
    for (size_t li = n_early; li < s.launches.size(); ++li) {
        LevelLaunch& ll = s.launches[li];
        const bool timed = idx->profile && !ll.small;
        const bool group_start = timed && (li == n_early || s.launches[li - 1].small);
        const bool group_end = timed && (li + 1 == s.launches.size() || s.launches[li + 1].small);
        ll.ev = -1;
        if (group_start) { ll.ev = (int)s.prof_used; HIPCHECK(prof_event(s, st)); }
        launch_level(ll, st);
        if (group_end) HIPCHECK(prof_event(s, st));
    }
    // the ordering pass runs on a side stream: it only occupies nq CUs, and the main stream is free to start the
    // next batch's kernels meanwhile (it uses the other slot's buffers).  It stores [QueryOut[nq]][entries] straight
    // into the slot's pinned host block, so no device-to-host copy waits in the DMA queue behind this batch (a
    // queued copy with an unmet dependency stalls every later copy of the process, see the upload above).
    if (!alone) {
        hipStream_t main_st = st;
        if (!s.ev_scanned) HIPCHECK(hipEventCreateWithFlags(&s.ev_scanned, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_scanned, main_st));
        st = idx->sort_stream;
        HIPCHECK(hipStreamWaitEvent(st, s.ev_scanned, 0));
    }
    launch_sort_cands(s.d_qs, s.d_cands.p, s.cap_q, nq, s.d_qout, s.d_entries, s.out_cap, s.d_hdr, st,
                      dev_stream ? s.d_stream.p : nullptr);
    s.heaps_ready = s.dev_replay && !s.dist_batch;           // (a merge batch is replayed after the gather, not here)
    if (s.heaps_ready) {
        uint64_t* d_heaps = reinterpret_cast<uint64_t*>(d_result + off_heaps);
        uint32_t* d_sizes = reinterpret_cast<uint32_t*>(d_result + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
        if ((uint32_t)s.R <= replay_wave_max_R())     // one wave per query, all lanes at work (heap in registers)
            HIPCHECK(launch_replay_heap_wave_states(s.d_qs, s.d_stream.p, s.out_cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st));
        else                                                              // one wave per query, lane 0 pushing into an LDS heap (any R)
            launch_replay_heap(s.d_qs, s.d_stream.p, s.out_cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st);
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(take_launch_error());
    if (s.dist_batch && !s.rerun && s.mode != 1)
        if (int rc = enqueue_merge(idx, s, st)) return rc;

    if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(s.ev_done, st));
    return QADC_OK;
}
}  // namespace host
}  // namespace qadc

namespace {

int submit_common(qadc_index* idx, int slot_i, int nq, int ma, const int32_t* assign, float* tables,
                  const int8_t* qtables, int R, int mode = 0, int slice = 0, int nslices = 1,
                  const float* inj_vals = nullptr, int inj_n = 0) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    if (nq <= 0 || ma <= 0 || R <= 0 || !assign) return fail(QADC_E_ARG, "nq, ma, R must be > 0 and assign non-null");
    if (nq >= (1 << 24)) return fail(QADC_E_ARG, "nq must be < 2^24");
    if (ma >= (1 << 14)) return fail(QADC_E_ARG, "ma must be < 16384");
    if (!tables && !qtables) return fail(QADC_E_ARG, "tables is null");
    if (mode == 1 && (nslices < 1 || slice < 0 || slice >= nslices)) return fail(QADC_E_ARG, "need 0 <= slice < nslices");
    if (mode == 1 && slot_i > 1) return fail(QADC_E_ARG, "pre-scan slot must be 0 or 1");
    if (mode == 2 && (!inj_vals || inj_n < 1)) return fail(QADC_E_ARG, "prescan values missing");
    Slot& s = mode == 1 ? idx->pre_slot[slot_i] : idx->slot[slot_i];
    if (s.busy) return fail(QADC_E_STATE, "slot still holds an uncollected batch");
    if (int rc = use_device(idx)) return rc;
    s.mode = mode;
    s.pre_slice = slice;
    s.pre_nslices = nslices;
    s.inj_n = (uint32_t)inj_n;
    if (mode == 2) s.inj_vals.assign(inj_vals, inj_vals + (size_t)nq * inj_n);
    s.nq = nq;
    s.ma = ma;
    s.R = R;
    s.float_path = tables != nullptr;
    s.tables = tables;
    s.device_tables = false;
    s.assign.assign(assign, assign + (size_t)nq * ma);
    if (qtables) {
        const size_t nt = (size_t)nq * ma * idx->M * 16;
        for (size_t i = 0; i < nt; ++i)
            if (qtables[i] < 0)
                return fail(QADC_E_ARG, "int8 tables must lie in [0,127] (QuantizerMAX<int8_t> output, db_query_4.cpp:37-71)");
        s.qtables_in.assign(qtables, qtables + nt);
    }
    s.full_prescan = false;
    s.rerun = false;
    s.group_fell_back = false;
    s.front_sharded = false;
    s.assign_on_device = false;
    {   // which path: levels (long shared lists) or one workgroup per query (IVF batches, small lists)
        uint64_t max_codes = 0;
        for (int q = 0; q < nq; ++q) {
            uint64_t c = 0;
            for (int a = 0; a < ma; ++a) {
                const int p = s.assign[(size_t)q * ma + a];
                if (p < 0 || p >= (int)idx->parts.size()) return fail(QADC_E_ARG, "assign[] names a partition that does not exist");
                c += idx->parts[p].n;
            }
            max_codes = std::max(max_codes, c);
        }
        // (the rules for a LONE call of one or two queries hold when nothing else is in flight: a caller that pipelines single
        // queries over the slots gets them overlapped by the level path, not serialised on the query kernel's one stream)
        bool alone = mode == 0;
        for (int i = 0; i < kSlots; ++i) alone = alone && (&idx->slot[i] == &s || !idx->slot[i].busy);
        alone = alone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
        s.wgq = wgq_eligible(idx, nq, ma, R, mode, max_codes, tables != nullptr, alone);
        s.wgq_codes = max_codes;
    }
    if (s.wgq) s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
    s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
    if (!s.wgq) s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)nq * 8192u, 1ull << 30));
    if (int rc = plan_and_launch(idx, s)) return rc;
    s.busy = true;
    return QADC_OK;
}

}  // namespace

namespace qadc {
namespace host {

// Waits for the batch, regrows and re-runs on overflow, and lays the ordered candidate streams
// (padding-lane replays expanded) out in s.out_*.  Queries the device could not sort (more than
// kSortCap candidates) are sorted here.
// from_dist: the caller is qadc_dist_collect, which consumes the ordered streams where they lie in device memory; a batch
// submitted under the multi-GPU merge but collected by a plain collect call has no device-side heaps and, on the
// one-workgroup-per-query path, no host copy of its streams: they are fetched and replayed on the host.
int collect_common(qadc_index* idx, int slot_i, bool need_stream, bool from_dist) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    Slot& s = idx->slot[slot_i];
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no batch");
    if (int rc = use_device(idx)) return rc;
    if (s.dist_batch && !from_dist) {
        need_stream = true;
        if (idx->dist && idx->dist->slot[slot_i].pending)
            if (int rc = flush_merges(idx, idx->dist->slot[slot_i].seq)) return rc;
        if (idx->dist && idx->dist->slot[slot_i].pending_share)
            if (int rc = flush_shares(idx, idx->dist->slot[slot_i].seq)) return rc;
        if (idx->dist && idx->dist->slot[slot_i].enqueued) {     // a merge was enqueued with the batch: let it finish (its result is unused)
            HIPCHECK(hipEventSynchronize(idx->dist->slot[slot_i].ev_done));
            idx->dist->slot[slot_i].enqueued = false;
        }
    }
    const uint32_t sort_limit_max = kSortCap;
    uint64_t total_sorted = 0;
    if (s.assign_on_device) {                                  // qadc_search: assign[] comes back for the caller (and the planner)
        if (s.front_sharded) {                                   // gathered from the ranks; front_unpack_kernel stored it in mapped memory
            HIPCHECK(hipEventSynchronize(s.ev_fb));
            const int32_t* ha = reinterpret_cast<const int32_t*>(s.h_fmap.p);
            s.assign.assign(ha, ha + (size_t)s.nq * s.ma);
        } else {
            HIPCHECK(hipEventSynchronize(s.ev_assign));
            s.assign.assign(s.h_assign.p, s.h_assign.p + (size_t)s.nq * s.ma);
        }
        s.assign_on_device = false;
    }
    for (int attempt = 0; s.wgq; ++attempt) {                  // one workgroup per query: per-query stream capacity only
        bool seen = false;
        s.appended = s.replayed = false;
        if (s.poll) {
            // a lone small batch: the kernel's workgroups set the done bit of their records in the mapped result block as
            // they finish; watching those spares the completion-signal path (end-of-kernel release, signal, wake-up).
            // Bounded: a batch that takes longer is waited for the ordinary way.
            // The records are taken in workgroup order — the order the sub-streams are concatenated in — and a sub-stream is
            // appended to the batch's stream the moment its record shows the bit: the lines the GPU has just written are misses
            // (the copy of a lone query's 1.3 K entries: 2.8 us), and this way the first workgroups' — workgroup 0's ~500 of them —
            // are fetched while the last workgroups are still being waited for.
            const auto t_poll = std::chrono::steady_clock::now();
            const int nsub = s.nq * s.wgq_G;
            if (s.out_entries.size() < (size_t)nsub * s.wgq_cap) s.out_entries.resize((size_t)nsub * s.wgq_cap);
            int next = 0;
            size_t filled = 0, pushed = 0;
            bool appended = true;                                // (false: some sub-stream overflowed — sorted out below, nothing appended counts)
            // ... and a lone query whose caller wants the heap only (the synchronous query_scan) is replayed as it arrives: the
            // reference's pushes of the first workgroups' entries — kv_binheap::push, binheap.hpp:75-116, after the (0,127)
            // sentinel of db_query_4.cpp:276 — run while the last workgroups finish
            const bool replay_now = s.nq == 1 && !need_stream;
            if (replay_now) {
                if (s.early_heap.capacity() != s.R) s.early_heap.reset_capacity(s.R);
                s.early_heap.reset();
                s.early_heap.push(0, 127);
            }
            for (uint32_t spins = 0; !seen; ++spins) {
                while (next < nsub && (reinterpret_cast<const volatile uint32_t*>(&s.h_qout[next].flags)[0] & 4u) != 0) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    const QueryOut& qo = s.h_qout[next];
                    appended = appended && qo.count <= s.wgq_cap && !(qo.flags & 32u);
                    if (appended) {
                        std::memcpy(s.out_entries.data() + filled, s.h_entries + qo.out_off, sizeof(uint64_t) * qo.count);
                        filled += qo.count;
                    }
                    ++next;
                }
                // (pushes only while nothing waits to be appended: the copies are the misses and go first — with the pushes in
                // front of them the call was 3-7 us SLOWER than without any early replay, same box)
                if (replay_now && appended && next < nsub) {
                    const uint64_t* e_ = s.out_entries.data();
                    for (const size_t stop = std::min(filled, pushed + 32); pushed < stop; ++pushed)
                        s.early_heap.push((uint32_t)e_[pushed], (int8_t)(e_[pushed] >> 32));
                }
                seen = next == nsub;
                if (!seen && (spins & 63u) == 63u &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count() > 300e-6)
                    break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            s.appended = seen && appended;
            s.replayed = s.appended && replay_now;
            if (s.replayed) {
                const uint64_t* e_ = s.out_entries.data();
                for (; pushed < filled; ++pushed) s.early_heap.push((uint32_t)e_[pushed], (int8_t)(e_[pushed] >> 32));
            }
        }
        if (!seen) {
            if (s.ev_valid) HIPCHECK(hipEventSynchronize(s.ev_done));
            else HIPCHECK(hipStreamSynchronize(idx->wgq_stream));    // (a polled batch records no event: launch_wgq_batch)
        }
        uint64_t max_count = 0;
        bool cand_overflow = false;
        for (int q = 0; q < s.nq * s.wgq_G; ++q) {
            max_count = std::max<uint64_t>(max_count, s.h_qout[q].count);
            cand_overflow |= (s.h_qout[q].flags & 32u) != 0;
        }
        if (cand_overflow) {
            // some query emitted more candidates than a workgroup sorts in LDS (adversarial order, all-equal tables):
            // the level-structured path has the machinery for that (regrow, host sort) — run the batch through it
            idx->prof.regrows++;
            if (s.wgq_grouped) {
                idx->prof.group_fallbacks++;
                // (auto mode gives the second phase up after two such batches.  Under the multi-GPU merge that count decides
                // whether the NEXT batches issue the sharded front's all-gather, so it must move on every rank alike: the
                // strike travels in the gathered headers and qadc_dist_collect counts it, not this rank by itself)
                if (idx->dist) s.group_fell_back = true;
                else idx->group.strikes++;
            }
            s.wgq_grouped = false;
            s.wgq = false;
            s.wgq_G = 1;
            if (s.front_sharded) s.float_path = false;           // the level path takes the gathered int8 tables as they lie on the device
            s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)s.nq * 8192u, 1ull << 30));
            s.rerun = true;
            if (int rc = plan_and_launch(idx, s)) {
                s.busy = false;
                return rc;
            }
            break;
        }
        if (max_count <= s.wgq_cap) break;
        if (attempt >= 2 || max_count + 64 > (1ull << 31)) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "candidate stream overflow persists");
        }
        s.wgq_cap = (uint32_t)(max_count + max_count / 8 + 64);  // every entry was counted: size for all of them, run again
        idx->prof.regrows++;
        s.rerun = true;
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    for (int attempt = 0; !s.wgq; ++attempt) {
        HIPCHECK(hipEventSynchronize(s.ev_done));
        const uint32_t limit = std::min<uint32_t>(s.cap_q, sort_limit_max);
        uint64_t max_count = 0;
        total_sorted = 0;
        for (int q = 0; q < s.nq; ++q) {
            const QueryOut& qs = s.h_qout[q];
            max_count = std::max<uint64_t>(max_count, qs.count);
            if (qs.count <= limit) total_sorted += (uint64_t)qs.count + qs.reps;
        }
        const bool region_overflow = max_count > s.cap_q;     // emit_candidate counts every request
        const bool out_overflow = total_sorted > s.out_cap;
        bool prescan_overflow = false;
        for (int q = 0; q < s.nq; ++q) prescan_overflow |= (s.h_qout[q].flags & 8u) != 0;
        if (!region_overflow && !out_overflow && !prescan_overflow) break;
        if (attempt >= 4 || max_count + 64 > (1ull << 31) || total_sorted > (1ull << 31)) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "candidate buffers overflow persists (adversarial scan order?)");
        }
        // every emitted candidate was counted: size the buffers for all of them and run the batch again
        if (region_overflow) s.cap_q = (uint32_t)std::max<uint64_t>(max_count + 64, (uint64_t)s.cap_q * 2);
        if (out_overflow) s.out_cap = (uint32_t)(total_sorted + total_sorted / 4 + 1024);
        if (prescan_overflow) s.full_prescan = true;   // adversarially ordered starts: pre-scan all of them unfiltered
        idx->prof.regrows++;
        s.rerun = true;                                // (no merge is enqueued behind a re-run: the collect-time merge follows)
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    s.busy = false;
    if (idx->profile && s.wgq) {
        float ms = 0;
        if (s.prof_used >= 2) HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[s.prof_used - 1]));
        idx->prof.wgq_ms += ms;
        if (s.wgq_grouped && s.prof_used >= 6) {                 // [0] start, [1..4] around head / grouped scan / ordering, [5] end
            float a = 0, b = 0, c = 0;
            HIPCHECK(hipEventElapsedTime(&a, s.prof_ev[1], s.prof_ev[2]));
            HIPCHECK(hipEventElapsedTime(&b, s.prof_ev[2], s.prof_ev[3]));
            HIPCHECK(hipEventElapsedTime(&c, s.prof_ev[3], s.prof_ev[4]));
            idx->prof.group_head_ms += a;
            idx->prof.group_scan_ms += b;
            idx->prof.group_order_ms += c;
            idx->prof.group_batches++;
            // the work in those launches, recounted from assign[] the way the device draws the line (ivf_for_grouped_pairs)
            std::vector<uint32_t> cnt(idx->parts.size(), 0);
            for (int q = 0; q < s.nq; ++q) {
                int seen = 0;
                for (int a_ = 0; a_ < s.ma; ++a_) {
                    const int p = s.assign[(size_t)q * s.ma + a_];
                    const uint32_t n = idx->parts[p].n;
                    if (!n) continue;
                    if (seen < s.group_head_slots) idx->prof.group_head_codes += n;
                    else cnt[p]++;
                    ++seen;
                }
            }
            for (size_t p = 0; p < cnt.size(); ++p) {
                if (!cnt[p]) continue;
                const uint64_t n = idx->parts[p].n, full = cnt[p] / 8, rem = cnt[p] % 8;
                idx->prof.group_pairs += cnt[p];
                idx->prof.group_pass_codes8 += full * n;
                idx->prof.group_seats += full * 8;
                if (rem) {
                    const bool narrow = rem <= 4;
                    (narrow ? idx->prof.group_pass_codes4 : idx->prof.group_pass_codes8) += n;
                    idx->prof.group_seats += narrow ? 4 : 8;
                }
            }
        }
        idx->prof.wgq_launches++;
        for (int q = 0; q < s.nq * s.wgq_G; ++q) {
            idx->prof.wgq_front_cycles += (uint64_t)(s.h_qout[q].pad[0] & 0xffffu) << 6;
            idx->prof.wgq_sort_cycles += (uint64_t)(s.h_qout[q].pad[0] >> 16) << 6;
            idx->prof.wgq_scan_cycles += (uint64_t)s.h_qout[q].pad[1] << 4;
        }
        idx->prof.wgq_queries += (uint64_t)s.nq;
        for (int q = 0; q < s.nq; ++q)
            for (int a = 0; a < s.ma; ++a) idx->prof.wgq_codes += idx->parts[s.assign[(size_t)q * s.ma + a]].n;
        if (s.float_path) idx->prof.start_codes += s.start_codes;
    }
    if (idx->profile && !s.wgq) {
        float ms = 0;
        if (s.prof_used >= 2) {
            HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[1]));
            if (s.float_path) idx->prof.start_ms += ms;
        }
        for (auto& ll : s.launches) {
            if (ll.small || ll.early) {                      // counted, not timed (see plan_and_launch)
                idx->prof.small_launches++;
                idx->prof.small_codes += ll.codes;
                continue;
            }
            idx->prof.scan_launches++;
            idx->prof.scan_codes += ll.codes;
            idx->prof.mq_launches += ll.mq ? 1 : 0;
            idx->prof.pass_codes += ll.mq ? ll.codes / (uint64_t)ll.nitems * (uint64_t)((ll.nitems + 7) / 8) : ll.codes;
            if (ll.ev < 0 || (size_t)ll.ev + 1 >= s.prof_used) continue;   // not the first launch of its timed group
            HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[ll.ev], s.prof_ev[ll.ev + 1]));
            idx->prof.scan_ms += ms;
        }
        if (s.float_path) idx->prof.start_codes += s.start_codes;
    }
    const auto t0 = std::chrono::steady_clock::now();
    s.out_off.assign((size_t)s.nq + 1, 0);                       // (out_entries keeps its size as a high-water mark: out_off[nq] is the length)
    s.skipped_streams = false;
    const bool streams_stay = s.dist_batch && from_dist;         // consumed in device memory by the gather's pack kernel
    if (s.wgq && s.dev_replay && !streams_stay) {
        // the streams stayed in device memory; fetch them only if the caller wants them (or a query could not be
        // replayed on the device): one strided copy of the used part of every query's region
        bool need = need_stream;
        uint32_t max_count = 0;
        for (int q = 0; q < s.nq; ++q) {
            need = need || (s.heaps_ready && s.h_heap_sizes[q] == 0xffffffffu);
            max_count = std::max(max_count, s.h_qout[q].count);
        }
        if (need && max_count) {
            HIPCHECK(s.h_fetch.ensure((size_t)s.nq * s.wgq_cap));
            HIPCHECK(hipMemcpy2DAsync(s.h_fetch.p, sizeof(uint64_t) * s.wgq_cap, s.d_stream.p, sizeof(uint64_t) * s.wgq_cap,
                                      sizeof(uint64_t) * max_count, s.nq, hipMemcpyDeviceToHost, idx->copy_stream));
            HIPCHECK(hipStreamSynchronize(idx->copy_stream));
            s.h_entries = s.h_fetch.p;
        }
    }
    // Common case — every query's stream lies ready in the pinned result block (ordered on the device; a split query's
    // sub-streams in workgroup order) or is not wanted (heap built on the device): lengths first, then the pool's threads
    // copy disjoint ranges (the block was just written by the GPU: a single thread reads it at ~9 GB/s).
    {
        const bool split = s.wgq && s.wgq_G > 1;
        bool simple = true, skipped = false;
        uint64_t total = 0, ncand = 0;
        for (int q = 0; q < s.nq && simple; ++q) {
            s.out_off[q] = total;
            if (split) {
                for (int g = 0; g < s.wgq_G; ++g) total += s.h_qout[(size_t)q * s.wgq_G + g].count;
                continue;
            }
            const QueryOut& qs = s.h_qout[q];
            if ((!need_stream && s.heaps_ready && s.h_heap_sizes[q] != 0xffffffffu) ||   // heap already built on the device
                (streams_stay && (qs.flags & 4u))) {                                     // ... or the gather takes it from there
                skipped = true;
                continue;
            }
            if (qs.flags & 4u) total += (uint64_t)qs.count + qs.reps;
            else simple = false;
        }
        if (simple) {
            s.out_off[s.nq] = total;
            s.skipped_streams = skipped;
            for (int q = 0; q < s.nq * (split ? s.wgq_G : 1); ++q) ncand += s.h_qout[q].count;
            idx->prof.candidates += ncand;
            if (split && s.appended) {                           // (a polled batch: the poll loop appended the sub-streams as they finished)
                idx->prof.host_replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return QADC_OK;
            }
            if (s.out_entries.size() < total) s.out_entries.resize(total);
            auto copy = [&](int q0, int q1) {
                for (int q = q0; q < q1; ++q) {
                    uint64_t* dst = s.out_entries.data() + s.out_off[q];
                    if (split) {
                        for (int g = 0; g < s.wgq_G; ++g) {
                            const QueryOut& qs = s.h_qout[(size_t)q * s.wgq_G + g];
                            std::memcpy(dst, s.h_entries + qs.out_off, sizeof(uint64_t) * qs.count);
                            dst += qs.count;
                        }
                    } else if (s.out_off[q + 1] > s.out_off[q]) {
                        std::memcpy(dst, s.h_entries + s.h_qout[q].out_off, sizeof(uint64_t) * (s.out_off[q + 1] - s.out_off[q]));
                    }
                }
            };
            int nt = (int)std::min<unsigned>(host_threads(), 8);
            nt = std::max(1, std::min(nt, s.nq / 2));
            if (total < 16384) nt = 1;
            const int per = std::max(1, s.nq / (nt * 4));
            idx->pool.run((s.nq + per - 1) / per, nt, [&](int t) { copy(t * per, std::min(s.nq, (t + 1) * per)); });
            idx->prof.host_replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return QADC_OK;
        }
        s.out_off.assign((size_t)s.nq + 1, 0);                   // a query needs the host sort: the sequential path below
        s.out_entries.clear();
    }
    for (int q = 0; q < s.nq && s.wgq && s.wgq_G > 1; ++q) {   // sub-streams of a query's workgroups, in workgroup order
        s.out_off[q] = s.out_entries.size();
        for (int g = 0; g < s.wgq_G; ++g) {
            const QueryOut& qs = s.h_qout[(size_t)q * s.wgq_G + g];
            idx->prof.candidates += qs.count;
            s.out_entries.insert(s.out_entries.end(), s.h_entries + qs.out_off, s.h_entries + qs.out_off + qs.count);
        }
    }
    for (int q = 0; q < s.nq && !(s.wgq && s.wgq_G > 1); ++q) {
        const QueryOut& qs = s.h_qout[q];
        s.out_off[q] = s.out_entries.size();
        idx->prof.candidates += qs.count;
        if ((!need_stream && s.heaps_ready && s.h_heap_sizes[q] != 0xffffffffu) ||   // heap already built on the device
            (streams_stay && (qs.flags & 4u))) {
            s.skipped_streams = true;
            continue;
        }
        if (qs.flags & 4u) {                                  // ordered and expanded on the device
            const size_t n = (size_t)qs.count + qs.reps;
            s.out_entries.insert(s.out_entries.end(), s.h_entries + qs.out_off, s.h_entries + qs.out_off + n);
            continue;
        }
        // host fallback: fetch the raw region, sort by (level, assign slot, position), expand replays
        idx->prof.host_sorted_queries++;
        HIPCHECK(s.h_cands.ensure(qs.count));
        HIPCHECK(hipMemcpyAsync(s.h_cands.p, s.d_cands.p + (size_t)q * s.cap_q, sizeof(Cand) * qs.count,
                                hipMemcpyDeviceToHost, idx->copy_stream));
        HIPCHECK(hipStreamSynchronize(idx->copy_stream));
        Cand* c = s.h_cands.p;
        std::sort(c, c + qs.count, [](const Cand& a, const Cand& b) {
            const uint32_t oa = a.order & 0xfffffu, ob = b.order & 0xfffffu;
            if (oa != ob) return oa < ob;
            return a.pos < b.pos;
        });
        for (uint32_t i = 0; i < qs.count; ++i) {
            const uint32_t reps = 1u + ((c[i].order >> 20) & 15u);
            for (uint32_t r = 0; r < reps; ++r)
                s.out_entries.push_back((uint64_t)c[i].key | ((uint64_t)(c[i].val & 0xffu) << 32) |
                                        ((uint64_t)(c[i].order & 0x3fffu) << 40));
        }
    }
    s.out_off[s.nq] = s.out_entries.size();
    idx->prof.host_replay_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return QADC_OK;
}

void finish_float_outputs(qadc_index* idx, Slot& s, int32_t* status, float* qmin, float* qmax) {
    const size_t per_q = (size_t)s.ma * idx->M * 16;
    const int stride = s.wgq ? s.wgq_G : 1;                      // (a query's workgroups report the same qmin / qmax / flags)
    for (int q = 0; q < s.nq; ++q) {
        QueryOut qs = s.h_qout[(size_t)q * stride];
        if (s.front_sharded) {                                   // the front ran on rank q / per: its verdict came with the gather
            const uint32_t* fr = reinterpret_cast<const uint32_t*>(s.h_fmap.p + (size_t)s.nq * s.ma * 4) + 4 * (size_t)q;
            qs.flags = (qs.flags & ~3u) | (fr[0] & 3u);
            std::memcpy(&qs.qmin, &fr[1], 4);
            std::memcpy(&qs.qmax, &fr[2], 4);
        }
        if (status) status[q] = (qs.flags & 1u) ? 1 : 0;
        if (qmin) qmin[q] = qs.qmin;
        if (qmax) qmax[q] = qs.qmax;
        if ((qs.flags & 2u) && s.tables) {  // in-place clamp of the caller's tables (db_query_4.cpp:262-269)
            float* t = s.tables + (size_t)q * per_q;
            for (size_t i = 0; i < per_q; ++i)
                if (t[i] < 0) t[i] = 0;
        }
    }
}

int replay_outputs(qadc_index* idx, Slot& s, uint32_t* keys, int8_t* values, int32_t* sizes, const int32_t* status) {
    ScopedMs timer(idx->prof.host_heap_ms);
    auto work = [&](int q0, int q1) {
        kv_heap<uint32_t, int8_t> bh(s.R);
        for (int q = q0; q < q1; ++q) {
            bh.reset();
            if (status && status[q]) {
                if (sizes) sizes[q] = 0;
                continue;
            }
            if (s.heaps_ready && s.skipped_streams && s.h_heap_sizes[q] != 0xffffffffu) {   // replayed by replay_heap_kernel
                const uint32_t sz = s.h_heap_sizes[q];
                const uint64_t* hv = s.h_heaps + (size_t)q * s.R;
                if (sizes) sizes[q] = (int32_t)sz;
                for (uint32_t i = 0; i < sz; ++i) {
                    if (keys) keys[(size_t)q * s.R + i] = (uint32_t)hv[i];
                    if (values) values[(size_t)q * s.R + i] = (int8_t)(hv[i] >> 32);
                }
                continue;
            }
            if (s.replayed && s.wgq && s.wgq_G > 1) {            // a lone polled query: replayed while it arrived (collect_common)
                if (sizes) sizes[q] = s.early_heap.size();
                if (keys) std::memcpy(keys + (size_t)q * s.R, s.early_heap.keys(), sizeof(uint32_t) * s.early_heap.size());
                if (values) std::memcpy(values + (size_t)q * s.R, s.early_heap.values(), s.early_heap.size());
                continue;
            }
            bh.push(0, 127);  // db_query_4.cpp:276
            for (uint64_t i = s.out_off[q]; i < s.out_off[q + 1]; ++i)
                bh.push((uint32_t)s.out_entries[i], (int8_t)(s.out_entries[i] >> 32));
            if (sizes) sizes[q] = bh.size();
            if (keys) std::memcpy(keys + (size_t)q * s.R, bh.keys(), sizeof(uint32_t) * bh.size());
            if (values) std::memcpy(values + (size_t)q * s.R, bh.values(), bh.size());
        }
    };
    // queries are independent: large batches (IVF) are replayed by a few host threads, the caller still
    // drives the library from one thread
    const uint64_t pushes = s.out_off[s.nq];
    int nt = (int)std::min<unsigned>(host_threads(), 16);   // (10M-code list, 32 queries per step: 8 -> 16 threads 0.260 -> 0.231 ms per step)
    nt = std::max(1, std::min(nt, s.nq / 2));
    if (pushes < 4000) nt = 1;                                 // (waking the workers costs about as much as 4 K pushes)
    // tasks of a few queries each, handed out dynamically: candidate counts differ from query to query
    const int per = std::max(1, s.nq / (nt * 4));
    const int tasks = (s.nq + per - 1) / per;
    idx->pool.run(tasks, nt, [&](int t) { work(t * per, std::min(s.nq, (t + 1) * per)); });
    return QADC_OK;
}
}  // namespace host
}  // namespace qadc

namespace {

// ---- the streams of an index ----------------------------------------------------------------------------------------------
// The streaming launches fill every CU for milliseconds.  They go on the LOWEST-priority stream so that the short work that
// must overlap them is dispatched as soon as a workgroup slot frees up instead of waiting for the whole batch: the previous
// batch's candidate ordering, the next batch's front (own streams, highest priority), the collectives of the multi-GPU merge.
//
// What was measured about streams (round 4; tools/stream_order_ab*.sh, profiles/r04_stream_order.txt):
//  * every HIP stream created here gets a hardware queue of its own (the runtime keeps at most four per priority,
//    GPU_MAX_HW_QUEUES), the queues take hardware slots in creation order, and slot i is served by compute pipe i mod 4;
//  * queues on one pipe are served in turn, whatever their priority: a dispatch that waits for free CUs — anything launched
//    beside a long scan — holds up the other queues of its pipe.  So what matters is WHICH streams share a pipe:
//        pipe 0: S scan stream of the level path (lowest priority), W scan stream of the one-workgroup-per-query batches (option
//                "wgq_stream"; highest priority since round 5 — see the last item —; a batch uses one or the other)
//        pipe 1: C copy / coarse assignment, L the merge's collectives         (both short; C normal priority since round 5)
//        pipe 2: O ordering pass + replays of single-GPU batches (normal priority since round 5), M the merge's interleave +
//                replay (highest since round 5; O idles under the merge)
//        pipe 3: F the next batch's front — alone: it decides when the next scan can start
//    One of 8 ranks' batch (loopback stand-in) with this order against round 3's (the merge's nine streams created at
//    qadc_dist_init, one merge stream per slot): C3 shape 0.84 -> 0.51-0.53 ms, C5 1.26 -> 0.89-0.90.  With the front on the
//    scan stream's pipe, or on a merge stream's: 0.78-0.83 / 1.20-1.23.  More merge streams only put one of them on the
//    front's or the scan's pipe (C5 shape: 1 stream 0.89 ms, 2: 0.93, 3: 1.06);
//  * the layout a process gets is only the one above for the FIRST set of streams it creates: an index created after another
//    one was destroyed — in either order of stream destruction — lands elsewhere (same shapes: 0.81 / 1.20 ms; the single-GPU
//    C3 leg 0.75 -> 1.08 us per query; that was bench.py's IVF leg, the third index of its process).  Round 3's "figures
//    depend on the process's history" was this;
//  * (round 5, qadc_stream_probe: profiles/r05_stream_probe.txt) on one pipe only a HIGHER-priority queue's waiting dispatch
//    holds up a lower-priority queue; queues of equal priority cost each other a few microseconds.  And the runtime's four
//    queues of the highest priority are all a process has: a fifth highest-priority stream — a communicator's, or a fifth of
//    ours — shares one of them.  Hence the priorities below (round 4 had C and O at the highest, W normal, M lowest, and every
//    batch on S): FOUR highest-priority streams — W, the scan stream of the one-workgroup-per-query (IVF) batches, which
//    nobody can hold up then; F; L; M — C and O at normal priority, S (the level path's long launches) lowest as before.
//    Same-box A/B, every leg of bench.py and the one-of-8 stand-ins (profiles/r05_stream_priorities.txt, "candidate 4"):
//    nothing slower, the flat 125 M-code step 1.13 -> 1.08 ms, and one of 8 ranks' IVF batch in a process that created a
//    torch "nccl" group BEFORE the set 0.73 -> 0.49 ms (C3) / 1.03 -> 0.83 (C5) instead of 0.38 / 0.72 in a fresh one: the
//    order rule (qadc_device_prepare first) still holds, breaking it costs half as much.
// Hence: ONE set of streams per process and device, created back to back in the order above by the first index on that
// device, shared by every later index and never destroyed.  Indexes of one device that are driven at the same time share
// these streams (stream order is then a superset of what each needs: correct, possibly serialised); the expected use is one
// index at a time per GPU (include/qadc.h: one host thread drives one index, one process per GPU).
struct StreamSet {
    hipStream_t stream = nullptr, wgq = nullptr, copy = nullptr, sort = nullptr, front = nullptr, coll = nullptr;
    hipStream_t merge[kMergeStreams] = {};
    std::vector<hipStream_t> created;                           // creation order
    std::string order, report;                                  // the creation order that was kept, and what the layout probe saw
};
std::mutex g_streams_mu;
std::vector<std::pair<int, StreamSet*>> g_streams;              // (device, set): lives until the process ends

// Creates the set's streams back to back, in the order and with the priorities the comment above derives:
// S (level-path scan, lowest), C (copy, normal), O (ordering, normal), F (front), W (query-kernel scan), L (collectives), M (merge) —
// the last four at the highest priority: the four highest-priority hardware queues a process has.
hipError_t create_stream_set(StreamSet& ss) {
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int prio_normal = (prio_least + prio_greatest) / 2;
    struct Want { hipStream_t* dst; int prio; };
    const Want wants[] = {{&ss.stream, prio_least}, {&ss.copy, prio_normal}, {&ss.sort, prio_normal}, {&ss.front, prio_greatest},
                          {&ss.wgq, prio_greatest}, {&ss.coll, prio_greatest}, {&ss.merge[0], prio_greatest}};
    hipError_t e = hipSuccess;
    for (const Want& w : wants) {
        e = hipStreamCreateWithPriority(w.dst, hipStreamNonBlocking, w.prio);
        if (e != hipSuccess) break;
        ss.created.push_back(*w.dst);
    }
    if (e != hipSuccess) {
        for (size_t i = ss.created.size(); i-- > 0;) (void)hipStreamDestroy(ss.created[i]);
        ss = StreamSet();
        return e;
    }
    for (int i = 1; i < kMergeStreams; ++i) ss.merge[i] = ss.merge[0];
    return hipSuccess;
}

// One probe: microseconds the marker on `b` waited behind the CU-hungry launch on `a` (min of `reps`: other work on the GPU only
// ever adds to it).  d_t: 4 u64 of device scratch.
hipError_t probe_wait_us(hipStream_t a, hipStream_t b, unsigned long long* d_t, int ncu, int reps, double* out) {
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const unsigned long long init[4] = {~0ull, 0, 0, 0};
        hipError_t e = hipMemcpy(d_t, init, sizeof(init), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_stream_probe(d_t, 8 * ncu, 3000, a, b);       // 4 rounds of 30 us at two workgroups per CU
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess) e = hipStreamSynchronize(b);
        unsigned long long t[4];
        if (e == hipSuccess) e = hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        best = std::min(best, ((double)t[2] - (double)t[0]) / 100.0);
    }
    *out = best;
    return hipSuccess;
}

// How many of the pairs that must not obstruct each other do (-1: the probe itself failed).  A marker that waits more than 40 us
// of the ~120 us launch sits behind it on the same pipe (~100 us: it runs when the last workgroup has been dispatched) or in the
// same hardware queue (~135 us: when the launch is over).  Must stay free: the scan stream of {copy, ordering, front, collectives,
// merge}; the front stream of {copy, ordering, collectives}; the collectives' stream of {ordering, front}.  (By design, and not
// counted: ordering holds up merge, the idle alternative scan stream shares the scan's pipe, copy and collectives share one.)
int layout_violations(StreamSet& ss, int device, std::string* report) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    unsigned long long* d_t = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_t), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    struct Pair { hipStream_t a, b; const char* name; };
    const Pair pairs[] = {
        {ss.coll, ss.stream, "collectives>scan"}, {ss.copy, ss.stream, "copy>scan"}, {ss.sort, ss.stream, "ordering>scan"},
        {ss.front, ss.stream, "front>scan"}, {ss.merge[0], ss.stream, "merge>scan"},
        {ss.copy, ss.front, "copy>front"}, {ss.sort, ss.front, "ordering>front"}, {ss.coll, ss.front, "collectives>front"},
        {ss.sort, ss.coll, "ordering>collectives"}, {ss.front, ss.coll, "front>collectives"}};
    int v = 0;
    double warm = 0;
    (void)probe_wait_us(ss.copy, ss.front, d_t, prop.multiProcessorCount, 1, &warm);   // (first launch: code object load, LDS opt-in)
    for (const Pair& p : pairs) {
        double w = 0;
        if (probe_wait_us(p.a, p.b, d_t, prop.multiProcessorCount, 2, &w) != hipSuccess) { v = -1; break; }
        if (w > 40.0) {
            ++v;
            if (report) *report += std::string(report->empty() ? "" : ", ") + p.name + " " + std::to_string((int)w) + " us";
        }
    }
    (void)hipFree(d_t);
    if (report && v == 0) *report = "ok";
    return v;
}

// The per-device stream set of the process: found or created (the caller holds no lock).  A cached set whose streams the
// runtime no longer knows (the application called hipDeviceReset between two indexes) is dropped and rebuilt.
int shared_stream_set(int device, StreamSet** out) {
    std::lock_guard<std::mutex> lock(g_streams_mu);
    for (auto it = g_streams.begin(); it != g_streams.end(); ++it) {
        if (it->first != device) continue;
        const hipError_t q = hipStreamQuery(it->second->stream);
        if (q == hipSuccess || q == hipErrorNotReady) { *out = it->second; return QADC_OK; }
        (void)hipGetLastError();                                 // dead handles: forget them (they cannot be destroyed any more)
        delete it->second;
        g_streams.erase(it);
        break;
    }
    StreamSet* ss = new StreamSet();
    const hipError_t e = create_stream_set(*ss);
    if (e != hipSuccess) {
        delete ss;
        return fail(QADC_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    // (the layout this order gives is CHECKED on demand — qadc_stream_layout: layout_violations above — not repaired: a search
    // over pad streams in front of re-created sets was built and measured in round 5, and in a process that created a torch "nccl"
    // group first every candidate left exactly one of the four highest-priority streams on the scan's pipe; what works is creating
    // the set BEFORE the communicator: qadc_device_prepare)
    ss->order = "S,C,O,F,W,L,M0";
    g_streams.emplace_back(device, ss);
    *out = ss;
    return QADC_OK;
}

int attach_streams(qadc_index* idx) {
    StreamSet* ss = nullptr;
    if (int rc = shared_stream_set(idx->device, &ss)) return rc;
    idx->stream = ss->stream;
    idx->wgq_stream = ss->wgq;
    idx->copy_stream = ss->copy;
    idx->sort_stream = ss->sort;
    idx->front_stream = ss->front;
    idx->coll_stream = ss->coll;
    for (int i = 0; i < kMergeStreams; ++i) idx->merge_streams[i] = ss->merge[i];
    return QADC_OK;
}

}  // namespace

extern "C" {

const char* qadc_last_error(void) { return g_err.c_str(); }
const char* qadc_version(void) { return "qadc-mi355x 0.1 (gfx950)"; }

int qadc_device_prepare(int device_id) {
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(QADC_E_HIP, "no HIP device: the Quick-ADC engine has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(QADC_E_ARG, "device_id out of range");
    HIPCHECK(hipSetDevice(device_id));
    StreamSet* ss = nullptr;
    return shared_stream_set(device_id, &ss);
}

const char* qadc_stream_layout(int device_id) {
    static thread_local std::string out;
    if (qadc_device_prepare(device_id)) return "";
    StreamSet* ss = nullptr;
    if (shared_stream_set(device_id, &ss)) return "";
    std::lock_guard<std::mutex> lock(g_streams_mu);              // (the report cache is shared by every index of the device)
    if (ss->report.empty()) {                                    // probed once per set, on demand — by THIS entry point only
        std::string rep;
        const int v = layout_violations(*ss, device_id, &rep);
        ss->report = v < 0 ? "probe failed" : rep;
    }
    out = ss->order + " | " + ss->report;
    return out.c_str();
}

int qadc_stream_probe(int device_id, int a, int b, double* wait_us, double* spin_us) {
    if (a < 0 || a > 6 || b < 0 || b > 6 || !wait_us) return fail(QADC_E_ARG, "streams are numbered 0 .. 6 (scan, copy, ordering, front, alternative scan, collectives, merge)");
    if (int rc = qadc_device_prepare(device_id)) return rc;
    StreamSet* ss = nullptr;
    if (int rc = shared_stream_set(device_id, &ss)) return rc;
    hipStream_t st[7] = {ss->stream, ss->copy, ss->sort, ss->front, ss->wgq, ss->coll, ss->merge[0]};
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device_id));
    DevBuf<unsigned long long> d_t;
    HIPCHECK(d_t.ensure(4));
    const unsigned long long init[4] = {~0ull, 0, 0, 0};
    for (int i = 0; i < 7; ++i) HIPCHECK(hipStreamSynchronize(st[i]));
    HIPCHECK(hipMemcpy(d_t.p, init, sizeof(init), hipMemcpyHostToDevice));
    // 8 workgroups per CU at two resident per CU: four rounds of 30 us each
    HIPCHECK(launch_stream_probe(d_t.p, 8 * prop.multiProcessorCount, 3000, st[a], st[b]));
    HIPCHECK(hipStreamSynchronize(st[a]));
    HIPCHECK(hipStreamSynchronize(st[b]));
    unsigned long long t[4];
    HIPCHECK(hipMemcpy(t, d_t.p, sizeof(t), hipMemcpyDeviceToHost));
    d_t.release();
    *wait_us = ((double)t[2] - (double)t[0]) / 100.0;           // marker start after the first spin workgroup's start
    if (spin_us) *spin_us = ((double)t[1] - (double)t[0]) / 100.0;
    return QADC_OK;
}

int qadc_index_create(qadc_index** out, int M, int device_id) {
    if (!out) return fail(QADC_E_ARG, "out is null");
    if (M != 16 && M != 32)
        return fail(QADC_E_ARG, "Unsupported (nsq,nsq_bits) configuration. Supported configurations are: (16,4) (32,4).");
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(QADC_E_HIP, "no HIP device: the Quick-ADC engine has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(QADC_E_ARG, "device_id out of range");
    HIPCHECK(hipSetDevice(device_id));
    qadc_index* idx = new qadc_index();
    idx->M = M;
    idx->cs = M / 2;
    if (M == 16) idx->head_wg = 512;                           // the partition-major batches' head at 16x4: 8 waves per query (same-box A/B,
                                                               // profiles/r06_head_wg_ab.txt: C3 0.656 -> 0.612 us per query; 32x4 keeps 16 waves)
    idx->device = device_id;
    // Test hooks (tests/conftest.py runs every parity test through every scan path this way).  They only apply when
    // QADC_TEST_HOOKS=1 is set as well, so that a stray QADC_* variable in a deployment cannot change the scan path.
    const char* hooks = std::getenv("QADC_TEST_HOOKS");
    if (hooks && std::atoi(hooks) == 1) {
        if (const char* e = std::getenv("QADC_WGQ")) idx->wgq = std::atoi(e);   // force (2) / forbid (0) the one-workgroup-per-query path
        if (const char* e = std::getenv("QADC_HEAD_LEVEL")) idx->head_level = std::max(0, std::min(std::atoi(e), kMaxLevels - 1));
    }
    if (int rc = attach_streams(idx)) {
        delete idx;
        return rc;
    }
    *out = idx;
    return QADC_OK;
}

int qadc_index_destroy(qadc_index* idx) {
    if (!idx) return QADC_OK;
    (void)hipSetDevice(idx->device);
    // a pre-scan (front stream) or an on-demand copy may still be in flight: drain all four streams before freeing
    for (hipStream_t st : {idx->stream, idx->wgq_stream, idx->front_stream, idx->copy_stream, idx->sort_stream})
        if (st) (void)hipStreamSynchronize(st);
    (void)qadc_dist_shutdown(idx);      // drains the merge's stream and frees the communicator while the index's streams and
                                        // the slot buffers the pack kernel reads are still alive
    for (auto& p : idx->parts) {
        if (p.own) {
            if (p.d_codes) (void)hipFree(p.d_codes);
            if (p.d_labels) (void)hipFree(p.d_labels);
        }
        if (p.d_starts) (void)hipFree(p.d_starts);
    }
    idx->feed.d_codebooks.release();
    idx->feed.d_rotation.release();
    idx->feed.d_coarse.release();
    idx->d_partdesc.release();
    Slot* all_slots[kSlots + 2];
    for (int i = 0; i < kSlots; ++i) all_slots[i] = &idx->slot[i];
    all_slots[kSlots] = &idx->pre_slot[0];
    all_slots[kSlots + 1] = &idx->pre_slot[1];
    for (Slot* sp : all_slots) {
        Slot& s = *sp;
        s.d_in.release(); s.h_in.release(); s.d_state.release(); s.h_result.release();
        s.d_ftables.release(); s.d_qtables.release(); s.d_cands.release(); s.d_fc.release(); s.d_lfstate.release();
        s.h_cands.release(); s.d_stream.release(); s.d_qflags.release(); s.d_fvals.release(); s.d_qcands.release(); s.h_fetch.release();
        s.d_fblock.release(); s.d_fgathered.release(); s.d_front_all.release(); s.h_fmap.release();
        if (s.ev_fa) (void)hipEventDestroy(s.ev_fa);
        if (s.ev_fb) (void)hipEventDestroy(s.ev_fb);
        if (s.ev_assign) (void)hipEventDestroy(s.ev_assign);
        s.d_queries.release(); s.d_assign.release(); s.d_cdist.release(); s.h_queries.release(); s.h_assign.release();
        if (s.ev_feed) (void)hipEventDestroy(s.ev_feed);
        if (s.ev_pre) (void)hipEventDestroy(s.ev_pre);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_up) (void)hipEventDestroy(s.ev_up);
        if (s.ev_front) (void)hipEventDestroy(s.ev_front);
        if (s.ev_scanned) (void)hipEventDestroy(s.ev_scanned);
        for (auto e : s.prof_ev) (void)hipEventDestroy(e);
    }
    // (the streams belong to the process: attach_streams)
    delete idx;
    return QADC_OK;
}

// has_labels is what partition 0 hands out, EVEN IF IT IS EMPTY (compute_sizes, db_query_4.cpp:105-110); every non-empty
// partition must then agree (118-124).  An index_db whose partition 0 is empty returns a null labels pointer for it
// (an empty std::vector's data()), so the reference refuses such a database: same here, same message.
static int check_labels_mode(qadc_index* idx, bool has_labels) {
    if (idx->labeled < 0) idx->labeled = has_labels ? 1 : 0;
    if ((idx->labeled == 1) != has_labels)
        return fail(QADC_E_ARG, "Cannot prepare database. Some partitions have labels and some have not");
    return QADC_OK;
}

static int alloc_part(qadc_index* idx, Part& pt, uint32_t n, bool labels) {
    pt.n = n;
    pt.global_n = n;
    pt.starts_cap = n;
    const size_t bytes = ((size_t)n * idx->cs + 15) / 16 * 16 + 64;  // tail padding for 16-byte vector reads
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_codes), bytes));
    HIPCHECK(hipMemsetAsync(pt.d_codes + (bytes - 80), 0, 80, idx->stream));
    if (labels) HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_labels), std::max<size_t>(n, 1) * sizeof(uint32_t)));
    return QADC_OK;
}

int qadc_index_add_partitions(qadc_index* idx, int part_count, const uint8_t* const* codes,
                              const uint32_t* const* labels, const uint32_t* sizes) {
    if (!idx || part_count < 0 || !sizes || (part_count && !codes)) return fail(QADC_E_ARG, "bad arguments");
    if (int rc = use_device(idx)) return rc;
    for (int p = 0; p < part_count; ++p) {
        Part pt;
        const bool has_labels = labels != nullptr && labels[p] != nullptr;
        if (sizes[p] == 0) {  // "Warning: Partition i is empty" (db_query_4.cpp:113-116)
            if (idx->parts.empty() && idx->labeled < 0) idx->labeled = has_labels ? 1 : 0;
            idx->parts.push_back(pt);
            continue;
        }
        if (int rc = check_labels_mode(idx, has_labels)) return rc;
        if (int rc = alloc_part(idx, pt, sizes[p], has_labels)) return rc;
        HIPCHECK(hipMemcpyAsync(pt.d_codes, codes[p], (size_t)sizes[p] * idx->cs, hipMemcpyHostToDevice, idx->stream));
        if (has_labels)
            HIPCHECK(hipMemcpyAsync(pt.d_labels, labels[p], (size_t)sizes[p] * 4, hipMemcpyHostToDevice, idx->stream));
        HIPCHECK(hipStreamSynchronize(idx->stream));
        idx->parts.push_back(pt);
    }
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_interleaved(qadc_index* idx, const uint8_t* interleaved, const uint32_t* labels, uint32_t size) {
    if (!idx || (size && !interleaved)) return fail(QADC_E_ARG, "bad arguments");
    if (int rc = use_device(idx)) return rc;
    Part pt;
    if (size == 0) {
        if (idx->parts.empty() && idx->labeled < 0) idx->labeled = labels != nullptr ? 1 : 0;
        idx->parts.push_back(pt);
        return QADC_OK;
    }
    if (int rc = check_labels_mode(idx, labels != nullptr)) return rc;
    if (int rc = alloc_part(idx, pt, size, labels != nullptr)) return rc;
    const size_t ibytes = (size_t)((size + 15u) / 16u) * idx->cs * 16;
    uint8_t* d_tmp = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_tmp), ibytes));
    HIPCHECK(hipMemcpyAsync(d_tmp, interleaved, ibytes, hipMemcpyHostToDevice, idx->stream));
    launch_deinterleave(pt.d_codes, d_tmp, size, idx->cs, idx->stream);
    if (labels) HIPCHECK(hipMemcpyAsync(pt.d_labels, labels, (size_t)size * 4, hipMemcpyHostToDevice, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    HIPCHECK(hipFree(d_tmp));
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_device(qadc_index* idx, const void* d_codes, const void* d_labels, uint32_t size) {
    if (!idx || (size && !d_codes)) return fail(QADC_E_ARG, "bad arguments");
    if ((reinterpret_cast<uintptr_t>(d_codes) & 15u) != 0) return fail(QADC_E_ARG, "d_codes must be 16-byte aligned");
    Part pt;
    if (!size && idx->parts.empty() && idx->labeled < 0) idx->labeled = d_labels != nullptr ? 1 : 0;
    if (size) {
        if (int rc = check_labels_mode(idx, d_labels != nullptr)) return rc;
        pt.d_codes = const_cast<uint8_t*>(static_cast<const uint8_t*>(d_codes));
        pt.d_labels = const_cast<uint32_t*>(static_cast<const uint32_t*>(d_labels));
        pt.n = size;
        pt.global_n = size;
        pt.starts_cap = size;
        pt.own = false;
    }
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_synthetic(qadc_index* idx, uint32_t size, uint64_t seed, uint64_t first_word) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (int rc = use_device(idx)) return rc;
    Part pt;
    if (size) {
        if (int rc = check_labels_mode(idx, false)) return rc;
        if (int rc = alloc_part(idx, pt, size, false)) return rc;
        const uint64_t nwords = ((uint64_t)size * idx->cs + 7) / 8;
        launch_fill_codes(pt.d_codes, first_word, nwords, seed, idx->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

static int attach_starts(qadc_index* idx, Part& pt, const uint8_t* starts_host, uint32_t starts_count, uint64_t seed,
                         bool synthetic) {
    if (pt.n && pt.first_pos == 0 && starts_count <= pt.n) {  // the local range begins with the starts
        pt.starts_cap = pt.n;
        return QADC_OK;
    }
    if (starts_count == 0) return fail(QADC_E_ARG, "a shard that does not begin the partition needs a starts replica");
    const size_t bytes = ((size_t)starts_count * idx->cs + 15) / 16 * 16 + 64;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&pt.d_starts), bytes));
    if (synthetic) {
        launch_fill_codes(pt.d_starts, 0, ((uint64_t)starts_count * idx->cs + 7) / 8, seed, idx->stream);
        HIPCHECK(hipGetLastError());
    } else {
        HIPCHECK(hipMemcpyAsync(pt.d_starts, starts_host, (size_t)starts_count * idx->cs, hipMemcpyHostToDevice, idx->stream));
    }
    HIPCHECK(hipStreamSynchronize(idx->stream));
    pt.starts_cap = starts_count;
    return QADC_OK;
}

int qadc_index_add_partition_shard(qadc_index* idx, const uint8_t* codes, const uint32_t* labels, uint32_t local_n,
                                   uint32_t global_n, uint32_t first_pos, const uint8_t* starts, uint32_t starts_count) {
    if (!idx || (local_n && !codes) || global_n == 0 || (uint64_t)first_pos + local_n > global_n)
        return fail(QADC_E_ARG, "bad shard range");
    if ((uint64_t)first_pos * idx->cs % 16 != 0) return fail(QADC_E_ARG, "first_pos must keep the shard 16-byte aligned");
    if (int rc = use_device(idx)) return rc;
    if (local_n)
        if (int rc = check_labels_mode(idx, labels != nullptr)) return rc;
    Part pt;
    if (local_n) {
        if (int rc = alloc_part(idx, pt, local_n, labels != nullptr)) return rc;
        HIPCHECK(hipMemcpyAsync(pt.d_codes, codes, (size_t)local_n * idx->cs, hipMemcpyHostToDevice, idx->stream));
        if (labels) HIPCHECK(hipMemcpyAsync(pt.d_labels, labels, (size_t)local_n * 4, hipMemcpyHostToDevice, idx->stream));
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    pt.global_n = global_n;      // local_n == 0: the rank holds only the partition's starts replica
    pt.first_pos = first_pos;
    if (int rc = attach_starts(idx, pt, starts, starts_count, 0, false)) return rc;
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_add_partition_synthetic_shard(qadc_index* idx, uint32_t global_n, uint32_t first_pos, uint32_t local_n,
                                             uint64_t seed, uint32_t starts_count) {
    if (!idx || global_n == 0 || (uint64_t)first_pos + local_n > global_n) return fail(QADC_E_ARG, "bad shard range");
    if ((uint64_t)first_pos * idx->cs % 16 != 0) return fail(QADC_E_ARG, "first_pos must keep the shard 16-byte aligned");
    if (int rc = use_device(idx)) return rc;
    if (int rc = check_labels_mode(idx, false)) return rc;
    Part pt;
    if (local_n) {                                              // (local_n == 0: only the partition's starts replica lives here)
        if (int rc = alloc_part(idx, pt, local_n, false)) return rc;
        launch_fill_codes(pt.d_codes, (uint64_t)first_pos * idx->cs / 8, ((uint64_t)local_n * idx->cs + 7) / 8, seed, idx->stream);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(idx->stream));
    }
    pt.global_n = global_n;
    pt.first_pos = first_pos;
    if (int rc = attach_starts(idx, pt, nullptr, starts_count, seed, true)) return rc;
    idx->parts.push_back(pt);
    idx->finalized = false;
    return QADC_OK;
}

int qadc_index_set_key_base(qadc_index* idx, int part, uint32_t key_base) {
    if (!idx || part < 0 || part >= (int)idx->parts.size()) return fail(QADC_E_ARG, "bad partition");
    idx->parts[part].key_base = key_base;
    if (idx->finalized && (size_t)part < idx->h_partdesc.size()) idx->h_partdesc[part].key_base = key_base;
    if (idx->finalized && idx->d_partdesc.p) {                  // keep the device partition table in step
        if (int rc = use_device(idx)) return rc;
        HIPCHECK(hipMemcpy(reinterpret_cast<unsigned char*>(idx->d_partdesc.p + part) + offsetof(PartDesc, key_base), &key_base,
                           sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHECK(hipDeviceSynchronize());                        // (as in qadc_index_finalize)
    }
    return QADC_OK;
}

int qadc_index_finalize(qadc_index* idx, float keep) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (idx->parts.empty()) return fail(QADC_E_STATE, "no partitions");
    idx->keep = keep;
    for (auto& p : idx->parts) {
        if (p.global_n == 0) {
            p.start_n = 0;
            continue;
        }
        // std::max(1u, static_cast<unsigned>(size * keep)) with unsigned*float -> float (db_query_4.cpp:125-126)
        const float prod = static_cast<float>(p.global_n) * keep;
        const unsigned s = static_cast<unsigned>(prod);
        p.start_n = std::min<uint32_t>(std::max(1u, s), p.global_n);
        if (p.start_n > p.starts_cap)
            return fail(QADC_E_ARG, "shard holds fewer start codes than max(1, unsigned(global_size * keep))");
    }
    // device partition table for the one-workgroup-per-query kernel
    if (int rc = use_device(idx)) return rc;
    std::vector<PartDesc>& pd = idx->h_partdesc;
    pd.assign(idx->parts.size(), PartDesc{});
    idx->max_start_n = 0;
    idx->total_codes = 0;
    idx->total_global_codes = 0;
    idx->max_part_n = 0;
    for (size_t i = 0; i < idx->parts.size(); ++i) {
        const Part& p = idx->parts[i];
        PartDesc& d = pd[i];
        d.codes = p.d_codes;
        d.labels = p.d_labels;
        d.starts = p.d_starts;
        d.n = p.n;
        d.global_n = p.global_n;
        d.first_pos = p.first_pos;
        d.key_base = p.key_base;
        d.start_n = p.start_n;
        d.pad = 0;
        idx->max_start_n = std::max(idx->max_start_n, p.start_n);
        idx->max_part_n = std::max(idx->max_part_n, p.n);
        idx->total_codes += p.n;
        idx->total_global_codes += p.global_n;
    }
    HIPCHECK(idx->d_partdesc.ensure(pd.size()));
    HIPCHECK(hipMemcpy(idx->d_partdesc.p, pd.data(), pd.size() * sizeof(PartDesc), hipMemcpyHostToDevice));
    // (the query kernels run on non-blocking streams, which do not wait for the null stream this copy from pageable memory is issued
    // on; whatever the runtime's staging does, nothing of the table is in flight when the first query is launched)
    HIPCHECK(hipDeviceSynchronize());
    idx->finalized = true;
    return QADC_OK;
}

int qadc_index_partition_count(const qadc_index* idx) { return idx ? (int)idx->parts.size() : 0; }
uint32_t qadc_index_partition_size(const qadc_index* idx, int part) {
    return (idx && part >= 0 && part < (int)idx->parts.size()) ? idx->parts[part].n : 0;
}
uint32_t qadc_index_start_size(const qadc_index* idx, int part) {
    return (idx && part >= 0 && part < (int)idx->parts.size()) ? idx->parts[part].start_n : 0;
}

// The 30 options (include/qadc.h documents each; tests/test_capi_host.py pins this list, tests/test_gpu_fuzz.py draws them)
const char* qadc_option_names(void) {
    return "quant_mode,sum_mode,table_form,profile,"                                       // parity (float half), diagnostics
           "wgq,wgq_group,wgq_group_head,head_level,head_wg,"                              // which path scans a batch, and its head
           "mq,share_variant,variant,front_run_max,wgs_per_item,"                          // level path: launch forms
           "cand_capacity,level_base,level_growth,small_run,prescan_sample,"               // level path: sizes
           "device_replay_nq,device_replay_alone_nq,"                                      // where the heap replay runs
           "wgq_split,wgq_split_codes,wgq_capacity,wgq_cand_cap,"                          // query-kernel path: sizes
           "dist_cap_entries,dist_device_nq,dist_shard_replay,dist_shard_front,dist_inject_failure";   // multi-GPU merge
}

int qadc_set_option(qadc_index* idx, const char* name, double value) {
    if (!idx || !name) return fail(QADC_E_ARG, "bad arguments");
    const std::string n(name);
    if (n.compare(0, 5, "dist_") == 0 && !idx->dist) return fail(QADC_E_STATE, "qadc_dist_init has not been called");
    if (n == "quant_mode") idx->quant_mode = value != 0 ? 1 : 0;
    else if (n == "sum_mode") idx->sum_mode = value != 0 ? 1 : 0;
    else if (n == "table_form") idx->feed.table_form = std::max(0, std::min((int)value, 2));
    else if (n == "profile") idx->profile = value != 0;
    else if (n == "wgq") idx->wgq = (int)value;
    else if (n == "wgq_group") { idx->group.mode = (int)std::max(0.0, std::min(value, 2.0)); idx->group.strikes = 0; }
    else if (n == "wgq_group_head") idx->group.head = idx->group.head_dist = (int)std::max(1.0, std::min(value, 4096.0));
    else if (n == "head_level") idx->head_level = (int)std::max(0.0, std::min(value, (double)(kMaxLevels - 1)));
    else if (n == "head_wg") idx->head_wg = value == 512 ? 512 : 0;
    else if (n == "mq") idx->mq = value != 0;
    else if (n == "share_variant") idx->share_variant = (int)value;
    else if (n == "variant") idx->variant = (int)value;
    else if (n == "front_run_max") idx->front_run_max = (uint64_t)std::max(value, 0.0);
    else if (n == "wgs_per_item") idx->wgs_per_item = (int)value;
    else if (n == "cand_capacity") {
        idx->cand_capacity = (uint32_t)std::max(16.0, std::min(value, 2147483648.0 - 1));
        for (auto& sl : idx->slot) sl.cap_q = 0;
    }
    else if (n == "level_base") idx->level_base = (uint64_t)std::max(16.0, value);
    else if (n == "level_growth") idx->level_growth = (uint64_t)std::max(2.0, value);
    else if (n == "small_run") idx->small_run = (uint32_t)std::max(0.0, value);
    else if (n == "prescan_sample") idx->prescan_sample = (uint32_t)std::max(256.0, value);
    else if (n == "device_replay_nq") idx->device_replay_nq = (int)std::max(value, 0.0);
    else if (n == "device_replay_alone_nq") idx->device_replay_alone_nq = (int)std::max(0.0, value);
    else if (n == "wgq_split") idx->wgq_split = (int)std::max(1.0, std::min(value, 64.0));
    else if (n == "wgq_split_codes") idx->wgq_split_codes = (uint32_t)std::max(value, 1024.0);
    else if (n == "wgq_capacity") {
        idx->wgq_capacity = (uint32_t)std::max(16.0, std::min(value, 1048576.0));
        for (auto& sl : idx->slot) sl.wgq_cap = 0;
    }
    else if (n == "wgq_cand_cap") {                            // both candidate capacities of the query-kernel path
        idx->wgq_cand_cap = (uint32_t)std::max(1.0, std::min(value, (double)kQueryCandCap));          // in-workgroup sort
        idx->group.cand_cap = (uint32_t)std::max(64.0, std::min(value, (double)kOrderCandCap));       // partition-major batches
    }
    else if (n == "dist_cap_entries") idx->dist->cap_entries = (uint32_t)std::max(16.0, std::min(value, 1073741824.0));
    else if (n == "dist_device_nq") idx->dist->device_nq = (int)std::max(1.0, value);
    else if (n == "dist_shard_replay") idx->dist->shard_replay = value != 0;
    else if (n == "dist_shard_front") idx->dist->shard_front = value == 2 ? 2 : value != 0;   // (2: also with a world of one: tests)
    else if (n == "dist_inject_failure") idx->dist->inject_failure = value != 0;              // (tests: this rank's next collect fails before the gather)
    else return fail(QADC_E_ARG, "unknown option: " + n);
    return QADC_OK;
}

int qadc_index_read_codes(qadc_index* idx, int part, uint32_t first, uint32_t count, uint8_t* out) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !out) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if ((uint64_t)first + count > p.n) return fail(QADC_E_ARG, "range outside the partition");
    if (int rc = use_device(idx)) return rc;
    HIPCHECK(hipMemcpy(out, p.d_codes + (size_t)first * idx->cs, (size_t)count * idx->cs, hipMemcpyDeviceToHost));
    return QADC_OK;
}

int qadc_query_scan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R);
}

/* ---- sharded pre-scan (multi-GPU): every rank pre-scans one slice of the starts, the R smallest values per query
 * are gathered by the caller (RCCL), and the batch is submitted with the gathered values in place of its own
 * pre-scan.  The R-th smallest of the union of the per-slice R smallest IS the R-th smallest of all starts, so
 * qmax — and everything after it — is bit-identical to the unsharded path. ---- */
int qadc_prescan_submit(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables, int R,
                        int slice, int nslices) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R, 1, slice, nslices);
}

int qadc_prescan_collect(qadc_index* idx, int slot, float* vals) {
    if (!idx || slot < 0 || slot > 1 || !vals) return fail(QADC_E_ARG, "bad arguments");
    Slot& s = idx->pre_slot[slot];
    if (!s.busy) return fail(QADC_E_STATE, "slot holds no pre-scan");
    if (int rc = use_device(idx)) return rc;
    for (int attempt = 0;; ++attempt) {
        HIPCHECK(hipEventSynchronize(s.ev_done));
        bool overflow = false;
        for (int q = 0; q < s.nq; ++q) overflow |= (s.h_export_flags[q] & 8u) != 0;
        if (!overflow) break;
        if (attempt >= 1) {
            s.busy = false;
            return fail(QADC_E_CAPACITY, "pre-scan survivor buffer overflow persists");
        }
        s.full_prescan = true;               // adversarially ordered starts: evaluate the slice unfiltered
        idx->prof.regrows++;
        if (int rc = plan_and_launch(idx, s)) {
            s.busy = false;
            return rc;
        }
    }
    s.busy = false;
    if (idx->profile && s.prof_used >= 2) {
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, s.prof_ev[0], s.prof_ev[1]));
        idx->prof.start_ms += ms;
        idx->prof.start_codes += s.start_codes;
    }
    std::memcpy(vals, s.h_export, sizeof(float) * (size_t)s.nq * s.R);
    return QADC_OK;
}

int qadc_query_scan_submit_prescanned(qadc_index* idx, int slot, int nq, int ma, const int32_t* assign, float* tables,
                                      int R, const float* prescan_vals, int nvals) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    return submit_common(idx, slot, nq, ma, assign, tables, nullptr, R, 2, 0, 1, prescan_vals, nvals);
}

int qadc_query_scan_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                            float* qmin, float* qmax, int8_t* qtables) {
    if (int rc = collect_common(idx, slot, /*need_stream=*/false)) return rc;
    Slot& s = idx->slot[slot];
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(s.nq);
        status = st_local.data();
    }
    finish_float_outputs(idx, s, status, qmin, qmax);
    if (qtables) {
        HIPCHECK(hipMemcpyAsync(qtables, s.d_qt, (size_t)s.nq * s.ma * idx->M * 16, hipMemcpyDeviceToHost,
                                idx->copy_stream));
        HIPCHECK(hipStreamSynchronize(idx->copy_stream));
    }
    return replay_outputs(idx, s, keys, values, sizes, status);
}

static int copy_stream(Slot& s, uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                       uint16_t* cand_slots = nullptr);

int qadc_query_scan_collect_candidates(qadc_index* idx, int slot, uint64_t cand_capacity, uint32_t* cand_keys,
                                       int8_t* cand_vals, uint16_t* cand_slots, uint64_t* offsets, int32_t* status,
                                       float* qmin, float* qmax) {
    if (!idx || slot < 0 || slot >= kSlots) return fail(QADC_E_ARG, "bad arguments");
    Slot& s = idx->slot[slot];
    if (s.busy) {
        if (int rc = collect_common(idx, slot)) return rc;
        s.has_result = true;
    } else if (!s.has_result) {
        return fail(QADC_E_STATE, "slot holds no batch");
    }
    finish_float_outputs(idx, s, status, qmin, qmax);
    // QADC_E_CAPACITY keeps the result: call again with buffers of offsets[nq] entries
    const int rc = copy_stream(s, cand_capacity, cand_keys, cand_vals, offsets, cand_slots);
    if (rc == QADC_OK) s.has_result = false;
    return rc;
}

int qadc_query_scan(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R, uint32_t* keys,
                    int8_t* values, int32_t* sizes, int32_t* status, float* qmin, float* qmax, int8_t* qtables) {
    if (int rc = qadc_query_scan_submit(idx, 0, nq, ma, assign, tables, R)) return rc;
    return qadc_query_scan_collect(idx, 0, keys, values, sizes, status, qmin, qmax, qtables);
}

static int copy_stream(Slot& s, uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                       uint16_t* cand_slots) {
    if (!offsets) return fail(QADC_E_ARG, "offsets is null");
    std::memcpy(offsets, s.out_off.data(), sizeof(uint64_t) * (s.nq + 1));
    const uint64_t total = s.out_off[s.nq];
    if (total > cand_capacity) return fail(QADC_E_CAPACITY, "candidate output buffers too small (see offsets[nq])");
    for (uint64_t i = 0; i < total; ++i) {
        if (cand_keys) cand_keys[i] = (uint32_t)s.out_entries[i];
        if (cand_vals) cand_vals[i] = (int8_t)(s.out_entries[i] >> 32);
        if (cand_slots) cand_slots[i] = (uint16_t)((s.out_entries[i] >> 40) & 0x3fffu);
    }
    return QADC_OK;
}

int qadc_query_scan_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, float* tables, int R,
                               uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets,
                               int32_t* status, float* qmin, float* qmax) {
    if (!tables) return fail(QADC_E_ARG, "tables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, tables, nullptr, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    Slot& s = idx->slot[0];
    std::vector<int32_t> st_local(nq);
    finish_float_outputs(idx, s, st_local.data(), qmin, qmax);
    if (status) std::memcpy(status, st_local.data(), sizeof(int32_t) * nq);
    for (int q = 0; q < nq; ++q)
        if (st_local[q]) {  // the reference never reaches the scan for such a query
            // keep offsets monotone but expose no candidates: handled by the caller through status
        }
    return copy_stream(s, cand_capacity, cand_keys, cand_vals, offsets);
}

int qadc_scan_i8(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R, uint32_t* keys,
                 int8_t* values, int32_t* sizes) {
    if (!qtables) return fail(QADC_E_ARG, "qtables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, nullptr, qtables, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    return replay_outputs(idx, idx->slot[0], keys, values, sizes, nullptr);
}

int qadc_scan_i8_candidates(qadc_index* idx, int nq, int ma, const int32_t* assign, const int8_t* qtables, int R,
                            uint64_t cand_capacity, uint32_t* cand_keys, int8_t* cand_vals, uint64_t* offsets) {
    if (!qtables) return fail(QADC_E_ARG, "qtables is null");
    if (int rc = submit_common(idx, 0, nq, ma, assign, nullptr, qtables, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    return copy_stream(idx->slot[0], cand_capacity, cand_keys, cand_vals, offsets);
}

int qadc_scan_start(qadc_index* idx, int nq, int ma, const int32_t* assign, const float* tables, int R, float* qmax) {
    if (!idx || !tables || !qmax || !assign || nq <= 0 || ma <= 0 || R <= 0) return fail(QADC_E_ARG, "bad arguments");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    // Runs the float chain of a batch and reads back qmax only (copy: the caller's tables stay untouched).
    std::vector<float> copy(tables, tables + (size_t)nq * ma * idx->M * 16);
    std::vector<float> qm(nq);
    if (int rc = submit_common(idx, 0, nq, ma, assign, copy.data(), nullptr, R)) return rc;
    if (int rc = collect_common(idx, 0)) return rc;
    for (int q = 0; q < nq; ++q) qmax[q] = idx->slot[0].h_qout[(size_t)q * (idx->slot[0].wgq ? idx->slot[0].wgq_G : 1)].qmax;
    return QADC_OK;
}

int qadc_replay_i8(uint64_t n, const uint32_t* keys, const int8_t* vals, int R, int push_sentinel, uint32_t* out_keys,
                   int8_t* out_vals, int32_t* out_size) {
    if (R <= 0 || (n && (!keys || !vals)) || !out_size) return fail(QADC_E_ARG, "bad arguments");
    kv_heap<uint32_t, int8_t> bh(R);
    if (push_sentinel) bh.push(0, 127);
    for (uint64_t i = 0; i < n; ++i) bh.push(keys[i], vals[i]);
    *out_size = bh.size();
    if (out_keys) std::memcpy(out_keys, bh.keys(), sizeof(uint32_t) * bh.size());
    if (out_vals) std::memcpy(out_vals, bh.values(), bh.size());
    return QADC_OK;
}

int qadc_sort_keys_i8(int size, const uint32_t* heap_keys, const int8_t* heap_vals, uint32_t* out_keys) {
    if (size < 0 || (size && (!heap_keys || !heap_vals || !out_keys))) return fail(QADC_E_ARG, "bad arguments");
    kv_heap<uint32_t, int8_t> bh(std::max(size, 1));
    bh.assign(heap_keys, heap_vals, size);
    bh.sort_keys(out_keys);
    return QADC_OK;
}

int qadc_candidates_i8(qadc_index* idx, int part, const int8_t* qtable, int8_t* out) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !qtable || !out) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if (p.n == 0) return QADC_OK;
    for (int i = 0; i < idx->M * 16; ++i)
        if (qtable[i] < 0) return fail(QADC_E_ARG, "int8 tables must lie in [0,127]");
    if (int rc = use_device(idx)) return rc;
    int8_t *d_t = nullptr, *d_o = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_t), idx->M * 16));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_o), p.n));
    HIPCHECK(hipMemcpyAsync(d_t, qtable, idx->M * 16, hipMemcpyHostToDevice, idx->stream));
    launch_candidates_i8(idx->M, p.d_codes, p.n, d_t, d_o, idx->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(take_launch_error());
    HIPCHECK(hipMemcpyAsync(out, d_o, p.n, hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    HIPCHECK(hipFree(d_t));
    HIPCHECK(hipFree(d_o));
    return QADC_OK;
}

int qadc_float_top1(qadc_index* idx, int part, const float* table, uint32_t* out_key, uint32_t* out_pos, float* out_dist) {
    if (!idx || part < 0 || part >= (int)idx->parts.size() || !table) return fail(QADC_E_ARG, "bad arguments");
    const Part& p = idx->parts[part];
    if (p.n == 0) return fail(QADC_E_ARG, "empty partition");
    if (int rc = use_device(idx)) return rc;
    const int blocks = (int)std::min<uint32_t>((p.n + 255) / 256, 2048);
    float *d_t = nullptr, *d_v = nullptr;
    uint32_t* d_p = nullptr;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_t), idx->M * 16 * sizeof(float)));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_v), blocks * sizeof(float)));
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&d_p), blocks * sizeof(uint32_t)));
    HIPCHECK(hipMemcpyAsync(d_t, table, idx->M * 16 * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    launch_float_top1(idx->M, idx->sum_mode, p.d_codes, p.n, d_t, d_v, d_p, blocks, idx->stream);
    HIPCHECK(hipGetLastError());
    std::vector<float> hv(blocks);
    std::vector<uint32_t> hp(blocks);
    HIPCHECK(hipMemcpyAsync(hv.data(), d_v, blocks * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipMemcpyAsync(hp.data(), d_p, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(hipStreamSynchronize(idx->stream));
    int b = 0;
    for (int i = 1; i < blocks; ++i)
        if (hv[i] < hv[b] || (hv[i] == hv[b] && hp[i] < hp[b])) b = i;
    uint32_t key = p.key_base + p.first_pos + hp[b];
    if (p.d_labels) HIPCHECK(hipMemcpy(&key, p.d_labels + hp[b], sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (out_key) *out_key = key;
    if (out_pos) *out_pos = hp[b];
    if (out_dist) *out_dist = hv[b];
    HIPCHECK(hipFree(d_t));
    HIPCHECK(hipFree(d_v));
    HIPCHECK(hipFree(d_p));
    return QADC_OK;
}

int qadc_slot_assign(qadc_index* idx, int slot, int32_t* assign_out) {
    if (!idx || slot < 0 || slot >= kSlots || !assign_out) return fail(QADC_E_ARG, "bad arguments");
    const Slot& s = idx->slot[slot];
    if (s.busy) return fail(QADC_E_STATE, "collect the batch first");
    if (s.assign.size() != (size_t)s.nq * s.ma) return fail(QADC_E_STATE, "slot has held no batch");
    std::memcpy(assign_out, s.assign.data(), sizeof(int32_t) * s.assign.size());
    return QADC_OK;
}

int qadc_slot_qtables(qadc_index* idx, int slot, int q_first, int q_count, int8_t* out) {
    if (!idx || slot < 0 || slot >= kSlots || !out || q_first < 0 || q_count < 0) return fail(QADC_E_ARG, "bad arguments");
    const Slot& s = idx->slot[slot];
    if (s.busy) return fail(QADC_E_STATE, "collect the batch first");
    if (!s.d_qt || s.nq <= 0) return fail(QADC_E_STATE, "slot has held no batch");
    if ((int64_t)q_first + q_count > s.nq) return fail(QADC_E_ARG, "query range outside the batch");
    if (int rc = use_device(idx)) return rc;
    const size_t per_q = (size_t)s.ma * idx->M * 16;
    HIPCHECK(hipMemcpyAsync(out, s.d_qt + (size_t)q_first * per_q, (size_t)q_count * per_q, hipMemcpyDeviceToHost, idx->copy_stream));
    HIPCHECK(hipStreamSynchronize(idx->copy_stream));
    return QADC_OK;
}

int qadc_profile_read(qadc_index* idx, qadc_profile* out) {
    if (!idx || !out) return fail(QADC_E_ARG, "bad arguments");
    *out = idx->prof;
    return QADC_OK;
}

int qadc_profile_reset(qadc_index* idx) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    idx->prof = qadc_profile{};
    return QADC_OK;
}

}  // extern "C"
