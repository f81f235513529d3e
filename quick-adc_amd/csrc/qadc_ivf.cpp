// One workgroup per query (IVF batches, small lists), the partition-major second phase of large IVF batches, and the
// device-side feeders behind qadc_search: index_db::assign_compute_residuals (databases.hpp:201-211),
// the float distance tables (distances.hpp:277-311) and the whole scanner_4::query_scan chain (db_query_4.cpp:245-309)
// of a batch without a host round trip.  State: qadc_index::feed (FeederState), qadc_index::group (GroupState).
#include "qadc_host.h"

using namespace qadc;
using namespace qadc::host;

namespace qadc {
namespace host {

// Which float-table form qadc_search builds (option "table_form"): 0 = direct ||x - c||^2 always
// (compute_dists_single_simd_cg, distances.hpp:294-311), 1 = BLAS expansion always (nns_engine_batch,
// query_common.hpp:194-213), 2 = the rule of nns_engine (query_common.hpp:292-297): direct for ma == 1, expansion otherwise.
int table_expansion(const qadc_index* idx, int ma) {
    return idx->feed.table_form == 1 || (idx->feed.table_form == 2 && ma > 1);
}

// Decides whether a batch takes the one-workgroup-per-query path.  codes_per_query: exact maximum when the host
// knows assign[], an estimate (ma x mean partition size) when assign[] is produced on the GPU.
// Slices of the sliced front (lone_front_kernel) for a lone query whose one probed partition has `starts` start codes; 0 = it does
// not apply (a single slice: the walk's own front does as well; more than 64 slices or S x R keys beyond what the last workgroup ranks)
uint32_t lone_front_slices(uint32_t starts, int R) {
    const uint32_t S = (starts + kLoneFrontSlice - 1) / kLoneFrontSlice;
    return S >= 2 && S <= 64 && (uint64_t)S * (uint64_t)R <= (uint64_t)kLoneFrontMaxKeys && R <= kLoneFrontSlice ? S : 0u;
}

bool wgq_eligible(const qadc_index* idx, int nq, int ma, int R, int mode, uint64_t codes_per_query, bool host_float_tables, bool alone) {
    if (idx->wgq == 0 || mode != 0 || ma > 4096 || R <= 0) return false;
    if (idx->wgq >= 2) return true;
    if (codes_per_query <= kWgqSmallCodes) return true;
    // a synchronous query or two on a longer list: the level path answers with a dependent chain of ~10 launches (front, head, two
    // or three levels, sort), the query kernel with ONE launch of 32 workgroups per query whose chunks refresh their bounds as they
    // go.  Same box, one query, flat list (profiles/r06_lone_query_latency_ab.txt): 3 x 10^5 codes 105 -> 38 us, 10^6 138 -> 63,
    // 2 x 10^6 148 -> 82, 4 x 10^6 163 -> 122, 6 x 10^6 177 -> 151; at 10^7 the level path is ahead again (203 against 224)
    if (alone && nq <= 2 && codes_per_query <= kWgqLoneCodes) return true;
    // ... and ONE query on one long partition, float tables from the caller: with the front sliced over workgroups of its own in a
    // launch in front of the walk (lone_front_kernel) the query kernel stays ahead of the level path far beyond that — same box:
    // 10^7 codes 203 -> 136 us (BASELINE configs[1], a synchronous query_scan), 2 x 10^7 229 -> 198
    if (alone && nq == 1 && ma == 1 && host_float_tables && !idx->dist && !idx->profile && codes_per_query <= kWgqLoneSlicedCodes &&
        lone_front_slices((uint32_t)std::max(1.0f, (float)codes_per_query * idx->keep), R))
        return true;
    // an IVF batch (several partitions, several probes per query: the queries walk different codes) — the query kernel is
    // ahead of the level path at every batch size (C3 shape, synchronous: 1 query 217 -> 175 us, 64 queries 0.85 -> 0.73 ms)
    if (ma > 1 && idx->parts.size() > 1 && codes_per_query <= kWgqMaxCodes) return true;
    return nq >= kWgqMinNq && codes_per_query <= kWgqMaxCodes;
}

// One launch per batch: scan_query_kernel (one workgroup per query), then — for batches large enough to replay on
// the device — replay_heap_wave_kernel on the side stream.  No host planning: the kernel walks assign[] and the
// device partition table itself.
int launch_wgq_batch(qadc_index* idx, Slot& s) {
    ScopedMs timer(idx->prof.host_plan_ms);
    const int M = idx->M, nq = s.nq, ma = s.ma;
    const size_t table_dim = (size_t)M * 16, nt = (size_t)nq * ma * table_dim;
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    s.launches.clear();
    s.start_codes = 0;
    // ---- upload block: [assign i32 nq*ma (host-assign path)][float or int8 tables (host-table path)] ----
    const size_t assign_bytes = s.assign_on_device ? 0 : sizeof(int32_t) * (size_t)nq * ma;
    const size_t off_tables = align16(assign_bytes);
    const size_t tables_bytes = s.float_path ? (s.device_tables ? 0 : nt * sizeof(float)) : nt;
    const size_t in_bytes = align16(off_tables + tables_bytes);
    HIPCHECK(s.h_in.ensure(std::max<size_t>(in_bytes, 16)));
    HIPCHECK(s.d_in.ensure(std::max<size_t>(in_bytes, 16)));
    auto fill_upload = [&]() {
        if (assign_bytes) std::memcpy(s.h_in.p, s.assign.data(), assign_bytes);
        if (s.float_path && !s.device_tables) std::memcpy(s.h_in.p + off_tables, s.tables, nt * sizeof(float));
        if (!s.float_path) std::memcpy(s.h_in.p + off_tables, s.qtables_in.data(), nt);
    };

    // ---- result block in pinned, device-mapped host memory: [QueryOut[nq]][streams u64[nq][cap] unless they stay
    // on the device][heaps u64[nq][R]][sizes u32[nq]] ----
    const uint32_t cap = s.wgq_cap;
    bool alone = true;
    for (int i = 0; i < kSlots; ++i) alone = alone && (&idx->slot[i] == &s || !idx->slot[i].busy);
    alone = alone && !idx->pre_slot[0].busy && !idx->pre_slot[1].busy;
    // The lane-per-query replay takes ~1.3 ms whatever the batch size (one query's pushes are sequential): in a pipeline
    // that latency hides under the next batches and the host stays free, but a batch submitted while nothing else is in
    // flight — a synchronous call — is answered sooner by the host's threads up to a few hundred queries (C3 shape,
    // synchronous: 64 queries 1.41 -> 0.87 ms, 256: 1.89 -> 1.44, 512: 2.26 vs 2.41)
    const int replay_from = alone ? std::max(idx->device_replay_nq, idx->device_replay_alone_nq) : idx->device_replay_nq;
    s.dev_replay = idx->device_replay_nq > 0 && nq >= replay_from && (uint32_t)s.R <= replay_wave_max_R();
    s.dist_batch = idx->dist != nullptr;
    s.heaps_ready = s.dev_replay && !s.dist_batch;
    if (s.dist_batch) s.dev_replay = true;                   // streams stay on the device for the gather (qadc_dist_collect)
    // A batch too small to fill the GPU splits every query's scan order over G workgroups (each tightens its bound on
    // the query's first block, then scans its own chunk); the sub-streams are concatenated in workgroup order.
    int G = 1;
    if (!s.dev_replay && nq * 2 <= 256) {
        // A query or two have the GPU to themselves and latency decides: up to wgq_split (32) workgroups each, down to
        // wgq_split_codes (2048) codes per workgroup — 10^5 codes, one query: 12 workgroups 36.1 us, 32: 32.6.  From three queries
        // on every further workgroup repeats a front for a shorter chunk while the host's replay of the queries, one after the
        // other, sets the call's time: at most kSplitBatch workgroups per query, four times the codes each (same box, 10^5 codes,
        // 32 against 12 workgroups: 2 queries 35.8 / 37.0 us, 4: 52.9 / 50.0, 8: 71 / 67, 16: 122 / 97)
        const bool few = nq <= 2;
        // (from 16 Mi codes on the walk outweighs what further workgroups repeat — with the front sliced off: 2 x 10^7 codes 194 -> 163 us
        //  at 64 workgroups, 2.5 x 10^7 220 -> 179; below, 32 and 64 are within 5 % of each other either way)
        const int split = few && s.wgq_codes >= (16ull << 20) ? std::min(2 * idx->wgq_split, kMaxSplit) : idx->wgq_split;
        G = std::min<int>(few ? split : std::min(idx->wgq_split, kSplitBatch), 256 / nq);
        G = (int)std::min<uint64_t>((uint64_t)G, s.wgq_codes / ((uint64_t)idx->wgq_split_codes * (few ? 1u : 4u)));
        G = std::max(G, 1);
    }
    s.wgq_G = G;
    const int nsub = nq * G;
    const size_t stream_entries = (size_t)nsub * cap;
    if (stream_entries >= (1ull << 32)) return fail(QADC_E_CAPACITY, "candidate stream capacity exceeds 2^32 entries");
    s.out_cap = (uint32_t)stream_entries;
    const size_t host_stream_bytes = s.dev_replay ? 0 : sizeof(uint64_t) * stream_entries;
    const size_t off_heaps = sizeof(QueryOut) * (size_t)nsub + host_stream_bytes;
    const size_t heaps_bytes = s.dev_replay ? (sizeof(uint64_t) * (size_t)s.R + sizeof(uint32_t)) * (size_t)nq : 0;
    HIPCHECK(s.h_result.ensure(off_heaps + heaps_bytes + 16, hipHostMallocMapped | hipHostMallocCoherent));
    if (s.h_result.p != s.h_result_mapped) {
        HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_result_mapped), s.h_result.p, 0));
        s.h_result_mapped = s.h_result.p;
    }
    unsigned char* d_result = s.d_result_mapped;
    s.d_qout = reinterpret_cast<QueryOut*>(d_result);
    s.h_qout = reinterpret_cast<QueryOut*>(s.h_result.p);
    s.d_entries = reinterpret_cast<uint64_t*>(d_result + sizeof(QueryOut) * (size_t)nsub);
    s.h_entries = reinterpret_cast<uint64_t*>(s.h_result.p + sizeof(QueryOut) * (size_t)nsub);
    s.h_heaps = reinterpret_cast<uint64_t*>(s.h_result.p + off_heaps);
    s.h_heap_sizes = reinterpret_cast<uint32_t*>(s.h_result.p + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
    if (s.dev_replay) {
        HIPCHECK(s.d_stream.ensure(stream_entries));
        HIPCHECK(s.d_qflags.ensure((size_t)nq * 4));
    }
    HIPCHECK(s.d_qtables.ensure(nt));
    // pre-scan values beyond the kernel's LDS budget go to a global scratch
    uint64_t fcap = 0;
    if (s.float_path) {
        uint64_t worst = (uint64_t)ma * idx->max_start_n;
        if (!s.assign_on_device) {
            worst = 0;
            for (int q = 0; q < nq; ++q) {
                uint64_t t = 0;
                for (int a = 0; a < ma; ++a) t += idx->parts[s.assign[(size_t)q * ma + a]].start_n;
                worst = std::max(worst, t);
            }
        }
        if (worst > query_kernel_lds_values(M)) fcap = worst;
        for (int q = 0; q < nq && !s.assign_on_device; ++q)
            for (int a = 0; a < ma; ++a) s.start_codes += idx->parts[s.assign[(size_t)q * ma + a]].start_n;
    }
    s.wgq_fcap = fcap;
    if (fcap) HIPCHECK(s.d_fvals.ensure((size_t)nsub * fcap));
    const uint32_t ccap = std::min<uint32_t>(idx->wgq_cand_cap, kQueryCandCap);
    HIPCHECK(s.d_qcands.ensure((size_t)nsub * ccap));

    hipStream_t st = idx->wgq_stream;
    // A lone small query (the synchronous single-query call): its input — which partitions, their descriptors, the
    // float tables — fits the kernel-argument segment and rides in the dispatch packet; no copy precedes the launch.
    alignas(16) unsigned char inl[kInlineBytes];
    size_t inl_bytes = 0, inl_off_parts = 0, inl_off_tables = 0;
    // The sliced front (lone_front_kernel): a lone query on ONE long partition whose starts fill at least two slices.  Its walk
    // launch then carries no float tables (the int8 table and {flags, qmin, qmax} come from the front launch in front of it).
    uint32_t lf_slices = 0;
    if (alone && G > 1 && nq == 1 && ma == 1 && s.float_path && !s.device_tables && !s.assign_on_device && !idx->profile && !s.dev_replay) {
        lf_slices = lone_front_slices(idx->parts[s.assign[0]].start_n, s.R);
    }
    if (alone && G > 1 && s.float_path && !s.device_tables && !s.assign_on_device) {
        const size_t na = (size_t)nq * ma;
        inl_off_parts = align16(sizeof(int32_t) * na);
        inl_off_tables = lf_slices ? 0 : align16(inl_off_parts + sizeof(PartDesc) * na);
        const size_t total = lf_slices ? inl_off_parts + sizeof(PartDesc) * na : inl_off_tables + nt * sizeof(float);
        if (total <= kInlineBytes) {
            for (size_t i = 0; i < na; ++i) {
                reinterpret_cast<int32_t*>(inl)[i] = (int32_t)i;
                std::memcpy(inl + inl_off_parts + sizeof(PartDesc) * i, &idx->h_partdesc[s.assign[i]], sizeof(PartDesc));
            }
            if (!lf_slices) std::memcpy(inl + inl_off_tables, s.tables, nt * sizeof(float));
            inl_bytes = total;
        }
    }
    if (lf_slices && !inl_bytes) lf_slices = 0;
    if (!inl_bytes) fill_upload();                               // (an inline query's input rides in the dispatch packet instead)
    if (in_bytes && !inl_bytes) {
        if (alone) {
            HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, st));
        } else {                                            // never queue a copy behind the previous batch's kernels
            HIPCHECK(hipMemcpyAsync(s.d_in.p, s.h_in.p, in_bytes, hipMemcpyHostToDevice, idx->copy_stream));
            if (!s.ev_up) HIPCHECK(hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_up, idx->copy_stream));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_up, 0));
        }
    }
    s.prof_used = 0;
    // What a pipelined batch needs BEFORE its first scan launch but from nobody on the scan stream — the float tables (built
    // from assign[], which the copy stream's coarse kernels produce), the state clear, the partition-major plan (count /
    // offsets / scatter over assign[]) — is enqueued on the stream that produces assign[]: the copy stream, or the
    // collectives' stream behind the unpack of a sharded front.  On the scan stream those ~6 short launches cost it ~0.1 ms
    // per batch (C3 shape: a seventh of the batch) between the previous batch's ordering pass and this batch's head.
    // (Round 3 tried the front stream for this and lost: it shared a pipe with the scan stream then, and carries every other
    // replay — DESIGN.md section 5.)  A lone batch and a re-run keep everything on one stream.
    const bool pre = !alone && !s.rerun;
    hipStream_t pre_st = pre ? idx->copy_stream : st;
    bool pre_used = false, flush_later = false;
    QueryKernelArgs A{};
    A.parts = idx->d_partdesc.p;
    A.assign = s.assign_on_device ? s.d_assign.p : reinterpret_cast<const int32_t*>(s.d_in.p);
    A.ma = ma;
    A.ftables = nullptr;
    A.qtables = s.d_qtables.p;
    if (s.front_sharded) {
        // (tables come out of the sharded front below)
    } else if (s.float_path) {
        if (s.device_tables) {
            HIPCHECK(s.d_ftables.ensure(nt));
            HIPCHECK(hipStreamWaitEvent(pre_st, s.ev_feed, 0));
            launch_build_tables(s.d_queries.p, idx->feed.K ? idx->feed.d_coarse.p : nullptr, s.d_assign.p, idx->feed.d_codebooks.p,
                                idx->feed.has_rotation ? idx->feed.d_rotation.p : nullptr, nq, ma, M, idx->feed.dim, table_expansion(idx, ma), idx->sum_mode, s.d_ftables.p, pre_st);
            pre_used = pre_st != st;
            A.ftables = s.d_ftables.p;
        } else {
            A.ftables = reinterpret_cast<float*>(s.d_in.p + off_tables);
        }
    } else {
        A.qtables = reinterpret_cast<int8_t*>(s.d_in.p + off_tables);    // the caller's int8 tables, as uploaded
    }
    if (s.front_sharded) {
        // ---- sharded front: this rank's share -> gather -> the whole batch as an int8 batch with device-resident inputs ----
        DistState& d = *idx->dist;
        const size_t tab = table_dim * (size_t)ma;
        const size_t block = ((size_t)s.front_per * (tab + (size_t)ma * 4 + 16) + 15) & ~(size_t)15;
        int32_t* d_assign_share = reinterpret_cast<int32_t*>(s.d_fblock.p + (size_t)s.front_per * tab);
        uint32_t* d_front_share = reinterpret_cast<uint32_t*>(s.d_fblock.p + (size_t)s.front_per * (tab + (size_t)ma * 4));
        if (!s.rerun) {                                          // (a re-run from inside collect reuses the gathered arrays)
            idx->prof.front_sharded_batches++;
            // The share's front runs on the FRONT stream, under the previous batch's scan (it depends on nothing that batch
            // produces), and its gather is issued IN FRONT of that batch's merge gather (flush_merges below).
            hipStream_t fs = alone ? st : idx->front_stream;
            HIPCHECK(hipStreamWaitEvent(fs, s.ev_feed, 0));
            if (s.front_n) {
                const size_t nt_share = (size_t)s.front_n * tab;
                HIPCHECK(s.d_ftables.ensure(nt_share));
                launch_build_tables(s.d_queries.p, idx->feed.d_coarse.p, d_assign_share, idx->feed.d_codebooks.p,
                                    idx->feed.has_rotation ? idx->feed.d_rotation.p : nullptr, s.front_n, ma, M, idx->feed.dim, table_expansion(idx, ma),
                                    idx->sum_mode, s.d_ftables.p, fs);
                QueryKernelArgs F{};
                F.parts = idx->d_partdesc.p;
                F.assign = d_assign_share;
                F.ma = ma;
                F.ftables = s.d_ftables.p;
                F.qtables = reinterpret_cast<int8_t*>(s.d_fblock.p);
                F.fvals = fcap ? s.d_fvals.p : nullptr;
                F.fcap = (uint32_t)fcap;
                F.R = (uint32_t)s.R;
                F.quant_mode = idx->quant_mode;
                F.sum_mode = idx->sum_mode;
                F.head_codes = ~0ull;
                F.head_slots = 1;
                F.G = 1;
                F.front_only = 1;
                F.front_out = d_front_share;
                HIPCHECK(launch_scan_query(M, s.front_n, F, fs));
            }
            // the collectives of the merge live on ONE stream, in the order the host issues them (the same on every rank)
            if (!s.ev_fa) HIPCHECK(hipEventCreateWithFlags(&s.ev_fa, hipEventDisableTiming));
            if (!s.ev_fb) HIPCHECK(hipEventCreateWithFlags(&s.ev_fb, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_fa, fs));
            HIPCHECK(hipStreamWaitEvent(d.stream, s.ev_fa, 0));
            std::string gerr;
            if (d.gather(s.d_fblock.p, s.d_fgathered.p, block / 8, d.stream, gerr)) return fail(QADC_E_HIP, gerr);
            HIPCHECK(launch_front_unpack(s.d_fgathered.p, block, d.world, s.front_per, nq, ma, tab, s.d_qtables.p, s.d_assign.p,
                                         s.d_front_all.p, reinterpret_cast<int32_t*>(s.d_fmap),
                                         reinterpret_cast<uint32_t*>(s.d_fmap + (size_t)nq * ma * 4), d.stream));
            HIPCHECK(hipEventRecord(s.ev_fb, d.stream));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_fb, 0));
            if (pre) pre_st = d.stream;                          // (the plan follows the unpack on the collectives' stream)
            flush_later = true;                                  // the older batches' merges: behind this front gather (and this plan)
        }
        A.assign = s.d_assign.p;
        A.ftables = nullptr;                                     // from here on: an int8 batch
        A.qtables = s.d_qtables.p;
        A.front_in = s.d_front_all.p;
    }
    s.d_qt = A.qtables;
    A.fvals = fcap ? s.d_fvals.p : nullptr;
    A.fcap = (uint32_t)fcap;
    A.stream = s.dev_replay ? s.d_stream.p : s.d_entries;
    A.cap = cap;
    A.cands = s.d_qcands.p;
    A.ccap = ccap;
    A.qout = s.d_qout;
    A.qstate_flags = s.dev_replay ? s.d_qflags.p : nullptr;
    A.R = (uint32_t)s.R;
    A.quant_mode = idx->quant_mode;
    A.sum_mode = idx->sum_mode;
    A.nontemporal = idx->total_codes * (uint64_t)idx->cs > (200ull << 20);   // (same rule as the level path)
    A.G = G;
    A.pos_bits = (idx->max_part_n ? 64u - (uint32_t)__builtin_clzll((unsigned long long)idx->max_part_n) : 0u) << 16 |
                 256u;
    if (idx->profile) HIPCHECK(prof_event(s, st));
    s.poll = alone && G > 1 && !s.dev_replay && !idx->profile;
    if (s.poll)
        for (int i = 0; i < nsub; ++i) s.h_qout[i].flags = 0;
    A.inline_off_parts = (uint32_t)inl_off_parts;
    A.inline_off_tables = (uint32_t)inl_off_tables;
    // Large IVF batches: a (query, probe) pair lands on a partition several other queries of the batch probe too.  The
    // kernel then only walks the first probes of every query (head: front + a tight bound); the other pairs are
    // regrouped by partition on the device and scanned 8 queries per pass (see launch_ivf_plan), and a third kernel
    // orders every query's candidates into the stream layout the plain launch produces.
    // (under the merge the head counts probes WITH CODES ON THIS RANK; its length is an option of its own: whole-partition
    // placement is better off with 2 — one of 8 ranks, C5 shape: 1.53 vs 1.78 ms per batch — the range split with 4:
    // 1.35 vs 1.52, and at 2048-query batches a head of 2 short pieces bounds too loosely and the batches fall back)
    const int head_slots = std::min(idx->dist ? idx->group.head_dist : idx->group.head, ma);
    const size_t pairs = (size_t)nq * (size_t)(ma - head_slots);
    const size_t nparts = idx->parts.size();
    s.wgq_grouped = s.dev_replay && G == 1 && pairs > 0 && nparts < (1u << 24) &&
                    (idx->group.mode == 2 || (idx->group.mode == 1 && idx->group.strikes < 2 && nq >= 256 && pairs >= 2 * nparts));
    if (s.wgq_grouped) {
        const size_t state_bytes = 64 + sizeof(QueryState) * (size_t)nq;
        const size_t ngroups = ivf_max_groups(pairs, nparts);
        if (ngroups * 8 >= (1ull << 31)) return fail(QADC_E_CAPACITY, "too many (query, probe) pairs for one batch");
        // query states, plan counters and group items in ONE allocation: one clear instead of three (every launch between
        // two batches' scans costs the scan stream ~10 us)
        auto up256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t gplan_off = up256(state_bytes), gplan_bytes = sizeof(uint32_t) * (3 * nparts + 1);
        const size_t gitems_off = up256(gplan_off + gplan_bytes), gitems_bytes = sizeof(ScanItem) * ngroups * 8;
        HIPCHECK(s.d_state.ensure(gitems_off + gitems_bytes));
        uint32_t* d_gplan = reinterpret_cast<uint32_t*>(s.d_state.p + gplan_off);
        ScanItem* d_gitems = reinterpret_cast<ScanItem*>(s.d_state.p + gitems_off);
        // (the ordering pass of this path sorts up to kOrderCandCap candidates per query — twice what the query kernel's own tail
        // takes; a cap the caller lowered — the tests' way to force the fallback — is honoured)
        const uint32_t gcap = idx->wgq_cand_cap < kQueryCandCap ? idx->wgq_cand_cap : std::min<uint32_t>(idx->group.cand_cap, kOrderCandCap);
        HIPCHECK(s.d_cands.ensure((size_t)nq * gcap));
        s.d_hdr = reinterpret_cast<CandHeader*>(s.d_state.p);
        s.d_qs = reinterpret_cast<QueryState*>(s.d_state.p + 64);
        HIPCHECK(hipMemsetAsync(s.d_state.p, 0, gitems_off + gitems_bytes, pre_st));
        // (tried: the plan — two clears + count / offsets / scatter, needed by the second phase only — on a stream of its own
        // under the head launch: its workgroups then wait for head workgroups to retire and the second phase for them;
        // C3 0.75 -> 1.05 us per query, one of 8 ranks 0.72 -> 1.51 ms per batch.  Tried as well: table build, clears and
        // plan of a qadc_search batch on the front stream, enqueued under the PREVIOUS batch's scan — ~100 us of the scan
        // stream per batch to win, but the dozen small launches trickle through that scan so slowly that the next head
        // ends up waiting for them: C3 0.75 -> 0.91 us per query, C5 4.63 -> 4.82.)
        launch_ivf_plan(A.assign, idx->d_partdesc.p, nq, ma, head_slots, (int)nparts, d_gplan, d_gplan + 2 * nparts,
                        d_gplan + nparts, d_gitems, pre_st);
        pre_used = pre_used || pre_st != st;
        if (pre_used) {
            if (!s.ev_pre) HIPCHECK(hipEventCreateWithFlags(&s.ev_pre, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_pre, pre_st));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_pre, 0));
            pre_used = false;
        }
        if (flush_later) {
            if (int rc = flush_merges(idx, ~0ull)) return rc;
            flush_later = false;
        }
        QueryKernelArgs H = A;
        H.head_codes = ~0ull;
        H.head_slots = (uint32_t)head_slots;
        H.qstates = s.d_qs;
        H.cand_regions = s.d_cands.p;
        H.cand_cap = gcap;
        H.hdr = s.d_hdr;
        H.G = 1;
        H.head_wg = (uint32_t)idx->head_wg;
        // (profile: one event before and after each of the three launches — prof_ev[1..4]; ~10 us of stream time each)
        if (idx->profile) HIPCHECK(prof_event(s, st));
        HIPCHECK(launch_scan_query(M, nq, H, st));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        // (workgroups per group: 128 KiB of codes each — 16 K codes at 16x4, 8 K at 32x4.  One workgroup per partition — the multi-query
        //  scans' own 64 K codes — leaves the C3 shape's 4.4 K groups 2.5 rounds of the GPU's 1792 resident workgroups, and the last
        //  round mostly empty: 0.335 -> 0.315 ms per batch at C3 with 16 K; 8 K / 12 K / 24 K / 32 K measured again in round 6:
        //  0.350 / 0.342 / 0.340 / 0.364 against 0.337.  C5: 16 K 3.06 ms, 8 K 2.97 (round 6, same box, profiles/r06_head_wg_ab.txt))
        const uint64_t per_wg = kGroupBytesPerWg / (uint64_t)idx->cs;
        const int wgs = (int)std::max<uint64_t>(1, ((uint64_t)idx->max_part_n + per_wg - 1) / per_wg);
        launch_scan_i8_mq(M, d_gitems, (int)(ngroups * 8), wgs, A.qtables, s.d_qs, s.d_hdr, s.d_cands.p, gcap, (uint32_t)s.R, st,
                          /*narrow=*/1);
        if (idx->profile) HIPCHECK(prof_event(s, st));
        HIPCHECK(launch_order_cands(s.d_qs, s.d_cands.p, gcap, gcap, nq, s.d_stream.p, cap, s.d_qout, s.d_qflags.p, ma, A.pos_bits, st));
        if (idx->profile) HIPCHECK(prof_event(s, st));
        s.group_head_slots = head_slots;
        idx->prof.group_launches++;
    } else {
        if (pre_used) {                                          // (tables built on the copy stream, no plan)
            if (!s.ev_pre) HIPCHECK(hipEventCreateWithFlags(&s.ev_pre, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_pre, pre_st));
            HIPCHECK(hipStreamWaitEvent(st, s.ev_pre, 0));
        }
        if (flush_later)
            if (int rc = flush_merges(idx, ~0ull)) return rc;
        if (lf_slices) {
            const size_t words = 4 + (size_t)kLoneFrontMaxKeys;
            if (words > s.d_lfstate.cap) {
                HIPCHECK(s.d_lfstate.ensure(words));
                // (the counter: every front leaves it at zero again.  On the launches' own stream: a plain hipMemset runs on the null
                // stream, which these non-blocking streams do not wait for — the first front of an index then raced it about once in
                // 5000 fresh indexes: no workgroup counted itself last, and the walk read whatever lay in front_out)
                HIPCHECK(hipMemsetAsync(s.d_lfstate.p, 0, words * sizeof(uint32_t), st));
            }
            HIPCHECK(s.d_front_all.ensure(4));
            LoneFrontArgs F{};
            F.part = idx->d_partdesc.p + s.assign[0];
            F.R = (uint32_t)s.R;
            F.S = lf_slices;
            F.quant_mode = idx->quant_mode;
            F.sum_mode = idx->sum_mode;
            F.state = s.d_lfstate.p;
            F.qtables = s.d_qtables.p;
            F.front_out = s.d_front_all.p;
            std::memcpy(F.table, s.tables, (size_t)M * 16 * sizeof(float));
            HIPCHECK(launch_lone_front(M, F, st));
            idx->prof.lone_front_launches++;
            A.ftables = nullptr;                                 // the walk: an int8 query whose verdict comes from the front launch
            A.qtables = s.d_qtables.p;
            A.front_in = s.d_front_all.p;
            s.d_qt = s.d_qtables.p;
        }
        HIPCHECK(launch_scan_query(M, nq, A, st, inl_bytes ? inl : nullptr, inl_bytes));
    }
    if (idx->profile) HIPCHECK(prof_event(s, st));
    if (s.dist_batch && !s.rerun)
        if (int rc = enqueue_merge(idx, s, st)) return rc;
    if (s.dev_replay) {
        if (!alone) {
            // replay on a side stream, under the next batches' scans.  A 1024-query replay (16 waves, one lane per
            // query, ~1.5 K dependent pushes each) lasts about as long as the batch's scan: consecutive batches use two
            // side streams alternately, so that a replay never waits for the previous batch's replay.
            if (!s.ev_scanned) HIPCHECK(hipEventCreateWithFlags(&s.ev_scanned, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(s.ev_scanned, st));
            // Alternate by SUBMISSION order, not by slot: with three batches in flight slots 2 and 0 follow each other, and
            // on one stream the second replay would wait out the first (every third batch lost 0.7 ms that way).
            // (a replay stream of its own per slot was tried: 1.2 -> 1.9 us per query at the IVF shape)
            st = (idx->replay_seq++ & 1) ? idx->front_stream : idx->sort_stream;
            HIPCHECK(hipStreamWaitEvent(st, s.ev_scanned, 0));
        }
        if (s.heaps_ready) {
            uint64_t* d_heaps = reinterpret_cast<uint64_t*>(d_result + off_heaps);
            uint32_t* d_sizes = reinterpret_cast<uint32_t*>(d_result + off_heaps + sizeof(uint64_t) * (size_t)s.R * nq);
            // one wave per query, heap in registers (replay_heap_wave_kernel)
            HIPCHECK(launch_replay_heap_wave_qflags(s.d_qflags.p, s.d_stream.p, cap, nq, (uint32_t)s.R, d_heaps, d_sizes, st));
        }
    }
    // (a polled batch — the lone synchronous query — is read from its records' done bits; should they not show up in time the collect
    // call waits for the scan stream itself, so the ~1 us of an event record stays off the call)
    s.ev_valid = !s.poll;
    if (s.ev_valid) {
        if (!s.ev_done) HIPCHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_done, st));
    }
    return QADC_OK;
}

// Whether launch_wgq_batch will send a batch of this shape through the partition-major second phase (same test as there).
bool will_group(const qadc_index* idx, int nq, int ma, bool dev_replay) {
    const int head_slots = std::min(idx->dist ? idx->group.head_dist : idx->group.head, ma);
    const size_t pairs = (size_t)nq * (size_t)(ma - head_slots), nparts = idx->parts.size();
    return dev_replay && pairs > 0 && nparts < (1u << 24) &&
           (idx->group.mode == 2 || (idx->group.mode == 1 && idx->group.strikes < 2 && nq >= 256 && pairs >= 2 * nparts));
}

// ||c||^2 of the coarse centroids, once per centroid set and sum_mode, on the stream the coarse kernels run on
static int ensure_cnorm(qadc_index* idx, hipStream_t cs) {
    FeederState& f = idx->feed;
    if (f.cnorm_mode == idx->sum_mode) return QADC_OK;
    HIPCHECK(f.d_cnorm.ensure((size_t)f.K));
    launch_row_sqnorm(f.d_coarse.p, f.K, f.dim, idx->sum_mode, f.d_cnorm.p, cs);
    HIPCHECK(hipGetLastError());
    f.cnorm_mode = idx->sum_mode;
    return QADC_OK;
}

// N1: queries in.  Coarse assignment runs on the copy stream (so it does not queue behind the previous
// batch's scan), the host reads assign[] back to plan the work items, residuals and float tables are built
// on the GPU by the main stream.
int search_submit(qadc_index* idx, int slot_i, int nq, const float* queries, int ma, int R) {
    if (!idx || !queries) return fail(QADC_E_ARG, "null argument");
    if (slot_i < 0 || slot_i >= kSlots) return fail(QADC_E_ARG, "slot must be 0 .. 7");
    if (!idx->finalized) return fail(QADC_E_STATE, "qadc_index_finalize has not been called");
    if (idx->feed.dim == 0) return fail(QADC_E_STATE, "qadc_index_set_pq has not been called");
    if (nq <= 0 || ma <= 0 || R <= 0) return fail(QADC_E_ARG, "nq, ma, R must be > 0");
    if (nq >= (1 << 24) || ma >= (1 << 14)) return fail(QADC_E_ARG, "nq must be < 2^24 and ma < 16384");
    if (idx->feed.K && (idx->feed.K != (int)idx->parts.size() || ma > idx->feed.K))
        return fail(QADC_E_ARG, "coarse centroids must match the partitions one to one and ma <= K");
    if (!idx->feed.K && idx->parts.size() != 1) return fail(QADC_E_ARG, "a database without coarse centroids must be flat (1 partition)");
    Slot& s = idx->slot[slot_i];
    if (s.busy) return fail(QADC_E_STATE, "slot still holds an uncollected batch");
    if (int rc = use_device(idx)) return rc;
    s.mode = 0;
    const int dim = idx->feed.dim;
    s.nq = nq;
    s.ma = ma;
    s.R = R;
    s.float_path = true;
    s.device_tables = true;
    s.tables = nullptr;
    HIPCHECK(s.h_queries.ensure((size_t)nq * dim));
    HIPCHECK(s.d_queries.ensure((size_t)nq * dim));
    HIPCHECK(s.h_assign.ensure((size_t)nq * ma));
    HIPCHECK(s.d_assign.ensure((size_t)nq * ma));
    std::memcpy(s.h_queries.p, queries, sizeof(float) * (size_t)nq * dim);
    hipStream_t cs = idx->copy_stream;
    // Under the multi-GPU merge every rank receives the same queries; what is per QUERY rather than per code — coarse
    // assignment, residual tables, pre-scan, select, quantizer — is then split over the ranks: rank r does it for queries
    // [r * per, (r + 1) * per) and one all-gather ships assign[] + int8 tables + (flags, qmin, qmax) to everybody
    // (launch_wgq_batch).  Only for batches that take the one-workgroup-per-query head + partition-major second phase.
    s.front_sharded = false;
    s.group_fell_back = false;
    {
        // Whether this batch issues the front's all-gather must come out the same on every rank: the estimate is drawn from the
        // GLOBAL partition sizes (a rank's own total differs under whole-partition placement and for the last range), and the
        // strike count behind will_group moves only on gathered verdicts (qadc_dist_collect).
        const uint64_t est = idx->parts.empty() ? 0 : idx->total_global_codes / idx->parts.size() * (uint64_t)(idx->feed.K ? ma : 1);
        if (idx->dist && idx->dist->shard_front && (idx->dist->world > 1 || idx->dist->shard_front == 2) && idx->feed.K && nq >= 2 * idx->dist->world &&
            wgq_eligible(idx, nq, ma, R, 0, est) && will_group(idx, nq, ma, true)) {
            s.front_sharded = true;
            s.front_per = (nq + idx->dist->world - 1) / idx->dist->world;
            s.front_q0 = std::min(nq, idx->dist->rank * s.front_per);
            s.front_n = std::min(nq, s.front_q0 + s.front_per) - s.front_q0;
        }
    }
    if (s.front_sharded) {
        const size_t tab = (size_t)ma * idx->M * 16;
        const size_t block = ((size_t)s.front_per * (tab + (size_t)ma * 4 + 16) + 15) & ~(size_t)15;   // [qtables][assign][front] of `per` queries
        HIPCHECK(s.d_fblock.ensure(block));
        HIPCHECK(s.d_fgathered.ensure(block * idx->dist->world));
        HIPCHECK(s.d_front_all.ensure(4 * (size_t)nq));
        HIPCHECK(s.h_fmap.ensure((size_t)nq * ma * 4 + (size_t)nq * 16, hipHostMallocMapped | hipHostMallocCoherent));
        if (s.h_fmap.p != s.h_fmap_mapped) {
            HIPCHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d_fmap), s.h_fmap.p, 0));
            s.h_fmap_mapped = s.h_fmap.p;
        }
        if (s.front_n) {
            HIPCHECK(hipMemcpyAsync(s.d_queries.p, s.h_queries.p + (size_t)s.front_q0 * dim, sizeof(float) * (size_t)s.front_n * dim,
                                    hipMemcpyHostToDevice, cs));
            HIPCHECK(s.d_cdist.ensure((size_t)s.front_n * idx->feed.K + (size_t)s.front_n));   // (+ the queries' norms behind the distances)
            if (int rc = ensure_cnorm(idx, cs)) return rc;
            int32_t* d_assign_share = reinterpret_cast<int32_t*>(s.d_fblock.p + (size_t)s.front_per * tab);
            launch_coarse_assign(s.d_queries.p, idx->feed.d_coarse.p, s.front_n, idx->feed.K, dim, ma, s.d_cdist.p + (size_t)s.front_n * idx->feed.K,
                                 idx->feed.d_cnorm.p, idx->sum_mode, s.d_cdist.p, d_assign_share, cs);
            HIPCHECK(hipGetLastError());
        }
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));
        s.wgq = true;
        s.wgq_codes = idx->total_global_codes / idx->parts.size() * (uint64_t)ma;
        s.assign_on_device = true;
        s.full_prescan = false;
        s.rerun = false;
        s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
        s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
        if (int rc = plan_and_launch(idx, s)) return rc;
        s.busy = true;
        return QADC_OK;
    }
    HIPCHECK(hipMemcpyAsync(s.d_queries.p, s.h_queries.p, sizeof(float) * (size_t)nq * dim, hipMemcpyHostToDevice, cs));
    if (idx->feed.K) {
        HIPCHECK(s.d_cdist.ensure((size_t)nq * idx->feed.K + (size_t)nq));                     // (+ the queries' norms behind the distances)
        if (int rc = ensure_cnorm(idx, cs)) return rc;
        launch_coarse_assign(s.d_queries.p, idx->feed.d_coarse.p, nq, idx->feed.K, dim, ma, s.d_cdist.p + (size_t)nq * idx->feed.K,
                             idx->feed.d_cnorm.p, idx->sum_mode, s.d_cdist.p, s.d_assign.p, cs);
        HIPCHECK(hipGetLastError());
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));             // the tables need assign[] on the device, not the copy below
        HIPCHECK(hipMemcpyAsync(s.h_assign.p, s.d_assign.p, sizeof(int32_t) * (size_t)nq * ma, hipMemcpyDeviceToHost, cs));
    } else {
        HIPCHECK(hipMemsetAsync(s.d_assign.p, 0, sizeof(int32_t) * (size_t)nq * ma, cs));
        std::memset(s.h_assign.p, 0, sizeof(int32_t) * (size_t)nq * ma);
        if (!s.ev_feed) HIPCHECK(hipEventCreateWithFlags(&s.ev_feed, hipEventDisableTiming));
        HIPCHECK(hipEventRecord(s.ev_feed, cs));
    }
    if (!s.ev_assign) HIPCHECK(hipEventCreateWithFlags(&s.ev_assign, hipEventDisableTiming));
    HIPCHECK(hipEventRecord(s.ev_assign, cs));
    // one workgroup per query: the kernel reads assign[] on the device, the host only wants it back for the caller
    // (under the multi-GPU merge the estimate comes from the partitions' GLOBAL sizes, like the sharded front's above: the scan
    // path — hence which kernels a later collective may be keyed on — must not depend on how many codes THIS rank happens to hold;
    // qadc_host.h, "rank-invariant state")
    const uint64_t est_total = idx->dist ? idx->total_global_codes : idx->total_codes;
    const uint64_t est_codes = idx->parts.empty() ? 0 : est_total / idx->parts.size() * (uint64_t)(idx->feed.K ? ma : 1);
    s.wgq = wgq_eligible(idx, nq, ma, R, 0, est_codes);
    s.wgq_codes = est_codes;
    s.assign_on_device = s.wgq;
    if (!s.wgq) {
        HIPCHECK(hipStreamSynchronize(cs));                  // the planner needs assign[] on the host
        s.assign.assign(s.h_assign.p, s.h_assign.p + (size_t)nq * ma);
    }
    s.full_prescan = false;
    s.rerun = false;
    if (s.wgq) s.wgq_cap = std::max<uint32_t>(s.wgq_cap, idx->wgq_capacity);
    s.cap_q = std::max<uint32_t>(s.cap_q, idx->cand_capacity);
    if (!s.wgq) s.out_cap = std::max<uint32_t>(s.out_cap, (uint32_t)std::min<uint64_t>((uint64_t)nq * 8192u, 1ull << 30));
    if (int rc = plan_and_launch(idx, s)) return rc;
    s.busy = true;
    return QADC_OK;
}
}  // namespace host
}  // namespace qadc

extern "C" {

int qadc_index_set_pq(qadc_index* idx, int dim, const float* codebooks) {
    if (!idx || !codebooks || dim <= 0 || dim % idx->M != 0) return fail(QADC_E_ARG, "dim must be a positive multiple of M");
    if (int rc = use_device(idx)) return rc;
    const size_t n = (size_t)idx->M * 16 * (dim / idx->M);
    HIPCHECK(idx->feed.d_codebooks.ensure(n));
    HIPCHECK(hipMemcpy(idx->feed.d_codebooks.p, codebooks, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipDeviceSynchronize());                            // (setup copy from pageable memory: nothing in flight when the first batch's kernels start on their non-blocking streams)
    idx->feed.dim = dim;
    return QADC_OK;
}

int qadc_index_set_rotation(qadc_index* idx, const float* rotation) {
    if (!idx) return fail(QADC_E_ARG, "null index");
    if (idx->feed.dim == 0) return fail(QADC_E_STATE, "call qadc_index_set_pq first");
    if (!rotation) {
        idx->feed.has_rotation = false;
        return QADC_OK;
    }
    if (int rc = use_device(idx)) return rc;
    const size_t n = (size_t)idx->feed.dim * idx->feed.dim;
    HIPCHECK(idx->feed.d_rotation.ensure(n));
    HIPCHECK(hipMemcpy(idx->feed.d_rotation.p, rotation, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipDeviceSynchronize());                            // (setup copy from pageable memory: nothing in flight when the first batch's kernels start on their non-blocking streams)
    idx->feed.has_rotation = true;
    return QADC_OK;
}

int qadc_index_set_coarse(qadc_index* idx, int K, const float* centroids) {
    if (!idx || K <= 0 || !centroids) return fail(QADC_E_ARG, "bad arguments");
    if (idx->feed.dim == 0) return fail(QADC_E_STATE, "call qadc_index_set_pq first");
    if (int rc = use_device(idx)) return rc;
    HIPCHECK(idx->feed.d_coarse.ensure((size_t)K * idx->feed.dim));
    HIPCHECK(hipMemcpy(idx->feed.d_coarse.p, centroids, (size_t)K * idx->feed.dim * sizeof(float), hipMemcpyHostToDevice));
    HIPCHECK(hipDeviceSynchronize());                            // (setup copy from pageable memory: nothing in flight when the first batch's kernels start on their non-blocking streams)
    idx->feed.K = K;
    idx->feed.cnorm_mode = -1;
    return QADC_OK;
}

int qadc_search_submit(qadc_index* idx, int slot, int nq, const float* queries, int ma, int R) {
    return search_submit(idx, slot, nq, queries, ma, R);
}

int qadc_search_collect(qadc_index* idx, int slot, uint32_t* keys, int8_t* values, int32_t* sizes, int32_t* status,
                        int32_t* assign_out) {
    if (int rc = collect_common(idx, slot, /*need_stream=*/false)) return rc;
    Slot& s = idx->slot[slot];
    std::vector<int32_t> st_local;
    if (!status) {
        st_local.resize(s.nq);
        status = st_local.data();
    }
    finish_float_outputs(idx, s, status, nullptr, nullptr);
    if (assign_out) std::memcpy(assign_out, s.assign.data(), sizeof(int32_t) * s.assign.size());
    return replay_outputs(idx, s, keys, values, sizes, status);
}

int qadc_search(qadc_index* idx, int nq, const float* queries, int ma, int R, uint32_t* keys, int8_t* values, int32_t* sizes,
                int32_t* status, int32_t* assign_out) {
    if (int rc = search_submit(idx, 0, nq, queries, ma, R)) return rc;
    return qadc_search_collect(idx, 0, keys, values, sizes, status, assign_out);
}

}  // extern "C"
