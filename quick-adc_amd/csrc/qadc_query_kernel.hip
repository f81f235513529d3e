// One workgroup = one query: the whole scanner_4::query_scan chain (db_query_4.cpp:245-309) in ONE launch.
//
// The level-structured path (qadc_kernels.hip) cuts every query's scan order into bound levels, one launch per
// level, many workgroups per run: right for a flat list of 10^8..10^9 codes, wrong for an IVF batch, where a
// query probes a few 10^5 codes in dozens of short partitions, there are hundreds of queries in flight, and
// the dependent chain of launches, the per-item table builds, the candidate sort and the host planner cost more
// than the scan.  Here a query belongs to one 1024-thread workgroup that walks the query's scan order itself:
//
//   1. float pre-scan of the probed partitions' starts (scan_4<M>, query_common.hpp:59-90; same add order) into
//      LDS, R-th smallest by one histogram pass over a digit fitted to the keys + a ranking of the chosen bucket (a
//      threshold cut first when there are thousands of values; four fixed-digit passes when a bucket overflows)
//      -> qmax                                                                (db_query_4.cpp:230-242, 259)
//   2. qmin over all ma tables, negative clamp, QuantizerMAX<int8_t>           (db_query_4.cpp:37-71, 258-284)
//   3. int8 scan of every probed partition in assign[] order                   (simd_scan.hpp:125-187):
//      pair-fused 256-entry byte tables in LDS, bank-replicated so that any code data reads conflict-free,
//      cand = min(127, sum).  The scan order is cut into EPOCHS (64, 128, 256 ... codes, then one per <= 32 Ki
//      vectors of a partition); inside an epoch the 16 waves run free — no workgroup barrier per tile — against
//      the bound of the epochs that are FINISHED (a 128-bin histogram in LDS, swapped at the epoch boundary), and
//      append the qualifying codes unordered to the workgroup's candidate buffer in global memory.
//   4. the workgroup sorts its candidates by (assign slot, position) — bitonic, LDS-tiled — and writes the
//      ordered push stream, expanding the padding-lane replays of a partition's last code
//      (simd_layout.hpp:46-50, simd_scan.hpp:67) while writing.
//
// Output per query: the ordered push stream (entry = key | value << 32 | assign slot << 40, the format of the
// sorted output of sort_cands_kernel) and a QueryOut record; the heap replay follows (host for small batches,
// replay_heap_wave_kernel — one WAVE per query, the heap in registers — for large ones).  Exactness argument: every dropped code has
// cand >= the R-th smallest (<127) value of a set of codes that all PRECEDE it in this query's scan order
// (the finished epochs), DESIGN.md section 4.
//
// Variants of the same kernel: MULTI (a small batch spreads each query over G workgroups, which share the first
// block of the scan order for their bound — walked in ONE step from registers, workgroup 0 writing its candidates
// straight into the stream: see "the first block in ONE step" below) and HEAD (the first launch of the
// level-structured path, and the head of a partition-major IVF batch).  gfx950 only.
#include "qadc_kernels.h"
#include "qadc_float_sum.h"

#include <atomic>
#include <cfloat>
#include <cstddef>
#include <cstring>
#include <type_traits>

constexpr uint32_t kOrderBuckets = 1024;   // order_cands_kernel's bucket-sort scratch: (assign slots) x (up to 32 position sub-buckets)
// Helper launches that run BESIDE the scan kernels (plan, merge, replay) use workgroups of 4 waves.  The partition-major scan
// keeps 28 of a CU's 32 wave slots (256-thread workgroups, 7 waves per SIMD), so a 4-wave workgroup fits beside it at once;
// a 1024-thread one waits until 16 slots of ONE CU are free together — and while it waits at the head of its queue nothing
// else is dispatched there.  Round 4, same-box A/B on one of 8 ranks' IVF batch: replay 16 -> 4 waves per workgroup and the
// one-workgroup totals kernel 1024 -> 256 threads: C3 0.52 -> 0.45 ms per batch, C5 0.81 -> 0.78 (profiles/r04_side_wg_ab.txt).
constexpr int kSideWG = 256;

namespace qadc {

namespace {

constexpr int kQWG = 1024;          // threads per query workgroup (16 waves)
constexpr int kQWaves = kQWG / 64;

extern __shared__ __attribute__((aligned(16))) unsigned char qsmem[];

__device__ __forceinline__ uint32_t q_fkey(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float q_funkey(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

__device__ __forceinline__ void q_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)): in the scan loop that would stall all 16 waves for the write latency of the few stream
// entries one of them just stored — to device memory, or over PCIe to the host-mapped result block.
__device__ __forceinline__ void q_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS image.  The scan's pair tables are BANK-REPLICATED (conflict-free for any code data, see qadc_kernels.hip):
// for code byte b = 4g + j the table P_b[x] = T[2b][x & 15] + T[2b+1][x >> 4] sits at
//     region(g) + x*256 + (g&1)*128 + bank*4 + j,  bank = 0..31,  region(g) = (g>>1) * 64 KiB,
// one dword = the four tables of a code dword, replicated over the 32 banks of a ds_read lane group; a lane reads
// with bank = lane & 31.  64 KiB (16x4: two workgroups per CU) / 128 KiB (32x4: one).  The tables start at LDS
// address 0 (absolute addressing spares an add per lookup) and OVERLAY the pre-scan's buffers, which are dead by then.
// 32x4 (round 4): the same image with SIXTEEN replicas — row x (256 B) = [dword 0: 16 replicas][dword 1]
// [dword 2][dword 3], address x*256 + g*64 + (lane & 15)*4 + j — is 64 KiB instead of 128, so TWO workgroups share a CU
// (32 waves instead of 16 to hide the walk's latencies behind) at the price of a 2-way bank conflict on every lookup
// (lanes l and l + 16 of a 32-lane group share a replica): the walk is latency / issue bound, the LDS pipe 30 % busy.
// Round 5: the conflict is a property of the ORDER in which a lane takes its code's dwords, not of the image.
// Lanes with bit 4 set (h = 1) process the dwords as 1, 0, 3, 2: at the lookup (w, k) they read group w ^ 1 — so the 32 lanes of a
// pass touch banks (w * 16 + l) and ((w ^ 1) * 16 + l), l = 0..15: 32 distinct banks, no conflict, same 64 KiB image.  Price:
// four v_cndmask per code (the dword swap); the address bytes come from two lane constants, (lane & 15) * 4 + 64 h for even w
// and (lane & 15) * 4 + 64 (1 - h) for odd w, with the immediate offset (w & ~1) * 64 + k.
template <int M, int WG = 1024>
struct QCfg {
    static constexpr int CS = M / 2, DW = M / 8, CPL = 16 / CS;
    static constexpr int WAVES = WG / 64;                        // waves of the query workgroup (kQWaves, or 8: the IVF head's 512-thread form)
    static constexpr bool R16 = M == 32;                           // 16 replicas per dword, all four dwords in one 64 KiB region
    static constexpr bool SWZ = R16;                                 // the two 16-lane halves of a pass read different dword groups
    static constexpr int TABLE_BYTES = R16 ? 65536 : (M / 16) * 65536;
    static constexpr int WTAB_BYTES = WAVES * M * 16 * 4;        // float[waves][M*16] (pre-scan)
    static constexpr int FCAP = (TABLE_BYTES - WTAB_BYTES) / 4;  // pre-scan values kept in LDS: 12288 / 24576
    static constexpr int VALS_OFF = 0;                           // float[FCAP]        } pre-scan phase
    static constexpr int WTAB_OFF = FCAP * 4;                    // float[16][M*16]    }
    static constexpr int TQ_OFF = TABLE_BYTES;                   // u8[M*16] staged int8 table
    static constexpr int MISC_OFF = TABLE_BYTES + 512;
    // misc (u32 words): [0..255] radix histogram / [0..127] value histogram, then scalars
    static constexpr int LDS_BYTES = MISC_OFF + (256 + 64 + 64) * 4;
    // small batches (several workgroups per query): the first block's per-epoch value counters, behind everything else
    static constexpr int FB_EPOCHS = M == 16 ? 25 : 21;          // epoch 0 = codes [0, 128), then quarter octaves up to 4096 vectors
    static constexpr int FBH_BYTES = FB_EPOCHS * 128 * 4;
    static constexpr int OCC = (M == 16 || R16) ? 8 : 4;         // waves per SIMD the register budget is sized for
};

typedef const __attribute__((address_space(3))) unsigned char* q_lds_bytes_t;

// sum of the M/2 pair-table entries of one code (DW dwords at d): ONE v_perm_b32 forms each LDS address
// (byte0 = bank*4 of the lane, byte1 = code byte, byte2 = region), the table index rides in the immediate offset
template <int M>
__device__ __forceinline__ uint32_t q_pair_sum(const uint32_t* d, uint32_t lane_lo, uint32_t lane_hi) {
    // all M/2 addresses first, then all M/2 reads, then the adds: the reads of a code (and, unrolled, of the codes
    // around it) are in flight together instead of waiting out one LDS latency per pair
    uint32_t a[M / 2], v[M / 2];
    constexpr bool R16 = QCfg<M>::R16;                           // (then every dword lives in region 0)
    constexpr bool SWZ = QCfg<M>::SWZ;
    uint32_t e[M / 8];
#pragma unroll
    for (int w = 0; w < M / 8; ++w) e[w] = d[w];
    if constexpr (SWZ) {
        // lanes 16..31 / 48..63 take their code's dwords in the order 1, 0, 3, 2 (integer adds: any order gives the sum), and
        // lane_lo / lane_hi carry (lane & 15) * 4 + 64 h / + 64 (1 - h): in every 32-lane pass of a lookup the two halves read
        // DIFFERENT dword groups of the rows — 32 distinct banks instead of 16 banks twice
        const bool h = (lane_lo & 64u) != 0;
#pragma unroll
        for (int w = 0; w < M / 8; w += 2) {
            e[w] = h ? d[w + 1] : d[w];
            e[w + 1] = h ? d[w] : d[w + 1];
        }
    }
#pragma unroll
    for (int w = 0; w < M / 8; ++w) {
        const uint32_t lo = SWZ ? ((w & 1) ? lane_hi : lane_lo) : (!R16 && (w >> 1)) ? lane_hi : lane_lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[w * 4 + k] = __builtin_amdgcn_perm(e[w], lo, 0x0c020000u | ((4u + k) << 8));
    }
#pragma unroll
    for (int w = 0; w < M / 8; ++w)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[w * 4 + k] = *reinterpret_cast<q_lds_bytes_t>(
                static_cast<uintptr_t>(a[w * 4 + k] + (SWZ ? (w & ~1) * 64 : R16 ? w * 64 : (w & 1) * 128) + k));
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < M / 2; ++i) s += v[i];
    return s;
}

// The pre-scan's float ADC of one code (scan_4<M>, query_common.hpp:59-90; the grouping of the adds is sum_mode's — 1 = as
// the reference is compiled, 0 = source order: qadc_float_sum.h) against the wave's float table at absolute LDS address tb
// (256-byte aligned; table t at tb + t*64).  Sixteen lookups are issued before the first add waits: one read waited for
// per add made a code 32 LDS round trips (C5 shape: the pre-scan was 107 K of the front's 236 K cycles).  Addresses: the
// nibbles are masked four at a time, pre-multiplied by 4, and ONE v_perm_b32 per lookup puts the byte under the table's
// address (the table index rides in the read's immediate offset).
typedef const __attribute__((address_space(3))) float* q_lds_float_t;
template <int M>
__device__ __forceinline__ float q_prescan_sum(const uint32_t* d, uint32_t tb, int sum_mode) {
    constexpr int CS = M / 2;
    float cand = 0.0f;
#pragma unroll
    for (int h = 0; h < CS; h += 8) {
        float v[16];
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const uint32_t dw = d[(h >> 2) + w];
            const uint32_t dl = (dw << 2) & 0x3c3c3c3cu, dh = (dw >> 2) & 0x3c3c3c3cu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = h + 4 * w + k;                     // code byte: sub-quantizers 2b (low nibble) and 2b + 1
                const uint32_t al = __builtin_amdgcn_perm(dl, tb, 0x03020104u + (uint32_t)k);
                const uint32_t ah = __builtin_amdgcn_perm(dh, tb, 0x03020104u + (uint32_t)k);
                v[8 * w + 2 * k] = *reinterpret_cast<q_lds_float_t>(static_cast<uintptr_t>(al + (2 * b) * 64));
                v[8 * w + 2 * k + 1] = *reinterpret_cast<q_lds_float_t>(static_cast<uintptr_t>(ah + (2 * b + 1) * 64));
            }
        }
        if (sum_mode == 0) cand = adc_sum8_source(cand, v);
        else if (h == 0)   cand = adc_sum8_compiled_first(v);
        else               cand = adc_sum8_compiled_next(cand, v);
    }
    return cand;
}

// A partition descriptor whose fields are forced into scalar registers: the probed partition is the same for the
// whole workgroup, but the compiler cannot prove that loads through assign[] are uniform.
struct UDesc {
    const uint8_t* codes;
    const uint32_t* labels;
    uint32_t n, global_n, first_pos, key_base;
};
__device__ __forceinline__ uint32_t q_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t q_uni64(uint64_t v) {
    return ((uint64_t)q_uni((uint32_t)(v >> 32)) << 32) | q_uni((uint32_t)v);
}
__device__ __forceinline__ UDesc q_load_desc(const PartDesc* __restrict__ parts, int p) {
    const PartDesc& s = parts[p];
    UDesc d;
    d.codes = reinterpret_cast<const uint8_t*>(q_uni64(reinterpret_cast<uint64_t>(s.codes)));
    d.labels = reinterpret_cast<const uint32_t*>(q_uni64(reinterpret_cast<uint64_t>(s.labels)));
    d.n = q_uni(s.n);
    d.global_n = q_uni(s.global_n);
    d.first_pos = q_uni(s.first_pos);
    d.key_base = q_uni(s.key_base);
    return d;
}

// Wave-wide scans and reductions on the DPP path (row shifts inside the 16-lane rows, then row broadcasts: 7 VALU
// operations), not __shfl_*: those go through the LDS crossbar, ~100 cycles each in a dependent chain of six — the
// epoch ends, the select passes and the write-out of a single small query are chains of exactly these.
template <typename Op>
__device__ __forceinline__ uint32_t q_wave_scan_bits(uint32_t x, uint32_t identity, Op op) {
    const int id = (int)identity;
    uint32_t v = x;
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x113, 0xf, 0xf, false));   // row_shr:3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xf, 0xe, false));   // row_shr:4, banks 1-3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xf, 0xc, false));   // row_shr:8, banks 2-3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;                                                    // inclusive scan; lane 63 holds the reduction
}
__device__ __forceinline__ uint32_t q_wave_incl_sum(uint32_t x) {
    return q_wave_scan_bits(x, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint32_t q_wave_sum(uint32_t x) {     // the wave's total, in every lane
    return (uint32_t)__builtin_amdgcn_readlane((int)q_wave_incl_sum(x), 63);
}
__device__ __forceinline__ float q_wave_min(float x) {
    const uint32_t r = q_wave_scan_bits(__float_as_uint(x), __float_as_uint(FLT_MAX),
                                        [](uint32_t a, uint32_t b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); });
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)r, 63));
}

// Bound = smallest v such that at least R emitted candidates have value <= v, else 127 (lanes own bins 2l, 2l+1).
__device__ __forceinline__ uint32_t q_bound_from_hist(const uint32_t* hist, uint32_t R, uint32_t lane) {
    const uint32_t c0 = hist[2 * lane], c1 = hist[2 * lane + 1];
    const uint32_t incl = q_wave_incl_sum(c0 + c1);
    const uint32_t excl = incl - (c0 + c1);
    const uint64_t reached = __builtin_amdgcn_ballot_w64(incl >= R);
    if (reached == 0) return 127u;
    const uint32_t b = excl + c0 >= R ? 2 * lane : 2 * lane + 1;   // (meaningful in the first lane that reaches R)
    return min((uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(reached)), 127u);
}

// Orders one query's (one workgroup's) candidates by (assign slot, position) = scan order in LDS, expands the
// padding-lane replays and writes the push stream: the tail of scan_query_kernel, and all of order_cands_kernel.
// load(i) -> QCand of the i-th unordered candidate; needs ncand <= kQueryCandCap; returns the entries written
// (replays included; entries beyond `cap` are counted, not stored).  Uses qsmem[0 .. 64 KiB) and wcnt[16].
// LOGCAP: log2 of the most candidates taken — 12 inside scan_query_kernel, 13 in order_cands_kernel.
// PAYLDS: the payloads wait in LDS beside the sort keys (64 KiB in all at LOGCAP 12); false: they are fetched again through
// load() when the stream is written (order_cands_kernel: its candidates lie in global memory the second phase has just
// written — L2 — and keys alone let two workgroups share a CU at 8192 candidates, where keys + payloads left room for one).
// scr / scr_slots / nslots / pos_bits: 2 x scr_slots + 1 words of LDS scratch, the number of assign slots and the bits of
// the longest partition's length — for the BUCKET sort below (scr == nullptr: bitonic only).
template <int LOGCAP, bool PAYLDS, int WGS = 1024, typename Load>
__device__ __forceinline__ uint32_t q_order_and_write(Load load, uint32_t ncand, uint64_t* __restrict__ stream, uint32_t cap,
                                                      uint32_t* wcnt, uint32_t tid, uint32_t lane, uint32_t wave,
                                                      uint32_t* scr = nullptr, uint32_t scr_slots = 0, uint32_t nslots = 0,
                                                      uint32_t pos_bits = 0, uint32_t bucket_max = 256) {
    uint32_t out_count = 0;
    constexpr uint32_t kQWG = WGS, kQWaves = WGS / 64;               // (the caller's workgroup: shadows the 1024-thread default)
    constexpr uint32_t kCap = 1u << LOGCAP, kPer = kCap / kQWG;      // entries per thread in the write-out
    uint64_t* skey = reinterpret_cast<uint64_t*>(qsmem);             // [n2] (slot << (32 + LOGCAP) | pos << LOGCAP | index)
    uint64_t* spay = reinterpret_cast<uint64_t*>(qsmem + kCap * 8);  // [ncand] key | val << 32 | reps << 40 | slot << 48
    uint32_t n2 = 64;
    while (n2 < ncand) n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += kQWG) {
        uint64_t k = ~0ull;
        if (i < ncand) {
            const QCand c = load(i);
            k = ((uint64_t)c.slot << (32 + LOGCAP)) | ((uint64_t)c.pos << LOGCAP) | i;
            if (PAYLDS) spay[i] = (uint64_t)c.key | ((uint64_t)(c.val_reps & 0xfffu) << 32) | ((uint64_t)c.slot << 48);
        }
        skey[i] = k;
    }
    q_lds_barrier();
    // bitonic network, wave-major ownership: wave w owns elements [w*chunk, (w+1)*chunk).  An exchange at distance
    // j < chunk stays inside one wave's elements and needs no workgroup barrier (LDS traffic of a wave is in order);
    // only the few steps with j >= chunk synchronise the workgroup: 10 barriers instead of 55 for 1024 elements.
    constexpr uint32_t kRankSortMax = 512;
    const uint32_t chunk = max(64u, n2 / kQWaves);           // (fewer than 1024 elements: only the first n2/64 waves work)
    if (ncand <= kRankSortMax) {
        // few candidates (a small batch's workgroups): the network's ~n log^2 n dependent LDS round trips cost more
        // than counting — entry e's place is the number of smaller keys (keys are distinct: they end in the index)
        uint64_t* sorted = skey + kRankSortMax;              // (skey holds <= 512 entries here)
        if (tid < ncand) {
            const uint64_t mine = skey[tid];
            uint32_t rank = 0;
            // same address in every lane: broadcast reads.  Two keys per read, eight reads in flight (one key per iteration was one
            // LDS round trip per key: ~100 cycles each, 3.5 K for a lone query's 35 candidates); the keys behind ncand are ~0: never smaller
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const u64x2* s2 = reinterpret_cast<const u64x2*>(skey);
            for (uint32_t j = 0; j < n2 / 2; j += 8) {
                u64x2 kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kk[u] = s2[j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) rank += (kk[u].x < mine ? 1u : 0u) + (kk[u].y < mine ? 1u : 0u);
            }
            sorted[rank] = mine;
        }
        q_lds_barrier();
        if (tid < ncand) skey[tid] = sorted[tid];
    } else {
    // BUCKET sort (round 4): the network is 66 exchange steps at ~800 cycles for the 1.4 K candidates of a C3 query (~55 K
    // of the pass's ~66 K cycles: phase stamps).  A query's candidates spread over a few dozen assign slots, and inside a
    // slot their positions thin out like 1/pos (the heap's threshold tightens as the partition is walked): so a bucket =
    // (slot, quarter-octave of the position above 2^(pos_bits-7)) holds a few dozen entries at most, and the bucket index
    // is monotone in the key.  Count per bucket (LDS atomics), exclusive scan, scatter into the second half of the key
    // array, then an entry's place inside its bucket is the number of smaller keys there (the lanes that share a bucket
    // read them as broadcasts).  The keys are distinct (they end in the entry's index), so the order is the network's.
    // Not taken — the network runs — when the candidates exceed half the key array or one bucket holds more than
    // bucket_max entries (256).
    bool sorted_by_buckets = false;
    if (scr && nslots != 0 && nslots * 2 <= scr_slots && ncand <= kCap / 2) {
        uint32_t s_log = 5;                                      // sub-buckets per slot: 32, fewer when the scratch is short
        while ((nslots << s_log) > scr_slots) --s_log;
        const uint32_t nb = nslots << s_log;
        const uint32_t lo = pos_bits > 7 ? pos_bits - 7 : 0;
        auto bucket_of = [&](uint64_t k_) -> uint32_t {
            const uint32_t slot = (uint32_t)(k_ >> (32 + LOGCAP));
            const uint32_t p_ = (uint32_t)(k_ >> LOGCAP) >> lo;
            uint32_t sub = 0;
            if (p_) {
                const uint32_t e = 31u - (uint32_t)__builtin_clz(p_);
                sub = 1 + e * 4 + (e >= 2 ? (p_ >> (e - 2)) & 3u : (p_ << (2 - e)) & 3u);
            }
            sub = min(sub, 31u) >> (5 - s_log);
            return min(slot, nslots - 1) * (1u << s_log) + sub;
        };
        uint32_t* cnt = scr;                                     // counts, then the scatter's cursors
        uint32_t* base = scr + scr_slots;                        // nb + 1 bucket starts
        for (uint32_t s_ = tid; s_ < nb; s_ += kQWG) cnt[s_] = 0;
        q_lds_barrier();
        for (uint32_t i = tid; i < ncand; i += kQWG) atomicAdd(&cnt[bucket_of(skey[i])], 1u);
        q_lds_barrier();
        if (wave == 0) {                                         // exclusive scan of the counts + the largest bucket
            uint32_t run = 0, mx = 0;
            for (uint32_t s0 = 0; s0 < nb; s0 += 64) {
                const uint32_t c_ = s0 + lane < nb ? cnt[s0 + lane] : 0u;
                const uint32_t incl = q_wave_incl_sum(c_);
                if (s0 + lane < nb) {
                    base[s0 + lane] = run + incl - c_;
                    cnt[s0 + lane] = 0;
                }
                run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                mx = max(mx, c_);
            }
            mx = q_wave_scan_bits(mx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
            if (lane == 63) {
                base[nb] = run;
                wcnt[0] = mx;
            }
        }
        q_lds_barrier();
        sorted_by_buckets = wcnt[0] <= bucket_max;                    // (workgroup-uniform: read after the barrier)
        q_lds_barrier();                                         // (wcnt is reused by the write-out below)
        if (sorted_by_buckets) {
            uint64_t* bkt = skey + kCap / 2;
            for (uint32_t i = tid; i < ncand; i += kQWG) {
                const uint64_t k_ = skey[i];
                const uint32_t b_ = bucket_of(k_);
                bkt[base[b_] + atomicAdd(&cnt[b_], 1u)] = k_;
            }
            q_lds_barrier();
            for (uint32_t i = tid; i < ncand; i += kQWG) {
                const uint64_t k_ = bkt[i];
                const uint32_t b_ = bucket_of(k_);
                const uint32_t b0 = base[b_], b1 = base[b_ + 1];
                uint32_t rank = 0;
                for (uint32_t j = b0; j < b1; ++j) rank += bkt[j] < k_ ? 1u : 0u;
                skey[b0 + rank] = k_;
            }
        }
    }
    if (!sorted_by_buckets)
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
            const bool local = jj < chunk;
            if (!local) q_lds_barrier();
            for (uint32_t r = 0; r < chunk; r += 64) {
                const uint32_t i = wave * chunk + r + lane;
                const uint32_t pi = i ^ jj;
                if (i < n2 && pi > i) {
                    const uint64_t x = skey[i], y = skey[pi];
                    if ((x > y) == ((i & k) == 0)) { skey[i] = y; skey[pi] = x; }
                }
            }
            if (local) q_wave_lds_sync();
            else q_lds_barrier();
        }
    }
    q_lds_barrier();
    // expand the padding-lane replays while writing: thread t owns sorted entries [kPer * t, kPer * (t + 1))
    uint64_t pay[kPer];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tid * kPer + u;
        pay[u] = 0;
        if (i < ncand) {
            if (PAYLDS) {
                pay[u] = spay[skey[i] & (kCap - 1u)];
            } else {
                const QCand c = load((uint32_t)skey[i] & (kCap - 1u));
                pay[u] = (uint64_t)c.key | ((uint64_t)(c.val_reps & 0xfffu) << 32) | ((uint64_t)c.slot << 48);
            }
            mine += 1u + ((uint32_t)(pay[u] >> 40) & 15u);
        }
    }
    const uint32_t incl = q_wave_incl_sum(mine);
    if (lane == 63) wcnt[wave] = incl;
    q_lds_barrier();
    uint32_t wp = incl - mine;
    for (uint32_t w = 0; w < kQWaves; ++w) {
        const uint32_t c_ = wcnt[w];
        if (w < wave) wp += c_;
        out_count += c_;
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        const uint32_t i = tid * kPer + u;
        if (i < ncand) {
            const uint32_t reps = 1u + ((uint32_t)(pay[u] >> 40) & 15u);
            const uint64_t en = (pay[u] & 0xffffffffffull & ~(0xffull << 40)) | ((pay[u] >> 48) << 40);
            for (uint32_t r = 0; r < reps; ++r, ++wp)
                if (wp < cap) stream[wp] = en;
        }
    }
    return out_count;
}

// How the scan order (V vectors) of a query is shared by its G workgroups: everybody walks the first block [0, B) for the bound,
// workgroup g then takes [lo, hi) of the rest.  `handicap` (vectors, a multiple of 64; 0 = even split): what workgroup 0's extra
// work in the first block — it alone derives the per-epoch bounds and writes the block's candidates out — is worth in chunk
// vectors; its chunk is shorter by that much (possibly empty) and the others share the difference.  Lone query, 10^5 codes in 12
// workgroups: workgroup 0's first block 11.4 K cycles against 4.8 K, a chunk of 3.8 K vectors 5.5 K cycles (phase stamps).
// 32-bit arithmetic whenever the order fits (the 64-bit divisions are ~0.4 K cycles of scalar-free VALU each at kernel entry).
struct QRange { uint64_t lo, hi; };
__device__ __forceinline__ QRange q_chunk_range(uint64_t V, uint64_t B, int G, int g, uint32_t handicap) {
    QRange r;
    const uint64_t rest = V - B;
    if (G < 2 || handicap == 0) {
        const uint64_t chunk = rest < 0xffffff00ull ? (uint64_t)((((uint32_t)rest + (uint32_t)G - 1u) / (uint32_t)G + 63u) / 64u * 64u)
                                                    : ((rest + G - 1) / G + 63) / 64 * 64;
        r.lo = min(V, B + (uint64_t)g * chunk);
        r.hi = min(V, r.lo + chunk);
        return r;
    }
    uint64_t c0, per1;
    if (rest < 0xff000000ull) {
        const uint32_t per = (((uint32_t)rest + handicap + (uint32_t)G - 1u) / (uint32_t)G + 63u) / 64u * 64u;
        const uint32_t c0_ = min(per > handicap ? per - handicap : 0u, (uint32_t)rest);
        c0 = c0_;
        per1 = (((uint32_t)rest - c0_ + (uint32_t)G - 2u) / (uint32_t)(G - 1) + 63u) / 64u * 64u;
    } else {
        const uint64_t per = ((rest + handicap + G - 1) / G + 63) / 64 * 64;
        c0 = min(per > handicap ? per - handicap : (uint64_t)0, rest);
        per1 = ((rest - c0 + G - 2) / (G - 1) + 63) / 64 * 64;
    }
    if (g == 0) {
        r.lo = B;
        r.hi = B + c0;
    } else {
        r.lo = min(V, B + c0 + (uint64_t)(g - 1) * per1);
        r.hi = min(V, r.lo + per1);
    }
    return r;
}
constexpr uint32_t kWg0Handicap = 6144;

// OCC = waves per SIMD the register budget is sized for: 8 = two workgroups per CU (64 VGPRs), 4 = one (128 VGPRs).
// NT = non-temporal code loads (lists that stream from HBM anyway); a database that fits the 256 MiB Infinity Cache
// keeps the default policy and is re-read from the cache by every query.
// MULTI = several workgroups per query (small batches; A.G); the single-workgroup instantiation stays lean.
// HEAD  = the kernel serves as the HEAD of the level-structured path: it scans only the first A.head_codes codes of
//         every query's scan order (what the planner's bound levels 0..k0-1 would cover, same cuts) with the caller's
//         int8 tables, and emits straight into that path's structures — unordered Cand records in the query's region,
//         counts in QueryState::hist[level 0] — so that ONE launch replaces a dependent chain of k0 short level
//         launches; the later levels derive their bounds from it and sort_cands_kernel orders everything.
template <int M, int U, int OCC, bool NT, bool MULTI, bool HEAD, int WG = 1024>
__device__ __forceinline__ void scan_query_body(const QueryKernelArgs& A) {
    using C = QCfg<M, WG>;
    constexpr int kQWG = WG, kQWaves = WG / 64;                  // this instantiation's workgroup (shadows the 1024-thread default)
    if (reinterpret_cast<uintptr_t>((q_lds_bytes_t)qsmem) != 0) __builtin_trap();   // the lookups use absolute LDS addresses
    constexpr int DW = C::DW, CPL = C::CPL;
    float* vals = reinterpret_cast<float*>(qsmem + C::VALS_OFF);
    float* wtab = reinterpret_cast<float*>(qsmem + C::WTAB_OFF);
    unsigned char* tq = qsmem + C::TQ_OFF;
    uint32_t* misc = reinterpret_cast<uint32_t*>(qsmem + C::MISC_OFF);
    uint32_t* hist = misc;                       // [256] during the select, [128] value histogram during the scan
    uint32_t* wcnt = misc + 256;                 // [16] per-wave entry counts of one segment (+ pre-scan hand-over slots)
    uint32_t& s_nvals = misc[320];
    uint32_t& s_prefix = misc[321];
    uint32_t& s_k = misc[322];
    uint32_t& s_binc = misc[323];                // select: keys in the chosen digit's bin
    uint32_t& s_sel = misc[328];                 // select: keys collected by the short cut
    uint32_t& s_count = misc[325];
    float* redf = reinterpret_cast<float*>(misc + 336);   // [16] per-wave minima

    const int G = MULTI ? A.G : 1;                               // workgroups per query (small batches: > 1)
    const int q = (int)blockIdx.x / G, g = (int)blockIdx.x % G;
    const int wgi = (int)blockIdx.x;                             // index of this workgroup's output regions
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int ma = A.ma;
    const int32_t* __restrict__ assign = A.assign + (size_t)q * ma;
    const PartDesc* __restrict__ parts = A.parts;
    const size_t tbase = (size_t)q * ma * (M * 16);
    int8_t* __restrict__ qt_all = A.qtables + tbase;
    uint64_t* __restrict__ stream = A.stream + (size_t)wgi * A.cap;
    const uint32_t R = A.R;

    // A lone query's input lies in the kernel-argument segment, which the host has just written: every first touch of one of its
    // ~25 lines is a miss to memory (~2 K cycles), and the front touches them one dependent step at a time.  One load per lane
    // over the whole payload, issued here and not waited for before the pre-scan is through, brings the lines in together.
    uint32_t warm = 0;
    if constexpr (MULTI && !HEAD) {
        if (A.inline_input) {
            const uint32_t words = (A.inline_off_tables + (uint32_t)ma * (uint32_t)(M * 16) * 4u) / 4u;
            warm = reinterpret_cast<const uint32_t*>(A.assign)[min(tid * 8u, words - 1u)];   // (every 32 bytes: each 128-byte line four times)
        }
    }
    const uint64_t clk0 = __builtin_readcyclecounter();          // phase clocks (QueryOut::pad): 1/16 shader cycles
#ifdef QADC_STAMPS
    uint64_t stamps[16];
#define STAMP(k) stamps[k] = __builtin_readcyclecounter()
    for (int i_ = 0; i_ < 16; ++i_) stamps[i_] = clk0;
#else
#define STAMP(k) do {} while (0)
#endif
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) { s_nvals = 0; s_count = 0; misc[329] = 0xffffffffu; misc[330] = 0; }   // (329 / 330: smallest / largest pre-scan key)
    __syncthreads();

    // ---- small batches (MULTI): a query's time is LATENCY — every ramp epoch of the walk below would wait out one
    // memory round trip.  The codes a workgroup is going to walk first do not depend on the front at all: the first
    // block (kFirstBlock vectors, the bound's ramp) and the first kResident vectors of the workgroup's own chunk are
    // loaded into registers as soon as the pre-scan loop is through (prefetch() below), 8 / 12 loads per lane in flight
    // under the select and the quantizer, and the walk starts from registers.  Only when they lie in the query's first probed partition as complete vectors of
    // complete codes (a flat list, or a first partition of >= 4096 vectors); anything else takes the plain walk. ----
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gvec_t;
    constexpr uint32_t kFirstBlock = 4096, kResident = M == 16 ? 6144 : 8192;   // (vectors; 16x4: 90 -> 98 VGPRs of 128 with 6144)
    constexpr bool RES = MULTI && !HEAD;
    u32x4 fb[RES ? kFirstBlock / kQWG : 1], cf[RES ? kResident / kQWG : 1];
    uint64_t V_top = 0;
    bool resident = false;
    int a_res = 0;
    uint32_t res_lo = 0, res_hi = 0;                             // vectors [res_lo, res_hi) of partition a_res are in cf[]
    UDesc d_res{};
    if constexpr (RES) {
        auto part_of = [&](int a_) { return A.inline_input ? q * ma + a_ : assign[a_]; };   // (inline input: assign[] = 0, 1, 2 ...)
        if (ma == 1) {                                           // a flat list: nothing to add up
            V_top = (q_uni(parts[part_of(0)].n) + CPL - 1) / CPL;
        } else {
            uint32_t vmine = 0;
            for (int a_ = tid; a_ < ma; a_ += kQWG) vmine += (parts[part_of(a_)].n + CPL - 1) / CPL;
            vmine = q_wave_sum(vmine);
            if (lane == 0) wcnt[wave] = vmine;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kQWaves; ++w) V_top += wcnt[w];
            __syncthreads();
        }
        while (a_res < ma && q_uni(parts[part_of(a_res)].n) == 0) ++a_res;
        if (a_res < ma) {
            d_res = q_load_desc(parts, (int)q_uni((uint32_t)part_of(a_res)));
            const uint64_t B_ = min(V_top, (uint64_t)kFirstBlock);
            const uint32_t nfull0 = d_res.n / CPL;
            resident = B_ == kFirstBlock && nfull0 >= kFirstBlock && d_res.n > kFirstBlock * CPL;   // (the partition's last code — the one
                                                                 //  with padding-lane replays — never lies in the first block)
            const QRange mine_ = q_chunk_range(V_top, B_, G, g, resident ? kWg0Handicap : 0u);
            const uint64_t lo_ = mine_.lo, hi_ = mine_.hi;
            if (resident) {
                res_lo = res_hi = (uint32_t)min(lo_, (uint64_t)nfull0);
                if (lo_ < nfull0) res_hi = (uint32_t)min(min(hi_, (uint64_t)nfull0), lo_ + kResident);
            }
        }
    }
    STAMP(1);
    auto prefetch_fb = [&]() {                                   // the first block ...
        if constexpr (RES) {
            if (resident) {
                const gvec_t src0 = (gvec_t)(uintptr_t)d_res.codes;
#pragma unroll
                for (int j = 0; j < (int)(kFirstBlock / kQWG); ++j) fb[j] = src0[j * kQWG + tid];
            }
        }
    };
    auto prefetch_cf = [&]() {                                   // ... and the first vectors of the workgroup's own chunk
        if constexpr (RES) {
            if (resident) {
                const gvec_t src0 = (gvec_t)(uintptr_t)d_res.codes;
#pragma unroll
                for (int j = 0; j < (int)(kResident / kQWG); ++j) {
                    cf[j] = u32x4{0, 0, 0, 0};
                    if (res_lo + j * kQWG + tid < res_hi) cf[j] = src0[res_lo + j * kQWG + tid];
                }
            }
        }
    };

    // The record of a finished workgroup.  Small batches (RES): the host does not wait for the kernel's completion
    // signal but polls the records in its mapped result block — the done bit (flags & 4) is stored last, alone, behind a
    // system-scope release fence issued by this ONE lane after the workgroup barrier that follows the stream stores
    // (the release is cumulative: it covers what the other lanes stored before the barrier; every lane fencing for
    // itself costs 4 us, and merely waiting for the stores' acknowledgements is not enough — tried, the host then
    // sometimes sees the bit before the entries).
    auto publish = [&](QueryOut o) {
        if constexpr (RES) {
            const uint32_t fl = o.flags;
            o.flags = 0;
            A.qout[wgi] = o;
            __threadfence_system();                              // ONE lane's release covers the workgroup's stores (see the caller)
            __hip_atomic_store(&A.qout[wgi].flags, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            A.qout[wgi] = o;
        }
    };
    float qmin = 0.0f, qmax = 0.0f;
    uint32_t flags = 0;
    if (A.ftables) {
        float* __restrict__ ft_all = A.ftables + tbase;
        // ---- 1. float pre-scan of the starts.  waves_per_probe waves share a probe when ma < 16. ----
        // total starts of the query decide where the values live (LDS, or the global scratch when there are many)
        uint32_t total_starts = 0;
        if (RES && ma == 1) {                                    // a flat list: one descriptor, nothing to add up (two barriers less)
            const PartDesc& d = parts[A.inline_input ? q : assign[0]];
            total_starts = q_uni(d.global_n) ? q_uni(d.start_n) : 0u;
        } else {
            uint32_t mine = 0;
            for (int a = tid; a < ma; a += kQWG) {
                const PartDesc& d = parts[assign[a]];
                mine += d.global_n ? d.start_n : 0u;
            }
            mine = q_wave_sum(mine);
            if (lane == 0) wcnt[wave] = mine;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < kQWaves; ++w) total_starts += wcnt[w];
            __syncthreads();
        }
        const bool in_lds = total_starts <= (uint32_t)C::FCAP;
        float* __restrict__ gvals = A.fvals + (size_t)wgi * A.fcap;

        STAMP(2);
        const int wpp = ma >= kQWaves ? 1 : kQWaves / ma;        // waves per probe
        const int pstride = kQWaves / wpp;                       // probes in flight
        float* mytab = wtab + wave * (M * 16);
        const int sum_mode = A.sum_mode;                         // grouping of the float adds (qadc_float_sum.h)
        float lmin = FLT_MAX;
        uint32_t kmn = 0xffffffffu, kmx = 0;                     // (small batches) smallest / largest key among this lane's pre-scan values
        const int pslot = (int)wave / wpp, sub = (int)wave % wpp;
        // The pre-scan is LATENCY, not work: a wave's round used to be a chain of dependent memory round trips — assign[a],
        // the partition descriptor, the probe's float table, then one 16-byte code load per 64 starts, each waited for before the
        // next was issued (C5 shape: 4 rounds x ~12 round trips per wave; the front was half of the head launch).  Now the NEXT
        // round's descriptor and table are fetched into registers while the current round's starts are evaluated (round_in),
        // and the starts loop issues kPB code loads before it consumes the first.
        constexpr int kTV = M * 16 / 64;                         // table floats per lane
        constexpr int kPB = 4;                                   // start vectors in flight per lane
        struct RoundIn {
            float tv[kTV];
            uint32_t sn;
            const uint8_t* sc;
            bool active;
        };
        auto round_in = [&](int a0) {
            RoundIn r;
            const int a = a0 + pslot;
            r.active = pslot < pstride && a < ma;
            r.sn = 0;
            r.sc = nullptr;
#pragma unroll
            for (int k = 0; k < kTV; ++k) r.tv[k] = 0.0f;
            if (r.active) {
                const PartDesc& d = parts[assign[a]];            // (before the table: the code loads wait for these words only)
                const uint32_t gn = d.global_n, st_n = d.start_n;
                const uint8_t* st_p = d.starts;
                const uint8_t* co_p = d.codes;
                r.sn = gn ? st_n : 0u;
                r.sc = st_p ? st_p : co_p;
                const float* __restrict__ ft = ft_all + (size_t)a * (M * 16);
#pragma unroll
                for (int k = 0; k < kTV; ++k) r.tv[k] = ft[k * 64 + (int)lane];
            }
            return r;
        };
        RoundIn cur = round_in(0);
        // the loop bounds are workgroup-uniform (a wave without a probe idles through the barriers)
        for (int a0 = 0; a0 < ma; a0 += pstride) {
            RoundIn nxt = round_in(a0 + pstride);                // (a0 + pstride >= ma: inactive, no loads)
            const bool active = cur.active;
            uint32_t sn = 0, base = 0;
            const uint8_t* sc = nullptr;
            const uint32_t step = 64u * (uint32_t)wpp;
            const uint32_t i_first = (uint32_t)sub * 64u + lane;
            uint32_t dwv[kPB][DW];
            auto load_batch = [&](uint32_t i0) {
#pragma unroll
                for (int u = 0; u < kPB; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * step;
#pragma unroll
                    for (int w_ = 0; w_ < DW; ++w_) dwv[u][w_] = 0;
                    if (i < sn) {
                        if constexpr (M == 16) {
                            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                            const u32x2 v = ((const __attribute__((address_space(1))) u32x2*)(uintptr_t)sc)[i];
                            dwv[u][0] = v.x; dwv[u][1] = v.y;
                        } else {
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 v = ((const __attribute__((address_space(1))) u32x4*)(uintptr_t)sc)[i];
                            dwv[u][0] = v.x; dwv[u][1] = v.y; dwv[u][2] = v.z; dwv[u][3] = v.w;
                        }
                    }
                }
            };
            if (active) {
                // (round 6: the descriptor is fetched before the table in round_in, and the first start vectors are requested
                // here, BEFORE the wave waits for its table's floats: the lone query's pre-scan was two memory round trips in a
                // row — table, then codes — 7.2 K cycles for one code per lane)
                sn = q_uni(cur.sn);
                sc = reinterpret_cast<const uint8_t*>(q_uni64(reinterpret_cast<uint64_t>(cur.sc)));
                if constexpr (RES) load_batch(i_first);          // (small batches only: the 64-register head kernels would spill)
            }
            // the first block's vectors (small batches): requested behind the first start vectors, so that these 64 KiB — ~1 K
            // cycles of the CU's vector-memory issue — go out under the pre-scan's lookups, not after them (the own chunk's
            // vectors follow after the pre-scan: all of them in registers across it would spill)
            if (RES && a0 == 0) prefetch_fb();
            if (active) {
                q_wave_lds_sync();
#pragma unroll
                for (int k = 0; k < kTV; ++k) {
                    mytab[k * 64 + (int)lane] = cur.tv[k];
                    if (sub == 0) lmin = fminf(lmin, cur.tv[k]);
                }
                q_wave_lds_sync();
                if (sub == 0 && sn) {
                    if (lane == 0) base = atomicAdd(&s_nvals, sn);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (lane == 0) wcnt[16 + pslot] = base;      // hand the base to the probe's other waves
                }
            }
            if (wpp > 1) {
                q_lds_barrier();
                if (active && sn) base = wcnt[16 + pslot];
            }
            for (uint32_t i0 = i_first; i0 < sn; i0 += step * kPB) {
                if (!RES || i0 != i_first) load_batch(i0);
                const uint32_t tb = (uint32_t)C::WTAB_OFF + wave * (uint32_t)(M * 16 * 4);   // mytab, as an absolute LDS address
#pragma unroll
                for (int u = 0; u < kPB; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * step;
                    if (RES && i - lane >= sn) continue;         // (wave-uniform: none of the wave's lanes has a start here)
                    const float cand = q_prescan_sum<M>(dwv[u], tb, sum_mode);   // (lanes past the starts sum code 0: not stored)
                    if (i < sn) {
                        if (in_lds) vals[base + i] = cand;
                        else gvals[base + i] = cand;
                        if constexpr (RES) {
                            const uint32_t k_ = q_fkey(cand);
                            kmn = min(kmn, k_);
                            kmx = max(kmx, k_);
                        }
                    }
                }
            }
            if (wpp > 1) q_lds_barrier();                        // the hand-over slot is reused by the next round
            cur = nxt;
        }
        STAMP(3);
        if constexpr (RES) asm volatile("" ::"v"(warm));         // (the warm-up load's destination is live until here)
        // the quantizer's first kQB table entries per lane (all of them for up to 32 probes at 16x4): requested here, they
        // arrive under the select (they used to be waited for between the select and the quantizer: 2.4 K cycles for 256 floats)
        constexpr int kQB = 8;                                   // table entries in flight per lane (the clamp's store to the same
                                                                 // array keeps the compiler from overlapping the loads by itself)
        const int all = ma * M * 16;
        float qtv0[RES ? kQB : 1];                               // (small batches only: the 64-register head kernels would spill them)
        if constexpr (RES) {
#pragma unroll
            for (int u = 0; u < kQB; ++u) {
                const int i = (int)tid + u * kQWG;
                qtv0[u] = i < all ? ft_all[i] : 0.0f;
            }
        }
        prefetch_cf();                                           // in flight under the select and the quantizer
        // qmin = min over ALL ma tables (db_query_4.cpp:258)
        lmin = q_wave_min(lmin);
        if (lane == 0) redf[wave] = lmin;
        if constexpr (RES) {                                     // the keys' range, for the select's fitted digit (same barrier)
            kmn = q_wave_scan_bits(kmn, 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); });
            kmx = q_wave_scan_bits(kmx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
            if (lane == 63) {
                atomicMin(&misc[329], kmn);
                atomicMax(&misc[330], kmx);
            }
        }
        __threadfence_block();
        __syncthreads();
        qmin = redf[0];
#pragma unroll
        for (int w = 1; w < kQWaves; ++w) qmin = fminf(qmin, redf[w]);

        STAMP(4);
        // ---- R-th smallest of the pre-scan values = tmp_bh.max() (db_query_4.cpp:259); FLT_MAX if fewer than R ----
        // A 4-pass radix select (8-bit digits, histogram in LDS), with two short cuts measured on the phase stamps of round 4
        // (C3 shape, 7.8 K values: pass 5.6 K cycles, second pass 9 K — an LDS atomic per value on a handful of hot bins —;
        // C5, 39 K values in the global scratch: 28 K / 44 K per pass, 18 K of it reading the values):
        //  * THRESHOLD FIRST (n > 2048).  R is a percent or less of the values.  Wave 0 ranks 64 evenly spaced values in
        //    registers; the m-th smallest of them (m = 1.5 x the R-th value's expected rank in such a sample, + 3) is a
        //    threshold T some percent up the distribution.  ONE pass over the values keeps every key <= T (all of them:
        //    ranks below T stay exact) in the dead per-wave table area, and the radix passes run over those few hundred
        //    keys.  Fewer than R kept (an unlucky sample) or more than the area holds: the passes run over all values
        //    (tests/test_gpu_parity.py builds a list whose sample is unlucky on purpose.)
        //  * once the chosen digit's bin holds <= 256 keys they are collected and ranked by counting.
        // Either way the result is the exact R-th smallest key.
        const uint32_t n = s_nvals;
        if (n < R) {
            qmax = FLT_MAX;
        } else {
            uint32_t* list = reinterpret_cast<uint32_t*>(qsmem + C::WTAB_OFF);
            constexpr uint32_t kListCap = (uint32_t)C::WTAB_BYTES / 4u;
            uint32_t nlist = 0;                                  // != 0: the passes read list[0 .. nlist)
            // every value's key through f: the kept keys, the values in LDS, or the global scratch (more starts than the LDS
            // budget holds — the C5 shape: eight loads in flight per lane; one dependent load per iteration made this select
            // 2/3 of that shape's front)
            auto for_each_key = [&](auto f) {
                if (nlist) {
                    for (uint32_t i = tid; i < nlist; i += kQWG) f(list[i]);
                } else if (in_lds) {
                    for (uint32_t i = tid; i < n; i += kQWG) f(q_fkey(vals[i]));
                } else {
                    constexpr int kSB = 8;
                    for (uint32_t i0 = tid; i0 < n; i0 += kQWG * kSB) {
                        float fv[kSB];
#pragma unroll
                        for (int u = 0; u < kSB; ++u) {
                            const uint32_t i = i0 + (uint32_t)u * kQWG;
                            fv[u] = i < n ? gvals[i] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < kSB; ++u)
                            if (i0 + (uint32_t)u * kQWG < n) f(q_fkey(fv[u]));
                    }
                }
            };
            if (n > 2048u) {
                if (wave == 0) {
                    const uint32_t i = (uint32_t)((uint64_t)lane * n / 64u);   // (evenly spaced: the values lie probe by probe, nearest centroid first)
                    const uint32_t skey = q_fkey(in_lds ? vals[i] : gvals[i]);
                    const uint32_t m = min(64u, (uint32_t)((96ull * R + n - 1u) / n) + 3u);
                    uint32_t less = 0, le = 0;
#pragma unroll 8
                    for (int j = 0; j < 64; ++j) {
                        const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)skey, j);
                        less += kj < skey ? 1u : 0u;
                        le += kj <= skey ? 1u : 0u;
                    }
                    if (less < m && m <= le) s_prefix = skey;
                    if (lane == 0) s_sel = 0;
                }
                __syncthreads();
                const uint32_t T = s_prefix;
                for_each_key([&](uint32_t key) {
                    if (key <= T) {
                        const uint64_t hits = __builtin_amdgcn_ballot_w64(true);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                        uint32_t base = 0;
                        if (rank == 0) base = atomicAdd(&s_sel, (uint32_t)__popcll(hits));
                        base = q_uni(base);
                        if (base + rank < kListCap) list[base + rank] = key;
                    }
                });
                __syncthreads();
                const uint32_t c = s_sel;
                if (c >= R && c <= kListCap) nlist = c;          // (workgroup-uniform)
            } else if (in_lds && n <= 256u) {                     // a handful of values: they ARE the list
                for (uint32_t i = tid; i < n; i += kQWG) list[i] = q_fkey(vals[i]);
                __syncthreads();
                nlist = n;
            }
            // Round 6.  Up to 256 keys (kept by the threshold, or all there are) are ranked at once.  More: ONE histogram pass
            // over a digit fitted to the keys — bucket = (key - smallest key) >> sh, sh such that the keys' range fills 256
            // buckets — which puts a dozen keys into the bucket of rank R where the fixed top byte of a float key (sign and
            // seven exponent bits) puts nearly all of them into two or three; that bucket's keys are collected and ranked.
            // The lone query's 10^3 pre-scan values went through four histogram passes of three barriers each before:
            // 6.9 K cycles.  A bucket of more than 256 keys (tie-heavy tables) leaves it to the four fixed passes below.
            uint32_t& s_kmin = misc[329];
            uint32_t& s_kmax = misc[330];
            bool direct = nlist != 0 && nlist <= 256u;
            if (direct) {
                if (tid < nlist) {
                    const uint32_t key = list[tid];
                    const uint4* l4 = reinterpret_cast<const uint4*>(list);
                    uint32_t less = 0, le = 0, j = 0;
                    for (; j + 4 <= nlist; j += 4) {
                        const uint4 kk = l4[j >> 2];
                        less += (kk.x < key ? 1u : 0u) + (kk.y < key ? 1u : 0u) + (kk.z < key ? 1u : 0u) + (kk.w < key ? 1u : 0u);
                        le += (kk.x <= key ? 1u : 0u) + (kk.y <= key ? 1u : 0u) + (kk.z <= key ? 1u : 0u) + (kk.w <= key ? 1u : 0u);
                    }
                    for (; j < nlist; ++j) {
                        const uint32_t kj = list[j];
                        less += kj < key ? 1u : 0u;
                        le += kj <= key ? 1u : 0u;
                    }
                    if (less < R && R <= le) s_prefix = key;
                }
                __syncthreads();
            } else {
                // (small batches without a threshold cut: the pre-scan left the range of ALL keys in s_kmin / s_kmax, and the
                // histogram's words are still clear from the kernel's entry: two barriers less)
                if (!(RES && nlist == 0)) {
                    if (tid == 0) { s_kmin = 0xffffffffu; s_kmax = 0; }      // (s_sel — the threshold's count — may still be being read)
                    if (tid < 256) hist[tid] = 0;
                    __syncthreads();
                    uint32_t mn = 0xffffffffu, mx = 0;
                    for_each_key([&](uint32_t key) { mn = min(mn, key); mx = max(mx, key); });
                    mn = q_wave_scan_bits(mn, 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); });
                    mx = q_wave_scan_bits(mx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
                    if (lane == 63) {
                        atomicMin(&s_kmin, mn);
                        atomicMax(&s_kmax, mx);
                    }
                    __syncthreads();
                }
                const uint32_t kmin = s_kmin, range = s_kmax - kmin;
                const uint32_t sh = range >= 256u ? 24u - (uint32_t)__builtin_clz(range) : 0u;
                for_each_key([&](uint32_t key) { atomicAdd(&hist[(key - kmin) >> sh], 1u); });
                __syncthreads();
                if (tid < 64) {                                  // wave 0: 4 buckets per lane, pick the one holding rank R
                    const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                    const uint32_t sum4 = c0 + c1 + c2 + c3;
                    const uint32_t incl = q_wave_incl_sum(sum4);
                    const uint32_t excl = incl - sum4;
                    if (incl >= R && excl < R) {                 // exactly one lane
                        uint32_t run = excl, digit = 4 * tid;
                        const uint32_t cs4[4] = {c0, c1, c2, c3};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (run + cs4[j] >= R) { digit = 4 * tid + j; break; }
                            run += cs4[j];
                        }
                        s_prefix = digit;
                        s_k = R - run;
                        s_binc = hist[digit];
                    }
                    if (tid == 0) s_sel = 0;
                }
                __syncthreads();
                if (s_binc <= 256u) {                            // (workgroup-uniform)
                    const uint32_t b_ = s_prefix, k2 = s_k;
                    for_each_key([&](uint32_t key) {
                        if (((key - kmin) >> sh) == b_) hist[atomicAdd(&s_sel, 1u)] = key;
                    });
                    __syncthreads();
                    const uint32_t c = s_sel;
                    if (tid < c) {
                        const uint32_t key = hist[tid];
                        const uint4* h4 = reinterpret_cast<const uint4*>(hist);
                        uint32_t less = 0, le = 0, j = 0;
                        for (; j + 4 <= c; j += 4) {
                            const uint4 kk = h4[j >> 2];
                            less += (kk.x < key ? 1u : 0u) + (kk.y < key ? 1u : 0u) + (kk.z < key ? 1u : 0u) + (kk.w < key ? 1u : 0u);
                            le += (kk.x <= key ? 1u : 0u) + (kk.y <= key ? 1u : 0u) + (kk.z <= key ? 1u : 0u) + (kk.w <= key ? 1u : 0u);
                        }
                        for (; j < c; ++j) {
                            const uint32_t kj = hist[j];
                            less += kj < key ? 1u : 0u;
                            le += kj <= key ? 1u : 0u;
                        }
                        if (less < k2 && k2 <= le) s_prefix = key;
                    }
                    __syncthreads();
                    direct = true;
                }
            }
            if (!direct && tid == 0) { s_prefix = 0; s_k = R; }
            for (int pass = 0; pass < 4 && !direct; ++pass) {
                const int lo = 24 - 8 * pass;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const uint32_t prefix = s_prefix;
                const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (lo + 8));
                auto count_key = [&](uint32_t key) {
                    if ((key & himask) == prefix) {
                        // the first digit of float keys (sign + 7 exponent bits) puts almost every value into two or
                        // three bins: lanes that share the leading lane's digit add ONE count per wave
                        const uint32_t dg = (key >> lo) & 0xffu;
                        bool counted = false;
#pragma unroll
                        for (int it = 0; it < 3; ++it) {
                            if (!counted) {
                                const uint32_t lead = q_uni(dg);
                                if (dg == lead) {
                                    const uint64_t same = __builtin_amdgcn_ballot_w64(true);
                                    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u)) == 0)
                                        atomicAdd(&hist[lead], (uint32_t)__popcll(same));
                                    counted = true;
                                }
                            }
                        }
                        if (!counted) atomicAdd(&hist[dg], 1u);
                    }
                };
                for_each_key(count_key);
                __syncthreads();
                if (tid < 64) {                                  // wave 0: 4 bins per lane, pick the digit holding rank k
                    const uint32_t k = s_k;
                    const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                    const uint32_t sum4 = c0 + c1 + c2 + c3;
                    const uint32_t incl = q_wave_incl_sum(sum4);
                    const uint32_t excl = incl - sum4;
                    if (incl >= k && excl < k) {                 // exactly one lane
                        uint32_t run = excl, digit = 4 * tid;
                        const uint32_t cs4[4] = {c0, c1, c2, c3};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (run + cs4[j] >= k) { digit = 4 * tid + j; break; }
                            run += cs4[j];
                        }
                        s_prefix = prefix | (digit << lo);
                        s_k = k - run;
                        s_binc = hist[digit];
                    }
                    if (tid == 0) s_sel = 0;
                }
                __syncthreads();
                if (pass < 3 && s_binc <= 256u) {                // the bin's keys into the histogram's words, ranked by counting
                    const uint32_t prefix2 = s_prefix, k2 = s_k, mask2 = 0xffffffffu << lo;
                    for_each_key([&](uint32_t key) {
                        if ((key & mask2) == prefix2) hist[atomicAdd(&s_sel, 1u)] = key;
                    });
                    __syncthreads();
                    const uint32_t c = s_sel;
                    if (tid < c) {
                        const uint32_t key = hist[tid];
                        const uint4* h4 = reinterpret_cast<const uint4*>(hist);
                        uint32_t less = 0, le = 0, j = 0;
                        for (; j + 4 <= c; j += 4) {             // (four keys per LDS read: one dependent read per key made 256 keys 25 K cycles)
                            const uint4 kk = h4[j >> 2];
                            less += (kk.x < key ? 1u : 0u) + (kk.y < key ? 1u : 0u) + (kk.z < key ? 1u : 0u) + (kk.w < key ? 1u : 0u);
                            le += (kk.x <= key ? 1u : 0u) + (kk.y <= key ? 1u : 0u) + (kk.z <= key ? 1u : 0u) + (kk.w <= key ? 1u : 0u);
                        }
                        for (; j < c; ++j) {
                            const uint32_t kj = hist[j];
                            less += kj < key ? 1u : 0u;
                            le += kj <= key ? 1u : 0u;
                        }
                        if (less < k2 && k2 <= le) s_prefix = key;
                    }
                    __syncthreads();
                    break;
                }
            }
            qmax = q_funkey(s_prefix);
        }
        STAMP(5);
        // ---- 2. qmin / clamp / QuantizerMAX (db_query_4.cpp:258-284, 37-71) ----
        if (qmin < 0) { qmin = 0; flags |= 2u; }
        if ((double)qmax > 1e30) flags |= 1u;                   // (a double compare, as db_query_4.cpp:271: 1e30f itself is above 1e30)
        const float delta = (qmax - qmin) / 127;
        const float scale = 127.0f / (qmax - qmin);
        for (int i0 = tid; i0 < all; i0 += kQWG * kQB) {
            float tv[kQB];
#pragma unroll
            for (int u = 0; u < kQB; ++u) {
                const int i = i0 + u * kQWG;
                if constexpr (RES) tv[u] = i0 == (int)tid ? qtv0[u] : (i < all ? ft_all[i] : 0.0f);
                else tv[u] = i < all ? ft_all[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kQB; ++u) {
                const int i = i0 + u * kQWG;
                if (i >= all) continue;
                float v = tv[u];
                if (v < 0) { v = 0; if (G == 1) ft_all[i] = 0; }    // (G > 1: another workgroup may still pre-scan the unclamped tables)
                int8_t o;
                if (flags & 1u) o = 127;
                else if (v >= qmax) o = 127;
                else o = (int8_t)(int)(A.quant_mode == 0 ? (v - qmin) / delta : (v - qmin) * scale);
                qt_all[i] = o;
                if constexpr (RES) {                             // the first probed partition's table: straight into LDS as well
                    if (resident && i / (M * 16) == a_res) tq[i % (M * 16)] = (unsigned char)o;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (HEAD && A.front_only) {                              // this rank's share of a sharded front: no walk here
            if (tid == 0 && g == 0) {
                uint32_t* fo = A.front_out + 4 * (size_t)q;
                fo[0] = flags & 3u;
                fo[1] = __float_as_uint(qmin);
                fo[2] = __float_as_uint(qmax);
                fo[3] = 0;
            }
            return;
        }
        // (the select's histogram words become the scan's value histograms: cleared with the rest of them below)
    } else {
        prefetch_fb();
        prefetch_cf();
        if (A.front_in) {                                        // the front ran elsewhere (another rank; lone_front_kernel): its verdict travels with the tables
            flags = A.front_in[4 * (size_t)q] & 3u;
            qmin = __uint_as_float(A.front_in[4 * (size_t)q + 1]);
            qmax = __uint_as_float(A.front_in[4 * (size_t)q + 2]);
        }
    }

    if (flags & 1u) {                                            // the reference prints a warning and exits: no scan
        if (HEAD) {
            if (tid == 0 && g == 0) {
                QueryState* qs = A.qstates + q;
                qs->qmin = qmin;
                qs->qmax = qmax;
                qs->flags = flags & 3u;
            }
            return;
        }
        if (tid == 0) {
            QueryOut o;
            o.count = 0; o.reps = 0; o.flags = flags | 4u; o.out_off = (uint32_t)((size_t)wgi * A.cap);
            o.qmin = qmin; o.qmax = qmax; o.pad[0] = o.pad[1] = 0;
            publish(o);
        }
        return;
    }

    const uint64_t clk1 = __builtin_readcyclecounter();
    // (HEAD with a slot limit: the walk ends with the head_slots-th probed partition that has codes HERE — on a multi-GPU
    // shard a rank may hold nothing of a query's first probes, and the grouped second phase needs a bound from local
    // candidates; ivf_count / ivf_scatter draw the same line.  The front above covered all ma.)
    int mas = ma;
    if (HEAD && A.head_slots) {
        int seen = 0;
        mas = ma;
        for (int a0 = 0; a0 < ma; a0 += 64) {                    // every wave computes it for itself: lanes = slots
            const int a_ = a0 + (int)lane;
            const bool ne = a_ < ma && parts[assign[a_]].n != 0;
            const uint64_t m = __builtin_amdgcn_ballot_w64(ne);
            const int cnt = __popcll(m);
            if (seen + cnt >= (int)A.head_slots) {               // the head_slots-th non-empty slot lies in this chunk
                uint64_t mm = m;
                for (int k = seen + 1; k < (int)A.head_slots; ++k) mm &= mm - 1;   // drop the lower set bits
                mas = a0 + (int)__builtin_ctzll(mm) + 1;
                break;
            }
            seen += cnt;
        }
    }
    // ---- 3. int8 scan in assign[] order: free-running waves, epochs, one in-workgroup sort at the end ----
    // The query's scan order is cut into EPOCHS: 64, 128, 256, ... vectors at the start (while the bound is loose),
    // then one epoch per probed partition (long partitions: one per kEpochVec vectors).  Inside an epoch the 16 waves
    // run WITHOUT any barrier: wave w takes the tiles (round*16 + w) of 64 vectors, kRounds 16-byte loads per lane in
    // flight, sums every code and compares with the epoch's bound — the R-th smallest value among the candidates of all
    // EARLIER epochs, i.e. of codes that precede every code of this epoch in scan order (the prefix-bound rule,
    // DESIGN.md section 4).  The rare qualifying code is appended, UNORDERED, to the query's candidate list in global
    // memory (one LDS atomic for the slot) and counted in the epoch's value histogram.  An epoch ends with two
    // barriers: histogram merge + the next partition's pair tables between them, the new bound after them.
    // When the walk is over, the workgroup sorts its candidates (typically ~1 K, at most kCandCap) by
    // (assign slot, position) with a bitonic network in the LDS the tables no longer need, expands the padding-lane
    // replays and writes the ordered stream.  Exactness does not depend on how the waves interleave: the bound of an
    // epoch is fixed before its first code is tested, and the order is restored by the sort.
    constexpr int kRounds = U;                                   // loads per lane in flight
    constexpr uint32_t kEpochVec = 32768;                        // longest epoch (vectors): bounds how stale a bound gets
    uint32_t* hist_done = misc;                                  // [128] candidates of the finished epochs, by value
    uint32_t* hist_cur = misc + 128;                             // [128] candidates of the running epoch
    uint32_t& s_ccount = misc[326];                              // candidates appended so far
    uint32_t& s_hreps = misc[324];                               // (HEAD, one workgroup per query) padding-lane replays among them
    QCand* __restrict__ cands = A.cands + (size_t)wgi * A.ccap;
    // small batches: the first kLdsCands candidates of a workgroup wait in LDS (where the first block's epoch counters lay) —
    // a workgroup has ~10^2 of them, and through global memory the ordering pass began with a store drain and an L2 round trip
    constexpr uint32_t kLdsCands = RES ? (uint32_t)C::FBH_BYTES / 16u : 0u;
    QCand* lcands = reinterpret_cast<QCand*>(qsmem + C::LDS_BYTES);
    auto put_cand = [&](uint32_t slot, const QCand& qc) {
        if (RES && slot < kLdsCands) lcands[slot] = qc;
        else cands[slot] = qc;
    };
    auto next_part = [&](int a_) {
        ++a_;
        while (a_ < mas && q_uni(parts[assign[a_]].n) == 0) ++a_; // empty partition (db_query_4.cpp:291-293) / no local codes
        return a_;
    };
    auto table_word = [&](int a_) -> uint32_t {
        return tid < M * 4 ? reinterpret_cast<const uint32_t*>(qt_all + (size_t)a_ * (M * 16))[tid] : 0u;
    };
    auto write_tables = [&]() {                                  // replicated pair tables from the staged int8 table (layout: QCfg)
        // The image is written in ADDRESS order — thread t stores the 16-byte units t, t + 1024, ... — so a wave's
        // ds_write_b128 covers 1 KiB contiguous: every bank once, no conflict.  (Round 3 gave a thread one (dword g, byte
        // value x) pair and let it write that pair's 128-byte row by itself: the rows of a wave's lanes lie 256 bytes apart,
        // i.e. in the SAME banks — a 32-way conflict on every store, ~4 K LDS cycles per 64 KiB image instead of ~256; the
        // 15 % bank-conflict cycles the PMC pass of round 3 charged to this kernel.)
        // A wave's units of iteration k belong to 8 (g, x) pairs, 8 lanes each; over all iterations the wave needs
        // 8 * kIter pairs.  Lane l computes pair l's dword once; ds_bpermute hands it to the lanes that replicate it.
        constexpr int kIter = C::TABLE_BYTES / 16 / kQWG;        // 16x4: 4, 32x4: 8 (4 with 16 replicas) units per thread
        constexpr int kLPP = C::R16 ? 4 : 8;                     // lanes (16-byte units) per pair's row piece
        constexpr int kPPI = 64 / kLPP;                          // pairs per wave and iteration
        constexpr int kPairs = kPPI * kIter;                     // (g, x) pairs a wave needs: 32 or 64 — or 128 (32x4 in 8 waves)
        constexpr int kNW = (kPairs + 63) / 64;                  // pair dwords a lane computes
        uint32_t w[kNW];
#pragma unroll
        for (int hf = 0; hf < kNW; ++hf) {
            const uint32_t pi = (uint32_t)hf * 64u + (lane & (uint32_t)(min(kPairs, 64) - 1));   // (16x4 in 16 waves: lanes 32..63 repeat 0..31)
            const uint32_t k = pi / kPPI, within = pi % kPPI;
            uint32_t x, g;
            constexpr uint32_t kRows = kQWG / 16;                // 256-byte rows one iteration of the workgroup covers
            constexpr uint32_t kKPR = 256u / kRows;              // iterations per 64 KiB region
            if (C::R16) {                                        // a wave-iteration = 1 KiB = rows x .. x+3, each [g0][g1][g2][g3]
                x = k * kRows + wave * 4u + (within >> 2);
                g = within & 3u;
            } else {
                x = (k % kKPR) * kRows + wave * 4u + (within >> 1);
                g = (k / kKPR) * 2u + (within & 1u);
            }
            w[hf] = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t b = 4u * g + jj;
                w[hf] |= ((uint32_t)tq[(2 * b) * 16 + (x & 15u)] + (uint32_t)tq[(2 * b + 1) * 16 + (x >> 4)]) << (8 * jj);
            }
        }
#pragma unroll
        for (int k = 0; k < kIter; ++k) {
            const uint32_t wk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((((uint32_t)(k * kPPI) & 63u) + (lane / kLPP)) * 4u),
                                                                       (int)w[(k * kPPI) / 64]);
            *reinterpret_cast<uint4*>(qsmem + ((size_t)k * kQWG + tid) * 16) = make_uint4(wk, wk, wk, wk);
        }
    };
    const uint32_t lane_lo = C::SWZ ? (tid & 15u) * 4u + ((tid >> 4) & 1u) * 64u : C::R16 ? (tid & 15u) * 4u : (tid & 31u) * 4u;
    const uint32_t lane_hi = C::SWZ ? (tid & 15u) * 4u + (1u - ((tid >> 4) & 1u)) * 64u : lane_lo | 0x10000u;
    if (tid < 256) misc[tid] = 0;                                // both histograms
    if (tid == 0) { s_ccount = 0; s_hreps = 0; }
    // ---- how the query's scan order is shared by the G workgroups of the query (G = 1: everything is "mine") ----
    // total vectors of the probed partitions, in scan order
    // (HEAD: codes of a partition the head covers = the planner's cut of bound level k0 in that partition: everything if
    // the partition ends before head_codes, else (head_codes - codes before it) rounded down to whole 16-byte vectors)
    auto eff_n = [&](uint32_t n_, uint64_t cbase_) -> uint32_t {
        if (!HEAD) return n_;
        if (A.head_codes >= cbase_ + n_) return n_;
        const uint64_t cut = A.head_codes > cbase_ ? A.head_codes - cbase_ : 0;
        return (uint32_t)(cut - cut % CPL);
    };
    uint64_t V = 0;
    if (HEAD) {
        uint64_t cb_ = 0;
        for (int a_ = 0; a_ < mas && cb_ < A.head_codes; ++a_) {
            const uint32_t n_ = q_uni(parts[assign[a_]].n);
            V += (eff_n(n_, cb_) + CPL - 1) / CPL;
            cb_ += n_;
        }
    } else if (RES) {
        V = V_top;
    } else {
        uint32_t vmine = 0;
        for (int a_ = tid; a_ < mas; a_ += kQWG) vmine += (parts[assign[a_]].n + CPL - 1) / CPL;
        vmine = q_wave_sum(vmine);
        if (lane == 0) wcnt[wave] = vmine;
        q_lds_barrier();
#pragma unroll
        for (int w = 0; w < kQWaves; ++w) V += wcnt[w];
        q_lds_barrier();
    }
    // Every workgroup walks the first block [0, B) to tighten its bound (only workgroup 0 emits from it), then its own
    // chunk of the rest.  The bound a workgroup uses for a code is the R-th smallest value of candidates from the
    // first block and from its own chunk before that code: a subset of the code's scan-order prefix, hence valid.
    const uint64_t B = MULTI ? min(V, (uint64_t)kFirstBlock) : 0;
    const QRange mine = MULTI ? q_chunk_range(V, B, G, g, (RES && resident) ? kWg0Handicap : 0u) : QRange{0, V};
    const uint64_t my_lo = mine.lo, my_hi = mine.hi;

    uint32_t bound = 127, ramp = MULTI ? 256 : 64, last_cnt = 0;   // (several workgroups per query: fewer, larger ramp epochs)
    uint32_t& s_dirty = misc[327];                               // bumped by a histogram-only walk that found a candidate
    int tables_of = -1;                                          // probe whose pair tables are in LDS
    if (tid == 0) s_dirty = 0;
    q_lds_barrier();
    STAMP(7);
    uint64_t lo1 = my_lo;                                        // where pass 1 of the plain walk starts
    uint32_t nfb = 0;                                            // entries workgroup 0 wrote straight into its stream (the first block's)
    const uint32_t dup_pos_res = RES && d_res.first_pos + d_res.n == d_res.global_n ? d_res.n - 1u : 0xffffffffu;
    const uint32_t dup_reps_res = RES ? (16u - d_res.global_n % 16u) % 16u : 0u;
    if constexpr (RES) {
        if (resident) {
            // ---- the first block in ONE step (round 6) ----
            // Until round 6 the block's 4096 vectors (in registers since the front) were walked in five ramp epochs — lookups,
            // emission, two barriers and a bound each: 13-15 K cycles — and workgroup 0, which emits the block's ~850 candidates,
            // then SORTED them (37 K cycles of its 88 K; the other workgroups order ~70): workgroup 0 was the critical path of
            // the lone query, by 14 us (phase stamps, profiles/r06_lone_query_stamps.txt).  Now every lane looks its codes up
            // once; the values below 127 are counted per EPOCH of the block (epoch 0 = codes [0, 128), then quarter octaves:
            // 25 epochs at 8192 codes) in LDS; after one barrier the waves derive every epoch's bound in parallel — the R-th
            // smallest value of ALL codes of the earlier epochs, which all precede the epoch's codes in scan order (the
            // prefix-bound rule) — and after a second one workgroup 0 writes the qualifying codes STRAIGHT
            // into its stream at their scan-order index (prefix sums over (vector row, wave, lane): no candidate buffer, no
            // sort).  The finer epochs also emit fewer candidates than the ramp did.
            // the float path staged this partition's int8 table in the quantizer; the int8 path stages it here
            if (!A.ftables) {
                if (tid < M * 4) reinterpret_cast<uint32_t*>(tq)[tid] = table_word(a_res);
                q_lds_barrier();
            }
            write_tables();
            uint32_t* fbh = reinterpret_cast<uint32_t*>(qsmem + C::LDS_BYTES);    // [kFbEpochs][128] (the launch adds FBH_BYTES)
            uint32_t* s_fbound = misc + 352;                                     // [kFbEpochs + 1]
            constexpr int kFbCodes = (int)kFirstBlock * CPL;
            constexpr int kFbEpochs = C::FB_EPOCHS;                              // epochs of the block; bound index kFbEpochs = behind it
            static_assert((1 << (kFbEpochs + 27) / 4) == kFbCodes, "epochs of the first block");
            if (g == 0)
                for (uint32_t i = tid; i < (uint32_t)kFbEpochs * 128u; i += kQWG) fbh[i] = 0;
            q_lds_barrier();                                     // image and counters ready
            tables_of = a_res;
            STAMP(8);
            auto epoch_of = [](uint32_t p) -> uint32_t {
                if (p < 128u) return 0u;
                const uint32_t lg = 31u - (uint32_t)__builtin_clz(p);
                return 1u + 4u * (lg - 7u) + ((p >> (lg - 2u)) & 3u);
            };
            constexpr int kRows = (int)(kFirstBlock / kQWG);
            uint32_t cvs[kRows];                                 // the lane's values, a byte per code
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                uint32_t dd[4] = {fb[j].x, fb[j].y, fb[j].z, fb[j].w};
                asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]));   // (keeps the lookup addresses of all rows from being formed at once)
                cvs[j] = 0;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const uint32_t cv = min(q_pair_sum<M>(dd + c * DW, lane_lo, lane_hi), 127u);
                    cvs[j] |= cv << (8 * c);
                    if (cv < 127u) {
                        const uint32_t p = ((uint32_t)j * kQWG + tid) * CPL + c;
                        atomicAdd(g == 0 ? &fbh[epoch_of(p) * 128u + cv] : &hist_done[cv], 1u);   // (the others need the block's total only)
                    }
                }
            }
            q_lds_barrier();
            if (g == 0) {
                // wave w: bounds of the epochs w + 1 and w + 17 — counts of the epochs before them, bins 2 lane and 2 lane + 1
                uint32_t c0 = 0, c1 = 0;
                int e_done = 0;
                for (int e = (int)wave + 1; e <= kFbEpochs; e += kQWaves) {
                    for (; e_done < e; ++e_done) {
                        const uint2 h2 = *reinterpret_cast<const uint2*>(&fbh[e_done * 128 + 2 * (int)lane]);
                        c0 += h2.x;
                        c1 += h2.y;
                    }
                    const uint32_t incl = q_wave_incl_sum(c0 + c1);
                    const uint32_t excl = incl - (c0 + c1);
                    const uint64_t reached = __builtin_amdgcn_ballot_w64(incl >= R);
                    uint32_t b_ = 127u;
                    if (reached != 0) {
                        const uint32_t bl = excl + c0 >= R ? 2 * lane : 2 * lane + 1;
                        b_ = min((uint32_t)__builtin_amdgcn_readlane((int)bl, (int)__builtin_ctzll(reached)), 127u);
                    }
                    if (lane == 0) s_fbound[e] = b_;
                    if (e == kFbEpochs) {                        // the block's totals: what the rest of the walk counts on
                        hist_done[2 * lane] = c0;
                        hist_done[2 * lane + 1] = c1;
                    }
                }
                if (tid == 0) s_fbound[0] = 127u;
                q_lds_barrier();
                bound = q_uni(s_fbound[kFbEpochs]);
                // emission in scan order: row j, then wave, then lane, then the code inside the vector
                const uint32_t key_base = d_res.key_base + d_res.first_pos;
                uint32_t passm = 0, incl_j[kRows];               // passm: bit (j * CPL + c) = that code qualifies
#pragma unroll
                for (int j = 0; j < kRows; ++j) {
                    uint32_t k_ = 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const uint32_t p = ((uint32_t)j * kQWG + tid) * CPL + c;
                        const uint32_t cv = (cvs[j] >> (8 * c)) & 0xffu;
                        if (cv < s_fbound[epoch_of(p)]) {
                            passm |= 1u << (j * CPL + c);
                            ++k_;
                        }
                    }
                    incl_j[j] = q_wave_incl_sum(k_);
                    if (lane == 63) fbh[j * kQWaves + (int)wave] = incl_j[j];   // (the counters are dead: every bound is out)
                }
                q_lds_barrier();
                if (wave == 0) {                                 // exclusive scan over the (row, wave) totals, in that order
                    const uint32_t t_ = lane < (uint32_t)(kRows * kQWaves) ? fbh[lane] : 0u;
                    const uint32_t in_ = q_wave_incl_sum(t_);
                    fbh[64 + lane] = in_ - t_;
                    if (lane == 63) fbh[128] = in_;
                }
                q_lds_barrier();
                nfb = fbh[128];
#pragma unroll
                for (int j = 0; j < kRows; ++j) {
                    uint32_t wp = fbh[64 + j * kQWaves + (int)wave] + incl_j[j];
#pragma unroll
                    for (int c = CPL - 1; c >= 0; --c) {         // (wp counts down from behind the lane's last entry of the row)
                        if (passm & (1u << (j * CPL + c))) {
                            --wp;
                            const uint32_t p = ((uint32_t)j * kQWG + tid) * CPL + c;
                            const uint32_t key = d_res.labels ? d_res.labels[p] : key_base + p;
                            const uint64_t en = (uint64_t)key | ((uint64_t)((cvs[j] >> (8 * c)) & 0xffu) << 32) | ((uint64_t)a_res << 40);
                            if (wp < A.cap) stream[wp] = en;
                        }
                    }
                }
                q_lds_barrier();                                 // (the counters' words hold candidates from here on)
            } else {
                bound = q_uni(q_bound_from_hist(hist_done, R, lane));
            }
            STAMP(9);
            auto proc = [&](const u32x4& v, uint32_t vec) {
                uint32_t dd[4] = {v.x, v.y, v.z, v.w};
                // (opaque copy: keeps the compiler from hoisting the 16 lookup addresses of every resident vector
                // out of the epoch loop, which would spill)
                asm volatile("" : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]));
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const uint32_t cv = min(q_pair_sum<M>(dd + c * DW, lane_lo, lane_hi), 127u);
                    if (__builtin_expect(cv < bound, 0)) {
                        const uint32_t p = vec * CPL + c;
                        const uint32_t slot = atomicAdd(&s_ccount, 1u);
                        if (slot < A.ccap) {
                            QCand qc;
                            qc.key = d_res.labels ? d_res.labels[p] : d_res.key_base + d_res.first_pos + p;
                            qc.val_reps = cv | ((p == dup_pos_res ? dup_reps_res : 0u) << 8);
                            qc.pos = p;
                            qc.slot = (uint32_t)a_res;
                            put_cand(slot, qc);
                        }
                        atomicAdd(&hist_cur[cv], 1u);
                    }
                }
            };
            auto end_epoch_r = [&]() {
                q_lds_barrier();
                const uint32_t cnt = s_ccount + s_dirty;
                if (cnt != last_cnt && tid < 128) {
                    hist_done[tid] += hist_cur[tid];
                    hist_cur[tid] = 0;
                }
                q_lds_barrier();
                if (cnt != last_cnt) {
                    bound = q_uni(q_bound_from_hist(hist_done, R, lane));
                    last_cnt = cnt;
                }
            };
            if (res_lo < res_hi) {                               // this workgroup's chunk, as far as it lies in registers: one epoch
#pragma unroll
                for (int j = 0; j < (int)(kResident / kQWG); ++j) {
                    if (res_lo + j * kQWG < res_hi) {            // (workgroup-uniform)
                        const uint32_t vec = res_lo + j * kQWG + tid;
                        if (vec < res_hi) proc(cf[j], vec);
                    }
                }
                end_epoch_r();
            }
            STAMP(10);
            ramp = max(ramp, 2u * kFirstBlock);
            lo1 = max(my_lo, (uint64_t)res_hi);
        }
    }
    // pass 0: the first block [0, B) (bound only unless this is workgroup 0); pass 1: this workgroup's chunk.
    // (one loop body for both: inlining the walk twice doubles the register pressure of the hot loop)
    for (int pass = (MULTI && B && !(RES && resident)) ? 0 : ((RES && resident && lo1 >= my_hi) ? 2 : 1); pass < 2; ++pass) {
        const uint64_t lo = pass == 0 ? 0 : lo1, hi = pass == 0 ? B : my_hi;
        const bool do_emit = pass == 1 || g == 0;
        uint64_t pbase = 0, cbase = 0;                           // vectors / codes of the scan order before partition a
        int a = HEAD ? 0 : next_part(-1);                        // (HEAD walks every slot: cbase counts empty-here partitions too)
        if (MULTI && !HEAD && V < 0xffffffffull) {
            // the partition `lo` lies in, found by the lanes together (lanes = assign slots, 64 per step: two loads deep).  The
            // loop below takes one slot per iteration, two DEPENDENT loads each: workgroup g of a lone query over 32 probes
            // spent up to ~1.4 us per skipped slot there — the launch lasted 74 us where its first workgroups needed 31
            // (rocprofv3 against the phase stamps of workgroups 0 and 1, round 6)
            uint32_t run = 0;
            a = mas;
            for (int a0 = 0; a0 < mas; a0 += 64) {
                const int a_ = a0 + (int)lane;
                const uint32_t nv_ = a_ < mas ? (parts[assign[a_]].n + CPL - 1) / CPL : 0u;
                const uint32_t incl = q_wave_incl_sum(nv_);
                const uint32_t before = run + incl - nv_;
                const uint64_t hit = __builtin_amdgcn_ballot_w64(nv_ != 0 && (uint64_t)before + nv_ > lo);
                if (hit) {
                    const int l_ = (int)__builtin_ctzll(hit);
                    a = a0 + l_;
                    pbase = (uint32_t)__builtin_amdgcn_readlane((int)before, l_);
                    break;
                }
                run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                pbase = run;
            }
        }
        while (MULTI && a < mas) {                                // skip the partitions that end before lo
            const uint32_t n_ = q_uni(parts[assign[a]].n);
            const uint32_t nv = (eff_n(n_, cbase) + CPL - 1) / CPL;
            if (pbase + nv > lo) break;
            pbase += nv;
            cbase += n_;
            a = HEAD ? a + 1 : next_part(a);
        }
        while (a < mas && (!MULTI || pbase < hi)) {
            UDesc d = q_load_desc(parts, (int)q_uni((uint32_t)assign[a]));
            const uint32_t n_full = d.n;
            if (HEAD) {
                d.n = eff_n(n_full, cbase);
                if (d.n == 0) {                                  // nothing of this partition lies inside the head
                    cbase += n_full;
                    ++a;
                    continue;
                }
            }
            const uint32_t n = d.n;
            const uint32_t nvec = (n + CPL - 1) / CPL;
            const uint32_t t_begin = MULTI ? (uint32_t)(max(lo, pbase) - pbase) : 0u;
            const uint32_t t_end = MULTI ? (uint32_t)(min(hi, pbase + nvec) - pbase) : nvec;
            if (tables_of != a) {                                // (the partition switch inside end_epoch covers the common case)
                q_lds_barrier();
                if (tid < M * 4) reinterpret_cast<uint32_t*>(tq)[tid] = table_word(a);
                q_lds_barrier();
                write_tables();
                q_lds_barrier();
                tables_of = a;
            }
            const int a_next = HEAD ? a + 1 : next_part(a);
            const bool more = !HEAD && a_next < mas && (!MULTI || pbase + nvec < hi);   // the walk continues in the next probed partition
            const uint32_t tqv = more ? table_word(a_next) : 0u; // in flight during this partition's epochs
            const gvec_t src = (gvec_t)(uintptr_t)d.codes;
            const uint32_t dup_pos = (d.first_pos + n_full == d.global_n && n == n_full) ? n_full - 1u : 0xffffffffu;
            const uint32_t dup_reps = (16u - d.global_n % 16u) % 16u;
            const uint32_t key_base = d.key_base + d.first_pos;
            auto emit = [&](uint32_t cv, uint32_t p) {           // rare: count (and append, unordered) one candidate
                if (HEAD && do_emit && G == 1) {
                    // this workgroup is the only writer of the query's region: slots, replays and the histogram are
                    // counted in LDS and published once, at the end (a global atomic per candidate — its return value
                    // stalls the wave for a memory round trip — is what the shared-region form below pays)
                    const uint32_t reps = p == dup_pos ? dup_reps : 0u;
                    const uint32_t slot = atomicAdd(&s_ccount, 1u);
                    if (slot < A.cand_cap) {
                        Cand c_;
                        c_.order = (uint32_t)a | (reps << 20);   // level 0, assign slot a
                        c_.pos = p;
                        c_.key = d.labels ? d.labels[p] : key_base + p;
                        c_.val = cv;
                        A.cand_regions[(size_t)q * A.cand_cap + slot] = c_;
                    }
                    if (reps) atomicAdd(&s_hreps, reps);
                } else if (HEAD && do_emit) {                    // into the level path's structures (emit_candidate's twin)
                    QueryState* qs = A.qstates + q;
                    const uint32_t reps = p == dup_pos ? dup_reps : 0u;
                    const uint32_t slot = atomicAdd(&qs->count, 1u);
                    if (slot < A.cand_cap) {
                        Cand c_;
                        c_.order = (uint32_t)a | (reps << 20);   // level 0, assign slot a
                        c_.pos = p;
                        c_.key = d.labels ? d.labels[p] : key_base + p;
                        c_.val = cv;
                        A.cand_regions[(size_t)q * A.cand_cap + slot] = c_;
                    } else {
                        atomicAdd(&A.hdr->overflow, 1u);
                    }
                    if (reps) atomicAdd(&qs->reps, reps);
                    atomicAdd(&qs->hist[cv], 1u);
                    atomicAdd(&s_dirty, 1u);
                } else if (do_emit) {
                    const uint32_t slot = atomicAdd(&s_ccount, 1u);
                    if (slot < A.ccap) {
                        QCand qc;
                        qc.key = d.labels ? d.labels[p] : key_base + p;
                        qc.val_reps = cv | ((p == dup_pos ? dup_reps : 0u) << 8);
                        qc.pos = p;
                        qc.slot = (uint32_t)a;
                        put_cand(slot, qc);
                    }
                } else {
                    atomicAdd(&s_dirty, 1u);                     // (histogram-only walk: still makes the epoch "dirty")
                }
                atomicAdd(&hist_cur[cv], 1u);
            };
            // FULL = every lane of every tile of the iteration holds CPL complete codes: no predicates, so all the
            // lookups of the iteration sit in one basic block and overlap (a predicated copy handles ragged ends)
            auto load_tiles = [&](u32x4 (&v)[kRounds], uint32_t t0, uint32_t width, uint32_t tl, auto full) {
#pragma unroll
                for (int i = 0; i < kRounds; ++i) {
                    const uint32_t off = (tl + (uint32_t)i * kQWaves) * 64u + lane;
                    if (decltype(full)::value) {
                        v[i] = NT ? __builtin_nontemporal_load(src + t0 + off) : src[t0 + off];
                    } else {
                        v[i] = u32x4{0, 0, 0, 0};
                        if (off < width) v[i] = NT ? __builtin_nontemporal_load(src + t0 + off) : src[t0 + off];
                    }
                }
            };
            auto sums_of = [&](uint32_t (&cand)[kRounds * CPL], const u32x4 (&v)[kRounds], uint32_t t0, uint32_t width,
                               uint32_t tl, auto full) {
                uint32_t best = 127u;
#pragma unroll
                for (int i = 0; i < kRounds; ++i) {
                    const uint32_t dd[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                    const uint32_t off = (tl + (uint32_t)i * kQWaves) * 64u + lane;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const uint32_t sum = q_pair_sum<M>(dd + c * DW, lane_lo, lane_hi);
                        uint32_t cv = min(sum, 127u);
                        if (!decltype(full)::value) {
                            const bool live = off < width && (t0 + off) * CPL + c < n;
                            cv = live ? cv : 127u;
                        }
                        cand[i * CPL + c] = cv;
                        best = min(best, cv);
                    }
                }
                return best;
            };
            auto end_epoch = [&](bool switch_part) {
                if (switch_part && tid < M * 4) reinterpret_cast<uint32_t*>(tq)[tid] = tqv;
                q_lds_barrier();                                 // every candidate of the epoch is counted; the tables are idle
                const uint32_t cnt = s_ccount + s_dirty;
                if (cnt != last_cnt && tid < 128) {
                    hist_done[tid] += hist_cur[tid];
                    hist_cur[tid] = 0;
                }
                if (switch_part) write_tables();
                q_lds_barrier();
                if (cnt != last_cnt) {
                    bound = q_uni(q_bound_from_hist(hist_done, R, lane));
                    last_cnt = cnt;
                }
            };
            uint32_t t0 = t_begin;
            while (t0 < t_end) {
                const uint32_t rest = t_end - t0;
                uint32_t width;
                if (ramp < kEpochVec) {
                    width = min(ramp, rest);
                    ramp <<= 1u;
                } else {
                    width = min(rest, kEpochVec);
                }
                const uint32_t tiles = (width + 63u) / 64u;      // 64-vector tiles of the epoch; wave w takes w, w+16, ...
                // (tried and dropped: loads of the next tiles in flight across iterations, and a register-resident first
                // block for the ramp — both cost registers the lookups need; the walk is bound by issue, not by memory)
                // tiles [0, full_tiles) hold only complete vectors of complete codes
                const uint32_t full_tiles = min(width, n / CPL - min(n / CPL, t0)) / 64u;
                auto step = [&](uint32_t tl, auto full) {
                    u32x4 v[kRounds];
                    load_tiles(v, t0, width, tl, full);
                    uint32_t cand[kRounds * CPL];
                    const uint32_t best = sums_of(cand, v, t0, width, tl, full);
                    if (__builtin_expect(best < bound, 0)) {     // rare
#pragma unroll
                        for (int i = 0; i < kRounds; ++i)
#pragma unroll
                            for (int c = 0; c < CPL; ++c)
                                if (cand[i * CPL + c] < bound)
                                    emit(cand[i * CPL + c], (t0 + (tl + (uint32_t)i * kQWaves) * 64u + lane) * CPL + c);
                    }
                };
                uint32_t tl = wave;
                {
                    // software pipeline over the full tiles: the NEXT iteration's loads are issued before the current tiles are
                    // summed, so a wave has loads in flight while it works the LDS pipe — without it the 16 (x 2 workgroups) waves of
                    // a CU, which issued their loads together, get them back together, and the memory system idles while they all
                    // look up (the head walk measured 3.6 TB/s of reads: about half of the time nothing was in flight)
                    auto consume = [&](const u32x4 (&v)[kRounds], uint32_t tl_) {
                        uint32_t cand[kRounds * CPL];
                        const uint32_t best = sums_of(cand, v, t0, width, tl_, std::integral_constant<bool, true>());
                        if (__builtin_expect(best < bound, 0)) {
#pragma unroll
                            for (int i = 0; i < kRounds; ++i)
#pragma unroll
                                for (int c = 0; c < CPL; ++c)
                                    if (cand[i * CPL + c] < bound)
                                        emit(cand[i * CPL + c], (t0 + (tl_ + (uint32_t)i * kQWaves) * 64u + lane) * CPL + c);
                        }
                    };
                    if (tl + (kRounds - 1) * kQWaves < full_tiles) {
                        u32x4 v[kRounds];
                        load_tiles(v, t0, width, tl, std::integral_constant<bool, true>());
                        for (;;) {
                            const uint32_t tn = tl + kQWaves * kRounds;
                            const bool more_full = tn + (kRounds - 1) * kQWaves < full_tiles;
                            u32x4 nv[kRounds];
                            if (more_full) load_tiles(nv, t0, width, tn, std::integral_constant<bool, true>());
                            consume(v, tl);
                            tl = tn;
                            if (!more_full) break;
#pragma unroll
                            for (int i = 0; i < kRounds; ++i) v[i] = nv[i];
                        }
                    }
                }
                for (; tl < tiles; tl += kQWaves * kRounds) step(tl, std::integral_constant<bool, false>());
                t0 += width;
                const bool sw = t0 >= t_end && more;
                end_epoch(sw);
                if (sw) tables_of = a_next;
            }
            if (MULTI) pbase += nvec;
            if (HEAD) cbase += n_full;
            a = a_next;
        }
    }
    if (HEAD) {                                                  // the level path (or order_cands_kernel) orders and reports
        if (G == 1) {                                            // counts kept in LDS (see emit): published here
            QueryState* qs = A.qstates + q;
            q_lds_barrier();
            if (tid < 128) qs->hist[tid] = hist_done[tid] + hist_cur[tid];
            if (tid == 0) {
                qs->count = s_ccount;
                qs->reps = s_hreps;
                if (s_ccount > A.cand_cap) atomicAdd(&A.hdr->overflow, s_ccount - A.cand_cap);
            }
        }
        if ((A.ftables || A.front_in) && tid == 0 && g == 0) {   // front run here (or gathered): its results travel in the QueryState
            QueryState* qs = A.qstates + q;
            qs->qmin = qmin;
            qs->qmax = qmax;
            qs->flags = flags & 3u;
#ifdef QADC_STAMPS
            if (blockIdx.x == 1) {
                const uint64_t clk2s = __builtin_readcyclecounter();
                printf("HEAD STAMPS: prologue %llu starts-known %llu prescan %llu qmin %llu select %llu quantize %llu walk %llu\n",
                       (unsigned long long)(stamps[1] - clk0), (unsigned long long)(stamps[2] - stamps[1]), (unsigned long long)(stamps[3] - stamps[2]),
                       (unsigned long long)(stamps[4] - stamps[3]), (unsigned long long)(stamps[5] - stamps[4]), (unsigned long long)(clk1 - stamps[5]),
                       (unsigned long long)(clk2s - clk1));
            }
#endif
            if (A.head_slots) {                                  // the IVF head: phase clocks for the profile (order_cands_kernel forwards
                const uint64_t clk2 = __builtin_readcyclecounter();   // them; the select fields are idle on this path)
                qs->sel_prefix = (uint32_t)min((clk1 - clk0) >> 6, (uint64_t)0xffff);
                qs->sel_k = (uint32_t)((clk2 - clk1) >> 4);
            }
        }
        return;
    }
    // ---- 4. order the candidates: (assign slot, position) ascending = scan order ----
    const uint64_t clk2 = __builtin_readcyclecounter();
    __syncthreads();                                             // the candidate stores of every wave are complete
    const uint32_t ncand = s_ccount;
    uint32_t out_count = 0;
    if (ncand > A.ccap) {
        flags |= 32u;                                            // more candidates than the in-workgroup sort takes: host falls back
    } else if (ncand) {
        // (bucket-sort scratch: the value histograms — misc[0..255] — are dead by now)
        // (behind the first block's entries, which workgroup 0 wrote in order as it found them)
        out_count = q_order_and_write<12, true, WG>([&](uint32_t i) { return (RES && i < kLdsCands) ? lcands[i] : cands[i]; }, ncand, stream + min(nfb, A.cap),
                                                A.cap - min(nfb, A.cap), wcnt, tid, lane, wave,
                                                misc, 127u, (uint32_t)ma, A.pos_bits >> 16, A.pos_bits & 0xffffu);
    }
    out_count += nfb;
    STAMP(14);
    s_count = out_count;
    if constexpr (RES) __syncthreads();                          // every lane's stream entries happen-before lane 0's release in publish()
    if (tid == 0) {
        QueryOut o;
        o.count = s_count;                                       // entries requested (replays included); > cap = overflow
        o.reps = 0;
        o.flags = flags | 4u;
        o.out_off = (uint32_t)((size_t)wgi * A.cap);
        o.qmin = qmin;
        o.qmax = qmax;
        const uint64_t clk3 = __builtin_readcyclecounter();
#ifdef QADC_STAMPS
        if (blockIdx.x <= 1 || blockIdx.x + 1 == gridDim.x) {   // (workgroup 0 emits the first block; the last one skips the most partitions)
            stamps[6] = clk1; stamps[11] = clk2; stamps[15] = clk3;
            for (int i_ = 12; i_ < 14; ++i_) stamps[i_] = stamps[11];
            printf("STAMPS wg=%u ncand=%u out=%u:", (unsigned)blockIdx.x, ncand, out_count);
            for (int i_ = 1; i_ < 16; ++i_) printf(" %d:%llu", i_, (unsigned long long)(stamps[i_] - stamps[i_ - 1]));
            printf("\n");
        }
#endif
        // phase clocks: pad[0] = (pre-scan + select + quantizer) >> 6 | (sort + ordered write) >> 6 << 16; pad[1] = scan >> 4
        o.pad[0] = (uint32_t)min((clk1 - clk0) >> 6, (uint64_t)0xffff) | ((uint32_t)min((clk3 - clk2) >> 6, (uint64_t)0xffff) << 16);
        o.pad[1] = (uint32_t)((clk2 - clk1) >> 4);
        publish(o);
        if (A.qstate_flags) {                                    // what replay_heap_wave_kernel reads
            A.qstate_flags[4 * wgi + 0] = flags | 4u;
            A.qstate_flags[4 * wgi + 1] = s_count;
        }
    }
}

template <int M, int U, int OCC, bool NT, bool MULTI, bool HEAD, int WG = 1024>
__global__ __launch_bounds__(WG, OCC) void scan_query_kernel(QueryKernelArgs A) {
    scan_query_body<M, U, OCC, NT, MULTI, HEAD, WG>(A);
}

// Small batches (G > 1): the same kernel with room for inline input behind its arguments (QueryKernelArgs::inline_input).
template <int M, int U, int OCC, bool NT>
__global__ __launch_bounds__(kQWG, OCC) void scan_query_inline_kernel(QueryKernelInline IA) {
    QueryKernelArgs A = IA.a;
    if (A.inline_input) {
        // (read through the kernel-argument segment pointer: indexing IA.payload would copy the struct to scratch)
        const unsigned char* ka = (const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(QueryKernelInline, payload);
        A.assign = reinterpret_cast<const int32_t*>(ka);
        A.parts = reinterpret_cast<const PartDesc*>(ka + A.inline_off_parts);
        A.ftables = A.inline_off_tables ? reinterpret_cast<float*>(const_cast<unsigned char*>(ka) + A.inline_off_tables)
                                        : nullptr;               // (no tables in the payload: the front ran in lone_front_kernel)
    }
    scan_query_body<M, U, OCC, NT, true, false>(A);
}

// ---------------------------------------------------------------------------------------------
// The sliced front of a lone query on a long flat list (qadc_kernels.h: LoneFrontArgs).
// The k-th smallest (k >= 1) of the n >= k u32 keys in LDS, exactly, by the 1024 threads of the workgroup: histogram passes over a
// digit fitted to the range the answer still lies in ((key - lo) >> sh, 256 buckets) until the bucket holding rank k has at most 256
// keys, which are then ranked by counting (a 63 us first version sorted the keys: ~130 barrier steps of a bitonic network).
// w: 256 + 16 words of LDS scratch.
__device__ __forceinline__ uint32_t lf_select(const uint32_t* keys, uint32_t n, uint32_t k, uint32_t* w, uint32_t tid) {
    uint32_t* hist = w;                                          // [256]
    uint32_t& s_lo = w[256];
    uint32_t& s_hi = w[257];
    uint32_t& s_k = w[258];
    uint32_t& s_cnt = w[259];
    uint32_t& s_sel = w[260];
    const uint32_t lane = tid & 63u;
    if (tid == 0) { s_lo = 0xffffffffu; s_hi = 0; s_k = k; }
    q_lds_barrier();
    {
        uint32_t mn = 0xffffffffu, mx = 0;
        for (uint32_t i = tid; i < n; i += kQWG) { mn = min(mn, keys[i]); mx = max(mx, keys[i]); }
        mn = q_wave_scan_bits(mn, 0xffffffffu, [](uint32_t a, uint32_t b) { return min(a, b); });
        mx = q_wave_scan_bits(mx, 0u, [](uint32_t a, uint32_t b) { return max(a, b); });
        if (lane == 63) { atomicMin(&s_lo, mn); atomicMax(&s_hi, mx); }
    }
    for (;;) {
        if (tid < 256) hist[tid] = 0;
        q_lds_barrier();
        const uint32_t lo = s_lo, hi = s_hi, kk = s_k;
        if (lo == hi) return lo;                                 // (every key left is the same)
        const uint32_t range = hi - lo;
        const uint32_t sh = range >= 256u ? 24u - (uint32_t)__builtin_clz(range) : 0u;
        for (uint32_t i = tid; i < n; i += kQWG) {
            const uint32_t key = keys[i];
            if (key >= lo && key <= hi) atomicAdd(&hist[(key - lo) >> sh], 1u);
        }
        q_lds_barrier();
        if (tid < 64) {                                          // wave 0: 4 buckets per lane, the one holding rank kk
            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const uint32_t sum4 = c0 + c1 + c2 + c3;
            const uint32_t incl = q_wave_incl_sum(sum4);
            const uint32_t excl = incl - sum4;
            if (incl >= kk && excl < kk) {
                uint32_t run = excl, digit = 4 * tid;
                const uint32_t cs4[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (run + cs4[j] >= kk) { digit = 4 * tid + j; break; }
                    run += cs4[j];
                }
                const uint32_t top = (digit << sh) + ((1u << sh) - 1u);   // (offset of the bucket's last key; compared before it is added: no wrap)
                s_lo = lo + (digit << sh);
                s_hi = top >= range ? hi : lo + top;
                s_k = kk - run;
                s_cnt = hist[digit];
            }
            if (tid == 0) s_sel = 0;
        }
        q_lds_barrier();
        if (s_cnt <= 256u) break;
    }
    const uint32_t lo = s_lo, hi = s_hi, kk = s_k;
    q_lds_barrier();
    for (uint32_t i = tid; i < n; i += kQWG) {
        const uint32_t key = keys[i];
        if (key >= lo && key <= hi) hist[atomicAdd(&s_sel, 1u)] = key;
    }
    q_lds_barrier();
    const uint32_t c = s_sel;
    if (tid < c) {
        const uint32_t key = hist[tid];
        const uint4* h4 = reinterpret_cast<const uint4*>(hist);
        uint32_t less = 0, le = 0, j = 0;
        for (; j + 4 <= c; j += 4) {                             // (four keys per LDS read)
            const uint4 kq = h4[j >> 2];
            less += (kq.x < key ? 1u : 0u) + (kq.y < key ? 1u : 0u) + (kq.z < key ? 1u : 0u) + (kq.w < key ? 1u : 0u);
            le += (kq.x <= key ? 1u : 0u) + (kq.y <= key ? 1u : 0u) + (kq.z <= key ? 1u : 0u) + (kq.w <= key ? 1u : 0u);
        }
        for (; j < c; ++j) {
            const uint32_t kj = hist[j];
            less += kj < key ? 1u : 0u;
            le += kj <= key ? 1u : 0u;
        }
        if (less < kk && kk <= le) s_lo = key;
    }
    q_lds_barrier();
    return s_lo;
}

template <int M>
__global__ __launch_bounds__(kQWG, 2) void lone_front_kernel(LoneFrontArgs A) {
    constexpr int CS = M / 2, DW = M / 8;
    float* tab = reinterpret_cast<float*>(qsmem);                       // [M*16] at LDS address 0 (q_prescan_sum wants it absolute)
    uint32_t* keys = reinterpret_cast<uint32_t*>(qsmem + M * 16 * 4);   // [kLoneFrontMaxKeys]
    uint32_t& s_last = keys[kLoneFrontMaxKeys];                         // (no static LDS in this file: the tables start at address 0)
    float* s_red = reinterpret_cast<float*>(keys + kLoneFrontMaxKeys + 16);   // [16]
    uint32_t* scr = keys + kLoneFrontMaxKeys + 32;                      // [272] lf_select's scratch
    if (reinterpret_cast<uintptr_t>((q_lds_bytes_t)qsmem) != 0) __builtin_trap();
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, s = blockIdx.x;
    const float* kt = reinterpret_cast<const float*>((const unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(LoneFrontArgs, table));
    if (tid < M * 16) tab[tid] = kt[tid];
    const PartDesc d = *A.part;
    const uint32_t sn = q_uni(d.global_n) ? q_uni(d.start_n) : 0u;
    const uint8_t* sc = reinterpret_cast<const uint8_t*>(q_uni64(reinterpret_cast<uint64_t>(d.starts ? d.starts : d.codes)));
    const uint32_t lo = min(sn, s * (uint32_t)kLoneFrontSlice), hi = min(sn, lo + (uint32_t)kLoneFrontSlice);
    for (uint32_t i = tid; i < (uint32_t)kLoneFrontSlice; i += kQWG) keys[i] = 0xffffffffu;
    q_lds_barrier();
    {   // the slice's float ADC: kLoneFrontSlice / 1024 = 4 codes per lane, all loads first
        constexpr int kPer = kLoneFrontSlice / kQWG;
        uint32_t dw[kPer][DW];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t i = lo + tid + (uint32_t)u * kQWG;
#pragma unroll
            for (int w = 0; w < DW; ++w) dw[u][w] = 0;
            if (i < hi) {
                if constexpr (M == 16) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = ((const __attribute__((address_space(1))) u32x2*)(uintptr_t)sc)[i];
                    dw[u][0] = v.x; dw[u][1] = v.y;
                } else {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = ((const __attribute__((address_space(1))) u32x4*)(uintptr_t)sc)[i];
                    dw[u][0] = v.x; dw[u][1] = v.y; dw[u][2] = v.z; dw[u][3] = v.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t i = lo + tid + (uint32_t)u * kQWG;
            const float cand = q_prescan_sum<M>(dw[u], 0u, A.sum_mode);
            if (i < hi) keys[i - lo] = q_fkey(cand);
        }
        (void)CS;
    }
    q_lds_barrier();
    // the slice's R smallest keys (any R of them under ties; padded with ~0 when the slice has fewer values): everything below the
    // slice's R-th smallest, then copies of it
    // (device-scope stores, and device-scope loads in the last workgroup: the slices' key ranges share cache lines at their seams,
    // their workgroups sit on different XCDs, and with plain accesses behind the fences the last workgroup read stale words about once
    // in 6000 queries — a NaN qmax in the soak of tools/soak_lone_long.py)
    uint32_t* out = A.state + 4 + (size_t)s * A.R;
    auto put = [&](uint32_t i, uint32_t v) { __hip_atomic_store(&out[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const uint32_t nsl = hi - lo;
    if (nsl <= A.R) {
        for (uint32_t i = tid; i < A.R; i += kQWG) put(i, i < nsl ? keys[i] : 0xffffffffu);
    } else {
        const uint32_t T = lf_select(keys, nsl, A.R, scr, tid);
        uint32_t& s_below = scr[261];
        if (tid == 0) s_below = 0;
        q_lds_barrier();
        for (uint32_t i = tid; i < nsl; i += kQWG)
            if (keys[i] < T) put(atomicAdd(&s_below, 1u), keys[i]);
        q_lds_barrier();
        for (uint32_t i = s_below + tid; i < A.R; i += kQWG) put(i, T);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_last = atomicAdd(A.state, 1u) == A.S - 1u ? 1u : 0u;
        if (s_last) __threadfence();
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the last workgroup: R-th smallest of the union, qmin / clamp / QuantizerMAX ----
    const uint32_t nk = A.S * A.R;
    for (uint32_t i = tid; i < nk; i += kQWG) keys[i] = __hip_atomic_load(&A.state[4 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    q_lds_barrier();
    float qmax = FLT_MAX;                                        // (fewer starts than R: the reference's exit path)
    if (sn >= A.R) qmax = q_funkey(lf_select(keys, nk, A.R, scr, tid));
    float lmin = FLT_MAX;
    for (uint32_t i = tid; i < (uint32_t)(M * 16); i += kQWG) lmin = fminf(lmin, tab[i]);
    lmin = q_wave_min(lmin);
    if (lane == 0) s_red[wave] = lmin;
    __syncthreads();
    float qmin = s_red[0];
#pragma unroll
    for (int w = 1; w < kQWaves; ++w) qmin = fminf(qmin, s_red[w]);
    uint32_t flags = 0;
    if (qmin < 0) { qmin = 0; flags |= 2u; }
    if ((double)qmax > 1e30) flags |= 1u;
    const float delta = (qmax - qmin) / 127;
    const float scale = 127.0f / (qmax - qmin);
    for (uint32_t i = tid; i < (uint32_t)(M * 16); i += kQWG) {
        float v = tab[i];
        if (v < 0) v = 0;
        int8_t o;
        if (flags & 1u) o = 127;
        else if (v >= qmax) o = 127;
        else o = (int8_t)(int)(A.quant_mode == 0 ? (v - qmin) / delta : (v - qmin) * scale);
        A.qtables[i] = o;
    }
    if (tid == 0) {
        A.front_out[0] = flags;
        A.front_out[1] = __float_as_uint(qmin);
        A.front_out[2] = __float_as_uint(qmax);
        A.front_out[3] = 0;
        __hip_atomic_store(A.state, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// Large IVF batches, partition-major second phase.  A batch of 1024 queries x 32 probes lands ~8 times on each of
// 4096 partitions: walking query by query reads (and looks up) every partition 8 times.  After the head launch
// (scan_query_kernel in HEAD mode: front + the first head_slots probes of every query, which fixes a tight bound per
// query) the remaining (query, probe) pairs are regrouped BY PARTITION on the device — count, offsets, scatter: three
// small kernels, no host round trip — into groups of up to 8 pairs that share one pass of scan_i8_mq_kernel over the
// partition (one LDS lookup serves 8 queries).  Every pair is bound level 1: its bound derives from the head's
// candidates only, i.e. from codes that precede it in its query's scan order, whatever order the groups run in.
// order_cands_kernel then restores scan order per query and hands the streams to replay_heap_wave_kernel.
// ---------------------------------------------------------------------------------------------
// One wave per query, lanes = assign slots.  A (query, probe) pair belongs to the second phase iff its partition has codes
// here and at least `h` such probes precede it in the query's assign[] (those are the head's: scan_query_kernel, HEAD).
template <typename F>
__device__ __forceinline__ void ivf_for_grouped_pairs(const int32_t* __restrict__ assign_q, const PartDesc* __restrict__ parts,
                                                      int ma, int h, uint32_t lane, F f) {
    int seen = 0;
    for (int a0 = 0; a0 < ma; a0 += 64) {
        const int a = a0 + (int)lane;
        int p = 0;
        bool ne = false;
        if (a < ma) {
            p = assign_q[a];
            ne = parts[p].n != 0;
        }
        const uint64_t m = __builtin_amdgcn_ballot_w64(ne);
        const int before = seen + __popcll(m & ((1ull << lane) - 1ull));
        if (ne && before >= h) f(a, p);
        seen += __popcll(m);
    }
}

__global__ __launch_bounds__(256) void ivf_count_kernel(const int32_t* __restrict__ assign, const PartDesc* __restrict__ parts,
                                                        int nq, int ma, int s0, uint32_t* __restrict__ cnt) {
    const int q = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (q >= nq) return;
    ivf_for_grouped_pairs(assign + (size_t)q * ma, parts, ma, s0, threadIdx.x & 63u, [&](int, int p) { atomicAdd(&cnt[p], 1u); });
}

// goff[p] = first group of partition p, goff[K] = groups in all; one workgroup
// (256 threads, like every helper launch that runs beside the scans: see kSideWG)
__global__ __launch_bounds__(kSideWG) void ivf_offsets_kernel(const uint32_t* __restrict__ cnt, int K, uint32_t* __restrict__ goff) {
    __shared__ uint32_t wsum[kSideWG / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int per = ((K + kSideWG - 1) / kSideWG + 3) & ~3;      // a multiple of 4: the runs start 16-byte aligned if the arrays do
    const bool vec = ((reinterpret_cast<uintptr_t>(cnt) | reinterpret_cast<uintptr_t>(goff)) & 15u) == 0;   // (goff = cnt + 2K: K even)
    const int lo = min(K, (int)tid * per), hi = min(K, lo + per);
    uint32_t mine = 0;
    {
        // (four counters per load, all of a thread's loads in flight: one dependent load per counter made this one-workgroup
        //  launch — K / 256 = 64 counters per thread at the C5 shape — a chain of 64 round trips on the plan's critical path)
        int p = lo;
        for (; vec && p + 16 <= hi; p += 16) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(cnt + p + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) mine += (v[u].x + 7u) / 8u + (v[u].y + 7u) / 8u + (v[u].z + 7u) / 8u + (v[u].w + 7u) / 8u;
        }
        for (; p < hi; ++p) mine += (cnt[p] + 7u) / 8u;
    }
    const uint32_t incl = q_wave_incl_sum(mine);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - mine, total = 0;
    for (uint32_t w = 0; w < kSideWG / 64; ++w) {
        if (w < wave) base += wsum[w];
        total += wsum[w];
    }
    {
        int p = lo;
        for (; vec && p + 16 <= hi; p += 16) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(cnt + p + 4 * u);   // (L2 hits: read a moment ago)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint4 o;
                o.x = base; base += (v[u].x + 7u) / 8u;
                o.y = base; base += (v[u].y + 7u) / 8u;
                o.z = base; base += (v[u].z + 7u) / 8u;
                o.w = base; base += (v[u].w + 7u) / 8u;
                *reinterpret_cast<uint4*>(goff + p + 4 * u) = o;
            }
        }
        for (; p < hi; ++p) {
            goff[p] = base;
            base += (cnt[p] + 7u) / 8u;
        }
    }
    if (tid == 0) goff[K] = total;
}

// items must arrive zeroed: an item with n == 0 is an empty seat of its group (scan_i8_mq_kernel), a group whose first
// seat is empty does not exist.  The seat order inside a partition is whatever the atomics give — the candidates are
// ordered afterwards, so the result does not depend on it.
__global__ __launch_bounds__(256) void ivf_scatter_kernel(const int32_t* __restrict__ assign, const PartDesc* __restrict__ parts,
                                                          int nq, int ma, int s0, const uint32_t* __restrict__ goff,
                                                          uint32_t* __restrict__ fill, ScanItem* __restrict__ items) {
    const int q = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (q >= nq) return;
    ivf_for_grouped_pairs(assign + (size_t)q * ma, parts, ma, s0, threadIdx.x & 63u, [&](int a, int p) {
        const PartDesc d = parts[p];
        const uint32_t seat = atomicAdd(&fill[p], 1u);
        ScanItem it;
        it.codes = d.codes;
        it.labels = d.labels;
        it.n = d.n;
        it.pos0 = 0;
        it.key_base = d.key_base + d.first_pos;
        it.table = (uint32_t)(q * ma + a);
        it.query = (uint32_t)q;
        it.order = (1u << 16) | (uint32_t)a;                     // bound level 1, assign slot a
        it.dup_pos = d.first_pos + d.n == d.global_n ? d.n - 1u : 0xffffffffu;
        it.dup_reps = (16u - d.global_n % 16u) % 16u;
        items[(size_t)(goff[p] + seat / 8u) * 8u + seat % 8u] = it;
    });
}

// One workgroup per query: the query's unordered Cand records (head + grouped pairs) -> ordered push stream in the
// layout scan_query_kernel writes ([nq][cap] entries, QueryOut, {flags, entries} for the device replay).
__global__ __launch_bounds__(kQWG) void order_cands_kernel(const QueryState* __restrict__ qstates, const Cand* __restrict__ regions,
                                                           uint32_t cand_cap, uint32_t ccap, uint64_t* __restrict__ stream,
                                                           uint32_t cap, QueryOut* __restrict__ qout, uint32_t* __restrict__ qflags,
                                                           uint32_t ma, uint32_t pos_bits) {
    __shared__ uint32_t wcnt[kQWaves];
    __shared__ uint32_t bscr[2 * kOrderBuckets + 1];                 // bucket-sort scratch (q_order_and_write)
    const int q = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const QueryState* qs = qstates + q;
    const uint32_t n = qs->count;
    uint32_t flags = qs->flags & 3u, out_count = 0;
    const Cand* __restrict__ region = regions + (size_t)q * cand_cap;
    if (n > min(cand_cap, ccap)) {
        flags |= 32u;                                            // more than the in-workgroup sort takes (or the region held): host falls back
    } else if (n) {
        out_count = q_order_and_write<13, false>(
            [&](uint32_t i) {
                const Cand c = region[i];
                QCand o;
                o.key = c.key;
                o.val_reps = (c.val & 0xffu) | (((c.order >> 20) & 15u) << 8);
                o.pos = c.pos;
                o.slot = c.order & 0x3fffu;
                return o;
            },
            n, stream + (size_t)q * cap, cap, wcnt, tid, lane, wave, bscr, kOrderBuckets, ma, pos_bits >> 16, pos_bits & 0xffffu);
    }
    if (tid == 0) {
        QueryOut o;
        o.count = out_count;
        o.reps = 0;
        o.flags = flags | 4u;
        o.out_off = (uint32_t)((size_t)q * cap);
        o.qmin = qs->qmin;
        o.qmax = qs->qmax;
        o.pad[0] = qs->sel_prefix & 0xffffu;                     // the head's phase clocks (front >> 6, walk >> 4): scan_query_kernel, HEAD
        o.pad[1] = qs->sel_k;
        qout[q] = o;
        if (qflags) {
            qflags[4 * q + 0] = flags | 4u;
            qflags[4 * q + 1] = out_count;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU: the code list is range-sharded over the ranks; every rank's ordered push stream is a
// superset of the pushes the sequential scan's heap accepts from that range, so replaying the streams in GLOBAL scan
// order (assign slot, rank, position) reproduces the reference heap array for array.  Two kernels around ONE
// ncclAllGather (host side: qadc_dist_collect):
//   dist_pack_kernel        this rank's streams -> one contiguous block  [nq x {offset, count, flags, -}][entries][extra]
//   dist_totals_kernel      per-query totals over the gathered blocks, flags, 64-bit prefix, a status word for the host
//   dist_interleave_kernel  the world's streams of a query -> ONE stream in global scan order
//   replay_heap_wave_kernel that stream -> the heap, one wave per query (every rank computes every heap: there is no
//                           second collective)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dist_pack_kernel(const uint32_t* __restrict__ src_off, const uint32_t* __restrict__ src_cnt,
                                                        const uint32_t* __restrict__ src_flags, int nq,
                                                        const uint64_t* __restrict__ stream,
                                                        const uint64_t* __restrict__ fix, uint32_t cap_entries,
                                                        const float* __restrict__ extra, uint32_t extra_n,
                                                        uint64_t* __restrict__ block, const uint32_t* __restrict__ qflags,
                                                        uint32_t qcap) {
    __shared__ uint32_t red[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t* hdr = reinterpret_cast<uint32_t*>(block);         // [nq][4]
    uint64_t* ent = block + 2 * (size_t)nq;
    if (q == nq) {                                              // the extra payload rides behind the entries
        float* dst = reinterpret_cast<float*>(ent + cap_entries);
        for (uint32_t i = tid; i < extra_n; i += 256) dst[i] = extra[i];
        return;
    }
    // qflags != nullptr: the merge was enqueued with the batch (no host in between) — the streams are described by what
    // the query kernels left in device memory: {flags, entries} per query, stream q at q * qcap.  A stream that overflowed
    // its region or a query with a fallback pending (bit5) ships nothing and raises bit7: every rank then redoes the merge
    // at collect time, after this rank re-ran its batch.
    auto cnt_of = [&](int p) -> uint32_t {
        if (!qflags) return src_cnt[p];
        const uint32_t n_ = qflags[4 * p + 1];
        return (n_ > qcap || (qflags[4 * p] & 32u)) ? 0u : n_;
    };
    uint32_t part = 0;
    for (int p = tid; p < q; p += 256) part += cnt_of(p);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const uint32_t off = red[0] + red[1] + red[2] + red[3];
    const uint32_t n = cnt_of(q);
    uint32_t fl = qflags ? (qflags[4 * q] & 0x3fu) : src_flags[q];
    if (qflags && (qflags[4 * q + 1] > qcap || (qflags[4 * q] & 32u))) fl |= 128u;
    const bool fits = (uint64_t)off + n <= cap_entries;
    if (tid == 0) {
        hdr[4 * q + 0] = off;
        hdr[4 * q + 1] = n;
        hdr[4 * q + 2] = fl | (fits ? 0u : 64u);                // bit6: this rank's block was too small; bit7: the rank's batch
                                                                // failed before the gather / must be re-run first
        hdr[4 * q + 3] = 0;
    }
    if (!fits) return;
    // bit8: the rank ordered this query on the host (more candidates than the device sort takes); its stream lies in `fix`
    const uint64_t* __restrict__ s = qflags ? stream + (size_t)q * qcap : ((fl & 256u) ? fix : stream) + src_off[q];
    for (uint32_t i = tid; i < n; i += 256) ent[off + i] = s[i];
}

// Level-path batches (sort_cands_kernel ordered the streams): {offset, count, flags}[nq] for dist_pack_kernel from the query
// states, as qadc_dist_collect builds them on the host from the QueryOut records.  bit7 = "the collect call has work to do
// before this rank's streams are final" (a query the device did not order, an overflowed output or pre-scan buffer): every
// rank then redoes the merge at collect time.
__global__ __launch_bounds__(256) void dist_src_from_states_kernel(const QueryState* __restrict__ qstates, int nq, uint32_t out_cap,
                                                                   uint32_t* __restrict__ src) {
    for (int q = threadIdx.x; q < nq; q += 256) {
        const QueryState* qs = qstates + q;
        const uint32_t fl = qs->flags, n = qs->count + qs->reps, off = qs->out_off;
        const bool skip = (fl & 1u) != 0, ordered = (fl & 4u) != 0;
        const bool bad = (fl & 8u) != 0 || (!skip && (!ordered || (uint64_t)off + n > out_cap));
        src[q] = (ordered && !bad) ? off : 0u;
        src[nq + q] = (ordered && !bad && !skip) ? n : 0u;
        src[2 * nq + q] = (fl & 0x3fu) | (bad ? 128u : 0u);
    }
}

// ---------------------------------------------------------------------------------------------
// kv_binheap<unsigned,int8_t>::push (binheap.hpp:75-116) with ONE WAVE per query and the heap in REGISTERS (value, key
// register pairs; see WaveHeap).  Everything about a push is wave-uniform — the value, the path it sinks along — and the
// 64 lanes work out all levels of a sink at once (WaveHeap::sift) instead of following it through dependent LDS round
// trips.  The lanes also FILTER: a chunk of 64 stream entries is loaded coalesced, one ballot finds the entries below
// the current root, and only those are pushed (in order; the mask is re-filtered as the root drops).  An entry that is
// not below the root when its chunk arrives would be rejected by push() at its turn as well (the root never rises once
// the heap is full), so the heap array is the sequential one, entry for entry.
// ---------------------------------------------------------------------------------------------
template <int NREG>
struct WaveHeap {
    // Heap element e lives at POSITION e + 1 (1-based: children of p are 2p and 2p + 1, siblings share an even/odd lane
    // pair of one register); position p = lane p & 63 of register p >> 6.  Positions past `size` (and position 0) hold
    // value 0: "a child that is not there" and "a child <= the value" then stop the sift the same way.
    uint32_t hv[NREG], hk[NREG];
    uint32_t lane;
    uint32_t R, size;                                            // wave-uniform
    static constexpr uint32_t kPos = (uint32_t)NREG * 64u;

    __device__ __forceinline__ uint32_t val_at(uint32_t p) const {
        uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)hv[0], (int)(p & 63u));
#pragma unroll
        for (int j = 1; j < NREG; ++j) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)hv[j], (int)(p & 63u));
            r = (p >> 6) == (uint32_t)j ? t : r;
        }
        return r;
    }
    __device__ __forceinline__ uint32_t key_at(uint32_t p) const {
        uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)hk[0], (int)(p & 63u));
#pragma unroll
        for (int j = 1; j < NREG; ++j) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)hk[j], (int)(p & 63u));
            r = (p >> 6) == (uint32_t)j ? t : r;
        }
        return r;
    }
    __device__ __forceinline__ void set(uint32_t p, uint32_t value, uint32_t key) {
#pragma unroll
        for (int j = 0; j < NREG; ++j) {                         // one compare + two selects per register pair
            const bool own = (lane + 64u * (uint32_t)j) == p;
            hv[j] = own ? value : hv[j];
            hk[j] = own ? key : hk[j];
        }
    }
    __device__ __forceinline__ uint32_t root() const { return (uint32_t)__builtin_amdgcn_readlane((int)hv[0], 1); }

    // all arguments wave-uniform.  Room left: append, bubble up past strictly smaller parents (R pushes per query: the
    // element-by-element form is good enough).
    __device__ __forceinline__ void append(uint32_t key, uint32_t value) {
        uint32_t p = ++size;
        while (p != 1) {
            const uint32_t parent = p >> 1;
            const uint32_t pv = val_at(parent);
            if (!(value > pv)) break;
            set(p, pv, key_at(parent));
            p = parent;
        }
        set(p, value, key);
    }
    // Full heap, value strictly below the root: the root goes, the value sinks (binheap.hpp:75-116 — the right child only
    // if strictly greater, stop at a child <= the value).  Element by element this is a dependent chain of ~25
    // readlane / scalar / select instructions per LEVEL (~2000 cycles per push at R = 100).  Here the 64 lanes decide
    // all levels at once: every position works out whether it is the child its parent would descend to AND above the
    // value (one DPP sibling exchange, two compares), one ballot per register turns that into bit masks, the path from
    // the root is then ~8 scalar instructions per level on those masks, and every position on the path takes over its
    // chosen child's element through ONE bpermute per register (issued before the walk: its latency passes under it).
    __device__ __forceinline__ void sift(uint32_t key, uint32_t value) {
        uint64_t T[NREG];
        uint32_t tv[NREG], tk[NREG];
        const uint32_t even = (lane & 1u) ^ 1u;
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
            const uint32_t sv = (uint32_t)__builtin_amdgcn_mov_dpp((int)hv[j], 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]: the sibling
            const uint32_t sk = (uint32_t)__builtin_amdgcn_mov_dpp((int)hk[j], 0xB1, 0xf, 0xf, false);
            // the parent descends to the right child only if strictly greater: left (even lane) wins with >=, right with >
            const bool take = hv[j] + even > sv && hv[j] > value;
            T[j] = __builtin_amdgcn_ballot_w64(take);
            tv[j] = take ? hv[j] : sv;                           // both lanes of a pair: the element the parent would take
            tk[j] = take ? hk[j] : sk;
        }
        // position p of register j (2j < NREG) pulls from the pair (2p, 2p + 1): register 2j for lanes < 32, 2j + 1 above;
        // the even lanes of the operand carry register 2j's pairs, the odd lanes register 2j + 1's (xor-and select: a
        // select between two array elements is turned into a dynamically indexed load, and the arrays into LDS)
        uint32_t gv[(NREG + 1) / 2], gk[(NREG + 1) / 2];
        const int src = (int)(lane < 32u ? 8u * lane : 8u * (lane - 32u) + 4u);   // byte address of the source lane
        const uint32_t oddmask = 0u - (lane & 1u);
#pragma unroll
        for (int j = 0; 2 * j < NREG; ++j) {
            uint32_t zv = tv[2 * j], zk = tk[2 * j];
            if (2 * j + 1 < NREG) {
                zv ^= (zv ^ tv[2 * j + 1 < NREG ? 2 * j + 1 : 0]) & oddmask;
                zk ^= (zk ^ tk[2 * j + 1 < NREG ? 2 * j + 1 : 0]) & oddmask;
            }
            gv[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)zv);
            gk[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)zk);
        }
        // the walk: q at depth k has its children in registers (2 << k) >> 6 .. ((4 << k) - 1) >> 6 — static per level
        uint32_t q = 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if ((2u << k) >= kPos) break;                        // (compile time)
            const uint32_t l = 2u * q;
            if (((4u << k) > kPos) && l >= kPos) break;          // (only the last level can run past the registers)
            const int clo = (int)((2u << k) >> 6), chi = (int)(min(kPos - 1u, (4u << k) - 1u) >> 6);
            uint64_t t = T[clo];
#pragma unroll
            for (int j = clo + 1; j <= chi; ++j) t = (l >> 6) == (uint32_t)j ? T[j] : t;
            const uint32_t pair = (uint32_t)(t >> (l & 63u)) & 3u;
            if (pair == 0) break;
            q = l + (pair >> 1);
        }
        // the path is q's ancestors: p is one (or q itself) iff q >> (depth(q) - depth(p)) == p — deeper p are > q and can
        // not match whatever the shift; position 0 is kept out by its stand-in value
        const uint32_t clzq = (uint32_t)__builtin_clz(q);
#pragma unroll
        for (int j = 0; j < NREG; ++j) {
            const uint32_t p = (j == 0 && lane == 0) ? 0xffffffffu : lane + 64u * (uint32_t)j;
            const bool own = p == q;
            if (2 * j < NREG) {
                const bool onpath = (q >> (((uint32_t)__builtin_clz(p) - clzq) & 31u)) == p;
                const uint32_t nv = own ? value : gv[j < (NREG + 1) / 2 ? j : 0];
                const uint32_t nk = own ? key : gk[j < (NREG + 1) / 2 ? j : 0];
                hv[j] = onpath ? nv : hv[j];
                hk[j] = onpath ? nk : hk[j];
            } else {
                hv[j] = own ? value : hv[j];
                hk[j] = own ? key : hk[j];
            }
        }
    }
};

// One wave per query: stream[off[q] .. off[q] + n[q]) (entries key | value << 32 | ...), after the (0,127) sentinel
// (db_query_4.cpp:276).  info[q]: bit0 = skip (qmax too high: the reference exits, size 0), bit1 = not replayable here
// (a block overflowed / an unordered query: size 0xffffffff, the host regrows or falls back).
// QF: the single-GPU layout instead — query q's stream at q * cap, {flags, entries} in info[4q], info[4q + 1]
// (scan_query_kernel / order_cands_kernel write those): bit0 of the flags = qmax too high, bit5 = fallback pending.
// kReplayWaves queries per workgroup.  The waves live for about half a millisecond (a query's pushes are a dependent scalar
// chain) beside the scan kernels.  Round 3 packed 16 to a workgroup so that they tie up few CUs; but a 16-wave workgroup must
// first FIND half a CU free, and beside the partition-major scan (28 of 32 wave slots taken, in 4-wave workgroups) it waits
// for that at the head of its queue.  Four to a workgroup start at once in the slots that scan leaves free (see kSideWG).
constexpr int kReplayWaves = 4;                  // (round 3 packed 16 to a workgroup; see kSideWG)
template <int NREG, int QF>
__global__ __launch_bounds__(kReplayWaves * 64) void replay_heap_wave_kernel(const uint64_t* __restrict__ stream, const uint64_t* __restrict__ off,
                                                               const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ info,
                                                               uint32_t cap, int nq, uint32_t R, uint64_t* __restrict__ heaps,
                                                               uint32_t* __restrict__ heap_sizes, int q0, int qstep) {
    // (raised wave priority — s_setprio 3 here and in the two merge kernels — was measured in round 4: no change in one of 8
    // ranks' batch; what the merge chain waits for beside the scans is not the issue port)
    const uint32_t lane = threadIdx.x & 63u;
    // wave w of the launch replays query q and writes heap w.  QF 0 (the multi-GPU merge): q = q0 + w * qstep — a rank that replays
    // only ITS share of a batch's queries (q = rank, rank + world, ...) packs their heaps densely; elsewhere q = w.
    const int w = (int)(blockIdx.x * (uint32_t)kReplayWaves + (threadIdx.x >> 6));
    if (w >= nq) return;
    const int q = QF == 0 ? q0 + w * qstep : w;
    uint32_t fl, n;
    uint64_t o;
    if (QF == 2) {
        // the level path's layout: what sort_cands_kernel left in the query states (info = the QueryState array, cap = the
        // capacity of the compact ordered output); not ordered on the device / past the output: the host replays
        const QueryState* qs = reinterpret_cast<const QueryState*>(info) + q;
        const uint32_t qf = (uint32_t)__builtin_amdgcn_readfirstlane((int)qs->flags);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qs->count + qs->reps));
        o = (uint32_t)__builtin_amdgcn_readfirstlane((int)qs->out_off);
        fl = (qf & 1u) | ((!(qf & 4u) || o + n > cap) ? 2u : 0u);
    } else if (QF == 1) {
        const uint32_t qf = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[4 * q]);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[4 * q + 1]);
        fl = (qf & 1u) | ((n > cap || (qf & 32u)) ? 2u : 0u);    // overflowed stream / fallback pending: the host re-runs the batch
        o = (uint64_t)q * cap;
    } else {
        fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)info[q]);
        n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt[q]);
        o = off[q];
    }
    if (fl & 2u) {
        if (lane == 0) heap_sizes[w] = 0xffffffffu;
        return;
    }
    if (fl & 1u) {
        if (lane == 0) heap_sizes[w] = 0;
        return;
    }
    const uint64_t* __restrict__ src = stream + (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)o));
    WaveHeap<NREG> h;
#pragma unroll
    for (int j = 0; j < NREG; ++j) h.hv[j] = h.hk[j] = 0;
    h.lane = lane;
    h.R = R;
    h.size = 0;
    h.append(0u, 127u);                                          // the sentinel
    uint64_t nxt = lane < n ? src[lane] : 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint64_t cur = nxt;
        const uint32_t m = min(64u, n - base);                   // entries of this chunk
        nxt = base + 64 + lane < n ? src[base + 64 + lane] : 0;  // the next chunk is in flight while this one is pushed
        const uint32_t key = (uint32_t)cur, val = (uint32_t)(cur >> 32) & 0xffu;
        uint32_t j0 = 0;
        while (h.size != R && j0 < m) {                          // the heap is still filling: every entry goes in
            h.append((uint32_t)__builtin_amdgcn_readlane((int)key, (int)j0), (uint32_t)__builtin_amdgcn_readlane((int)val, (int)j0));
            ++j0;
        }
        if (j0 >= m) continue;
        uint32_t root = h.root();
        uint64_t mask = __builtin_amdgcn_ballot_w64(lane >= j0 && lane < m && val < root);
        while (mask) {
            const uint32_t j = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1;
            h.sift((uint32_t)__builtin_amdgcn_readlane((int)key, (int)j), (uint32_t)__builtin_amdgcn_readlane((int)val, (int)j));
            root = h.root();
            mask &= __builtin_amdgcn_ballot_w64(val < root);     // the root dropped: later entries may no longer qualify
        }
    }
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
        const uint32_t p = (uint32_t)j * 64u + lane;             // element p - 1
        if (p >= 1 && p <= h.size) heaps[(size_t)w * R + (p - 1)] = (uint64_t)h.hk[j] | ((uint64_t)h.hv[j] << 32);
    }
    if (lane == 0) heap_sizes[w] = h.size;
}

// ---- sharded replay of the multi-GPU merge: the ranks' heap shares -> the batch's heaps in query order ----
// Rank r replayed queries r, r + world, ...: share block = heaps u64[per][R], then sizes u32[per] (per = ceil(nq / world)) padded
// to whole u64 words, then ONE more word: != 0 if one of the rank's queries had a push stream that was not grouped by assign slot
// (dist_interleave_kernel) — every rank then fails the batch (status word 2 = sizes[nq + 2]), not only the owner of the query.
__global__ __launch_bounds__(256) void dist_heaps_unpack_kernel(const uint64_t* __restrict__ all, size_t share_words, int world, int per, int nq,
                                                                uint32_t R, uint64_t* __restrict__ heaps, uint32_t* __restrict__ sizes) {
    const int q = blockIdx.x;                                    // one workgroup per query
    const int r = q % world, j = q / world;
    const uint64_t* blk = all + (size_t)r * share_words;
    const uint32_t sz = reinterpret_cast<const uint32_t*>(blk + (size_t)per * R)[j];
    const uint32_t n = sz == 0xffffffffu ? 0u : min(sz, R);
    for (uint32_t i = threadIdx.x; i < n; i += 256) heaps[(size_t)q * R + i] = blk[(size_t)j * R + i];
    if (threadIdx.x == 0) {
        sizes[q] = sz;
        if (reinterpret_cast<const uint32_t*>(blk + (size_t)per * R)[((per + 1) / 2) * 2]) sizes[nq + 2] = 1u;
    }
}

// ---- sharded front: the ranks' shares -> the batch's arrays in query order ----
__global__ __launch_bounds__(256) void front_unpack_kernel(const unsigned char* __restrict__ gathered, size_t block_bytes, int world, int per,
                                                           int nq, int ma, size_t tab, int8_t* __restrict__ qt, int32_t* __restrict__ assign,
                                                           uint32_t* __restrict__ front, int32_t* __restrict__ h_assign,
                                                           uint32_t* __restrict__ h_front) {
    const int q = blockIdx.x;                                    // one workgroup per query
    const int r = q / per, i = q % per;
    const unsigned char* blk = gathered + (size_t)r * block_bytes;
    const uint4* src = reinterpret_cast<const uint4*>(blk + (size_t)i * tab);      // (tab is a multiple of 256 bytes)
    uint4* dst = reinterpret_cast<uint4*>(qt + (size_t)q * tab);
    for (size_t w = threadIdx.x; w < tab / 16; w += 256) dst[w] = src[w];
    const int32_t* sa = reinterpret_cast<const int32_t*>(blk + (size_t)per * tab) + (size_t)i * ma;
    for (int a = threadIdx.x; a < ma; a += 256) {
        const int32_t v = sa[a];
        assign[(size_t)q * ma + a] = v;
        h_assign[(size_t)q * ma + a] = v;
    }
    const uint32_t* sf = reinterpret_cast<const uint32_t*>(blk + (size_t)per * tab + (size_t)per * ma * 4) + 4 * (size_t)i;
    if (threadIdx.x < 4) {
        front[4 * (size_t)q + threadIdx.x] = sf[threadIdx.x];
        h_front[4 * (size_t)q + threadIdx.x] = sf[threadIdx.x];
    }
}

// ---- the world's gathered blocks -> ONE stream per query in global scan order (assign slot, rank, position) ----
// Per query: totals and flags over the ranks, exclusive prefix over the queries (one workgroup).
constexpr int kTotalsThreads = 256;
__global__ __launch_bounds__(1024) void dist_totals_kernel(const uint64_t* __restrict__ gathered, size_t block_words, int world, int nq,
                                                           uint64_t* __restrict__ moff, uint32_t* __restrict__ mcnt,
                                                           uint32_t* __restrict__ info, uint32_t* __restrict__ status,
                                                           uint32_t* __restrict__ share_flag) {
    __shared__ uint64_t wsum[16];
    __shared__ unsigned long long rank_tot[16];
    __shared__ uint32_t bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int nthr = (int)blockDim.x;
    const int per = (nq + nthr - 1) / nthr;
    const int lo = min(nq, (int)tid * per), hi = min(nq, lo + per);
    if (tid < 16) rank_tot[tid] = 0;
    if (tid == 0) bad = 0;
    __syncthreads();
    uint64_t mine = 0;
    uint32_t my_bad = 0;
    // (every rank's header of a query in ONE round trip — 16 bytes each, up to 16 ranks in flight: the loop used to wait for
    //  each rank's two words in turn, twice: 16 dependent round trips made this one-workgroup kernel 40 us alone on the GPU)
    unsigned long long rt[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) rt[g] = 0;
    for (int q = lo; q < hi; ++q) {
        uint32_t tot = 0, fl = 0;
        for (int g0 = 0; g0 < world; g0 += 16) {
            uint32_t hc[16], hf[16];                             // header words 1 (entries) and 2 (flags); a block is 8-byte aligned
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                hc[u] = 0;
                hf[u] = 4u;
                if (g0 + u < world) {
                    const uint64_t* hq = gathered + (size_t)(g0 + u) * block_words + 2 * (size_t)q;
                    hc[u] = (uint32_t)(hq[0] >> 32);
                    hf[u] = (uint32_t)hq[1];
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (g0 + u >= world) continue;
                tot += hc[u];
                if (g0 == 0) rt[u] += hc[u];
                const uint32_t f = hf[u];
                if (f & 1u) fl |= 1u;                            // qmax too high (identical on every rank)
                if ((f & (64u | 128u)) || !(f & (4u | 1u))) fl |= 2u; // block too small / the rank failed / not ordered
                my_bad |= (f & (64u | 128u)) | ((f & (4u | 1u)) ? 0u : 256u);
            }
        }
        if (fl) tot = 0;
        mcnt[q] = tot;
        info[q] = fl;
        mine += tot;
    }
    if (status) {                                                // entries every rank wanted to ship: sizes a regrow
#pragma unroll
        for (int g = 0; g < 16; ++g)
            if (g < world && rt[g]) atomicAdd(&rank_tot[g], rt[g]);
    }
    // exclusive prefix of the per-thread sums (64-bit: a batch may gather more than 2^32 entries)
    uint64_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = (uint64_t)__shfl_up((unsigned long long)incl, (unsigned)d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t base = incl - mine;
    for (uint32_t w = 0; w < wave; ++w) base += wsum[w];
    for (int q = lo; q < hi; ++q) {
        moff[q] = base;
        base += mcnt[q];
    }
    if (share_flag && threadIdx.x == 0) *share_flag = 0;         // (sharded replay: this rank's "a stream of mine was broken" word, see below)
    if (status) {                                                // (host-mapped: the caller reads it after the batch's event)
        if (my_bad) atomicOr(&bad, my_bad);
        __syncthreads();
        if (tid == 0) {
            unsigned long long need = 0;
            for (int g = 0; g < world; ++g) need = max(need, rank_tot[g]);
            status[0] = bad;                                     // bit6 / bit7 / bit8 of any header: the merge must be redone
            status[1] = (uint32_t)min(need, 0xffffffffull);      // entries the fullest rank block needs
            status[2] = 0;                                       // set by dist_interleave_kernel: a stream not grouped by assign slot
        }
    }
}

// One workgroup per query.  Rank g's entries of the query are sorted by assign slot already; the merged position of its
// i-th entry (slot s) is  #entries of every rank with a lower slot  +  #entries of lower ranks with slot s  +  its index
// inside rank g's run of slot s.  Both terms are prefix sums over the count matrix cnt[s][g]: one in (s, g) order, one in
// (g, s) order.  Dynamic LDS: 2 x ma x world counters.
__global__ __launch_bounds__(256) void dist_interleave_kernel(const uint64_t* __restrict__ gathered, size_t block_words, int world,
                                                              int nq, int ma, const uint64_t* __restrict__ moff,
                                                              const uint32_t* __restrict__ info, uint64_t* __restrict__ merged, int q0, int qstep,
                                                              uint32_t* __restrict__ status, uint32_t* __restrict__ share_flag) {
    uint32_t* cnt_sg = reinterpret_cast<uint32_t*>(qsmem);        // [ma][world] -> exclusive prefix in (s, g) order
    uint32_t* cnt_gs = cnt_sg + (size_t)ma * world;               // [world][ma] -> exclusive prefix in (g, s) order
    __shared__ uint32_t wtot[4];
    const int q = q0 + (int)blockIdx.x * qstep;                   // (a rank that merges only its share of the queries: rank, rank + world, ...)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (info[q]) return;
    const int cells = ma * world;
    for (int i = tid; i < 2 * cells; i += 256) cnt_sg[i] = 0;
    __syncthreads();
    if (ma > 1) {
        // A rank's stream is in scan order, i.e. sorted by slot: a slot's count is the length of its RUN.  The lanes at a run's
        // first and last entry store its bounds (plain LDS stores: a few dozen per rank); counting every entry with two LDS
        // atomics — 64 lanes of a wave on the same counter, since neighbours share the slot — serialised every one of them.
        // (Round 4.  It did not move a rank's pipelined batch: beside the scans this launch waits for wave slots)
        for (int g = 0; g < world; ++g) {
            const uint32_t* hdr = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * block_words) + 4 * (size_t)q;
            const uint64_t* __restrict__ ent = gathered + (size_t)g * block_words + 2 * (size_t)nq + hdr[0];
            const uint32_t n = hdr[1];
            constexpr int kIB = 4;                               // rows of 64 entries in flight per wave (one round trip, not four)
            for (uint32_t i0 = wave * 64u; i0 < n; i0 += 256 * kIB) {   // (wave-uniform bounds: the shuffles need every lane)
                uint32_t slv[kIB], pv[kIB], nx[kIB];
#pragma unroll
                for (int u = 0; u < kIB; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * 256u + lane;
                    slv[u] = i < n ? (uint32_t)(ent[i] >> 40) & 0x3fffu : 0xffffffffu;
                    pv[u] = nx[u] = 0xffffffffu;
                    if (lane == 0 && i && i < n) pv[u] = (uint32_t)(ent[i - 1] >> 40) & 0x3fffu;
                    if (lane == 63 && i + 1 < n) nx[u] = (uint32_t)(ent[i + 1] >> 40) & 0x3fffu;
                }
#pragma unroll
                for (int u = 0; u < kIB; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * 256u + lane;
                    const uint32_t sl = slv[u];
                    uint32_t prev = (uint32_t)__shfl_up((int)sl, 1), next = (uint32_t)__shfl_down((int)sl, 1);
                    if (lane == 0) prev = pv[u];
                    if (lane == 63) next = nx[u];
                    // The counts below come from run BOUNDS, so they are right only if a rank's stream is grouped by slot, slots
                    // ascending and below ma — what the ordering passes guarantee.  A stream that is not (a future pack path, a
                    // corrupted block) would scatter out of place: flag it (status[2]: the host fails the batch) and keep the
                    // indices inside the count arrays.
                    const bool live = i < n;
                    const bool broken = live && (sl >= (uint32_t)ma || (prev != 0xffffffffu && prev > sl));
                    if (broken && status) status[2] = 1u;
                    // (sharded replay: only the rank that owns the query interleaves it — the word travels in its heap share, and
                    //  dist_heaps_unpack_kernel raises status[2] on EVERY rank: the ranks must agree on the batch's error)
                    if (broken && share_flag) *share_flag = 1u;
                    if (live && sl >= (uint32_t)ma) continue;
                    if (live && prev != sl) cnt_gs[g * ma + sl] = i;             // the run's first entry
                    if (live && next != sl) cnt_sg[sl * world + g] = i + 1;      // one past its last
                }
            }
        }
        __syncthreads();
        for (int c = tid; c < cells; c += 256) {                 // cell (g, s): count = end - first (both 0 where the slot is absent)
            const int g = c / ma, s = c % ma;
            const uint32_t cnt = cnt_sg[s * world + g] - cnt_gs[c];
            cnt_gs[c] = cnt;
            cnt_sg[s * world + g] = cnt;
        }
    } else if ((int)tid < world) {                                // one probe: a rank's count is its header's
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(gathered + (size_t)tid * block_words) + 4 * (size_t)q;
        cnt_sg[tid] = cnt_gs[tid] = hdr[1];
    }
    __syncthreads();
    // two exclusive scans of `cells` counters each: thread t owns a contiguous run of each array
    for (int arr = 0; arr < 2; ++arr) {
        uint32_t* a = arr ? cnt_gs : cnt_sg;
        const int per = (cells + 255) / 256;
        const int lo = min(cells, (int)tid * per), hi = min(cells, lo + per);
        uint32_t mine = 0;
        for (int i = lo; i < hi; ++i) mine += a[i];
        const uint32_t incl = q_wave_incl_sum(mine);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t base = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) base += wtot[w];
        for (int i = lo; i < hi; ++i) {
            const uint32_t c = a[i];
            a[i] = base;
            base += c;
        }
        __syncthreads();
    }
    uint64_t* __restrict__ out = merged + moff[q];
    for (int g = 0; g < world; ++g) {
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(gathered + (size_t)g * block_words) + 4 * (size_t)q;
        const uint64_t* __restrict__ ent = gathered + (size_t)g * block_words + 2 * (size_t)nq + hdr[0];
        const uint32_t n = hdr[1];
        const uint32_t rank_first = cnt_gs[g * ma];              // entries of the ranks below g (= prefix at (g, slot 0))
        constexpr int kOB = 4;                                   // entries in flight per lane
        for (uint32_t i0 = tid; i0 < n; i0 += 256 * kOB) {
            uint64_t ev[kOB];
#pragma unroll
            for (int u = 0; u < kOB; ++u) {
                const uint32_t i = i0 + (uint32_t)u * 256u;
                ev[u] = i < n ? ent[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < kOB; ++u) {
                const uint32_t i = i0 + (uint32_t)u * 256u;
                if (i >= n) continue;
                const uint64_t e = ev[u];
                const uint32_t sl = ma > 1 ? min((uint32_t)(e >> 40) & 0x3fffu, (uint32_t)ma - 1u) : 0u;   // (clamped: see the guard above)
                const uint32_t run_first = cnt_gs[g * ma + sl] - rank_first;  // index of the slot's first entry in rank g's stream
                out[cnt_sg[sl * world + g] + (i - run_first)] = e;
            }
        }
    }
}

}  // namespace

// Dynamic-LDS opt-in above the default limit, once per (kernel, device); safe from several host threads (one per index).
static hipError_t dynamic_lds_optin(const void* fn, int bytes, std::atomic<uint64_t>& done_devices) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;                   // (beyond 64 devices: ask every time, it is cheap)
    const uint64_t bit = tracked ? 1ull << dev : 0;
    if (tracked && (done_devices.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (tracked) done_devices.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

uint32_t query_kernel_lds_values(int M) { return M == 16 ? QCfg<16>::FCAP : QCfg<32>::FCAP; }

template <int M, int U, int OCC, bool NT, bool MULTI, bool HEAD = false, int WG = 1024>
static hipError_t launch_scan_query_nt(int nq, const QueryKernelArgs& args, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    const size_t lds = QCfg<M, WG>::LDS_BYTES;
    const hipError_t e = dynamic_lds_optin(reinterpret_cast<const void*>(&scan_query_kernel<M, U, OCC, NT, MULTI, HEAD, WG>), (int)lds, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((scan_query_kernel<M, U, OCC, NT, MULTI, HEAD, WG>), dim3(nq * (MULTI ? args.G : 1)), dim3(WG), lds, stream, args);
    return hipGetLastError();
}

template <int M, int U, int OCC, bool NT>
static hipError_t launch_scan_query_inline(int nq, const QueryKernelArgs& args, hipStream_t stream, const void* payload, size_t bytes) {
    static std::atomic<uint64_t> done{0};
    const size_t lds = QCfg<M>::LDS_BYTES + QCfg<M>::FBH_BYTES;
    const hipError_t e = dynamic_lds_optin(reinterpret_cast<const void*>(&scan_query_inline_kernel<M, U, OCC, NT>), (int)lds, done);
    if (e != hipSuccess) return e;
    QueryKernelInline ia;
    ia.a = args;
    ia.a.inline_input = payload ? 1u : 0u;
    if (payload) std::memcpy(ia.payload, payload, bytes);
    hipLaunchKernelGGL((scan_query_inline_kernel<M, U, OCC, NT>), dim3(nq * args.G), dim3(kQWG), lds, stream, ia);
    return hipGetLastError();
}

template <int M, int U, int OCC>
static hipError_t launch_scan_query_v(int nq, const QueryKernelArgs& args, hipStream_t stream, const void* payload, size_t bytes) {
    if (payload && (args.head_codes || args.G <= 1 || bytes > kInlineBytes)) return hipErrorInvalidValue;
    // The IVF head in 512-thread workgroups (option "head_wg"; default at 16x4).  Same image, same walk, 8 waves per query: two
    // such workgroups fill a CU's LDS (2 x 64 KiB) with 16 of its 32 wave slots, so the other launches of the pipeline — the
    // next batches' coarse distances / tables / select, the previous batch's replay — find wave slots on every CU while the
    // head runs.  Alone on the GPU the head is SLOWER this way (C3: 0.172 -> 0.205 ms per 1024 queries, half the waves to hide
    // its round trips behind); in the pipelined batches it is faster (0.281 -> 0.235) and the batch with it: 0.656 -> 0.612 us
    // per query at C3, same box, profiles/r06_head_wg_ab.txt.  32x4 (C5) keeps 16 waves: 3.99 -> 4.02-4.05 us per query with 8.
    // 4 waves per query and 8 waves at 128 VGPRs with 4 tiles in flight were measured too (C3 0.70 / 0.63): not kept.
    if (args.head_codes && args.head_wg == 512 && U == 2)
        return args.nontemporal ? launch_scan_query_nt<M, 2, OCC, true, true, true, 512>(nq, args, stream)
                                : launch_scan_query_nt<M, 2, OCC, false, true, true, 512>(nq, args, stream);
    if (args.head_codes)
        return args.nontemporal ? launch_scan_query_nt<M, U, OCC, true, true, true>(nq, args, stream)
                                : launch_scan_query_nt<M, U, OCC, false, true, true>(nq, args, stream);
    if (args.G > 1)                                              // (small batches: one workgroup per CU at most, 128 VGPRs —
                                                                 //  the walk's first 12 vectors per lane live in registers)
        return args.nontemporal ? launch_scan_query_inline<M, U, 4, true>(nq, args, stream, payload, bytes)
                                : launch_scan_query_inline<M, U, 4, false>(nq, args, stream, payload, bytes);
    return args.nontemporal ? launch_scan_query_nt<M, U, OCC, true, false>(nq, args, stream)
                            : launch_scan_query_nt<M, U, OCC, false, false>(nq, args, stream);
}

// Two 64-vector tiles (16-byte loads per lane) in flight per wave and iteration (3, 4 and 6 were measured in round 4: no faster).
// 16x4: two workgroups per CU (64 KiB of tables each); 32x4: two as well (16 replicas).
hipError_t launch_scan_query(int M, int nq, const QueryKernelArgs& args, hipStream_t stream, const void* inline_payload,
                             size_t inline_bytes) {
    if (M == 16) return launch_scan_query_v<16, 2, QCfg<16>::OCC>(nq, args, stream, inline_payload, inline_bytes);
    return launch_scan_query_v<32, 2, QCfg<32>::OCC>(nq, args, stream, inline_payload, inline_bytes);
}

hipError_t launch_lone_front(int M, const LoneFrontArgs& args, hipStream_t stream) {
    if (args.S < 1 || args.S * args.R > (uint32_t)kLoneFrontMaxKeys || args.R > (uint32_t)kLoneFrontSlice) return hipErrorInvalidValue;
    const size_t lds = (size_t)M * 16 * 4 + (size_t)kLoneFrontMaxKeys * 4 + 128 + 272 * 4;
    if (M == 16) hipLaunchKernelGGL(lone_front_kernel<16>, dim3(args.S), dim3(kQWG), lds, stream, args);
    else hipLaunchKernelGGL(lone_front_kernel<32>, dim3(args.S), dim3(kQWG), lds, stream, args);
    return hipGetLastError();
}

void launch_ivf_plan(const int32_t* d_assign, const PartDesc* d_parts, int nq, int ma, int s0, int K, uint32_t* d_cnt,
                     uint32_t* d_goff, uint32_t* d_fill, ScanItem* d_items, hipStream_t stream) {
    if (nq <= 0 || ma <= s0) return;
    hipLaunchKernelGGL(ivf_count_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, d_assign, d_parts, nq, ma, s0, d_cnt);
    hipLaunchKernelGGL(ivf_offsets_kernel, dim3(1), dim3(kSideWG), 0, stream, d_cnt, K, d_goff);
    hipLaunchKernelGGL(ivf_scatter_kernel, dim3((nq + 3) / 4), dim3(256), 0, stream, d_assign, d_parts, nq, ma, s0, d_goff,
                       d_fill, d_items);
}

hipError_t launch_order_cands(const QueryState* d_qs, const Cand* d_regions, uint32_t cand_cap, uint32_t ccap, int nq,
                              uint64_t* d_stream, uint32_t cap, QueryOut* d_qout, uint32_t* d_qflags, int ma, uint32_t pos_bits, hipStream_t stream) {
    static std::atomic<uint64_t> done{0};
    const hipError_t e = dynamic_lds_optin(reinterpret_cast<const void*>(&order_cands_kernel), kOrderCandCap * 8, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(order_cands_kernel, dim3(nq), dim3(kQWG), kOrderCandCap * 8, stream, d_qs, d_regions, cand_cap, ccap, d_stream, cap,
                       d_qout, d_qflags, (uint32_t)ma, pos_bits);
    return hipGetLastError();
}

hipError_t launch_dist_pack(const uint32_t* d_src_off, const uint32_t* d_src_cnt, const uint32_t* d_src_flags, int nq,
                            const uint64_t* d_stream, const uint64_t* d_fix, uint32_t cap_entries, const float* d_extra,
                            uint32_t extra_n, uint64_t* d_block, hipStream_t stream) {
    hipLaunchKernelGGL(dist_pack_kernel, dim3(nq + 1), dim3(256), 0, stream, d_src_off, d_src_cnt, d_src_flags, nq, d_stream,
                       d_fix, cap_entries, d_extra, extra_n, d_block, (const uint32_t*)nullptr, 0u);
    return hipGetLastError();
}

// qadc_dist_init_loopback's stand-in for the all-gather: ONE kernel (as a real all-gather is), the block into every slot
__global__ __launch_bounds__(256) void replicate_block_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t words,
                                                              int world) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        const uint64_t v = src[i];
        for (int g = 0; g < world; ++g) dst[(size_t)g * words + i] = v;
    }
}
hipError_t launch_replicate_block(const void* d_src, void* d_dst, size_t words, int world, hipStream_t stream) {
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>((words + 255) / 256, 64));   // (a collective's few workgroups)
    hipLaunchKernelGGL(replicate_block_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const uint64_t*>(d_src),
                       static_cast<uint64_t*>(d_dst), words, world);
    return hipGetLastError();
}

hipError_t launch_dist_src_from_states(const QueryState* d_qs, int nq, uint32_t out_cap, uint32_t* d_src, hipStream_t stream) {
    hipLaunchKernelGGL(dist_src_from_states_kernel, dim3(1), dim3(256), 0, stream, d_qs, nq, out_cap, d_src);
    return hipGetLastError();
}

hipError_t launch_dist_pack_qflags(const uint32_t* d_qflags, int nq, const uint64_t* d_stream, uint32_t qcap, uint32_t cap_entries,
                                   uint64_t* d_block, hipStream_t stream) {
    hipLaunchKernelGGL(dist_pack_kernel, dim3(nq), dim3(256), 0, stream, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, nq, d_stream, (const uint64_t*)nullptr, cap_entries, (const float*)nullptr, 0u, d_block,
                       d_qflags, qcap);
    return hipGetLastError();
}

hipError_t launch_front_unpack(const unsigned char* d_gathered, size_t block_bytes, int world, int per, int nq, int ma, size_t tab,
                               int8_t* d_qt, int32_t* d_assign, uint32_t* d_front, int32_t* h_assign, uint32_t* h_front,
                               hipStream_t stream) {
    hipLaunchKernelGGL(front_unpack_kernel, dim3(nq), dim3(256), 0, stream, d_gathered, block_bytes, world, per, nq, ma, tab, d_qt,
                       d_assign, d_front, h_assign, h_front);
    return hipGetLastError();
}

uint32_t replay_wave_max_R() { return 320; }                     // positions 1 .. R in up to six register pairs of 64
size_t dist_interleave_max_cells() { return 8192; }              // ma x world counters, twice, in 64 KiB of LDS

// One wave per query over stream[off[q] .. +cnt[q]); R <= replay_wave_max_R().
template <int QF>
static hipError_t launch_replay_wave_t(const uint64_t* d_stream, const uint64_t* d_off, const uint32_t* d_cnt, const uint32_t* d_info,
                                       uint32_t cap, int nq, uint32_t R, uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream,
                                       int q0 = 0, int qstep = 1) {
    if (R == 0 || R > replay_wave_max_R()) return hipErrorInvalidValue;
    const dim3 grid((nq + kReplayWaves - 1) / kReplayWaves), block(kReplayWaves * 64);
    switch ((R + 64) / 64) {                                     // positions 1 .. R
        case 1: hipLaunchKernelGGL((replay_heap_wave_kernel<1, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
        case 2: hipLaunchKernelGGL((replay_heap_wave_kernel<2, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
        case 3: hipLaunchKernelGGL((replay_heap_wave_kernel<3, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
        case 4: hipLaunchKernelGGL((replay_heap_wave_kernel<4, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
        case 5: hipLaunchKernelGGL((replay_heap_wave_kernel<5, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
        default: hipLaunchKernelGGL((replay_heap_wave_kernel<6, QF>), grid, block, 0, stream, d_stream, d_off, d_cnt, d_info, cap, nq, R, d_heaps, d_heap_sizes, q0, qstep); break;
    }
    return hipGetLastError();
}
hipError_t launch_replay_heap_wave(const uint64_t* d_stream, const uint64_t* d_off, const uint32_t* d_cnt, const uint32_t* d_info,
                                   int nq, uint32_t R, uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream, int q0, int qstep) {
    return launch_replay_wave_t<0>(d_stream, d_off, d_cnt, d_info, 0, nq, R, d_heaps, d_heap_sizes, stream, q0, qstep);
}
hipError_t launch_replay_heap_wave_qflags(const uint32_t* d_qflags, const uint64_t* d_stream, uint32_t cap, int nq, uint32_t R,
                                          uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream) {
    return launch_replay_wave_t<1>(d_stream, nullptr, nullptr, d_qflags, cap, nq, R, d_heaps, d_heap_sizes, stream);
}
hipError_t launch_replay_heap_wave_states(const QueryState* d_qs, const uint64_t* d_stream, uint32_t out_cap, int nq, uint32_t R,
                                          uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream) {
    return launch_replay_wave_t<2>(d_stream, nullptr, nullptr, reinterpret_cast<const uint32_t*>(d_qs), out_cap, nq, R, d_heaps,
                                   d_heap_sizes, stream);
}

// The world's gathered blocks -> heaps: totals + prefix, interleave into global scan order, wave-per-query replay.
// d_moff [nq] u64, d_mcnt / d_info [nq] u32, d_merged [world * entries of a block] u64: scratch of the caller.
hipError_t launch_dist_merge(const uint64_t* d_gathered, size_t block_words, int world, int nq, int ma, uint32_t R,
                             uint64_t* d_moff, uint32_t* d_mcnt, uint32_t* d_info, uint64_t* d_merged, uint64_t* d_heaps,
                             uint32_t* d_heap_sizes, hipStream_t stream, uint32_t* d_status, int q0, int qstep) {
    if (world > 16 || (size_t)ma * world > dist_interleave_max_cells() || R > replay_wave_max_R() || q0 < 0 || qstep < 1) return hipErrorInvalidValue;
    const int nmine = q0 < nq ? (nq - q0 + qstep - 1) / qstep : 0;   // queries q0, q0 + qstep, ...: interleaved and replayed here
    static std::atomic<uint64_t> done{0};
    const hipError_t e = dynamic_lds_optin(reinterpret_cast<const void*>(&dist_interleave_kernel), 65536, done);
    if (e != hipSuccess) return e;
    // (sharded replay, qstep = world > 1: d_heap_sizes points into this rank's heap share; the share's flag word follows its sizes)
    uint32_t* d_share_flag = qstep > 1 ? d_heap_sizes + (((nq + qstep - 1) / qstep + 1) / 2) * 2 : nullptr;
    hipLaunchKernelGGL(dist_totals_kernel, dim3(1), dim3(kTotalsThreads), 0, stream, d_gathered, block_words, world, nq, d_moff, d_mcnt, d_info,
                       d_status, d_share_flag);
    if (nmine == 0) return hipGetLastError();
    hipLaunchKernelGGL(dist_interleave_kernel, dim3(nmine), dim3(256), (size_t)ma * world * 8, stream, d_gathered, block_words, world, nq, ma,
                       d_moff, d_info, d_merged, q0, qstep, d_status, d_share_flag);
    return launch_replay_heap_wave(d_merged, d_moff, d_mcnt, d_info, nmine, R, d_heaps, d_heap_sizes, stream, q0, qstep);
}

hipError_t launch_dist_heaps_unpack(const uint64_t* d_all, size_t share_words, int world, int per, int nq, uint32_t R, uint64_t* d_heaps,
                                    uint32_t* d_sizes, hipStream_t stream) {
    hipLaunchKernelGGL(dist_heaps_unpack_kernel, dim3(nq), dim3(256), 0, stream, d_all, share_words, world, per, nq, R, d_heaps, d_sizes);
    return hipGetLastError();
}

}  // namespace qadc
