// One workgroup = one query: the whole scanner_4::query_scan chain (db_query_4.cpp:245-309) in ONE launch.
//
// The level-structured path (qadc_kernels.hip) cuts every query's scan order into bound levels, one launch per
// level, many workgroups per run: right for a flat list of 10^8..10^9 codes, wrong for an IVF batch, where a
// query probes a few 10^5 codes in dozens of short partitions, there are hundreds of queries in flight, and
// the dependent chain of launches, the per-item table builds, the candidate sort and the host planner cost more
// than the scan.  Here a query belongs to one 1024-thread workgroup that walks the query's scan order
// SEQUENTIALLY, tile by tile:
//
//   1. float pre-scan of the probed partitions' starts (scan_4<M>, query_common.hpp:59-90; same add order) into
//      LDS, R-th smallest by a 4-pass radix select in LDS  -> qmax            (db_query_4.cpp:230-242, 259)
//   2. qmin over all ma tables, negative clamp, QuantizerMAX<int8_t>           (db_query_4.cpp:37-71, 258-284)
//   3. int8 scan of every probed partition in assign[] order                   (simd_scan.hpp:125-187):
//      pair-fused 256-entry byte tables in LDS (conflict-free: a 256-byte table covers every bank once),
//      cand = min(127, sum).  Because ONE workgroup sees the query's codes in scan order, the prefix bound
//       is refreshed after every tile from a 128-bin histogram in LDS — no levels, no launch
//      boundaries — and the qualifying codes are appended IN SCAN ORDER by a wave-ordered compaction, so there
//      is no sort pass either.  The padding-lane replays of a partition's last code (simd_layout.hpp:46-50,
//      simd_scan.hpp:67) are expanded while writing.
//
// Output per query: the ordered push stream (entry = key | value << 32 | assign slot << 40, the format of the
// sorted output of sort_cands_kernel) and a QueryOut record; the heap replay stays where it was (host for small
// batches, replay_heap_lanes_kernel for large ones).  Exactness argument: every dropped code has
// cand >= the R-th smallest (<127) value of a set of codes that all PRECEDE it in this query's scan order.
//
// The workgroups of a batch are independent and each streams its own codes: 256 CUs x 2 resident workgroups
// keep ~64 KiB of 16-byte loads in flight per CU, which is what it takes to pull HBM bandwidth with one query
// per workgroup.  gfx950 only.
#include "qadc_kernels.h"

#include <cfloat>

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

template <int M>
struct QCfg {
    static constexpr int CS = M / 2, DW = M / 8, CPL = 16 / CS;
    static constexpr int FCAP = M == 16 ? 12288 : 8192;          // pre-scan values kept in LDS
    static constexpr int VALS_OFF = 0;                           // float[FCAP]
    static constexpr int WTAB_OFF = VALS_OFF + FCAP * 4;         // float[16 waves][M*16] (pre-scan); later:
    static constexpr int PTAB_OFF = WTAB_OFF;                    //   u8[CS*256] pair tables
    static constexpr int TQ_OFF = PTAB_OFF + CS * 256;           //   u8[M*16] staged int8 table
    static constexpr int MISC_OFF = WTAB_OFF + kQWaves * M * 16 * 4;
    // misc (u32 words): [0..255] radix histogram / [0..127] value histogram, then scalars
    static constexpr int LDS_BYTES = MISC_OFF + (256 + 64 + 64) * 4;
};

// Bound = smallest v such that at least R emitted candidates have value <= v, else 127 (wave 0, lanes own bins 2l, 2l+1).
__device__ __forceinline__ uint32_t q_bound_from_hist(const uint32_t* hist, uint32_t R, uint32_t lane) {
    const uint32_t c0 = hist[2 * lane], c1 = hist[2 * lane + 1];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += o;
    }
    const uint32_t excl = incl - (c0 + c1);
    uint32_t b = 127;
    if (excl + c0 >= R) b = 2 * lane;
    else if (incl >= R) b = 2 * lane + 1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b = min(b, (uint32_t)__shfl_xor(b, d, 64));
    return min(b, 127u);
}

template <int M, int U>
__global__ __launch_bounds__(kQWG, 8) void scan_query_kernel(QueryKernelArgs A) {
    using C = QCfg<M>;
    constexpr int CS = C::CS, DW = C::DW, CPL = C::CPL;
    float* vals = reinterpret_cast<float*>(qsmem + C::VALS_OFF);
    float* wtab = reinterpret_cast<float*>(qsmem + C::WTAB_OFF);
    unsigned char* ptab = qsmem + C::PTAB_OFF;
    unsigned char* tq = qsmem + C::TQ_OFF;
    uint32_t* misc = reinterpret_cast<uint32_t*>(qsmem + C::MISC_OFF);
    uint32_t* hist = misc;                       // [256] during the select, [128] value histogram during the scan
    uint32_t* wcnt = misc + 256;                 // [U][16] per-wave entry counts of one emission round
    uint32_t& s_nvals = misc[320];
    uint32_t& s_prefix = misc[321];
    uint32_t& s_k = misc[322];
    uint32_t& s_bound = misc[323];
    uint32_t* s_any = misc + 328;                // [3] "some lane qualifies" flags, rotated per tile (see the scan loop)
    uint32_t& s_count = misc[325];
    float* redf = reinterpret_cast<float*>(misc + 336);   // [16] per-wave minima

    const int q = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int ma = A.ma;
    const int32_t* __restrict__ assign = A.assign + (size_t)q * ma;
    const PartDesc* __restrict__ parts = A.parts;
    const size_t tbase = (size_t)q * ma * (M * 16);
    int8_t* __restrict__ qt_all = A.qtables + tbase;
    uint64_t* __restrict__ stream = A.stream + (size_t)q * A.cap;
    uint64_t* __restrict__ stream2 = A.stream2 ? A.stream2 + (size_t)q * A.cap : nullptr;
    const uint32_t R = A.R;

    if (tid < 256) hist[tid] = 0;
    if (tid == 0) { s_nvals = 0; s_any[0] = s_any[1] = s_any[2] = 0; s_count = 0; s_bound = 127; }
    __syncthreads();

    float qmin = 0.0f, qmax = 0.0f;
    uint32_t flags = 0;
    if (A.ftables) {
        float* __restrict__ ft_all = A.ftables + tbase;
        // ---- 1. float pre-scan of the starts.  waves_per_probe waves share a probe when ma < 16. ----
        // total starts of the query decide where the values live (LDS, or the global scratch when there are many)
        uint32_t mine = 0;
        for (int a = tid; a < ma; a += kQWG) {
            const PartDesc& d = parts[assign[a]];
            mine += d.global_n ? d.start_n : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
        if (lane == 0) wcnt[wave] = mine;
        __syncthreads();
        uint32_t total_starts = 0;
#pragma unroll
        for (int w = 0; w < kQWaves; ++w) total_starts += wcnt[w];
        const bool in_lds = total_starts <= (uint32_t)C::FCAP;
        float* __restrict__ gvals = A.fvals + (size_t)q * A.fcap;
        __syncthreads();

        const int wpp = ma >= kQWaves ? 1 : kQWaves / ma;        // waves per probe
        const int pstride = kQWaves / wpp;                       // probes in flight
        float* mytab = wtab + wave * (M * 16);
        float lmin = FLT_MAX;
        const int pslot = (int)wave / wpp, sub = (int)wave % wpp;
        // the loop bounds are workgroup-uniform (a wave without a probe idles through the barriers)
        for (int a0 = 0; a0 < ma; a0 += pstride) {
            const int a = a0 + pslot;
            const bool active = pslot < pstride && a < ma;
            uint32_t sn = 0, base = 0;
            const uint8_t* sc = nullptr;
            if (active) {
                const float* __restrict__ ft = ft_all + (size_t)a * (M * 16);
                q_wave_lds_sync();
                for (int i = lane; i < M * 16; i += 64) {
                    const float v = ft[i];
                    mytab[i] = v;
                    if (sub == 0) lmin = fminf(lmin, v);
                }
                q_wave_lds_sync();
                const PartDesc d = parts[assign[a]];
                sn = d.global_n ? d.start_n : 0u;
                sc = d.starts ? d.starts : d.codes;
                if (sub == 0 && sn) {
                    if (lane == 0) base = atomicAdd(&s_nvals, sn);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (lane == 0) wcnt[16 + pslot] = base;      // hand the base to the probe's other waves
                }
            }
            if (wpp > 1) {
                __syncthreads();
                if (active && sn) base = wcnt[16 + pslot];
            }
            for (uint32_t i = (uint32_t)sub * 64u + lane; i < sn; i += 64u * (uint32_t)wpp) {
                uint32_t dw[DW];
                if constexpr (M == 16) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = ((const __attribute__((address_space(1))) u32x2*)(uintptr_t)sc)[i];
                    dw[0] = v.x; dw[1] = v.y;
                } else {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = ((const __attribute__((address_space(1))) u32x4*)(uintptr_t)sc)[i];
                    dw[0] = v.x; dw[1] = v.y; dw[2] = v.z; dw[3] = v.w;
                }
                // scan_4<M> order: byte by byte, low nibble then high nibble, accumulated from 0 (query_common.hpp:72-80)
                float cand = 0.0f;
#pragma unroll
                for (int b = 0; b < CS; ++b) {
                    const uint32_t byte = (dw[b >> 2] >> (8 * (b & 3))) & 0xffu;
                    cand += mytab[(2 * b) * 16 + (byte & 15u)];
                    cand += mytab[(2 * b + 1) * 16 + (byte >> 4)];
                }
                if (in_lds) vals[base + i] = cand;
                else gvals[base + i] = cand;
            }
            if (wpp > 1) __syncthreads();                        // the hand-over slot is reused by the next round
        }
        // qmin = min over ALL ma tables (db_query_4.cpp:258)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lmin = fminf(lmin, __shfl_xor(lmin, d, 64));
        if (lane == 0) redf[wave] = lmin;
        __threadfence_block();
        __syncthreads();
        qmin = redf[0];
#pragma unroll
        for (int w = 1; w < kQWaves; ++w) qmin = fminf(qmin, redf[w]);

        // ---- R-th smallest of the pre-scan values = tmp_bh.max() (db_query_4.cpp:259); FLT_MAX if fewer than R ----
        const uint32_t n = s_nvals;
        if (n < R) {
            qmax = FLT_MAX;
        } else {
            if (tid == 0) { s_prefix = 0; s_k = R; }
            for (int pass = 0; pass < 4; ++pass) {
                const int lo = 24 - 8 * pass;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                const uint32_t prefix = s_prefix;
                const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (lo + 8));
                for (uint32_t i = tid; i < n; i += kQWG) {
                    const uint32_t key = q_fkey(in_lds ? vals[i] : gvals[i]);
                    if ((key & himask) == prefix) atomicAdd(&hist[(key >> lo) & 0xffu], 1u);
                }
                __syncthreads();
                if (tid < 64) {                                  // wave 0: 4 bins per lane, pick the digit holding rank k
                    const uint32_t k = s_k;
                    const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                    const uint32_t sum4 = c0 + c1 + c2 + c3;
                    uint32_t incl = sum4;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t o = __shfl_up(incl, d, 64);
                        if (tid >= (uint32_t)d) incl += o;
                    }
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
                    }
                }
                __syncthreads();
            }
            qmax = q_funkey(s_prefix);
        }
        // ---- 2. qmin / clamp / QuantizerMAX (db_query_4.cpp:258-284, 37-71) ----
        if (qmin < 0) { qmin = 0; flags |= 2u; }
        if (qmax > 1e30f) flags |= 1u;
        const float delta = (qmax - qmin) / 127;
        const float scale = 127.0f / (qmax - qmin);
        const int all = ma * M * 16;
        for (int i = tid; i < all; i += kQWG) {
            float v = ft_all[i];
            if (v < 0) { v = 0; ft_all[i] = 0; }
            int8_t o;
            if (flags & 1u) o = 127;
            else if (v >= qmax) o = 127;
            else o = (int8_t)(int)(A.quant_mode == 0 ? (v - qmin) / delta : (v - qmin) * scale);
            qt_all[i] = o;
        }
        __threadfence_block();
        __syncthreads();
        if (tid < 256) hist[tid] = 0;                            // becomes the value histogram of the scan
        __syncthreads();
    }

    if (flags & 1u) {                                            // the reference prints a warning and exits: no scan
        if (tid == 0) {
            QueryOut o;
            o.count = 0; o.reps = 0; o.flags = flags | 4u; o.out_off = (uint32_t)((size_t)q * A.cap);
            o.qmin = qmin; o.qmax = qmax; o.pad[0] = o.pad[1] = 0;
            A.qout[q] = o;
        }
        return;
    }

    // ---- 3. int8 scan in assign[] order ----
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gvec_t;
    uint32_t ramp = 64;                                          // vectors of the next tile: 64, 128, ... up to U*1024
    // "some lane of the tile qualifies" flag: tile i uses s_any[i % 3].  After tile i's barrier every wave has read
    // tile i-1's flag, and nobody can set tile i+2's flag before passing tile i+1's barrier — so the flag of tile
    // i+2 (= the one of tile i-1) is cleared by thread 0 between those two barriers.
    uint32_t fi = 0;
    for (int a = 0; a < ma; ++a) {
        const PartDesc d = parts[assign[a]];
        if (d.n == 0) continue;                                  // empty partition (db_query_4.cpp:291-293) / no local codes
        // pair tables of this probe: P_b[x] = T[2b][x & 15] + T[2b+1][x >> 4]
        if (tid < M * 4) reinterpret_cast<uint32_t*>(tq)[tid] = reinterpret_cast<const uint32_t*>(qt_all + (size_t)a * (M * 16))[tid];
        __syncthreads();
        for (int e = tid; e < CS * 256; e += kQWG) {
            const int b = e >> 8, x = e & 255;
            ptab[e] = (unsigned char)(tq[(2 * b) * 16 + (x & 15)] + tq[(2 * b + 1) * 16 + (x >> 4)]);
        }
        __syncthreads();
        const gvec_t src = (gvec_t)(uintptr_t)d.codes;
        const uint32_t n = d.n;
        const uint32_t nvec = (n + CPL - 1) / CPL;
        const uint32_t dup_pos = (d.first_pos + d.n == d.global_n) ? d.n - 1u : 0xffffffffu;
        const uint32_t dup_reps = (16u - d.global_n % 16u) % 16u;
        const uint32_t key_base = d.key_base + d.first_pos;
        const uint64_t slot_bits = (uint64_t)(uint32_t)a << 40;

        for (uint32_t t0 = 0; t0 < nvec;) {
            const uint32_t width = min(ramp, (uint32_t)(U * kQWG));
            u32x4 v[U];
            uint32_t e[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t off = u * kQWG + tid;
                e[u] = t0 + off;
                v[u] = u32x4{0, 0, 0, 0};
                if (off < width && e[u] < nvec) v[u] = __builtin_nontemporal_load(src + e[u]);
                else e[u] = 0xffffffffu;
            }
            const uint32_t bound = s_bound;
            uint32_t cand[U * CPL];
            uint32_t best = 127u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t dd[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    uint32_t sum = 0;
#pragma unroll
                    for (int b = 0; b < CS; ++b) {
                        const uint32_t w = dd[c * DW + (b >> 2)];
                        sum += ptab[b * 256 + ((w >> (8 * (b & 3))) & 0xffu)];
                    }
                    const bool live = e[u] != 0xffffffffu && e[u] * CPL + c < n;
                    cand[u * CPL + c] = live ? min(sum, 127u) : 127u;
                    best = min(best, cand[u * CPL + c]);
                }
            }
            const bool hit = best < bound;
            if (__builtin_amdgcn_ballot_w64(hit) != 0 && lane == 0) s_any[fi] = 1;
            __syncthreads();
            const uint32_t any = s_any[fi];
            if (tid == 0) s_any[fi == 0 ? 2 : fi - 1] = 0;       // the flag of tile i+2
            fi = fi == 2 ? 0 : fi + 1;
            if (any) {
                // ---- ordered emission: tile u before tile u+1, wave w before w+1, lane order inside a wave ----
                uint32_t cnt[U], incl[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c_ = 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
                        if (cand[u * CPL + c] < bound) c_ += 1u + ((e[u] * CPL + c == dup_pos) ? dup_reps : 0u);
                    cnt[u] = c_;
                    uint32_t in_ = c_;
#pragma unroll
                    for (int dlt = 1; dlt < 64; dlt <<= 1) {
                        const uint32_t o = __shfl_up(in_, dlt, 64);
                        if (lane >= (uint32_t)dlt) in_ += o;
                    }
                    incl[u] = in_;
                    if (lane == 63) wcnt[u * kQWaves + wave] = in_;
                }
                __syncthreads();
                uint32_t before = s_count, total = 0;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int w = 0; w < kQWaves; ++w) {
                        const uint32_t c_ = wcnt[u * kQWaves + w];
                        total += c_;
                    }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (cnt[u]) {
                        uint32_t pos = before + incl[u] - cnt[u];
                        for (int uu = 0; uu < u; ++uu)
                            for (int w = 0; w < kQWaves; ++w) pos += wcnt[uu * kQWaves + w];
                        for (uint32_t w = 0; w < wave; ++w) pos += wcnt[u * kQWaves + w];
#pragma unroll
                        for (int c = 0; c < CPL; ++c) {
                            const uint32_t cv = cand[u * CPL + c];
                            if (cv < bound) {
                                const uint32_t p = e[u] * CPL + c;
                                const uint32_t key = d.labels ? d.labels[p] : key_base + p;
                                const uint64_t en = (uint64_t)key | ((uint64_t)cv << 32) | slot_bits;
                                const uint32_t reps = 1u + (p == dup_pos ? dup_reps : 0u);
                                for (uint32_t r = 0; r < reps; ++r, ++pos)
                                    if (pos < A.cap) {
                                        stream[pos] = en;
                                        if (stream2) stream2[pos] = en;
                                    }
                                atomicAdd(&hist[cv], 1u);
                            }
                        }
                    }
                }
                __syncthreads();
                if (wave == 0) {
                    const uint32_t b = q_bound_from_hist(hist, R, lane);
                    if (lane == 0) { s_bound = b; s_count = before + total; }
                }
                __syncthreads();
            }
            t0 += width;
            ramp = min(ramp * 2u, (uint32_t)(U * kQWG));
        }
        __syncthreads();                                         // ptab / tq are rebuilt for the next probe
    }
    if (tid == 0) {
        QueryOut o;
        o.count = s_count;                                       // entries requested (replays included); > cap = overflow
        o.reps = 0;
        o.flags = flags | 4u;
        o.out_off = (uint32_t)((size_t)q * A.cap);
        o.qmin = qmin;
        o.qmax = qmax;
        o.pad[0] = o.pad[1] = 0;
        A.qout[q] = o;
        if (A.qstate_flags) {                                    // what replay_heap_lanes_kernel reads
            A.qstate_flags[4 * q + 0] = flags | 4u;
            A.qstate_flags[4 * q + 1] = s_count;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kv_binheap<unsigned,int8_t>::push (binheap.hpp:75-116) for 64 queries per wave: lane l replays the ordered
// stream of query q0 + l through its own max-heap of capacity R in LDS, sentinel (0,127) first
// (db_query_4.cpp:276).  Heap slot i of lane l lives at (i*64 + l) * 8 bytes: the bank of an access depends on
// the lane only, so the 64 lanes never conflict, whatever slots they are at.  One query's pushes are inherently
// sequential (each is a dependent chain of LDS round trips); 64 of them per wave, and a few waves per batch,
// replay a 1024-query IVF batch in the shadow of the next batch's scan, on a handful of CUs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void replay_heap_lanes_kernel(const uint32_t* __restrict__ qflags,
                                                               const uint64_t* __restrict__ stream, uint32_t cap, int nq,
                                                               uint32_t R, uint64_t* __restrict__ heaps,
                                                               uint32_t* __restrict__ heap_sizes) {
    uint64_t* hv = reinterpret_cast<uint64_t*>(qsmem);          // [R][64]
    const uint32_t lane = threadIdx.x;
    const int q = blockIdx.x * 64 + (int)lane;
    const bool have = q < nq;
    uint32_t flags = 0, n = 0;
    if (have) { flags = qflags[4 * q]; n = qflags[4 * q + 1]; }
    bool host_replay = false;                                   // overflowed stream: the batch is re-run / replayed on the host
    if (have && n > cap) { host_replay = true; n = 0; }
    if (flags & 1u) n = 0;                                      // qmax too high: the reference exits, no result
    const uint64_t* __restrict__ src = stream + (size_t)(have ? q : 0) * cap;
    uint32_t size = 0;
    auto val_of = [](uint64_t e) { return (int32_t)((e >> 32) & 0xffu); };
    auto at = [&](uint32_t i) -> uint64_t& { return hv[i * 64u + lane]; };
    auto push = [&](uint64_t e) {
        const int32_t value = val_of(e);
        if (size != R) {
            uint32_t i = size++;
            while (i != 0) {
                const uint32_t parent = (i - 1) / 2;
                const uint64_t pe = at(parent);
                if (!(value > val_of(pe))) break;
                at(i) = pe;
                i = parent;
            }
            at(i) = e;
            return;
        }
        if (!(value < val_of(at(0)))) return;
        uint32_t i = 0;
        for (;;) {
            const uint32_t l = 2 * i + 1;
            if (l >= size) break;
            uint64_t ce = at(l);
            uint32_t c = l;
            if (l + 1 < size) {
                const uint64_t re = at(l + 1);
                if (val_of(re) > val_of(ce)) { ce = re; c = l + 1; }
            }
            if (val_of(ce) <= value) break;
            at(i) = ce;
            i = c;
        }
        at(i) = e;
    };
    if (have && !host_replay && !(flags & 1u)) push((uint64_t)127 << 32);   // the sentinel: key 0, value 127
    // four entries per lane in flight: the stream of a lane is sequential in memory, 8 bytes at a time
    for (uint32_t j = 0; j < n; j += 4) {
        uint64_t e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e4[u] = j + u < n ? src[j + u] & 0xffffffffffull : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j + u < n) push(e4[u]);
    }
    // write the heaps out query by query, coalesced
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = 0; l < 64; ++l) {
        const int qq = blockIdx.x * 64 + l;
        if (qq >= nq) break;
        const uint32_t sz = __shfl(size, l, 64);
        const uint32_t hr = __shfl((uint32_t)host_replay, l, 64);
        if (hr) {
            if (lane == 0) heap_sizes[qq] = 0xffffffffu;
            continue;
        }
        for (uint32_t i = lane; i < sz; i += 64) heaps[(size_t)qq * R + i] = hv[i * 64u + (uint32_t)l];
        if (lane == 0) heap_sizes[qq] = sz;
    }
}

}  // namespace

size_t query_kernel_lds_bytes(int M) { return M == 16 ? QCfg<16>::LDS_BYTES : QCfg<32>::LDS_BYTES; }
uint32_t query_kernel_lds_values(int M) { return M == 16 ? QCfg<16>::FCAP : QCfg<32>::FCAP; }

hipError_t launch_scan_query(int M, int nq, const QueryKernelArgs& args, hipStream_t stream) {
    // dynamic LDS above the default limit is opted into per (kernel, device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static uint64_t done16 = 0, done32 = 0;
    uint64_t& done = M == 16 ? done16 : done32;
    const size_t lds = query_kernel_lds_bytes(M);
    if (dev < 64 && !(done & (1ull << dev))) {
        e = M == 16 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_query_kernel<16, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                    : hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_query_kernel<32, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done |= 1ull << dev;
    }
    if (M == 16) hipLaunchKernelGGL((scan_query_kernel<16, 2>), dim3(nq), dim3(kQWG), lds, stream, args);
    else         hipLaunchKernelGGL((scan_query_kernel<32, 2>), dim3(nq), dim3(kQWG), lds, stream, args);
    return hipGetLastError();
}

uint32_t replay_lanes_max_R() { return 288; }                    // R * 512 B of LDS per wave (<= 144 KiB)

hipError_t launch_replay_heap_lanes(const uint32_t* d_qflags, const uint64_t* d_stream, uint32_t cap, int nq, uint32_t R,
                                    uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static uint64_t done = 0;
    if (dev < 64 && !(done & (1ull << dev))) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&replay_heap_lanes_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(replay_lanes_max_R() * 512));
        if (e != hipSuccess) return e;
        done |= 1ull << dev;
    }
    hipLaunchKernelGGL(replay_heap_lanes_kernel, dim3((nq + 63) / 64), dim3(64), (size_t)R * 512, stream, d_qflags, d_stream,
                       cap, nq, R, d_heaps, d_heap_sizes);
    return hipGetLastError();
}

}  // namespace qadc
