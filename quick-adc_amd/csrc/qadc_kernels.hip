// MI355X (gfx950 / CDNA4) kernels of the Quick-ADC scan engine.
//
// Replaces, on device, the reference's hot loop scan_avx_4<M> (simd_scan.hpp:125-187) and the
// float pre-scan scan_4<M> (query_common.hpp:59-90).  Nothing here is a translation of the AVX2
// code: the pshufb register lookup becomes an LDS gather against pair-fused, bank-replicated byte
// tables, the 16-code block transpose (simd_layout.hpp) is dropped in favour of plain row-major
// codes read with one coalesced 16-byte load per lane, and the sequential heap is replaced by a
// prefix-bound filter whose output is replayed on the host.
//
// The op is a gather + byte add bound by HBM reads: no MFMA.
#include "qadc_kernels.h"
#include "qadc_float_sum.h"
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <type_traits>

namespace qadc {

constexpr int kWG = 1024;       // threads per workgroup of the int8 scan (16 waves)

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// Dynamic-LDS opt-in above the default limit is a property of (kernel, DEVICE): it is requested once per device
// the kernel is launched on (an index on GPU 1 after one on GPU 0 in the same process needs its own opt-in), and a
// failure is kept for the host (take_launch_error) instead of being dropped.
static thread_local hipError_t g_launch_err = hipSuccess;

hipError_t take_launch_error() {
    const hipError_t e = g_launch_err;
    g_launch_err = hipSuccess;
    return e;
}

static void ensure_dynamic_lds(const void* fn, int bytes, std::atomic<uint64_t>& done_devices) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
    const uint64_t bit = 1ull << dev;
    if (done_devices.load(std::memory_order_acquire) & bit) return;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { g_launch_err = e; return; }
    if (dev != 63) done_devices.fetch_or(bit, std::memory_order_release);
}

// ---------------------------------------------------------------------------------------------
// int8 scan
//
// LDS image (per workgroup): for byte position b of a code (b = 4g + j) the pair table
//     P_b[x] = T[2b][x & 15] + T[2b+1][x >> 4]          (<= 254, one byte)
// is stored at   region(g) + x*256 + (g&1)*128 + bank*4 + j     for bank = 0..31,
// region(g) = (g>>1) * 64 KiB.  A lane reads with bank = lane & 31, so the 32 lanes of a
// ds_read lane group hit 32 different banks whatever their x: conflict-free by construction.
// One dword holds the four tables of a group, so the 32-fold replication costs 64 KiB per
// 8 code bytes (M=16: 64 KiB, M=32: 128 KiB).
// The byte address x*256 + bank*4 is formed by ONE v_perm_b32 from the code dword and the
// lane constant; (g&1)*128 + j rides in the ds_read immediate offset.
// ---------------------------------------------------------------------------------------------
template <int M>
struct ScanCfg {
    static constexpr int CS = M / 2;             // code bytes
    static constexpr int DW = M / 8;             // dwords per code
    static constexpr int CPL = 16 / CS;          // codes per 16-byte lane load
    static constexpr int TABLE_BYTES = (M / 16) * 65536;
    static constexpr int STAGE_OFF = TABLE_BYTES;          // int8 [M][16] staging copy
    static constexpr int HIST_OFF = STAGE_OFF + M * 16;    // u32 [128] prefix histogram
    static constexpr int BOUND_OFF = HIST_OFF + 512;       // u32 bound
    static constexpr int LDS_BYTES = BOUND_OFF + 16;
};

template <int M>
__device__ __forceinline__ void build_pair_tables(const int8_t* __restrict__ qt) {
    using C = ScanCfg<M>;
    const int t = threadIdx.x;
    // stage the int8 table (M*16 bytes) into LDS
    if (t < M * 4) reinterpret_cast<uint32_t*>(smem + C::STAGE_OFF)[t] = reinterpret_cast<const uint32_t*>(qt)[t];
    __syncthreads();
    const unsigned char* T = smem + C::STAGE_OFF;
    // The image is written in ADDRESS order: thread t stores the 16-byte units t, t + kWG, ...; a wave's ds_write_b128 covers
    // 1 KiB contiguous — every bank once.  (Until round 4 a thread wrote the 128-byte row of one (dword g, byte x) pair by
    // itself: the rows of a wave's lanes lie 256 bytes apart, i.e. in the SAME banks — a 32-way conflict on every store,
    // ~4 K LDS cycles per 64 KiB image instead of ~256.)  A wave's units of iteration k belong to 8 (g, x) pairs, 8 lanes each;
    // lane l computes pair l's dword once and ds_bpermute hands it to the lanes that replicate it.
    constexpr int kIter = C::TABLE_BYTES / 16 / kWG;     // 16x4: 4, 32x4: 8 units per thread
    const uint32_t lane = (uint32_t)t & 63u, wave = (uint32_t)t >> 6;
    uint32_t w = 0;
    {
        const uint32_t pi = lane & (8u * kIter - 1u);    // (16x4: lanes 32..63 repeat 0..31)
        const uint32_t k = pi >> 3, within = pi & 7u;
        const uint32_t x = (k & 3u) * 64u + wave * 4u + (within >> 1);
        const uint32_t g = (k >> 2) * 2u + (within & 1u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t b = 4u * g + j;
            const uint32_t v = (uint32_t)T[(2 * b) * 16 + (x & 15u)] + (uint32_t)T[(2 * b + 1) * 16 + (x >> 4)];
            w |= v << (8 * j);
        }
    }
#pragma unroll
    for (int k = 0; k < kIter; ++k) {
        const uint32_t wk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((k * 8u + (lane >> 3)) * 4u), (int)w);
        *reinterpret_cast<uint4*>(smem + ((size_t)k * kWG + t) * 16) = make_uint4(wk, wk, wk, wk);
    }
}

// inclusive wave scan on the DPP path (row shifts + row broadcasts, 7 VALU operations) — __shfl_up goes through the LDS
// crossbar, ~100 cycles per hop in a dependent chain of six
__device__ __forceinline__ uint32_t dpp_wave_incl_sum(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);   // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);   // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);   // row_shr:8, banks 2-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// bound = smallest v with #(emitted candidates of all earlier levels with value <= v) >= R, else 127.
// Valid for every code of this level because those candidates all precede it in scan order.
__device__ __forceinline__ uint32_t prefix_bound(const QueryState* qs, int level, uint32_t R, uint32_t* lds_hist,
                                                 uint32_t* lds_bound) {
    const int t = threadIdx.x;
    if (t < 128) {
        uint32_t c = 0;
        for (int l = 0; l < level; ++l) c += qs->hist[l * 128 + t];
        lds_hist[t] = c;
    }
    __syncthreads();
    if (t < 64) {
        const uint32_t c0 = lds_hist[2 * t], c1 = lds_hist[2 * t + 1];
        const uint32_t incl = dpp_wave_incl_sum(c0 + c1);
        const uint32_t excl = incl - (c0 + c1);
        const uint64_t reached = __builtin_amdgcn_ballot_w64(incl >= R);
        uint32_t b = excl + c0 >= R ? 2 * t : 2 * t + 1;         // (meaningful in the first lane that reaches R)
        b = reached ? min((uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(reached ? reached : 1)), 127u) : 127u;
        if (t == 0) *lds_bound = b;
    }
    __syncthreads();
    return *lds_bound;
}

__device__ __noinline__ void emit_candidate(QueryState* qs, CandHeader* hdr, Cand* __restrict__ region, uint32_t cap,
                                            const uint32_t* __restrict__ labels, uint32_t key_base, uint32_t order,
                                            uint32_t dup_pos, uint32_t dup_reps, uint32_t pos, uint32_t val) {
    const uint32_t reps = pos == dup_pos ? dup_reps : 0u;
    const uint32_t slot = atomicAdd(&qs->count, 1u);
    if (slot < cap) {
        Cand c;
        c.order = order | (reps << 20);
        c.pos = pos;
        c.key = labels ? labels[pos] : key_base + pos;
        c.val = val;
        region[slot] = c;
    } else {
        atomicAdd(&hdr->overflow, 1u);
    }
    if (reps) atomicAdd(&qs->reps, reps);
    atomicAdd(&qs->hist[(order >> 16) * 128 + val], 1u);
}

// The pair tables start at LDS address 0 (these kernels have no static __shared__, so the dynamic
// region begins there — checked once per workgroup by lds_base_is_zero()); the lookups therefore use
// ABSOLUTE LDS addresses, which spares one "add the array base" VALU instruction per lookup.
typedef const __attribute__((address_space(3))) unsigned char* lds_bytes_t;

__device__ __forceinline__ void lds_base_is_zero() {
    if (reinterpret_cast<uintptr_t>((lds_bytes_t)smem) != 0) __builtin_trap();
}

template <int M>
__device__ __forceinline__ uint32_t pair_sum(const uint32_t* d, uint32_t lane_lo, uint32_t lane_hi) {
    uint32_t s = 0;
#pragma unroll
    for (int w = 0; w < M / 8; ++w) {
        const uint32_t lo = (w >> 1) ? lane_hi : lane_lo;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // byte0 = bank*4, byte1 = code byte k, byte2 = region bit, byte3 = 0
            const uint32_t a = __builtin_amdgcn_perm(d[w], lo, 0x0c020000u | ((4u + k) << 8));
            s += *reinterpret_cast<lds_bytes_t>(static_cast<uintptr_t>(a + (w & 1) * 128 + k));
        }
    }
    return s;
}

// Compile-time forms: NT = non-temporal loads, CHUNK = each workgroup owns a contiguous run of tiles instead of a grid-stride,
// PROBE = replace the LDS lookups by a trivial reduction (the streaming-ceiling diagnostic of bench.py; results meaningless).
// U = 2 16-KiB tiles in flight per lane (1 and 4, and a software-prefetch form, were measured in rounds 1-2: profiles/README.md).
template <int M, bool NT, bool CHUNK, bool PROBE>
__global__ __launch_bounds__(kWG, (M == 16 ? 8 : 4)) void scan_i8_kernel(
    const ScanItem* __restrict__ items, const int8_t* __restrict__ qtables, QueryState* __restrict__ qstates,
    CandHeader* __restrict__ hdr, Cand* __restrict__ out, uint32_t cand_cap, uint32_t R, uint32_t sib_items) {
    using C = ScanCfg<M>;
    constexpr int U = 2;
    // Workgroup -> (run, position in the run).  2-D launch: blockIdx.y = run.  Sibling-major 1-D launch
    // (sib_items = runs in the launch, all over the SAME code range, one per query): the hardware deals consecutive
    // workgroup ids round-robin to the 8 XCDs, so ids that are equal mod 8 share an L2; the decode below puts the
    // sib_items workgroups that read the same tiles (one per query) on one XCD, dispatched back to back: the first
    // of them pulls a tile from HBM, the others find it in that XCD's L2.
    uint32_t bx = blockIdx.x, by = blockIdx.y, G = gridDim.x;
    if (sib_items) {
        G = gridDim.x / sib_items;
        if ((G & 7u) == 0) {
            const uint32_t r = blockIdx.x >> 3;
            by = r % sib_items;
            bx = (r / sib_items) * 8u + (blockIdx.x & 7u);
        } else {
            by = blockIdx.x % sib_items;
            bx = blockIdx.x / sib_items;
        }
    }
    const ScanItem it = items[by];
    QueryState* qs = qstates + it.query;
    out += (uint64_t)it.query * cand_cap;                       // this query's candidate region

    build_pair_tables<M>(qtables + (uint64_t)it.table * (M * 16));
    const uint32_t bound = prefix_bound(qs, it.order >> 16, R, reinterpret_cast<uint32_t*>(smem + C::HIST_OFF),
                                        reinterpret_cast<uint32_t*>(smem + C::BOUND_OFF));

    const uint32_t tid = threadIdx.x;
    const uint32_t lane_lo = (tid & 31u) * 4u;
    const uint32_t lane_hi = lane_lo | 0x10000u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // codes live in global memory: say so, or the non-temporal builtin falls back to FLAT loads
    typedef const __attribute__((address_space(1))) u32x4* gvec_t;
    const gvec_t src = (gvec_t)(uintptr_t)it.codes;
    lds_base_is_zero();
    const uint32_t n = it.n;
    const uint32_t nvec = (n + C::CPL - 1) / C::CPL;            // 16-byte vectors in the run
    const uint32_t ntiles = (nvec + kWG - 1) / kWG;
    // tile sequence of this workgroup: first, first+step, ... < last
    uint32_t first = bx, last = ntiles, step = G;
    if (CHUNK) {
        const uint32_t per = (ntiles + G - 1) / G;
        first = bx * per;
        last = min(ntiles, first + per);
        step = 1;
    }

    // FULL = every lane of every tile of the iteration holds CPL valid codes: no predicates, no branches
    // around the lookups, so the compiler can keep all U*CPL*CS ds_reads of an iteration in flight.
    auto load_tiles = [&](uint32_t t0, u32x4 (&v)[U], uint32_t (&e)[U], auto full) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u * step;
            e[u] = t * kWG + tid;                               // vector index
            if (decltype(full)::value) {
                v[u] = NT ? __builtin_nontemporal_load(src + e[u]) : src[e[u]];
            } else {
                v[u] = u32x4{0, 0, 0, 0};
                if (t < last && e[u] < nvec) v[u] = NT ? __builtin_nontemporal_load(src + e[u]) : src[e[u]];
                else e[u] = 0xffffffffu;
            }
        }
    };
    auto process = [&](const u32x4 (&v)[U], const uint32_t (&e)[U], auto full) {
        uint32_t cand[U * C::CPL];
        uint32_t best = 127u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int c = 0; c < C::CPL; ++c) {
                uint32_t s;
                if (PROBE) {
                    s = 0;
#pragma unroll
                    for (int w = 0; w < C::DW; ++w) s ^= d[c * C::DW + w];
                    s = (s == 0x12345678u) ? 0u : 200u;
                } else {
                    s = pair_sum<M>(d + c * C::DW, lane_lo, lane_hi);
                }
                uint32_t cv = min(s, 127u);
                if (!decltype(full)::value) {
                    // lanes past the end of the run never qualify: 127 is not < bound (bound <= 127)
                    const bool live = e[u] != 0xffffffffu && e[u] * C::CPL + c < n;
                    cv = live ? cv : 127u;
                }
                cand[u * C::CPL + c] = cv;
                best = min(best, cv);
            }
        }
        if (__builtin_expect(best < bound, 0)) {                // rare: one branch per U tiles
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < C::CPL; ++c)
                    if (cand[u * C::CPL + c] < bound)
                        emit_candidate(qs, hdr, out, cand_cap, it.labels, it.key_base, it.order, it.dup_pos, it.dup_reps,
                                       it.pos0 + e[u] * C::CPL + c, cand[u * C::CPL + c]);
        }
    };
    using full_t = std::integral_constant<bool, true>;
    using part_t = std::integral_constant<bool, false>;
    // tiles [0, tiles_full) contain only complete vectors of complete codes
    const uint32_t tiles_full = (n / C::CPL) / kWG;
    const uint32_t full_last = min(last, tiles_full);

    uint32_t t0 = first;
    for (; t0 + (U - 1) * step < full_last; t0 += step * U) {       // steady state: unpredicated
        u32x4 v[U];
        uint32_t e[U];
        load_tiles(t0, v, e, full_t());
        process(v, e, full_t());
    }
    for (; t0 < last; t0 += step * U) {                              // ragged end of the run
        u32x4 v[U];
        uint32_t e[U];
        load_tiles(t0, v, e, part_t());
        process(v, e, part_t());
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-query streaming scan: ONE pass over the codes serves up to 8 queries.
//
// Once the queries of a batch share the codes through L2 (sibling-major launch above), scan_i8_kernel is bound
// by the LDS lookup pipe: one ds_read_u8 per (code byte, query) = 2 LDS cycles per 64 lookups, and 3/4 of every
// bank access is thrown away.  Here a lookup returns the entries of EIGHT queries at once:
//
//   LDS image: for sub-quantizer t and centroid x, row (t, x) = { T_q[t][x] as u16 : q = 0..7 } = 16 bytes at
//              t*256 + x*16.  One ds_read_b128 per code nibble = 4 LDS cycles per 64 lanes x 8 queries,
//              i.e. 8 LDS cycles per (64 codes, query) against 16.
//   No replication: a 16-lane ds_read_b128 group conflicts only when two lanes want DIFFERENT rows in the same
//   banks; the 16 rows of a table tile the 64 banks exactly once, and lanes that want the same row are one
//   broadcast.  Conflict-free for any code data, in 4 KiB (16x4) / 8 KiB (32x4) of LDS instead of 64 / 128 KiB.
//   Sums: the 16-bit fields never overflow (<= 32 x 127), so a row is added as TWO 64-bit integers
//   (v_lshl_add_u64: 4 queries per instruction, half the instructions of v_pk_add_u16), no widening, no
//   per-query loop.
//   Test: adding (0x8000 - bound) to every field sets the field's top bit iff sum >= bound (no carry between
//   fields: <= 4064 + 32768); the results are AND-ed over the U x CPL codes of an iteration and tested with ONE
//   branch; the rare path emits per query exactly as scan_i8_kernel does.
//
// Exactness is untouched: the same integer sums, the same prefix bound per query, the same candidates.
// ---------------------------------------------------------------------------------------------
constexpr int kMQ = 8;          // queries per pass
constexpr int kMQWG = 256;      // threads per workgroup

typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u64x2* lds_row_t;

// wave-level prefix bound (no LDS, no barrier): lane l owns value bins 2l and 2l+1
__device__ __forceinline__ uint32_t prefix_bound_wave(const QueryState* qs, int level, uint32_t R, uint32_t lane) {
    uint32_t c0 = 0, c1 = 0;
    for (int l = 0; l < level; ++l) {
        const uint2 h = *reinterpret_cast<const uint2*>(&qs->hist[l * 128 + 2 * lane]);
        c0 += h.x;
        c1 += h.y;
    }
    const uint32_t incl = dpp_wave_incl_sum(c0 + c1);
    const uint32_t excl = incl - (c0 + c1);
    const uint64_t reached = __builtin_amdgcn_ballot_w64(incl >= R);
    if (reached == 0) return 127u;
    const uint32_t b = excl + c0 >= R ? 2 * lane : 2 * lane + 1;   // (meaningful in the first lane that reaches R)
    return min((uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(reached)), 127u);
}

// (byte K of d) & mask in ONE VALU instruction (sub-dword operand select); the compiler finds this form for only a
// quarter of the lookups by itself and spends a shift + and on the others
template <int K>
__device__ __forceinline__ uint32_t byte_and(uint32_t d, uint32_t mask) {
    uint32_t r;
    // (a plain v_and_b32 for byte 0 — full issue rate on gfx950, every SDWA form half — was tried in round 5: nothing at the IVF
    //  shapes, +0.6 % on the flat batched step, profiles/r05_byte0_and_ab.txt)
    if (K == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(mask), "v"(d));
    if (K == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(mask), "v"(d));
    if (K == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(mask), "v"(d));
    if (K == 3) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(mask), "v"(d));
    return r;
}

typedef uint64_t u64x1 __attribute__((ext_vector_type(1)));
template <int NQ> struct MqRow;
template <> struct MqRow<8> { typedef u64x2 type; };          // 8 queries: 16-byte rows, ds_read_b128
template <> struct MqRow<4> { typedef u64x1 type; };          // 4 queries:  8-byte rows, ds_read_b64 (half the LDS cycles)

// One group's pass with NQ = 8 or 4 query seats.  A group whose live seats fit 4 (the remainder groups of the IVF second
// phase: seats are filled from 0 upward) takes the 4-seat form: the same lookups return 8 bytes instead of 16, i.e. 4 LDS
// cycles per (64 codes, nibble pair) instead of 8, and half the adds.
template <int M, int U, int NQ, bool PIPE>
__device__ __forceinline__ void scan_mq_body(const ScanItem* __restrict__ its, const ScanItem& it, int nq, uint32_t bx, uint32_t G,
                                             const int8_t* __restrict__ qtables, QueryState* __restrict__ qstates,
                                             CandHeader* __restrict__ hdr, Cand* __restrict__ out, uint32_t cand_cap, uint32_t R) {
    constexpr int CS = M / 2, DW = M / 8, CPL = 16 / CS;
    constexpr int ROWB = 2 * NQ, NW = NQ / 4;                    // bytes per row, u64 words per row
    typedef typename MqRow<NQ>::type row_t;
    typedef const __attribute__((address_space(3))) row_t* lds_rowq_t;
    constexpr int TAB = M * 256;                                 // bytes reserved for the row image; then u32 bound[8]
    uint32_t* lbound = reinterpret_cast<uint32_t*>(smem + TAB);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

    // ---- row image: thread -> (t, x); the queries' entries widened to u16; absent queries read as 0 ----
    for (int e = tid; e < M * 16; e += kMQWG) {
        uint16_t row[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            uint16_t v = 0;
            if (j < nq && its[j].n != 0) v = (uint16_t)(uint8_t)qtables[(uint64_t)its[j].table * (M * 16) + e];   // (n == 0: empty seat)
            row[j] = v;
        }
        uint64_t* dst = reinterpret_cast<uint64_t*>(smem + e * ROWB);   // little-endian: query j = field j&3 of u64 j>>2
#pragma unroll
        for (int w = 0; w < NW; ++w)
            dst[w] = (uint64_t)row[4 * w] | ((uint64_t)row[4 * w + 1] << 16) | ((uint64_t)row[4 * w + 2] << 32) | ((uint64_t)row[4 * w + 3] << 48);
    }
    // ---- bounds: wave w computes queries w, w+4 (absent queries: 0 = nothing qualifies) ----
    for (int j = (int)wave; j < NQ; j += kMQWG / 64) {
        uint32_t b = 0;
        if (j < nq && its[j].n != 0) b = prefix_bound_wave(qstates + its[j].query, its[j].order >> 16, R, lane);
        if (lane == 0) lbound[j] = b;
    }
    __syncthreads();
    uint32_t bq[NQ];
    uint64_t bias[NW];                                           // (0x8000 - bound) in every 16-bit field
#pragma unroll
    for (int w = 0; w < NW; ++w) bias[w] = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        bq[j] = __builtin_amdgcn_readfirstlane(lbound[j]);
        bias[j >> 2] |= (uint64_t)(0x8000u - bq[j]) << (16 * (j & 3));
    }
    constexpr uint64_t kTop = 0x8000800080008000ull;

    typedef const __attribute__((address_space(1))) u32x4_t* gvec_t;
    const gvec_t src = (gvec_t)(uintptr_t)it.codes;
    const uint32_t n = it.n;
    const uint32_t nvec = (n + CPL - 1) / CPL;
    const uint32_t ntiles = (nvec + kMQWG - 1) / kMQWG;

    // row offset = x * ROWB.  16-byte rows: the high nibble of byte k is (byte k) & 0xf0, the low one the same of d << 4;
    // 8-byte rows: (d >> 1) & 0x78 and (d << 3) & 0x78 (the bits that cross a byte boundary fall outside the mask)
    constexpr uint32_t nib_mask = NQ == 8 ? 0xf0u : 0x78u;
    // sums of the NQ queries for one code (DW dwords at d)
    auto code_sums = [&](const uint32_t* d) -> row_t {
        row_t a = {};
#pragma unroll
        for (int w = 0; w < DW; ++w) {
            const uint32_t dl = NQ == 8 ? d[w] << 4 : d[w] << 3;
            const uint32_t dh = NQ == 8 ? d[w] : d[w] >> 1;
#define QADC_MQ_BYTE(k)                                                                                        \
            {                                                                                                  \
                /* code byte b = 4w + k: sub-quantizer 2b takes the low nibble, 2b+1 the high one */           \
                const int t0 = 2 * (4 * w + (k));                                                              \
                const uint32_t xl = byte_and<(k)>(dl, nib_mask), xh = byte_and<(k)>(dh, nib_mask);             \
                a += *reinterpret_cast<lds_rowq_t>(static_cast<uintptr_t>(xl + t0 * (16 * ROWB)));             \
                a += *reinterpret_cast<lds_rowq_t>(static_cast<uintptr_t>(xh + (t0 + 1) * (16 * ROWB)));       \
            }
            QADC_MQ_BYTE(0) QADC_MQ_BYTE(1) QADC_MQ_BYTE(2) QADC_MQ_BYTE(3)
#undef QADC_MQ_BYTE
        }
        return a;
    };

    auto run = [&](uint32_t t0, auto full) {
        u32x4_t v[U];
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u * G;
            e[u] = t * kMQWG + tid;
            if (decltype(full)::value) {
                v[u] = src[e[u]];
            } else {
                v[u] = u32x4_t{0, 0, 0, 0};
                if (t < ntiles && e[u] < nvec) v[u] = src[e[u]];
                else e[u] = 0xffffffffu;
            }
        }
        row_t sums[U * CPL];
        uint64_t all[NW];                                        // stays kTop while no field is below its bound
#pragma unroll
        for (int w = 0; w < NW; ++w) all[w] = kTop;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const row_t a = code_sums(d + c * DW);
                sums[u * CPL + c] = a;
                const bool live = decltype(full)::value || (e[u] != 0xffffffffu && e[u] * CPL + c < n);
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    const uint64_t t = a[w] + bias[w];           // field top bit: sum >= bound
                    all[w] &= live ? t : kTop;
                }
            }
        }
        uint64_t allw = all[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) allw &= all[w];
        if (__builtin_expect((allw & kTop) != kTop, 0)) {        // rare
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    if (!decltype(full)::value && !(e[u] != 0xffffffffu && e[u] * CPL + c < n)) continue;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const uint32_t sj = (uint32_t)(sums[u * CPL + c][j >> 2] >> (16 * (j & 3))) & 0xffffu;
                        if (sj < bq[j]) {
                            const ScanItem* ij = its + j;
                            emit_candidate(qstates + ij->query, hdr, out + (uint64_t)ij->query * cand_cap, cand_cap,
                                           it.labels, it.key_base, ij->order, it.dup_pos, it.dup_reps,
                                           it.pos0 + e[u] * CPL + c, sj);
                        }
                    }
                }
        }
    };
    using full_t = std::integral_constant<bool, true>;
    using part_t = std::integral_constant<bool, false>;
    const uint32_t tiles_full = (n / CPL) / kMQWG;           // tiles whose every lane holds CPL complete codes
    uint32_t t0 = bx;
    if constexpr (PIPE) {
        // software pipeline over the full tiles (the IVF second phase: runs of a few 10^4 codes that come from HBM, one or two
        // groups per partition): the next iteration's loads are issued before the current tiles are looked up, so a wave
        // keeps loads in flight while it works the LDS pipe.  Without it a wave holds 2 KB in flight for ~3 us and then
        // nothing while it computes: 28 waves x 2 KB per CU bound the launch below the LDS roof (C3 shape: 2.9 TB/s of reads)
        auto consume = [&](const u32x4_t (&v)[U], uint32_t tt) {
            row_t sums[U * CPL];
            uint64_t all[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) all[w] = kTop;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const row_t a = code_sums(d + c * DW);
                    sums[u * CPL + c] = a;
#pragma unroll
                    for (int w = 0; w < NW; ++w) all[w] &= a[w] + bias[w];
                }
            }
            uint64_t allw = all[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) allw &= all[w];
            if (__builtin_expect((allw & kTop) != kTop, 0)) {        // rare
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int c = 0; c < CPL; ++c)
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            const uint32_t sj = (uint32_t)(sums[u * CPL + c][j >> 2] >> (16 * (j & 3))) & 0xffffu;
                            if (sj < bq[j]) {
                                const ScanItem* ij = its + j;
                                emit_candidate(qstates + ij->query, hdr, out + (uint64_t)ij->query * cand_cap, cand_cap,
                                               it.labels, it.key_base, ij->order, it.dup_pos, it.dup_reps,
                                               it.pos0 + ((tt + u * G) * kMQWG + tid) * CPL + c, sj);
                            }
                        }
            }
        };
        if (t0 + (U - 1) * G < tiles_full) {
            u32x4_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[(t0 + u * G) * kMQWG + tid];
            for (;;) {
                const uint32_t tn = t0 + G * U;
                const bool more_full = tn + (U - 1) * G < tiles_full;
                u32x4_t nv[U];
                if (more_full) {
#pragma unroll
                    for (int u = 0; u < U; ++u) nv[u] = src[(tn + u * G) * kMQWG + tid];
                }
                consume(v, t0);
                t0 = tn;
                if (!more_full) break;
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = nv[u];
            }
        }
    } else {
        for (; t0 + (U - 1) * G < tiles_full; t0 += G * U) run(t0, full_t());
    }
    for (; t0 < ntiles; t0 += G * U) run(t0, part_t());
}

// Round 5: two more seat counts for the IVF second phase (scan_mq_body_x).  A group with 1-2 live seats reads 4-byte rows
// (one LDS cycle per 64 lookups instead of two, v_add_u32 adds at full issue rate), one with 5-6 reads 12 of a 16-byte row
// (ds_read_b64 + ds_read_b32: three LDS cycles instead of four, three dword adds instead of two 64-bit ones).  Rows are kept
// as dwords of two u16 fields; everything else — bounds, bias test, rare emit path, software pipeline — as in scan_mq_body.
template <int M, int U, int NQ>
__device__ __forceinline__ void scan_mq_body_x(const ScanItem* __restrict__ its, const ScanItem& it, int nq, uint32_t bx, uint32_t G,
                                               const int8_t* __restrict__ qtables, QueryState* __restrict__ qstates,
                                               CandHeader* __restrict__ hdr, Cand* __restrict__ out, uint32_t cand_cap, uint32_t R) {
    static_assert(NQ == 2 || NQ == 6, "seat counts of this body");
    constexpr int CS = M / 2, DW = M / 8, CPL = 16 / CS;
    constexpr int ROWB = NQ == 2 ? 4 : 16, ND = NQ / 2;          // bytes per row, dwords used per row
    constexpr int TAB = M * 256;
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32_t;
    typedef const __attribute__((address_space(3))) uint64_t* lds_u64_t;
    uint32_t* lbound = reinterpret_cast<uint32_t*>(smem + TAB);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int e = tid; e < M * 16; e += kMQWG) {
        uint32_t dwv[ND];
#pragma unroll
        for (int w = 0; w < ND; ++w) {
            uint32_t lo = 0, hi = 0;
            if (2 * w < nq && its[2 * w].n != 0) lo = (uint8_t)qtables[(uint64_t)its[2 * w].table * (M * 16) + e];
            if (2 * w + 1 < nq && its[2 * w + 1].n != 0) hi = (uint8_t)qtables[(uint64_t)its[2 * w + 1].table * (M * 16) + e];
            dwv[w] = lo | (hi << 16);
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(smem + e * ROWB);
#pragma unroll
        for (int w = 0; w < ND; ++w) dst[w] = dwv[w];
    }
    for (int j = (int)wave; j < NQ; j += kMQWG / 64) {
        uint32_t b = 0;
        if (j < nq && its[j].n != 0) b = prefix_bound_wave(qstates + its[j].query, its[j].order >> 16, R, lane);
        if (lane == 0) lbound[j] = b;
    }
    __syncthreads();
    uint32_t bq[NQ], bias[ND];
#pragma unroll
    for (int j = 0; j < NQ; ++j) bq[j] = __builtin_amdgcn_readfirstlane(lbound[j]);
#pragma unroll
    for (int w = 0; w < ND; ++w) bias[w] = (0x8000u - bq[2 * w]) | ((0x8000u - bq[2 * w + 1]) << 16);
    constexpr uint32_t kTop = 0x80008000u;

    typedef const __attribute__((address_space(1))) u32x4_t* gvec_t;
    const gvec_t src = (gvec_t)(uintptr_t)it.codes;
    const uint32_t n = it.n;
    const uint32_t nvec = (n + CPL - 1) / CPL;
    const uint32_t ntiles = (nvec + kMQWG - 1) / kMQWG;
    constexpr uint32_t nib_mask = NQ == 2 ? 0x3cu : 0xf0u;
    struct Row { uint32_t d[ND]; };
    auto code_sums = [&](const uint32_t* d) -> Row {
        Row a;
#pragma unroll
        for (int w = 0; w < ND; ++w) a.d[w] = 0;
#pragma unroll
        for (int w = 0; w < DW; ++w) {
            const uint32_t dl = NQ == 2 ? d[w] << 2 : d[w] << 4;
            const uint32_t dh = NQ == 2 ? d[w] >> 2 : d[w];
#define QADC_MQX_READ(addr)                                                                                    \
            {                                                                                                  \
                if (NQ == 2) {                                                                                 \
                    a.d[0] += *reinterpret_cast<lds_u32_t>(static_cast<uintptr_t>(addr));                      \
                } else {                                                                                       \
                    const uint64_t r01 = *reinterpret_cast<lds_u64_t>(static_cast<uintptr_t>(addr));           \
                    const uint32_t r2 = *reinterpret_cast<lds_u32_t>(static_cast<uintptr_t>((addr) + 8));      \
                    a.d[0] += (uint32_t)r01;                                                                   \
                    a.d[1] += (uint32_t)(r01 >> 32);                                                           \
                    a.d[ND - 1] += r2;                                                                         \
                }                                                                                              \
            }
#define QADC_MQX_BYTE(k)                                                                                       \
            {                                                                                                  \
                const int t0 = 2 * (4 * w + (k));                                                              \
                const uint32_t xl = byte_and<(k)>(dl, nib_mask), xh = byte_and<(k)>(dh, nib_mask);             \
                QADC_MQX_READ(xl + t0 * (16 * ROWB))                                                           \
                QADC_MQX_READ(xh + (t0 + 1) * (16 * ROWB))                                                     \
            }
            QADC_MQX_BYTE(0) QADC_MQX_BYTE(1) QADC_MQX_BYTE(2) QADC_MQX_BYTE(3)
#undef QADC_MQX_BYTE
#undef QADC_MQX_READ
        }
        return a;
    };
    // one iteration over U tiles: v = the tiles' vectors (zeros where not loaded), tt = first tile, FULL = every lane holds CPL codes
    auto consume = [&](const u32x4_t (&v)[U], uint32_t tt, auto full) {
        Row sums[U * CPL];
        uint32_t all = kTop;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            const uint32_t e = (tt + u * G) * kMQWG + tid;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const Row a = code_sums(d + c * DW);
                sums[u * CPL + c] = a;
                const bool live = decltype(full)::value || (tt + u * G < ntiles && e < nvec && e * CPL + c < n);
#pragma unroll
                for (int w = 0; w < ND; ++w) {
                    const uint32_t t = a.d[w] + bias[w];         // field top bit: sum >= bound
                    all &= live ? t : kTop;
                }
            }
        }
        if (__builtin_expect((all & kTop) != kTop, 0)) {         // rare
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const uint32_t e = (tt + u * G) * kMQWG + tid;
                    if (!decltype(full)::value && !(tt + u * G < ntiles && e < nvec && e * CPL + c < n)) continue;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const uint32_t sj = (sums[u * CPL + c].d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
                        if (sj < bq[j]) {
                            const ScanItem* ij = its + j;
                            emit_candidate(qstates + ij->query, hdr, out + (uint64_t)ij->query * cand_cap, cand_cap,
                                           it.labels, it.key_base, ij->order, it.dup_pos, it.dup_reps, it.pos0 + e * CPL + c, sj);
                        }
                    }
                }
        }
    };
    using full_t = std::integral_constant<bool, true>;
    using part_t = std::integral_constant<bool, false>;
    const uint32_t tiles_full = (n / CPL) / kMQWG;
    uint32_t t0 = bx;
    if (t0 + (U - 1) * G < tiles_full) {                         // software pipeline over the full tiles (see scan_mq_body)
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[(t0 + u * G) * kMQWG + tid];
        for (;;) {
            const uint32_t tn = t0 + G * U;
            const bool more_full = tn + (U - 1) * G < tiles_full;
            u32x4_t nv[U];
            if (more_full) {
#pragma unroll
                for (int u = 0; u < U; ++u) nv[u] = src[(tn + u * G) * kMQWG + tid];
            }
            consume(v, t0, full_t());
            t0 = tn;
            if (!more_full) break;
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = nv[u];
        }
    }
    for (; t0 < ntiles; t0 += G * U) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t t = t0 + u * G, e = t * kMQWG + tid;
            v[u] = u32x4_t{0, 0, 0, 0};
            if (t < ntiles && e < nvec) v[u] = src[e];
        }
        consume(v, t0, part_t());
    }
}

// Two builds of the kernel.  NARROW = false: the 8-seat body alone — the flat list's bound levels (every group of a flat
// batch is full but possibly the last).  NARROW = true: both bodies, a group whose seats 4..7 are empty taking the 4-seat
// one — the IVF second phase, whose remainder groups are mostly narrow.  They are separate kernels on purpose: with both
// bodies in it the register allocator lets the pair grow to 104 VGPRs unless 7 waves per SIMD are demanded, and demanded,
// it fits the 8-seat body into 66 with a schedule that costs the FLAT path 5-8 % (1B step 7.77 -> 8.17 ms, 125 M-code
// shard 1.12 -> 1.21 ms: measured against round 2's build on one box) for the 3 % the narrow form gives the IVF path.
template <int M, int U, bool NARROW>
__device__ __forceinline__ void scan_mq_kernel_body(const ScanItem* __restrict__ items, int nitems, const int8_t* __restrict__ qtables,
                                                    QueryState* __restrict__ qstates, CandHeader* __restrict__ hdr, Cand* __restrict__ out,
                                                    uint32_t cand_cap, uint32_t R, uint32_t ngroups) {
    // sibling-major decode over the query GROUPS (see scan_i8_kernel): groups that read the same tiles share an XCD
    const uint32_t G = gridDim.x / ngroups;
    uint32_t bx, grp;
    if ((G & 7u) == 0) {
        const uint32_t r = blockIdx.x >> 3;
        grp = r % ngroups;
        bx = (r / ngroups) * 8u + (blockIdx.x & 7u);
    } else {
        grp = blockIdx.x % ngroups;
        bx = blockIdx.x / ngroups;
    }
    const int first_item = (int)grp * kMQ;
    const int nq = min(kMQ, nitems - first_item);            // queries of this group (>= 1)
    const ScanItem* __restrict__ its = items + first_item;
    const ScanItem it = its[0];                              // codes / n / pos0 / labels / key_base / dup_*: shared
    if (it.n == 0) return;                                   // (device-planned launches are sized for the worst case: no such group)
    lds_base_is_zero();
    if (NARROW) {
        // seats are filled from 0 upward (a remainder group of the IVF second phase is mostly short); nlive = 1 + the highest
        // live seat, so that a form never drops a seat whatever the planner did
        int nlive = 0;
        for (int j = 0; j < nq; ++j) nlive = its[j].n != 0 ? j + 1 : nlive;
        if (nlive <= 2) {
            scan_mq_body_x<M, U, 2>(its, it, min(nq, 2), bx, G, qtables, qstates, hdr, out, cand_cap, R);
            return;
        }
        if (nlive <= 4) {
            scan_mq_body<M, U, 4, true>(its, it, min(nq, 4), bx, G, qtables, qstates, hdr, out, cand_cap, R);
            return;
        }
        if (nlive <= 6) {
            scan_mq_body_x<M, U, 6>(its, it, min(nq, 6), bx, G, qtables, qstates, hdr, out, cand_cap, R);
            return;
        }
    }
    scan_mq_body<M, U, 8, NARROW && true>(its, it, nq, bx, G, qtables, qstates, hdr, out, cand_cap, R);
}

template <int M, int U>
__global__ __launch_bounds__(kMQWG) void scan_i8_mq_kernel(
    const ScanItem* __restrict__ items, int nitems, const int8_t* __restrict__ qtables,
    QueryState* __restrict__ qstates, CandHeader* __restrict__ hdr, Cand* __restrict__ out, uint32_t cand_cap,
    uint32_t R, uint32_t ngroups) {
    scan_mq_kernel_body<M, U, false>(items, nitems, qtables, qstates, hdr, out, cand_cap, R, ngroups);
}

// (7 waves per SIMD asked for: each of the two bodies fits 68 VGPRs by itself; without the demand the occupancy the
// kernel's latency hiding was tuned for — 7 waves per SIMD — drops to 4)
template <int M, int U>
__global__ __launch_bounds__(kMQWG, 7) void scan_i8_mq_narrow_kernel(
    const ScanItem* __restrict__ items, int nitems, const int8_t* __restrict__ qtables,
    QueryState* __restrict__ qstates, CandHeader* __restrict__ hdr, Cand* __restrict__ out, uint32_t cand_cap,
    uint32_t R, uint32_t ngroups) {
    scan_mq_kernel_body<M, U, true>(items, nitems, qtables, qstates, hdr, out, cand_cap, R, ngroups);
}

template <int M>
static void launch_scan_mq_m(const ScanItem* d_items, int nitems, int wgs_per_group, const int8_t* d_qtables,
                             QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cand_cap, uint32_t R,
                             hipStream_t stream, int narrow) {
    const uint32_t ngroups = (uint32_t)(nitems + kMQ - 1) / kMQ;
    if (narrow)
        hipLaunchKernelGGL((scan_i8_mq_narrow_kernel<M, 2>), dim3(ngroups * (uint32_t)wgs_per_group), dim3(kMQWG), M * 256 + 64,
                           stream, d_items, nitems, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, ngroups);
    else
        hipLaunchKernelGGL((scan_i8_mq_kernel<M, 2>), dim3(ngroups * (uint32_t)wgs_per_group), dim3(kMQWG), M * 256 + 64,
                           stream, d_items, nitems, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, ngroups);
}

// Runs of the launch must all cover the same codes (same codes / n / pos0 / labels / key_base / dup_*): the
// planner guarantees it.  Groups of up to 8 consecutive runs share one pass.
void launch_scan_i8_mq(int M, const ScanItem* d_items, int nitems, int wgs_per_group, const int8_t* d_qtables,
                       QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cand_cap, uint32_t R,
                       hipStream_t stream, int narrow) {
    if (M == 16) launch_scan_mq_m<16>(d_items, nitems, wgs_per_group, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, stream, narrow);
    else         launch_scan_mq_m<32>(d_items, nitems, wgs_per_group, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, stream, narrow);
}

// ---------------------------------------------------------------------------------------------
// Small-run variant (IVF partitions, early bound levels): same pair-fused lookup, but the byte tables
// are NOT bank-replicated — 2 KiB (16x4) / 4 KiB (32x4) of LDS and a trivial build, 256-thread
// workgroups, so many workgroups per CU overlap their item/table/histogram fetch latencies.  A table
// spans 64 dwords = 2 per bank, so a lookup is at worst a 2-way bank conflict; that costs LDS cycles
// the streaming kernel cannot afford at 6 TB/s but a run of a few 10^4 codes is latency-bound anyway.
// ---------------------------------------------------------------------------------------------
template <int M, int SU>
__global__ __launch_bounds__(256) void scan_i8_small_kernel(const ScanItem* __restrict__ items,
                                                            const int8_t* __restrict__ qtables,
                                                            QueryState* __restrict__ qstates, CandHeader* __restrict__ hdr,
                                                            Cand* __restrict__ out, uint32_t cand_cap, uint32_t R) {
    constexpr int CS = M / 2, DW = M / 8, CPL = 16 / CS;
    __shared__ __attribute__((aligned(16))) unsigned char ptab[CS * 256];
    __shared__ __attribute__((aligned(16))) unsigned char tq[M * 16];
    __shared__ uint32_t lhist[128];
    __shared__ uint32_t lbound;
    const ScanItem it = items[blockIdx.y];
    QueryState* qs = qstates + it.query;
    out += (uint64_t)it.query * cand_cap;
    const uint32_t tid = threadIdx.x;
    if (tid < M * 4) reinterpret_cast<uint32_t*>(tq)[tid] =
        reinterpret_cast<const uint32_t*>(qtables + (uint64_t)it.table * (M * 16))[tid];
    __syncthreads();
    for (int e = tid; e < CS * 256; e += 256) {
        const int b = e >> 8, x = e & 255;
        ptab[e] = (unsigned char)(tq[(2 * b) * 16 + (x & 15)] + tq[(2 * b + 1) * 16 + (x >> 4)]);
    }
    const uint32_t bound = prefix_bound(qs, it.order >> 16, R, lhist, &lbound);   // syncs inside

    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gvec_t;
    const gvec_t src = (gvec_t)(uintptr_t)it.codes;
    const uint32_t n = it.n;
    const uint32_t nvec = (n + CPL - 1) / CPL;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t e0 = blockIdx.x * 256 + tid; e0 < nvec; e0 += SU * stride) {
        u32x4 v[SU];
        uint32_t e[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            e[u] = e0 + u * stride;
            v[u] = u32x4{0, 0, 0, 0};
            if (e[u] < nvec) v[u] = __builtin_nontemporal_load(src + e[u]);
            else e[u] = 0xffffffffu;
        }
        uint32_t cand[SU * CPL];
        uint32_t best = 127u;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                uint32_t sum = 0;
#pragma unroll
                for (int b = 0; b < CS; ++b) {
                    const uint32_t w = d[c * DW + (b >> 2)];
                    sum += ptab[b * 256 + ((w >> (8 * (b & 3))) & 0xffu)];
                }
                const bool live = e[u] != 0xffffffffu && e[u] * CPL + c < n;
                cand[u * CPL + c] = live ? min(sum, 127u) : 127u;
                best = min(best, cand[u * CPL + c]);
            }
        }
        if (__builtin_expect(best < bound, 0)) {
#pragma unroll
            for (int u = 0; u < SU; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (cand[u * CPL + c] < bound)
                        emit_candidate(qs, hdr, out, cand_cap, it.labels, it.key_base, it.order, it.dup_pos, it.dup_reps,
                                       it.pos0 + e[u] * CPL + c, cand[u * CPL + c]);
        }
    }
}

void launch_scan_i8_small(int M, const ScanItem* d_items, int nitems, int wgs_per_item, const int8_t* d_qtables,
                          QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cap_per_query, uint32_t R,
                          hipStream_t stream) {
    const dim3 grid(wgs_per_item, nitems), block(256);
    // 4 vectors (64 B) in flight per lane: these runs are latency-bound
    if (M == 16) hipLaunchKernelGGL((scan_i8_small_kernel<16, 4>), grid, block, 0, stream, d_items, d_qtables, d_qs, d_hdr, d_cands, cap_per_query, R);
    else         hipLaunchKernelGGL((scan_i8_small_kernel<32, 4>), grid, block, 0, stream, d_items, d_qtables, d_qs, d_hdr, d_cands, cap_per_query, R);
}

template <int M, bool NT, bool CHUNK, bool PROBE>
static void launch_scan_variant(dim3 grid, hipStream_t stream, const ScanItem* d_items, const int8_t* d_qtables,
                                QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cand_cap, uint32_t R,
                                uint32_t sib_items) {
    auto k = &scan_i8_kernel<M, NT, CHUNK, PROBE>;
    static std::atomic<uint64_t> done{0};
    ensure_dynamic_lds(reinterpret_cast<const void*>(k), ScanCfg<M>::LDS_BYTES, done);
    hipLaunchKernelGGL(k, grid, dim3(kWG), ScanCfg<M>::LDS_BYTES, stream, d_items, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, sib_items);
}

// variant bits: [2] NT (non-temporal loads)  [3] CHUNK  [4] PROBE (ceiling diagnostic)
// [6] sibling-major 1-D launch (every run of the launch covers the same codes; see the kernel's decode).  Other bits: ignored.
void launch_scan_i8(int M, int variant, const ScanItem* d_items, int nitems, int wgs_per_item,
                    const int8_t* d_qtables, QueryState* d_qs, CandHeader* d_hdr, Cand* d_cands, uint32_t cand_cap,
                    uint32_t R, hipStream_t stream) {
    const bool sib = (variant & 64) != 0;
    const dim3 grid = sib ? dim3((unsigned)wgs_per_item * (unsigned)nitems) : dim3(wgs_per_item, nitems);
    const uint32_t sib_items = sib ? (uint32_t)nitems : 0u;
#define QADC_ARGS grid, stream, d_items, d_qtables, d_qs, d_hdr, d_cands, cand_cap, R, sib_items
#define QADC_DISPATCH_P(MM, PR)                                             \
    switch ((variant >> 2) & 3) {                                           \
        case 0: launch_scan_variant<MM, false, false, PR>(QADC_ARGS); break; \
        case 1: launch_scan_variant<MM, true, false, PR>(QADC_ARGS); break;  \
        case 2: launch_scan_variant<MM, false, true, PR>(QADC_ARGS); break;  \
        default: launch_scan_variant<MM, true, true, PR>(QADC_ARGS); break;  \
    }
#define QADC_DISPATCH(MM)                                                   \
    if (variant & 16) { QADC_DISPATCH_P(MM, true) } else { QADC_DISPATCH_P(MM, false) }
    if (M == 16) { QADC_DISPATCH(16) } else { QADC_DISPATCH(32) }
#undef QADC_DISPATCH
#undef QADC_DISPATCH_P
#undef QADC_ARGS
}

// ---------------------------------------------------------------------------------------------
// Device-side ordering of the candidates: the host replay needs them in scan order
// (level, assign slot, position).  A query's region is already grouped by level — slots are handed
// out by an atomic counter and the levels are separate, stream-ordered launches — so each level's
// segment (<= R * growth entries, typically a few hundred) is sorted by ONE WAVE with a barrier-free
// bitonic network in its own LDS slice, all levels concurrently.  Segments above 512 entries fall
// back to a workgroup-wide bitonic sort of the whole region.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSegCap = 512;                        // entries one wave sorts: 4 KiB keys + 4 KiB payload per wave

__device__ __forceinline__ uint64_t cand_sort_key(const Cand& c, uint32_t idx) {
    const uint64_t ord = ((uint64_t)((c.order >> 16) & 15u) << 14) | (c.order & 0x3fffu);
    return (ord << 46) | ((uint64_t)c.pos << 14) | idx;
}

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Bitonic sort of 512 u64 keys held by one wave, 8 per lane, element e = r*64 + lane.  Exchanges at distance
// >= 64 are register swaps inside a lane, shorter ones are wave shuffles: no LDS round trips, no barriers.
__device__ __forceinline__ void wave_sort512(uint64_t (&v)[8], uint32_t lane) {
#pragma unroll
    for (uint32_t k = 2; k <= 512; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const uint32_t jr = j >> 6;
#pragma unroll
                for (uint32_t r = 0; r < 8; ++r) {
                    if ((r & jr) == 0) {
                        const bool asc = (((r << 6) | lane) & k) == 0;
                        const uint64_t a = v[r], b = v[r | jr];
                        const bool sw = (a > b) == asc;
                        v[r] = sw ? b : a;
                        v[r | jr] = sw ? a : b;
                    }
                }
            } else {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (uint32_t r = 0; r < 8; ++r) {
                    const bool asc = (((r << 6) | lane) & k) == 0;
                    const uint64_t a = v[r];
                    const uint64_t b = ((uint64_t)__shfl_xor((uint32_t)(a >> 32), j, 64) << 32) | __shfl_xor((uint32_t)a, j, 64);
                    const bool keep_min = lower == asc;
                    v[r] = keep_min ? (a < b ? a : b) : (a > b ? a : b);
                }
            }
        }
    }
}

__device__ __forceinline__ void write_query_out(QueryOut* __restrict__ o, const QueryState* qs, uint32_t off, uint32_t flags) {
    o->count = qs->count;
    o->reps = qs->reps;
    o->flags = flags;
    o->out_off = off;
    o->qmin = qs->qmin;
    o->qmax = qs->qmax;
}

__global__ __launch_bounds__(1024) void sort_cands_kernel(QueryState* __restrict__ qstates, const Cand* __restrict__ cands,
                                                          uint32_t cap, int nq, QueryOut* __restrict__ qout,
                                                          uint64_t* __restrict__ out_entries, uint32_t out_cap,
                                                          CandHeader* __restrict__ hdr,
                                                          uint64_t* __restrict__ dev_entries) {
    // all LDS in the dynamic region (keeps the 8-byte key array naturally aligned)
    uint64_t* lkey = reinterpret_cast<uint64_t*>(smem);
    uint32_t* scan = reinterpret_cast<uint32_t*>(smem + kSortCap * 8);   // [1024] + misc
    uint32_t& s_off = scan[1024];
    uint32_t* seg_cnt = scan + 1025;                         // [16]
    uint32_t* seg_off = scan + 1041;                         // [17]
    uint32_t* seg_reps = scan + 1058;                        // [16] then exclusive prefix
    uint32_t& s_fast = scan[1074];
    const int q = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    QueryState* qs = qstates + q;
    const uint32_t n = qs->count;
    const uint32_t limit = min(cap, kSortCap);
    if (n > limit) {                                             // host sorts this query's region
        if (tid == 0) write_query_out(qout + q, qs, 0, qs->flags);
        return;
    }
    // offset of this query in the compact output = candidates (+ replays) of the device-sorted queries before it
    uint32_t part = 0;
    for (int p = tid; p < q; p += 1024) {
        const uint32_t c = qstates[p].count;
        if (c <= limit) part += c + qstates[p].reps;
    }
    scan[tid] = part;
    // entries per level = sum of the level's value histogram (wave w <-> level w)
    {
        uint32_t c = qs->hist[wave * 128 + lane] + qs->hist[wave * 128 + 64 + lane];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) seg_cnt[wave] = c;
    }
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if (tid < d) scan[tid] += scan[tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        s_off = scan[0];
        uint32_t run = 0, fast = 1;
        for (int l = 0; l < kMaxLevels; ++l) {
            seg_off[l] = run;
            run += seg_cnt[l];
            if (seg_cnt[l] > kSegCap) fast = 0;
        }
        seg_off[kMaxLevels] = run;
        s_fast = (fast && run == n) ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t off = s_off;
    const Cand* __restrict__ region = cands + (uint64_t)q * cap;

    if (s_fast) {
        // ---- one wave per level segment, no workgroup barriers in the sort ----
        uint64_t* wkey = lkey + (size_t)wave * (2 * kSegCap);   // [kSegCap] sort keys
        uint64_t* wpay = wkey + kSegCap;                          // [kSegCap] payload: key | (val | replays << 8) << 32
        const uint32_t s0 = seg_off[wave], ns = seg_cnt[wave];
        // keys in registers (element e = r*64 + lane), payload in LDS indexed by the entry's position in the segment
        uint64_t v[8];
#pragma unroll
        for (uint32_t r = 0; r < 8; ++r) {
            const uint32_t i = r * 64 + lane;
            uint64_t k = ~0ull;
            if (i < ns) {
                const Cand c = region[s0 + i];
                k = cand_sort_key(c, i);
                wpay[i] = (uint64_t)c.key | ((uint64_t)(c.val | (((c.order >> 20) & 15u) << 8)) << 32) |
                          ((uint64_t)(c.order & 0x3fffu) << 48);   // key | value | replays << 40 | assign slot << 48
            }
            v[r] = k;
        }
        if (ns > 1) wave_sort512(v, lane);
#pragma unroll
        for (uint32_t r = 0; r < 8; ++r) wkey[r * 64 + lane] = v[r];
        wave_lds_sync();
        const uint32_t n2 = 512;
        // padding-lane replays: lane owns sorted entries [lane*chunk, (lane+1)*chunk) of the segment
        const uint32_t chunk = n2 / 64;
        const uint32_t b0 = min(ns, lane * chunk), b1 = min(ns, (lane + 1) * chunk);
        uint32_t myreps = 0;
        for (uint32_t i = b0; i < b1; ++i) myreps += (uint32_t)(wpay[wkey[i] & 0x3fffu] >> 40) & 15u;
        uint32_t incl = myreps;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) seg_reps[wave] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int l = 0; l < kMaxLevels; ++l) {
                const uint32_t r = seg_reps[l];
                seg_reps[l] = run;
                run += r;
            }
        }
        __syncthreads();
        uint32_t w = off + s0 + seg_reps[wave] + b0 + (incl - myreps);
        for (uint32_t i = b0; i < b1; ++i) {
            const uint64_t pay = wpay[wkey[i] & 0x3fffu];
            const uint32_t reps = 1u + ((uint32_t)(pay >> 40) & 15u);
            for (uint32_t r = 0; r < reps; ++r, ++w) {
                // entry = key | value << 32 | assign slot << 40
                if (w < out_cap) {
                    const uint64_t en = (pay & 0xffffffffffull) | ((pay >> 48) << 40);
                    out_entries[w] = en;
                    if (dev_entries) dev_entries[w] = en;        // copy in device memory for replay_heap_kernel
                } else {
                    atomicAdd(&hdr->out_overflow, 1u);
                }
            }
        }
        if (tid == 0) {
            write_query_out(qout + q, qs, off, qs->flags | 4u);
            qs->out_off = off;
            qs->flags |= 4u;
        }
        return;
    }

    // ---- fallback: workgroup-wide bitonic sort of the whole region ----
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += 1024) lkey[i] = i < n ? cand_sort_key(region[i], i) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n2; i += 1024) {
                const uint32_t p = i ^ j;
                if (p > i) {
                    const uint64_t a = lkey[i], b = lkey[p];
                    if ((a > b) == ((i & k) == 0)) { lkey[i] = b; lkey[p] = a; }
                }
            }
            __syncthreads();
        }
    // expand padding-lane replays while writing: thread t owns sorted entries [t*chunk, (t+1)*chunk)
    const uint32_t chunk = (n2 + 1023) / 1024;
    const uint32_t b0 = min(n, tid * chunk), b1 = min(n, (tid + 1) * chunk);
    uint32_t myreps = 0;
    for (uint32_t i = b0; i < b1; ++i) myreps += (region[lkey[i] & 0x3fffu].order >> 20) & 15u;
    scan[tid] = myreps;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t o = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += o;
        __syncthreads();
    }
    uint32_t w = off + b0 + (scan[tid] - myreps);
    for (uint32_t i = b0; i < b1; ++i) {
        const Cand c = region[lkey[i] & 0x3fffu];
        const uint32_t reps = 1u + ((c.order >> 20) & 15u);
        for (uint32_t r = 0; r < reps; ++r, ++w) {
            if (w < out_cap) {
                const uint64_t en = (uint64_t)c.key | ((uint64_t)(c.val & 0xffu) << 32) | ((uint64_t)(c.order & 0x3fffu) << 40);
                out_entries[w] = en;
                if (dev_entries) dev_entries[w] = en;
            } else {
                atomicAdd(&hdr->out_overflow, 1u);
            }
        }
    }
    if (tid == 0) {
        write_query_out(qout + q, qs, off, qs->flags | 4u);
        qs->out_off = off;
        qs->flags |= 4u;
    }
}

// ---------------------------------------------------------------------------------------------
// The reference's heap on the device (kv_binheap<unsigned,int8_t>::push, binheap.hpp:75-116): one wave per
// query replays the query's ordered candidate stream through a binary max-heap of capacity R kept in LDS —
// sentinel (0,127) first (db_query_4.cpp:276), then every entry in scan order.  The pushes of one query are
// inherently sequential (lane 0 executes them; the other lanes stage the stream through LDS 64 entries at a
// time and write the result out), but thousands of queries replay side by side, which is what an IVF batch
// needs: the host then only copies R entries per query instead of assembling and replaying the streams.
// Same arrays as the host replay (qadc_heap.hpp): appended and bubbled up past strictly smaller parents while
// there is room; afterwards accepted only if strictly below the root, sinking with the left child preferred on
// ties and stopping at a child <= the value.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void replay_heap_kernel(const QueryState* __restrict__ qstates,
                                                         const uint64_t* __restrict__ dev_entries, uint32_t out_cap,
                                                         uint32_t R, uint64_t* __restrict__ heaps,
                                                         uint32_t* __restrict__ heap_sizes) {
    uint64_t* hv = reinterpret_cast<uint64_t*>(smem);            // [R]  key | value << 32
    uint64_t* buf = hv + R;                                      // [64] staged stream entries
    const int q = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const QueryState* qs = qstates + q;
    const uint32_t flags = qs->flags;
    if (!(flags & 4u) || (uint64_t)qs->out_off + qs->count + qs->reps > out_cap) {   // not ordered on the device: host replays
        if (lane == 0) heap_sizes[q] = 0xffffffffu;
        return;
    }
    if (flags & 1u) {                                            // qmax too high: the reference exits, no result
        if (lane == 0) heap_sizes[q] = 0;
        return;
    }
    const uint32_t n = qs->count + qs->reps;
    const uint64_t* __restrict__ src = dev_entries + qs->out_off;
    uint32_t size = 0;                                           // meaningful on lane 0
    auto val_of = [](uint64_t e) { return (int32_t)((e >> 32) & 0xffu); };
    auto push = [&](uint64_t e) {
        const int32_t value = val_of(e);
        if (size != R) {
            uint32_t i = size++;
            while (i != 0) {
                const uint32_t parent = (i - 1) / 2;
                const uint64_t pe = hv[parent];
                if (!(value > val_of(pe))) break;
                hv[i] = pe;
                i = parent;
            }
            hv[i] = e;
            return;
        }
        if (!(value < val_of(hv[0]))) return;
        uint32_t i = 0;
        for (;;) {
            const uint32_t l = 2 * i + 1;
            if (l >= size) break;
            uint64_t ce = hv[l];
            uint32_t c = l;
            if (l + 1 < size) {
                const uint64_t re = hv[l + 1];
                if (val_of(re) > val_of(ce)) { ce = re; c = l + 1; }
            }
            if (val_of(ce) <= value) break;
            hv[i] = ce;
            i = c;
        }
        hv[i] = e;
    };
    if (lane == 0) push((uint64_t)127 << 32);                    // the sentinel: key 0, value 127
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t m = min(64u, n - base);
        if (lane < m) buf[lane] = src[base + lane] & 0xffffffffffull;   // key | value (assign slot dropped)
        wave_lds_sync();
        if (lane == 0)
            for (uint32_t j = 0; j < m; ++j) push(buf[j]);
        wave_lds_sync();
    }
    size = __shfl(size, 0, 64);
    for (uint32_t i = lane; i < size; i += 64) heaps[(uint64_t)q * R + i] = hv[i];
    if (lane == 0) heap_sizes[q] = size;
}

void launch_replay_heap(const QueryState* d_qs, const uint64_t* d_entries, uint32_t out_cap, int nq, uint32_t R,
                        uint64_t* d_heaps, uint32_t* d_heap_sizes, hipStream_t stream) {
    hipLaunchKernelGGL(replay_heap_kernel, dim3(nq), dim3(64), (size_t)(R + 64) * 8, stream, d_qs, d_entries, out_cap, R, d_heaps,
                       d_heap_sizes);
}

void launch_sort_cands(QueryState* d_qs, const Cand* d_cands, uint32_t cap_per_query, int nq, QueryOut* d_qout,
                       uint64_t* d_entries, uint32_t out_cap, CandHeader* d_hdr, hipStream_t stream,
                       uint64_t* d_dev_entries) {
    static std::atomic<uint64_t> done{0};
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sort_cands_kernel), kSortCap * 8 + 4352, done);
    hipLaunchKernelGGL(sort_cands_kernel, dim3(nq), dim3(1024), kSortCap * 8 + 4352, stream, d_qs, d_cands, cap_per_query, nq,
                       d_qout, d_entries, out_cap, d_hdr, d_dev_entries);
}

// All candidate values (diagnostic; used by parity tests and checksums at full size).
template <int M>
__global__ __launch_bounds__(kWG) void candidates_i8_kernel(const uint8_t* __restrict__ codes, uint64_t n,
                                                            const int8_t* __restrict__ qtable, int8_t* __restrict__ out) {
    using C = ScanCfg<M>;
    lds_base_is_zero();
    build_pair_tables<M>(qtable);
    __syncthreads();
    const uint32_t tid = threadIdx.x;
    const uint32_t lane_lo = (tid & 31u) * 4u, lane_hi = lane_lo | 0x10000u;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(codes);
    const uint64_t nvec = (n + C::CPL - 1) / C::CPL;
    for (uint64_t e = (uint64_t)blockIdx.x * kWG + tid; e < nvec; e += (uint64_t)gridDim.x * kWG) {
        const uint4 v = src[e];
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < C::CPL; ++c) {
            const uint64_t idx = e * C::CPL + c;
            if (idx < n) out[idx] = (int8_t)min(pair_sum<M>(d + c * C::DW, lane_lo, lane_hi), 127u);
        }
    }
}

void launch_candidates_i8(int M, const uint8_t* d_codes, uint64_t n, const int8_t* d_qtable, int8_t* d_out,
                          hipStream_t stream) {
    const uint64_t nvec = (n + (M == 16 ? 2 : 1) - 1) / (M == 16 ? 2 : 1);
    const int grid = (int)std::min<uint64_t>((nvec + kWG - 1) / kWG, 512);
    if (M == 16) {
        static std::atomic<uint64_t> done{0};
        ensure_dynamic_lds(reinterpret_cast<const void*>(&candidates_i8_kernel<16>), ScanCfg<16>::LDS_BYTES, done);
        hipLaunchKernelGGL(candidates_i8_kernel<16>, dim3(grid), dim3(kWG), ScanCfg<16>::LDS_BYTES, stream, d_codes, n,
                           d_qtable, d_out);
    } else {
        static std::atomic<uint64_t> done{0};
        ensure_dynamic_lds(reinterpret_cast<const void*>(&candidates_i8_kernel<32>), ScanCfg<32>::LDS_BYTES, done);
        hipLaunchKernelGGL(candidates_i8_kernel<32>, dim3(grid), dim3(kWG), ScanCfg<32>::LDS_BYTES, stream, d_codes, n,
                           d_qtable, d_out);
    }
}

// ---------------------------------------------------------------------------------------------
// float pre-scan of the "starts" (scan_4<M>, query_common.hpp:59-90): the same IEEE adds in the grouping
// sum_mode names (qadc_float_sum.h): 1 = the reference as compiled with its -ffast-math, 0 = source order.
// The 16 lanes-distinct entries of one table sit in 16 consecutive LDS dwords, so every lookup
// instruction (all lanes in the same table) is bank-conflict-free without replication.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

template <int M, int SM>
__global__ __launch_bounds__(256) void start_scan_f32_kernel(const StartItem* __restrict__ items,
                                                             const float* __restrict__ ftables, float* __restrict__ fc,
                                                             uint64_t fc_stride, const uint32_t* __restrict__ fc_init,
                                                             QueryState* __restrict__ qstates) {
    constexpr int kStage = 1024;                             // survivors staged in LDS per workgroup
    __shared__ float tab[M * 16];
    __shared__ float stage[kStage];
    __shared__ uint32_t stage_n, stage_base;
    __shared__ uint32_t red[8];
    const StartItem it = items[blockIdx.y];
    const float* __restrict__ ft = ftables + (uint64_t)it.table * (M * 16);
    if (threadIdx.x == 0) stage_n = 0;
    for (int i = threadIdx.x; i < M * 16; i += 256) tab[i] = ft[i];
    __syncthreads();
    float* __restrict__ dst = fc + (uint64_t)it.query * fc_stride;
    QueryState* qs = qstates + it.query;
    const float thr = it.filter ? qs->qmax : 0.0f;           // R-th smallest of the sample (phase A)
    const uint32_t base_n = fc_init[2 * it.query], cap = fc_init[2 * it.query + 1];   // survivors go behind the sample
    constexpr int DW = M / 8;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    const uint32_t stride = gridDim.x * 256;
    // two codes per lane and iteration: two independent float sums in flight
    for (uint32_t i0 = blockIdx.x * 256 + threadIdx.x; i0 < it.n; i0 += 2 * stride) {
        uint32_t d[2][DW];
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t i = i0 + u * stride;
            live[u] = i < it.n;
#pragma unroll
            for (int w = 0; w < DW; ++w) d[u][w] = 0;
            if (live[u]) {
                // explicit global address space: a pointer read from the item struct is generic otherwise
                if constexpr (M == 16) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 v = ((const __attribute__((address_space(1))) u32x2*)(uintptr_t)it.codes)[i];
                    d[u][0] = v.x; d[u][1] = v.y;
                } else {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = ((const __attribute__((address_space(1))) u32x4*)(uintptr_t)it.codes)[i];
                    d[u][0] = v.x; d[u][1] = v.y; d[u][2] = v.z; d[u][3] = v.w;
                }
            }
        }
        // scan_4<M>'s sum (query_common.hpp:72-80) in the grouping SM names
        float cand[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v[M];
#pragma unroll
            for (int b = 0; b < M / 2; ++b) {
                const uint32_t byte = (d[u][b >> 2] >> (8 * (b & 3))) & 0xffu;
                v[2 * b] = tab[(2 * b) * 16 + (byte & 15u)];
                v[2 * b + 1] = tab[(2 * b + 1) * 16 + (byte >> 4)];
            }
            cand[u] = adc_sum_code<M>(v, SM, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!live[u]) continue;
            if (it.filter) {
                // a value above the sample's R-th smallest cannot be the R-th smallest of the whole set
                if (!(cand[u] <= thr)) continue;
                const uint32_t ls = atomicAdd(&stage_n, 1u);
                if (ls < kStage) {
                    stage[ls] = cand[u];
                } else {                                     // staging full (rare): straight to the query's buffer
                    const uint32_t slot = base_n + atomicAdd(&qs->fc_n, 1u);
                    if (slot >= cap) { atomicOr(&qs->flags, 8u); continue; }
                    dst[slot] = cand[u];
                }
            } else {
                dst[it.out_off + i0 + u * stride] = cand[u];
            }
            const uint32_t k = fkey(cand[u]);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
    }
    // key range of the query's pre-scan values: lets the radix select skip the constant leading bits
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, d, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, d, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = kmin; red[4 + (threadIdx.x >> 6)] = kmax; }
    __syncthreads();
    const uint32_t ns = min(stage_n, (uint32_t)kStage);
    if (threadIdx.x == 0) {                                  // one atomic set per workgroup
        kmin = min(min(red[0], red[1]), min(red[2], red[3]));
        kmax = max(max(red[4], red[5]), max(red[6], red[7]));
        if (kmin <= kmax) {
            atomicMax(&qs->sel_nmin, ~kmin);
            atomicMax(&qs->sel_max, kmax);
        }
        if (ns) stage_base = base_n + atomicAdd(&qs->fc_n, ns);
    }
    __syncthreads();
    if (ns) {
        const uint32_t base = stage_base;
        for (uint32_t j = threadIdx.x; j < ns; j += 256) {
            if (base + j < cap) dst[base + j] = stage[j];
            else atomicOr(&qs->flags, 8u);
        }
    }
}

// Multi-query float pre-scan: the same sums for up to 8 queries in ONE pass over the starts (the float twin of
// scan_i8_mq_kernel).  LDS row (t, half, x) = { T_q[t][x] : q = 4*half .. 4*half+3 } = 16 bytes at
// t*512 + half*256 + x*16: two ds_read_b128 per code nibble, each conflict-free without replication (the 16 rows
// of a (t, half) block tile the 64 banks once; equal rows broadcast).  Every query's sum is grouped exactly as in
// the one-query kernel (SM: qadc_float_sum.h) — v_pk_add_f32 adds two queries per instruction with the rounding of
// v_add_f32 — so the values are bit-identical to that kernel's.  The as-compiled grouping is evaluated in a
// streaming order (pair sums as soon as both operands are loaded) so that at most six 8-query rows are live.
// Items of a group share codes / n / out_off / filter and differ in table / query.
typedef float fq_f32x4 __attribute__((ext_vector_type(4)));
struct FQ8 {                                                 // one value per query of the group
    fq_f32x4 lo, hi;
};
__device__ __forceinline__ FQ8 operator+(const FQ8& a, const FQ8& b) { return FQ8{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ FQ8 fq_row(uint32_t off) {
    typedef const __attribute__((address_space(3))) fq_f32x4* lds_frow_t;
    const fq_f32x4 lo = *reinterpret_cast<lds_frow_t>(static_cast<uintptr_t>(off));
    const fq_f32x4 hi = *reinterpret_cast<lds_frow_t>(static_cast<uintptr_t>(off + 256));
    return FQ8{lo, hi};
}
template <int M, int SM>
__global__ __launch_bounds__(256) void start_scan_mq_kernel(const StartItem* __restrict__ items, int nitems,
                                                            const float* __restrict__ ftables, float* __restrict__ fc,
                                                            uint64_t fc_stride, const uint32_t* __restrict__ fc_init,
                                                            QueryState* __restrict__ qstates) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int kStageQ = 512;                             // survivors staged in LDS per query and workgroup
    constexpr int TAB = M * 512;
    float* stage = reinterpret_cast<float*>(smem + TAB);                 // [8][kStageQ]
    uint32_t* stage_n = reinterpret_cast<uint32_t*>(smem + TAB + kMQ * kStageQ * 4);   // [8]
    uint32_t* stage_base = stage_n + kMQ;                                              // [8]
    float* red = reinterpret_cast<float*>(stage_base + kMQ);                            // [2][8][4 waves]
    const int first_item = (int)blockIdx.y * kMQ;
    const int nq = min(kMQ, nitems - first_item);
    const StartItem* __restrict__ its = items + first_item;
    const StartItem it = its[0];
    const uint32_t tid = threadIdx.x;
    lds_base_is_zero();
    if (tid < kMQ) stage_n[tid] = 0;
    for (int e = tid; e < M * 16; e += 256) {                // e = t*16 + x
        float row[kMQ];
#pragma unroll
        for (int j = 0; j < kMQ; ++j) row[j] = j < nq ? ftables[(uint64_t)its[j].table * (M * 16) + e] : 0.0f;
        const int t = e >> 4, x = e & 15;
        *reinterpret_cast<f32x4*>(smem + t * 512 + x * 16) = f32x4{row[0], row[1], row[2], row[3]};
        *reinterpret_cast<f32x4*>(smem + t * 512 + 256 + x * 16) = f32x4{row[4], row[5], row[6], row[7]};
    }
    __syncthreads();
    // per query: destination, threshold (phase B), sample size and capacity; absent queries never store
    float thr[kMQ];
#pragma unroll
    for (int j = 0; j < kMQ; ++j) thr[j] = (j < nq && it.filter) ? qstates[its[j].query].qmax : 0.0f;
    constexpr int DW = M / 8;
    const uint32_t nib_mask = 0xf0u;
    float vmin[kMQ], vmax[kMQ];
#pragma unroll
    for (int j = 0; j < kMQ; ++j) { vmin[j] = FLT_MAX; vmax[j] = -FLT_MAX; }
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + tid; i < it.n; i += stride) {
        uint32_t d[DW];
        if constexpr (M == 16) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = ((const __attribute__((address_space(1))) u32x2*)(uintptr_t)it.codes)[i];
            d[0] = v.x; d[1] = v.y;
        } else {
            const u32x4_t v = ((const __attribute__((address_space(1))) u32x4_t*)(uintptr_t)it.codes)[i];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        // scan_4<M>'s sum (query_common.hpp:72-80) for the 8 queries at once, in the grouping SM names
#define QADC_FQ_L(w, k) fq_row(byte_and<(k)>(d[w] << 4, nib_mask) + (2 * (4 * (w) + (k))) * 512)
#define QADC_FQ_H(w, k) fq_row(byte_and<(k)>(d[w], nib_mask) + (2 * (4 * (w) + (k)) + 1) * 512)
        FQ8 s8;
        if constexpr (SM == 0) {
            s8 = FQ8{f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#define QADC_FQ_BYTE(w, k) s8 = s8 + QADC_FQ_L(w, k); s8 = s8 + QADC_FQ_H(w, k);
#define QADC_FQ_WORD(w) QADC_FQ_BYTE(w, 0) QADC_FQ_BYTE(w, 1) QADC_FQ_BYTE(w, 2) QADC_FQ_BYTE(w, 3)
            QADC_FQ_WORD(0) QADC_FQ_WORD(1)
            if constexpr (M == 32) { QADC_FQ_WORD(2) QADC_FQ_WORD(3) }
#undef QADC_FQ_WORD
#undef QADC_FQ_BYTE
        } else {
            // A = (H2+L3)+(H3+L4), B = (H0+L1)+(H1+L2), C = (H5+L6)+(H4+L5), D = (H6+L7)+(H7+L0); s = ((A+B)+C)+D
            const FQ8 l0 = QADC_FQ_L(0, 0);
            FQ8 hp = QADC_FQ_H(0, 0);
            const FQ8 p0 = hp + QADC_FQ_L(0, 1); hp = QADC_FQ_H(0, 1);
            const FQ8 p1 = hp + QADC_FQ_L(0, 2); hp = QADC_FQ_H(0, 2);
            const FQ8 qb = p0 + p1;
            const FQ8 p2 = hp + QADC_FQ_L(0, 3); hp = QADC_FQ_H(0, 3);
            const FQ8 p3 = hp + QADC_FQ_L(1, 0); hp = QADC_FQ_H(1, 0);
            s8 = (p2 + p3) + qb;
            const FQ8 p4 = hp + QADC_FQ_L(1, 1); hp = QADC_FQ_H(1, 1);
            const FQ8 p5 = hp + QADC_FQ_L(1, 2); hp = QADC_FQ_H(1, 2);
            s8 = s8 + (p5 + p4);
            const FQ8 p6 = hp + QADC_FQ_L(1, 3); hp = QADC_FQ_H(1, 3);
            const FQ8 p7 = hp + l0;
            s8 = s8 + (p6 + p7);
            if constexpr (M == 32) {
                // s = s + ((L_{b+1}+H_{b+1}) + (L_b+H_b)), b = 8, 10, 12, 14
#define QADC_FQ_PAIR(w, k) s8 = s8 + ((QADC_FQ_L(w, (k) + 1) + QADC_FQ_H(w, (k) + 1)) + (QADC_FQ_L(w, k) + QADC_FQ_H(w, k)));
                QADC_FQ_PAIR(2, 0) QADC_FQ_PAIR(2, 2) QADC_FQ_PAIR(3, 0) QADC_FQ_PAIR(3, 2)
#undef QADC_FQ_PAIR
            }
        }
#undef QADC_FQ_L
#undef QADC_FQ_H
        const float cand[kMQ] = {s8.lo.x, s8.lo.y, s8.lo.z, s8.lo.w, s8.hi.x, s8.hi.y, s8.hi.z, s8.hi.w};
        if (!it.filter) {
#pragma unroll
            for (int j = 0; j < kMQ; ++j) {
                if (j >= nq) continue;
                fc[(uint64_t)its[j].query * fc_stride + it.out_off + i] = cand[j];
                vmin[j] = fminf(vmin[j], cand[j]);
                vmax[j] = fmaxf(vmax[j], cand[j]);
            }
        } else {
            bool any = false;
#pragma unroll
            for (int j = 0; j < kMQ; ++j) any |= (j < nq) && (cand[j] <= thr[j]);
            if (__builtin_expect(any, 0)) {
                // a value above the sample's R-th smallest cannot be the R-th smallest of the whole set
#pragma unroll
                for (int j = 0; j < kMQ; ++j) {
                    if (j >= nq || !(cand[j] <= thr[j])) continue;
                    vmin[j] = fminf(vmin[j], cand[j]);
                    vmax[j] = fmaxf(vmax[j], cand[j]);
                    const uint32_t ls = atomicAdd(&stage_n[j], 1u);
                    if (ls < (uint32_t)kStageQ) {
                        stage[j * kStageQ + ls] = cand[j];
                    } else {                                 // staging full (rare): straight to the query's buffer
                        QueryState* qs = qstates + its[j].query;
                        const uint32_t base_n = fc_init[2 * its[j].query], cap = fc_init[2 * its[j].query + 1];
                        const uint32_t slot = base_n + atomicAdd(&qs->fc_n, 1u);
                        if (slot >= cap) { atomicOr(&qs->flags, 8u); continue; }
                        fc[(uint64_t)its[j].query * fc_stride + slot] = cand[j];
                    }
                }
            }
        }
    }
    // key range of each query's values (float order == key order; a sum accumulated from +0 is never -0)
#pragma unroll
    for (int j = 0; j < kMQ; ++j) {
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            vmin[j] = fminf(vmin[j], __shfl_xor(vmin[j], dd, 64));
            vmax[j] = fmaxf(vmax[j], __shfl_xor(vmax[j], dd, 64));
        }
        if ((tid & 63) == 0) { red[j * 4 + (tid >> 6)] = vmin[j]; red[32 + j * 4 + (tid >> 6)] = vmax[j]; }
    }
    __syncthreads();
    if (tid < (uint32_t)nq) {                                // one atomic set per workgroup and query
        const int j = (int)tid;
        const float mn = fminf(fminf(red[j * 4], red[j * 4 + 1]), fminf(red[j * 4 + 2], red[j * 4 + 3]));
        const float mx = fmaxf(fmaxf(red[32 + j * 4], red[32 + j * 4 + 1]), fmaxf(red[32 + j * 4 + 2], red[32 + j * 4 + 3]));
        QueryState* qs = qstates + its[j].query;
        if (mn <= mx) {
            atomicMax(&qs->sel_nmin, ~fkey(mn));
            atomicMax(&qs->sel_max, fkey(mx));
        }
        const uint32_t ns = min(stage_n[j], (uint32_t)kStageQ);
        if (ns) stage_base[j] = fc_init[2 * its[j].query] + atomicAdd(&qs->fc_n, ns);
    }
    __syncthreads();
    for (int j = 0; j < nq; ++j) {
        const uint32_t ns = min(stage_n[j], (uint32_t)kStageQ);
        if (!ns) continue;
        QueryState* qs = qstates + its[j].query;
        const uint32_t base = stage_base[j], cap = fc_init[2 * its[j].query + 1];
        for (uint32_t k = tid; k < ns; k += 256) {
            if (base + k < cap) fc[(uint64_t)its[j].query * fc_stride + base + k] = stage[j * kStageQ + k];
            else atomicOr(&qs->flags, 8u);
        }
    }
}

// groups of up to 8 consecutive items share codes / n / out_off / filter (the planner guarantees it)
void launch_start_scan_mq(int M, int sum_mode, const StartItem* d_items, int nitems, int wgs_per_group, const float* d_ftables, float* d_fc,
                          uint64_t fc_stride, const uint32_t* d_fc_init, QueryState* d_qs, hipStream_t stream) {
    const dim3 grid(wgs_per_group, (nitems + kMQ - 1) / kMQ), block(256);
    const size_t lds = (size_t)M * 512 + kMQ * 512 * 4 + 2 * kMQ * 4 + 64 * 4;
#define QADC_SSM(MM, SS) hipLaunchKernelGGL((start_scan_mq_kernel<MM, SS>), grid, block, lds, stream, d_items, nitems, d_ftables, d_fc, fc_stride, d_fc_init, d_qs)
    if (M == 16) { if (sum_mode) QADC_SSM(16, 1); else QADC_SSM(16, 0); }
    else         { if (sum_mode) QADC_SSM(32, 1); else QADC_SSM(32, 0); }
#undef QADC_SSM
}

// Exact float-ADC nearest code of a partition (smallest distance, lowest position on ties):
// the ground truth Recall@R is measured against when float ground truth over the raw vectors is
// not available.  Same sum as scan_4<M>, in the grouping sum_mode names (qadc_float_sum.h).
template <int M>
__global__ __launch_bounds__(256) void float_top1_kernel(const uint8_t* __restrict__ codes, uint32_t n,
                                                         const float* __restrict__ ftable, float* __restrict__ out_val,
                                                         uint32_t* __restrict__ out_pos, int sum_mode) {
    __shared__ float tab[M * 16];
    __shared__ float rv[256];
    __shared__ uint32_t rp[256];
    for (int i = threadIdx.x; i < M * 16; i += 256) tab[i] = ftable[i];
    __syncthreads();
    constexpr int DW = M / 8;
    float best = FLT_MAX;
    uint32_t bpos = 0xffffffffu;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint32_t d[DW];
        if constexpr (M == 16) {
            const uint2 v = reinterpret_cast<const uint2*>(codes)[i];
            d[0] = v.x; d[1] = v.y;
        } else {
            const uint4 v = reinterpret_cast<const uint4*>(codes)[i];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        float v[M];
#pragma unroll
        for (int b = 0; b < M / 2; ++b) {
            const uint32_t byte = (d[b >> 2] >> (8 * (b & 3))) & 0xffu;
            v[2 * b] = tab[(2 * b) * 16 + (byte & 15u)];
            v[2 * b + 1] = tab[(2 * b + 1) * 16 + (byte >> 4)];
        }
        const float cand = adc_sum_code<M>(v, sum_mode, 0.0f);
        if (cand < best) { best = cand; bpos = i; }   // i ascends per thread: first minimum kept
    }
    rv[threadIdx.x] = best;
    rp[threadIdx.x] = bpos;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s) {
            const float ov = rv[threadIdx.x + s];
            const uint32_t op = rp[threadIdx.x + s];
            if (ov < rv[threadIdx.x] || (ov == rv[threadIdx.x] && op < rp[threadIdx.x])) {
                rv[threadIdx.x] = ov;
                rp[threadIdx.x] = op;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_val[blockIdx.x] = rv[0]; out_pos[blockIdx.x] = rp[0]; }
}

void launch_float_top1(int M, int sum_mode, const uint8_t* d_codes, uint32_t n, const float* d_ftable, float* d_val, uint32_t* d_pos,
                       int blocks, hipStream_t stream) {
    if (M == 16) hipLaunchKernelGGL(float_top1_kernel<16>, dim3(blocks), dim3(256), 0, stream, d_codes, n, d_ftable, d_val, d_pos, sum_mode);
    else         hipLaunchKernelGGL(float_top1_kernel<32>, dim3(blocks), dim3(256), 0, stream, d_codes, n, d_ftable, d_val, d_pos, sum_mode);
}

void launch_start_scan_f32(int M, int sum_mode, const StartItem* d_items, int nitems, int wgs_per_item, const float* d_ftables,
                           float* d_fc, uint64_t fc_stride, const uint32_t* d_fc_init, QueryState* d_qs, hipStream_t stream) {
    const dim3 grid(wgs_per_item, nitems), block(256);
#define QADC_SSF(MM, SS) hipLaunchKernelGGL((start_scan_f32_kernel<MM, SS>), grid, block, 0, stream, d_items, d_ftables, d_fc, fc_stride, d_fc_init, d_qs)
    if (M == 16) { if (sum_mode) QADC_SSF(16, 1); else QADC_SSF(16, 0); }
    else         { if (sum_mode) QADC_SSF(32, 1); else QADC_SSF(32, 0); }
#undef QADC_SSF
}

// ---------------------------------------------------------------------------------------------
// R-th smallest float per query (= tmp_bh.max() after query_scan_start, db_query_4.cpp:259):
// 4-pass MSD radix select on the order-preserving u32 image of the floats.
// ---------------------------------------------------------------------------------------------
// MSD radix select, one workgroup per query, on k' = key - min(key): only the bits below the range's
// top bit vary, so the first 8-bit digit already spreads the values over the LDS histogram.
// max_passes < 4 stops early and returns the UPPER edge of the digit bin reached: an upper bound of the
// R-th smallest, good enough (and valid) as the survivor filter of the pre-scan's second phase.
// QuantizerMAX<int8_t> (db_query_4.cpp:37-71) + qmin / clamp glue of query_scan (258-274), run by the query's
// select workgroup once qmax is known.
// quant_mode 1 = as compiled by the reference's flags: scale = 127/(max-min), trunc((v-min)*scale);
// quant_mode 0 = source level: trunc((v-min)/delta), delta = (max-min)/127.
// Strict IEEE float ops (build uses -ffp-contract=off, no fast-math), so results equal a CPU
// evaluation of the same expressions.
template <int BT>
__device__ void quantize_query(int table_dim_all, float* __restrict__ tb, int8_t* __restrict__ qt, QueryState* qs,
                               float qmax, int quant_mode, float* red /* [BT] LDS */) {
    const int t = threadIdx.x;
    float m = FLT_MAX;
    for (int i = t; i < table_dim_all; i += BT) m = fminf(m, tb[i]);
    red[t] = m;
    __syncthreads();
    for (int d = BT / 2; d >= 1; d >>= 1) {
        if (t < d) red[t] = fminf(red[t], red[t + d]);
        __syncthreads();
    }
    float qmin = red[0];
    uint32_t flags = 0;
    if (qmin < 0) { qmin = 0; flags |= 2u; }
    if ((double)qmax > 1e30) flags |= 1u;                       // (a double compare, as db_query_4.cpp:271: 1e30f itself is above 1e30)
    const float delta = (qmax - qmin) / 127;
    const float scale = 127.0f / (qmax - qmin);
    for (int i = t; i < table_dim_all; i += BT) {
        float v = tb[i];
        if (v < 0) { v = 0; tb[i] = 0; }
        int8_t o;
        if (flags & 1u) o = 127;   // query is skipped by the host; emit nothing
        else if (v >= qmax) o = 127;
        else o = (int8_t)(int)(quant_mode == 0 ? (v - qmin) / delta : (v - qmin) * scale);
        qt[i] = o;
    }
    if (t == 0) { qs->qmin = qmin; qs->flags |= flags; }   // keeps bit3 set by the pre-scan
}

// BT threads per workgroup: 1024 for the level path's few queries with many values each; 256 for a batch of many queries
// (the throughput front of a partition-major batch: a 16-wave workgroup would wait for half a CU beside the scans).
// front_out (optional): {flags & 3, qmin, qmax, 0} per query, the record scan_query_kernel's HEAD takes as front_in.
template <int BT>
__global__ __launch_bounds__(BT) void select_kth_kernel(const float* __restrict__ fc, uint64_t fc_stride,
                                                        const uint32_t* __restrict__ fc_init, uint32_t R,
                                                        QueryState* __restrict__ qstates, int max_passes,
                                                        float* __restrict__ ftables, int8_t* __restrict__ qtables,
                                                        int table_dim_all, int quant_mode,
                                                        float* __restrict__ export_vals,
                                                        uint32_t* __restrict__ export_flags, uint32_t* __restrict__ front_out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_k, s_hi, s_cnt;
    __shared__ float red[BT];
    const int q = blockIdx.x, tid = threadIdx.x;
    QueryState* qs = qstates + q;
    auto publish_front = [&]() {                             // (after quantize_query: thread 0 wrote qmin / flags / qmax itself)
        if (front_out && tid == 0) {
            uint32_t* fo = front_out + 4 * (size_t)q;
            fo[0] = qs->flags & 3u;
            fo[1] = __float_as_uint(qs->qmin);
            fo[2] = __float_as_uint(qs->qmax);
            fo[3] = 0;
        }
    };
    const uint32_t n = min(fc_init[2 * q] + qs->fc_n, fc_init[2 * q + 1]);
    if (export_flags && tid == 0) export_flags[q] = qs->flags;
    if (n < R) {                                           // heap never fills: max() stays the FLT_MAX sentinel
        if (export_vals) {                                 // sharded pre-scan: all n values, padded with the sentinel
            for (uint32_t i = tid; i < R; i += BT)
                export_vals[(uint64_t)q * R + i] = i < n ? fc[(uint64_t)q * fc_stride + i] : FLT_MAX;
        }
        if (tid == 0) qs->qmax = FLT_MAX;
        if (qtables) quantize_query<BT>(table_dim_all, ftables + (uint64_t)q * table_dim_all, qtables + (uint64_t)q * table_dim_all,
                                        qs, FLT_MAX, quant_mode, red);
        publish_front();
        return;
    }
    const uint32_t kmin = ~qs->sel_nmin, kmax = qs->sel_max;
    const uint32_t range = kmax >= kmin ? kmax - kmin : 0u;
    if (tid == 0) { s_prefix = 0; s_k = R; s_hi = range ? 32u - (uint32_t)__clz(range) : 0u; }
    __syncthreads();
    const float* __restrict__ src = fc + (uint64_t)q * fc_stride;
    for (int pass = 0; pass < max_passes; ++pass) {
        const uint32_t hi = s_hi, prefix = s_prefix, k = s_k;
        if (hi == 0) break;
        const uint32_t lo = hi > 8 ? hi - 8 : 0;
        const uint32_t dmask = (1u << (hi - lo)) - 1u;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        // 8 independent loads per thread and iteration: the loop is latency-bound otherwise (one workgroup)
        for (uint32_t i0 = tid; i0 < n; i0 += 8 * BT) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + u * BT;
                v[u] = i < n ? src[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u * BT >= n) continue;
                const uint32_t key = fkey(v[u]) - kmin;
                if (hi >= 32 || ((key ^ prefix) >> hi) == 0) atomicAdd(&hist[(key >> lo) & dmask], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {                                    // wave 0: 4 bins per lane, inclusive scan, pick the digit
            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const uint32_t mine = c0 + c1 + c2 + c3;
            const uint32_t incl = dpp_wave_incl_sum(mine);
            const uint32_t excl = incl - mine;
            if (incl >= k && excl < k) {                   // exactly one lane
                uint32_t run = excl, digit = 4 * tid;
                const uint32_t cs[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (run + cs[j] >= k) { digit = 4 * tid + j; break; }
                    run += cs[j];
                }
                s_prefix = prefix | (digit << lo);
                s_k = k - run;
                s_hi = lo;
            }
        }
        __syncthreads();
    }
    const uint32_t low = s_hi ? ((s_hi >= 32 ? 0u : (1u << s_hi)) - 1u) : 0u;       // undecided bits -> all ones
    const uint64_t key = (uint64_t)(s_prefix | low) + kmin;
    const float qmax = funkey(key > kmax ? kmax : (uint32_t)key);                  // never above the largest stored value
    if (tid == 0) qs->qmax = qmax;
    if (export_vals) {
        // the R smallest values as a multiset (needs the exact R-th smallest: max_passes = 4): every value below
        // qmax, then qmax itself as often as it takes — which of several equal values is dropped does not matter
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += BT) {
            const float v = src[i];
            if (v < qmax) export_vals[(uint64_t)q * R + atomicAdd(&s_cnt, 1u)] = v;    // fewer than R such values
        }
        __syncthreads();
        for (uint32_t i = s_cnt + tid; i < R; i += BT) export_vals[(uint64_t)q * R + i] = qmax;
    }
    if (qtables) quantize_query<BT>(table_dim_all, ftables + (uint64_t)q * table_dim_all, qtables + (uint64_t)q * table_dim_all,
                                    qs, qmax, quant_mode, red);
    publish_front();
}

// Sharded pre-scan, second half: the gathered smallest values of all ranks stand in for the pre-scan output of
// a query.  One workgroup per query records their key range (what start_scan_f32_kernel does for its own values).
__global__ __launch_bounds__(256) void prescan_minmax_kernel(const float* __restrict__ vals, uint32_t nvals,
                                                             QueryState* __restrict__ qstates) {
    __shared__ uint32_t rmin[4], rmax[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t kmin = 0xffffffffu, kmax = 0;
    for (uint32_t i = tid; i < nvals; i += 256) {
        const uint32_t k = fkey(vals[(uint64_t)q * nvals + i]);
        kmin = min(kmin, k);
        kmax = max(kmax, k);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, d, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, d, 64));
    }
    if ((tid & 63) == 0) { rmin[tid >> 6] = kmin; rmax[tid >> 6] = kmax; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { kmin = min(kmin, rmin[w]); kmax = max(kmax, rmax[w]); }
        qstates[q].sel_nmin = ~kmin;
        qstates[q].sel_max = kmax;
    }
}

void launch_prescan_minmax(const float* d_vals, uint32_t nvals, int nq, QueryState* d_qs, hipStream_t stream) {
    hipLaunchKernelGGL(prescan_minmax_kernel, dim3(nq), dim3(256), 0, stream, d_vals, nvals, d_qs);
}

void launch_select_kth(const float* d_fc, uint64_t fc_stride, const uint32_t* d_fc_init, int nq, uint32_t R, QueryState* d_qs,
                       int max_passes, float* d_ftables, int8_t* d_qtables, int table_dim_all, int quant_mode,
                       hipStream_t stream, float* export_vals, uint32_t* export_flags, uint32_t* d_front_out, int small_wg) {
    if (small_wg)
        hipLaunchKernelGGL((select_kth_kernel<256>), dim3(nq), dim3(256), 0, stream, d_fc, fc_stride, d_fc_init, R, d_qs, max_passes,
                           d_ftables, d_qtables, table_dim_all, quant_mode, export_vals, export_flags, d_front_out);
    else
        hipLaunchKernelGGL((select_kth_kernel<1024>), dim3(nq), dim3(1024), 0, stream, d_fc, fc_stride, d_fc_init, R, d_qs, max_passes,
                           d_ftables, d_qtables, table_dim_all, quant_mode, export_vals, export_flags, d_front_out);
}

// ---- stream-layout probe (qadc_stream_probe): does a dispatch that WAITS FOR CUs on stream A hold up stream B? ----
// probe_spin_kernel: many 256-thread workgroups that hold 64 KiB of LDS each (two per CU: most wave slots stay free) and spin
// for spin_ticks of the 100 MHz wall clock: the launch keeps its queue's dispatcher waiting for CUs for several rounds.
// t[0] = first workgroup's start, t[1] = last workgroup's end, t[2] = start of probe_mark_kernel (launched on B right after).
__global__ __launch_bounds__(256) void probe_spin_kernel(unsigned long long* __restrict__ t, uint32_t spin_ticks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char probe_lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        probe_lds[0] = 1;
        atomicMin(&t[0], t0);
    }
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(&t[1], wall_clock64());
}
__global__ void probe_mark_kernel(unsigned long long* __restrict__ t) {
    if (threadIdx.x == 0) t[2] = wall_clock64();
}
hipError_t launch_stream_probe(unsigned long long* d_t, int spin_wgs, uint32_t spin_ticks, hipStream_t a, hipStream_t b) {
    static std::atomic<int> opted{0};
    if (!opted.exchange(1)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(probe_spin_kernel, dim3(spin_wgs), dim3(256), 65536, a, d_t, spin_ticks);
    hipLaunchKernelGGL(probe_mark_kernel, dim3(1), dim3(64), 0, b, d_t);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Host feeders on the device: coarse assignment, residuals, float distance tables.
// Plain sequential float arithmetic (no FMA contraction), so a host evaluation of the same loops
// (host/query_driver.hpp) gives identical bits.
// ---------------------------------------------------------------------------------------------
// ---- find_k_neighbors' selection WITH exact float ties (neighbors.cpp:18-28, 47-71; binheap.hpp:75-127) ----
// The select kernels below pick "the ma smallest by (distance, index)".  That IS what the reference's heaps leave whenever no two
// of the kept-or-boundary distances are exactly equal (pinned: tests/test_oracle_float_ref.py).  With exact ties the reference's
// result depends on its heap's history — a kv_binheap<int, float> of capacity ma takes the K distances in index order (push: a
// full heap accepts only values strictly below its root, the root sinks preferring the LEFT child on equal children) — and on the
// order std::sort leaves equal keys in (libstdc++ introsort on the permutation 0..ma-1 of the heap array, comparator
// value[a] < value[b]: median-of-3 quicksort loop down to ranges of 16, depth limit 2 floor(log2 n) with heapsort below it, one
// final insertion sort — GCC 11.4 bits/stl_algo.h, the same restatement as the oracle's sort_keys).  A select kernel that SEES a
// tie (two equal values among its ma, or a value equal to the ma-th outside them) calls this for its query: wave 0, the heap and
// the permutation in LDS, lane 0 doing the sequential part, the other lanes only the loads and the pre-filter (a chunk of 64
// distances none of which is below the root is skipped with one ballot).  ma <= 256.
struct ExactSel {
    float* hv;      // [256] heap values
    int* hk;        // [256] heap keys
    int* perm;      // [256] permutation sorted by std::sort's algorithm; then [72]: the quicksort loop's pending ranges
};
constexpr int kExactSelInts = 256 + 72;     // (everything lives in LDS: a private array would give the select kernels a scratch segment)

__device__ __forceinline__ bool xs_less(const ExactSel& x, int a, int b) { return x.hv[a] < x.hv[b]; }

__device__ __forceinline__ void xs_push_heap(const ExactSel& x, int* first, int hole, int top, int value) {      // std::__push_heap
    int parent = (hole - 1) / 2;
    while (hole > top && xs_less(x, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ __forceinline__ void xs_adjust_heap(const ExactSel& x, int* first, int hole, int len, int value) {    // std::__adjust_heap
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (xs_less(x, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    xs_push_heap(x, first, hole, top, value);
}

__device__ __forceinline__ void xs_heapsort(const ExactSel& x, int* first, int* last) {                          // std::__partial_sort(first, last, last)
    const int len = (int)(last - first);
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            xs_adjust_heap(x, first, parent, len, first[parent]);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        const int value = *last;
        *last = *first;
        xs_adjust_heap(x, first, 0, (int)(last - first), value);
    }
}

__device__ __forceinline__ void xs_linear_insert(const ExactSel& x, int* last) {                                 // std::__unguarded_linear_insert
    const int val = *last;
    int* next = last - 1;
    while (xs_less(x, val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

__device__ __forceinline__ void xs_insertion_sort(const ExactSel& x, int* first, int* last) {                    // std::__insertion_sort
    if (first == last) return;
    for (int* i = first + 1; i != last; ++i) {
        if (xs_less(x, *i, *first)) {
            const int val = *i;
            for (int* p = i; p != first; --p) *p = *(p - 1);                                      // std::move_backward
            *first = val;
        } else {
            xs_linear_insert(x, i);
        }
    }
}

__device__ __forceinline__ void xs_std_sort(const ExactSel& x, int n) {
    int* perm = x.perm;
    int lg = 0;
    for (int m = n; m > 1; m >>= 1) ++lg;
    // std::__introsort_loop, its tail recursion on [cut, last) as an explicit stack (at most 2 lg + 1 ranges deep)
    int* st_first = x.perm + 256; int* st_last = st_first + 24; int* st_depth = st_last + 24;
    int sp = 0;
    st_first[0] = 0; st_last[0] = n; st_depth[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = st_first[sp], last = st_last[sp], depth = st_depth[sp];
        // (the recursion handles [cut, last) FIRST and then continues with [first, cut): ranges are disjoint, so the order in which
        // they are processed does not change the result; here the right part is stacked and the left one continues)
        while (last - first > 16) {
            if (depth == 0) {
                xs_heapsort(x, perm + first, perm + last);
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            {                                                                                     // std::__move_median_to_first(first, first + 1, mid, last - 1)
                int* r = perm + first; int* a = perm + first + 1; int* b = perm + mid; int* c = perm + last - 1;
                int* pick;
                if (xs_less(x, *a, *b)) pick = xs_less(x, *b, *c) ? b : (xs_less(x, *a, *c) ? c : a);
                else pick = xs_less(x, *a, *c) ? a : (xs_less(x, *b, *c) ? c : b);
                const int t = *r; *r = *pick; *pick = t;
            }
            int lo = first + 1, hi = last;                                                       // std::__unguarded_partition(first + 1, last, first)
            for (;;) {
                while (xs_less(x, perm[lo], perm[first])) ++lo;
                --hi;
                while (xs_less(x, perm[first], perm[hi])) --hi;
                if (!(lo < hi)) break;
                const int t = perm[lo]; perm[lo] = perm[hi]; perm[hi] = t;
                ++lo;
            }
            if (sp < 24) { st_first[sp] = lo; st_last[sp] = last; st_depth[sp] = depth; ++sp; }
            last = lo;
        }
    }
    if (n > 16) {                                                                                 // std::__final_insertion_sort
        xs_insertion_sort(x, perm, perm + 16);
        for (int* i = perm + 16; i != perm + n; ++i) xs_linear_insert(x, i);
    } else {
        xs_insertion_sort(x, perm, perm + n);
    }
}

// ||x - c||^2 over ds components as the reference's direct table form adds it: fmanorm<ds/8, ds%8> called by
// compute_dists_single_simd_cg (distances.hpp:60-76, 294-311) AS COMPILED with the reference's flags (pinned to that
// build through the oracle's orc_tables_direct; host twin: host/float_sum.hpp sqdist):
//   per AVX lane j: acc[j] = fma(d, d, acc[j]) over the ds/8 blocks, d = x - c;  reduceadd's tree acc[j] + acc[j+4],
//   then (r0 + r2) + (r1 + r3);  the scalar remainder is paired p_k = fma(d_2k, d_2k, r(d_2k+1^2)), d = c - x:
//   REM 4 -> (p0 + p1) + vec,  REM 6 -> (vec + p2) + (p0 + p1).
// sum_mode 0, or a remainder the reference has no instance of (sq_dim 3 of BASELINE configs[4]; its dispatch is
// distances.cpp:50-84): one sequential sum in ascending d.  X / C: anything indexable by int (pointer or register array).
template <typename X, typename C>
__device__ __forceinline__ float direct_sqdist(const X& x, const C& c, int ds, int sum_mode) {
    const int blocks = ds >> 3, rem = ds & 7;
    if (sum_mode == 0 || !(rem == 0 || rem == 4 || rem == 6)) {
        float s = 0.0f;
        for (int d = 0; d < ds; ++d) {
            const float t = x[d] - c[d];
            s += t * t;
        }
        return s;
    }
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int b = 0; b < blocks; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = x[b * 8 + j] - c[b * 8 + j];
            acc[j] = __fmaf_rn(d, d, acc[j]);
        }
    }
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    const float vec = (r0 + r2) + (r1 + r3);
    if (rem == 0) return vec;
    float p[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (2 * k < rem) {
            const float d0 = c[blocks * 8 + 2 * k] - x[blocks * 8 + 2 * k];
            const float d1 = c[blocks * 8 + 2 * k + 1] - x[blocks * 8 + 2 * k + 1];
            p[k] = __fmaf_rn(d0, d0, d1 * d1);
        }
    }
    if (rem == 4) return (p[0] + p[1]) + vec;
    return (vec + p[2]) + (p[0] + p[1]);
}

// The BLAS-expansion distance compute_cross_dists_blas<DSQ> (distances.hpp:151-215) leaves in dists[v][c]:
//   ||v||^2 + ||c||^2 (153-176), then cblas_sgemm(alpha = -2, beta = 1) adds -2 v.c (178-182).
// expansion_sqnorm = fmanorm<DSQ/8, DSQ%8>(vec) / norm_4(vec) AS COMPILED with the reference's flags (sum_mode 1): the
// grouping of direct_sqdist with c = 0 — pinned to the reference's own text compiled up to the sgemm call (oracle/_ref
// qadc_reff_cross_norms, the 14 dimensions of its dispatch, 16 centroids) through the oracle's orc_sqnorm; host twin:
// host/float_sum.hpp sqnorm.  sum_mode 0, or a remainder the reference has no instance of: one sequential sum.
// The product is OpenBLAS's in the reference (not in this image: restated, unpinned): one sequential dot in ascending d,
// then base + (-2 dot) — -2 dot is exact, so this is the single rounding of a gemm kernel's C += alpha * acc.
struct zero_vec {
    __device__ __forceinline__ float operator[](int) const { return 0.0f; }
};
template <typename X>
__device__ __forceinline__ float expansion_sqnorm(const X& x, int ds, int sum_mode) {
    return direct_sqdist(x, zero_vec{}, ds, sum_mode);
}
template <typename X, typename C>
__device__ __forceinline__ float expansion_dist(const X& x, const C& c, int ds, float vn, float cn) {
    float dot = 0.0f;
    for (int d = 0; d < ds; ++d) dot += x[d] * c[d];
    return (vn + cn) + (-2.0f * dot);
}

// called by every lane of wave 0 (lane = 0..63); dq = the query's K distances in global memory
__device__ __forceinline__ void coarse_exact_select(const float* __restrict__ dq, int K, int ma, int32_t* __restrict__ out,
                                                 float* hv, int* hk, int* perm, uint32_t lane) {
    const ExactSel x{hv, hk, perm};
    int size = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + (int)lane;
        const float v = k < K ? dq[k] : 0.0f;
        uint64_t todo = __builtin_amdgcn_ballot_w64(k < K && (size < ma || v < hv[0]));
        while (todo) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const float vj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j));
            if (lane == 0) {                                                                      // kv_binheap::push (binheap.hpp:75-116)
                if (size != ma) {
                    int i = size;
                    hv[i] = vj; hk[i] = k0 + j;
                    int parent = (i - 1) / 2;
                    while (i != 0 && hv[i] > hv[parent]) {
                        const float tv = hv[i]; hv[i] = hv[parent]; hv[parent] = tv;
                        const int tk = hk[i]; hk[i] = hk[parent]; hk[parent] = tk;
                        i = parent;
                        parent = (i - 1) / 2;
                    }
                } else if (vj < hv[0]) {
                    int i = 0;
                    hv[0] = vj; hk[0] = k0 + j;
                    for (;;) {
                        const int l = 2 * i + 1, r = 2 * i + 2;
                        if (l >= ma) break;
                        int c = l;
                        if (r < ma && hv[r] > hv[l]) c = r;
                        if (hv[c] <= hv[i]) break;
                        const float tv = hv[i]; hv[i] = hv[c]; hv[c] = tv;
                        const int tk = hk[i]; hk[i] = hk[c]; hk[c] = tk;
                        i = c;
                    }
                }
            }
            if (size != ma) ++size;                                                               // (uniform: every lane counts)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = (int)lane; i < size; i += 64) perm[i] = i;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) xs_std_sort(x, size);                                                          // kv_binheap::sort (binheap.hpp:118-127)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = (int)lane; i < size; i += 64) out[i] = hk[perm[i]];
}

// KPT = centroids per thread kept in registers (K <= 256 * KPT); KPT == 0: distances live in the global scratch.
// Every thread accumulates ITS centroid's dimensions in ascending order (bit-exact with the host loop); a
// thread walks one row sequentially, so each cache line it touches is reused 16 times from L1 and the K x dim
// matrix (shared by every query's workgroup) stays in L2.
// QB = queries per workgroup: a thread walks ITS centroid rows once and accumulates the QB queries' distances side by
// side (each in its own ascending-d order, so every sum is bit-exact with the host loop) — the K x dim matrix is read
// nq/QB times instead of nq times, which is what this kernel is bound by.
// ||row||^2 of n rows of `dim` floats, one thread per row, in the grouping the reference's compute_cross_dists_blas adds it
// (fmanorm<dim/8, dim%8> as compiled; expansion_sqnorm): the ||q||^2 and ||c||^2 of the coarse distances below.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ rows, int n, int dim, int sum_mode, float* __restrict__ out) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i < n) out[i] = expansion_sqnorm(rows + (size_t)i * dim, dim, sum_mode);
}

// The coarse distances are the reference's: find_k_neighbors (neighbors.cpp:30-76) takes them from compute_cross_dists_blas<dim>
// (distances.hpp:151-183): (||q||^2 + ||c||^2) — qnorm / cnorm, as compiled — then sgemm(alpha = -2, beta = 1) adds -2 q.c; the
// product is restated as ONE sequential dot in ascending d (OpenBLAS is not in this image), the sum rounds once.  They can come
// out slightly negative for q ~ c: the selections below order floats of either sign.
template <int KPT, int QB>
__global__ __launch_bounds__(256) void coarse_assign_kernel(const float* __restrict__ queries,
                                                            const float* __restrict__ coarse, int nq, int K, int dim, int ma,
                                                            const float* __restrict__ qnorm, const float* __restrict__ cnorm,
                                                            float* __restrict__ dist, int32_t* __restrict__ assign) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* q = reinterpret_cast<float*>(dyn);                 // [QB][dim]
    __shared__ float rv[4];
    __shared__ int rk[4];
    const int q0 = blockIdx.x * QB, tid = threadIdx.x;
    const int nqb = min(QB, nq - q0);
    for (int i = tid; i < QB * dim; i += 256) {
        const int b = i / dim, d = i - b * dim;
        q[i] = b < nqb ? queries[(size_t)(q0 + b) * dim + d] : 0.0f;
    }
    __syncthreads();
    constexpr int NR = KPT > 0 ? KPT : 1;
    float mine[NR][QB];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int b = 0; b < QB; ++b) mine[j][b] = FLT_MAX;
    const int nblk = (K + 255) / 256;
    auto row_dist = [&](int k, float (&s)[QB]) {
        const float* __restrict__ c = coarse + (size_t)k * dim;
#pragma unroll
        for (int b = 0; b < QB; ++b) s[b] = 0.0f;
        for (int d = 0; d < dim; ++d) {
            const float cv = c[d];
#pragma unroll
            for (int b = 0; b < QB; ++b) s[b] += q[b * dim + d] * cv;
        }
        const float cn = cnorm[k];
#pragma unroll
        for (int b = 0; b < QB; ++b) s[b] = (qnorm[min(q0 + b, nq - 1)] + cn) + (-2.0f * s[b]);
    };
    if (KPT > 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int k = j * 256 + tid;
            if (j < nblk && k < K) {
                row_dist(k, mine[j]);
                if (dist)                                        // (the exact-tie path below reads the row from memory)
#pragma unroll
                    for (int b = 0; b < QB; ++b)
                        if (b < nqb) dist[(size_t)(q0 + b) * K + k] = mine[j][b];
            }
        }
    } else {
        for (int k = tid; k < K; k += 256) {
            float s[QB];
            row_dist(k, s);
#pragma unroll
            for (int b = 0; b < QB; ++b)
                if (b < nqb) dist[(size_t)(q0 + b) * K + k] = s[b];
        }
    }
    __syncthreads();
    __shared__ float x_hv[256];
    __shared__ int x_hk[256], x_perm[kExactSelInts];
    // per query: ma rounds of "smallest (distance, index) strictly after the previous pick" (+ one more round that only looks
    // for a value equal to the ma-th outside the picks: an exact tie, see coarse_exact_select)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        if (b >= nqb) break;
        const float* __restrict__ dq = dist + (size_t)(q0 + b) * K;
        float last_v = -FLT_MAX;
        int last_k = -1;
        bool tie = false;
        const int rounds = (dist && ma > 1 && ma <= 256 && ma < K) ? ma + 1 : ma;
        for (int a = 0; a < rounds; ++a) {
            float bv = FLT_MAX;
            int bk = 0x7fffffff;
            if (KPT > 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int k = j * 256 + tid;
                    const float v = mine[j][b];
                    const bool after = v > last_v || (v == last_v && k > last_k);
                    if (k < K && after && (v < bv || (v == bv && k < bk))) { bv = v; bk = k; }
                }
            } else {
                for (int k = tid; k < K; k += 256) {
                    const float v = dq[k];
                    const bool after = v > last_v || (v == last_v && k > last_k);
                    if (after && (v < bv || (v == bv && k < bk))) { bv = v; bk = k; }
                }
            }
            // wave-level reduction first (no barrier), then across the 4 waves through LDS
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(bv, d, 64);
                const int ok = __shfl_xor(bk, d, 64);
                if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
            }
            if ((tid & 63) == 0) { rv[tid >> 6] = bv; rk[tid >> 6] = bk; }
            __syncthreads();
            bv = rv[0];
            bk = rk[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (rv[w] < bv || (rv[w] == bv && rk[w] < bk)) { bv = rv[w]; bk = rk[w]; }
            tie = tie || (a > 0 && bv == last_v);
            last_v = bv;
            last_k = bk;
            if (tid == 0 && a < ma) assign[(size_t)(q0 + b) * ma + a] = last_k;
            __syncthreads();
        }
        if (tie && dist && ma <= 256) {                          // (uniform: every thread saw the same picks)
            __threadfence_block();
            if (tid < 64) coarse_exact_select(dq, K, ma, assign + (size_t)(q0 + b) * ma, x_hv, x_hk, x_perm, (uint32_t)tid);
            __syncthreads();
        }
    }
}

// ---- the same arithmetic as two lean kernels (what every batch with dim % 4 == 0 and K <= 16384 takes) ----
// coarse_assign_kernel keeps a query group's K distances in registers (64 accumulators per lane at K = 4096) and reads
// its centroid rows one row per lane.  Under the pipelined IVF batches that is the most expensive kernel of the front.
// Here (1) coarse_dist_kernel computes [16 queries] x [256 centroids] distance tiles — centroid tiles staged through LDS
// with 16-byte coalesced loads (the next tile's loads in flight), transposed so that lane k reads tile[d][k] without
// bank conflicts, the 16 queries' components as LDS broadcasts, 16 accumulators per lane — into the [nq][K] scratch,
// and (2) coarse_select_kernel picks the ma nearest per query.  Every (query, centroid) sum still accumulates
// d = 0, 1, 2 ... in order, and the selection applies the same (distance, index) order: same assign[] bit for bit.
constexpr int kCDQ = 16, kCDC = 32, kCDStride = 257;

__device__ __forceinline__ uint32_t dpp_wave_min_u32(uint32_t x) {
    uint32_t v = x;
    const int id = -1;                                           // 0xffffffff: identity of min
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x113, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xf, 0xe, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xf, 0xc, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(256) void coarse_dist_kernel(const float* __restrict__ queries, const float* __restrict__ coarse,
                                                          int nq, int K, int dim, const float* __restrict__ qnorm,
                                                          const float* __restrict__ cnorm, float* __restrict__ dist) {
    __shared__ float tile[kCDC * kCDStride];                     // [d][centroid]
    __shared__ __attribute__((aligned(16))) float qt[kCDC * kCDQ];   // [d][query]
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 256, q0 = blockIdx.y * kCDQ;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;           // this thread's float4s of a tile: rows lrow + 32p, columns lcol..lcol+3
    float acc[kCDQ];
#pragma unroll
    for (int b = 0; b < kCDQ; ++b) acc[b] = 0.0f;
    float4 pre[8];
    float qpre[2];
    auto fetch = [&](int d0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = k0 + lrow + 32 * p, d = d0 + lcol;
            pre[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k < K && d < dim) pre[p] = *reinterpret_cast<const float4*>(coarse + (size_t)k * dim + d);   // (dim % 4 == 0)
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {                            // the query chunk [16][32]: element (b, dd) = tid + 256 e
            const int i = tid + 256 * e, b = i / kCDC, dd = i % kCDC;
            const int q = min(q0 + b, nq - 1), d = d0 + dd;      // (a ragged last group repeats its last query)
            qpre[e] = d < dim ? queries[(size_t)q * dim + d] : 0.0f;
        }
    };
    fetch(0);
    for (int d0 = 0; d0 < dim; d0 += kCDC) {
        const int dcn = min(kCDC, dim - d0);
        __syncthreads();                                         // the previous tile has been consumed
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = lrow + 32 * p;
            tile[(lcol + 0) * kCDStride + r] = pre[p].x;
            tile[(lcol + 1) * kCDStride + r] = pre[p].y;
            tile[(lcol + 2) * kCDStride + r] = pre[p].z;
            tile[(lcol + 3) * kCDStride + r] = pre[p].w;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = tid + 256 * e;
            qt[(i % kCDC) * kCDQ + i / kCDC] = qpre[e];
        }
        __syncthreads();
        if (d0 + kCDC < dim) fetch(d0 + kCDC);                   // in flight while this tile is consumed
        for (int dd = 0; dd < dcn; ++dd) {
            const float cv = tile[dd * kCDStride + tid];
            // two queries per instruction: v_pk_mul_f32 + v_pk_add_f32 — each component is the IEEE result of the scalar operation
            // (no contraction: -ffp-contract=off), the dots stay sequential in d
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 cv2 = {cv, cv};
#pragma unroll
            for (int b4 = 0; b4 < kCDQ / 4; ++b4) {
                const float4 t4 = *reinterpret_cast<const float4*>(&qt[dd * kCDQ + 4 * b4]);   // same address in every lane: broadcast
                const f32x2 q01 = {t4.x, t4.y}, q23 = {t4.z, t4.w};
                const f32x2 s01 = q01 * cv2, s23 = q23 * cv2;
                acc[4 * b4] += s01.x; acc[4 * b4 + 1] += s01.y; acc[4 * b4 + 2] += s23.x; acc[4 * b4 + 3] += s23.y;
            }
        }
    }
    const int k = k0 + tid;
    if (k < K) {
        const float cn = cnorm[k];
#pragma unroll
        for (int b = 0; b < kCDQ; ++b)
            if (q0 + b < nq) dist[(size_t)(q0 + b) * K + k] = (qnorm[q0 + b] + cn) + (-2.0f * acc[b]);
    }
}

// Distances of either sign as unsigned keys that order like the values (positive: sign bit set; negative: all bits flipped; -0 does
// not occur: a sum of a non-negative and a product rounds to +0); 0xffffffff stays above every finite key: the padding past K.
__device__ __forceinline__ uint32_t coarse_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

// ma rounds of "smallest (distance, index) strictly after the previous pick" over a query's K distances (as coarse_key);
// one workgroup per query, KPT distances per lane in registers
template <int KPT>
__global__ __launch_bounds__(256) void coarse_select_kernel(const float* __restrict__ dist, int K, int ma, int32_t* __restrict__ assign) {
    __shared__ uint32_t rv[2][4], rk[2][4];
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t mine[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = j * 256 + tid;
        mine[j] = k < K ? coarse_key(dist[(size_t)q * K + k]) : 0xffffffffu;
    }
    __shared__ float x_hv[256];
    __shared__ int x_hk[256], x_perm[kExactSelInts];
    uint32_t last_v = 0;
    int last_k = -1, par = 0;
    bool tie = false;
    const int rounds = (ma > 1 && ma <= 256 && ma < K) ? ma + 1 : ma;   // (the extra round looks for a value equal to the ma-th outside the picks)
    for (int a = 0; a < rounds; ++a, par ^= 1) {
        uint32_t bv = 0xffffffffu, bk = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t k = (uint32_t)(j * 256 + tid);
            const uint32_t v = mine[j];
            const bool after = v > last_v || (v == last_v && (int)k > last_k);
            if (k < (uint32_t)K && after && (v < bv || (v == bv && k < bk))) { bv = v; bk = k; }
        }
        const uint32_t wv = dpp_wave_min_u32(bv);
        const uint32_t wk = dpp_wave_min_u32(bv == wv ? bk : 0xffffffffu);
        if ((tid & 63) == 0) { rv[par][tid >> 6] = wv; rk[par][tid >> 6] = wk; }
        __syncthreads();                                         // (slots alternate: one barrier per round)
        bv = rv[par][0];
        bk = rk[par][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (rv[par][w] < bv || (rv[par][w] == bv && rk[par][w] < bk)) { bv = rv[par][w]; bk = rk[par][w]; }
        tie = tie || (a > 0 && bv == last_v && bk != 0xffffffffu);
        last_v = bv;
        last_k = (int)bk;
        if (tid == 0 && a < ma) assign[(size_t)q * ma + a] = last_k;
    }
    if (tie && ma <= 256) {                                      // exact float ties: the reference's heap history decides (coarse_exact_select)
        __syncthreads();
        if (tid < 64) coarse_exact_select(dist + (size_t)q * K, K, ma, assign + (size_t)q * ma, x_hv, x_hk, x_perm, (uint32_t)tid);
    }
}

// The same selection for ma <= 256 without the ma dependent rounds: (1) the ma-th smallest key by a 4-pass MSD radix
// select over the K distances (8 bits per pass, 256-bin histogram in LDS); (2) every key below it, and the needed number
// of keys EQUAL to it with the smallest indices, collected into LDS; (3) one bitonic sort of those ma (key, index) pairs.
// Result identical to the rounds above (ascending distance, lower index first on ties); K = 16384, ma = 64: 330 -> ~25 us
// for a workgroup on its own, which is what a rank's share of a sharded front waits for.
// An exact float tie — a value equal to the ma-th outside the picks, or two equal values among them — hands the query to
// coarse_exact_select (the reference's heap history decides); without ties the result is the reference's entry for entry.
template <int KPT>
__global__ __launch_bounds__(256) void coarse_select_radix_kernel(const float* __restrict__ dist, int K, int ma, int32_t* __restrict__ assign) {
    __shared__ uint32_t hist[256];
    __shared__ __attribute__((aligned(16))) uint64_t cand[256];
    __shared__ uint32_t ties[256];
    __shared__ uint32_t s_prefix, s_rank, s_nless, s_nties, s_kept, s_T0;
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t mine[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int k = j * 256 + tid;
        mine[j] = k < K ? coarse_key(dist[(size_t)q * K + k]) : 0xffffffffu;
    }
    if (tid == 0) { s_prefix = 0; s_rank = (uint32_t)ma; s_nless = 0; s_nties = 0; s_kept = 0; s_T0 = 0xffffffffu; }
    __syncthreads();
    // Threshold first (round 4, as in the query kernel's front select): ma is a percent or less of K, and counting ALL K
    // distances into digit bins is one LDS atomic per value on one or two hot bins — 16384 serialized adds per pass at the C5
    // shape.  Wave 0 ranks the first 64 distances (centroid order carries no distance order) and takes the m-th smallest,
    // m = 1.5 x the ma-th value's expected rank among 64, + 3; only distances <= that take part in the passes (a few
    // percent of K; the ma smallest are among them as long as at least ma are kept — counted; if not, no filter).
    if (tid < 64) {
        const uint32_t sk = mine[0];                             // (k = tid; 0xffffffff past K)
        const uint32_t m = min(64u, (uint32_t)((96ull * (uint32_t)ma + (uint32_t)K - 1u) / (uint32_t)K) + 3u);
        uint32_t less = 0, le = 0;
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)sk, j);
            less += kj < sk ? 1u : 0u;
            le += kj <= sk ? 1u : 0u;
        }
        if (less < m && m <= le) s_T0 = sk;
    }
    __syncthreads();
    uint32_t T0 = s_T0;
    {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < KPT; ++j) c += (j * 256 + tid < K && mine[j] <= T0) ? 1u : 0u;
        c = dpp_wave_incl_sum(c);
        if ((tid & 63) == 63) atomicAdd(&s_kept, c);
    }
    __syncthreads();
    if (s_kept < (uint32_t)ma) T0 = 0xffffffffu;                 // (an unlucky sample: everything takes part, as before)
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix, sh = 8u * (uint32_t)pass;
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const uint32_t v = mine[j];
            if (j * 256 + tid < K && v <= T0 && (pass == 3 || (v >> (sh + 8u)) == prefix)) {
                // the lanes that share the first lane's digit add ONE count (the top digits of distances are all but equal)
                const uint32_t dg = (v >> sh) & 255u;
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg);
                if (dg == lead) {
                    const uint64_t same = __builtin_amdgcn_ballot_w64(true);
                    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u)) == 0)
                        atomicAdd(&hist[lead], (uint32_t)__popcll(same));
                } else {
                    atomicAdd(&hist[dg], 1u);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {                                          // wave 0: 4 bins per lane, the digit that holds rank s_rank
            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const uint32_t sum4 = c0 + c1 + c2 + c3;
            const uint32_t incl = dpp_wave_incl_sum(sum4), excl = incl - sum4;
            const uint32_t r = s_rank;
            if (incl >= r && excl < r) {                         // exactly one lane
                uint32_t run = excl, digit = 4 * tid;
                const uint32_t cs4[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (run + cs4[jj] >= r) { digit = 4 * tid + jj; break; }
                    run += cs4[jj];
                }
                s_prefix = (prefix << 8) | digit;
                s_rank = r - run;                                // rank inside the chosen digit
            }
        }
        __syncthreads();
    }
    const uint32_t T = s_prefix, need_ties = s_rank;            // need_ties keys equal to T complete the ma smallest
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const uint32_t v = mine[j], k = (uint32_t)(j * 256 + tid);
        if (k >= (uint32_t)K) continue;
        if (v < T) cand[atomicAdd(&s_nless, 1u)] = ((uint64_t)v << 32) | k;
        else if (v == T) {
            const uint32_t slot = atomicAdd(&s_nties, 1u);
            if (slot < 256u) ties[slot] = k;
        }
    }
    __syncthreads();
    const uint32_t nless = s_nless, nties = s_nties;            // nless == ma - need_ties
    __shared__ float x_hv[256];
    __shared__ int x_hk[256], x_perm[kExactSelInts];
    __shared__ uint32_t s_tie;
    if (nties > need_ties) {                                     // a value equal to the ma-th outside the picks: an exact tie at the boundary —
        // the reference's heap history decides which of them stay and in which order (coarse_exact_select); uniform branch
        if (tid < 64) coarse_exact_select(dist + (size_t)q * K, K, ma, assign + (size_t)q * ma, x_hv, x_hk, x_perm, (uint32_t)tid);
        return;
    }
    if (tid == 0) s_tie = 0;
    // ties: the need_ties smallest indices among them; then the ma survivors in (distance, index) order.  Both by COUNTING — an
    // entry's place is the number of smaller entries, read as LDS broadcasts (indices and keys are distinct) — instead of two
    // 256-element bitonic networks: 72 barrier steps were most of this kernel (87 us alone on the GPU at the C5 shape for 67 MB of
    // distances; round 4).
    if ((uint32_t)tid < nties) {
        const uint32_t mine_k = ties[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nties; ++j) rank += ties[j] < mine_k ? 1u : 0u;
        if (rank < need_ties) cand[nless + rank] = ((uint64_t)T << 32) | mine_k;
    }
    __syncthreads();
    if (tid < ma) {
        const uint64_t key = cand[tid];
        const uint32_t kv = (uint32_t)(key >> 32);
        uint32_t rank = 0, same = 0;
        int j = 0;
        for (; j + 2 <= ma; j += 2) {                            // (two keys per LDS read)
            const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&cand[j]);
            rank += (kk.x < key ? 1u : 0u) + (kk.y < key ? 1u : 0u);
            same += ((uint32_t)(kk.x >> 32) == kv ? 1u : 0u) + ((uint32_t)(kk.y >> 32) == kv ? 1u : 0u);
        }
        for (; j < ma; ++j) {
            rank += cand[j] < key ? 1u : 0u;
            same += (uint32_t)(cand[j] >> 32) == kv ? 1u : 0u;
        }
        assign[(size_t)q * ma + rank] = (int32_t)(uint32_t)key;
        if (same > 1u) s_tie = 1u;                               // two picks with the same distance: an exact tie inside
    }
    __syncthreads();
    if (s_tie && tid < 64) coarse_exact_select(dist + (size_t)q * K, K, ma, assign + (size_t)q * ma, x_hv, x_hk, x_perm, (uint32_t)tid);
}

void launch_row_sqnorm(const float* d_rows, int n, int dim, int sum_mode, float* d_out, hipStream_t stream) {
    if (n > 0) hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_rows, n, dim, sum_mode, d_out);
}

void launch_coarse_assign(const float* d_queries, const float* d_coarse, int nq, int K, int dim, int ma, float* d_qnorm,
                          const float* d_cnorm, int sum_mode, float* d_dist, int32_t* d_assign, hipStream_t stream) {
    const int kpt = (K + 255) / 256;
    launch_row_sqnorm(d_queries, nq, dim, sum_mode, d_qnorm, stream);                       // ||q||^2 (the centroids': the caller's, once)
    // large batches share every centroid row between 4 queries (registers: KPT x 4 distances per thread)
#define QADC_CA(N, QB) hipLaunchKernelGGL((coarse_assign_kernel<N, QB>), dim3((nq + QB - 1) / QB), dim3(256), (size_t)QB * dim * sizeof(float), \
                                          stream, d_queries, d_coarse, nq, K, dim, ma, d_qnorm, d_cnorm, d_dist, d_assign)
    // (any batch size: even one query — 15 of its group's 16 rows idle — is through sooner than with a row per lane)
    if (kpt <= 64 && ma <= K && dim % 4 == 0 && (reinterpret_cast<uintptr_t>(d_coarse) & 15) == 0 && d_dist) {
        hipLaunchKernelGGL(coarse_dist_kernel, dim3(kpt, (nq + kCDQ - 1) / kCDQ), dim3(256), 0, stream, d_queries, d_coarse, nq, K, dim, d_qnorm,
                           d_cnorm, d_dist);
        if (ma <= 256 && ma >= 8) {                              // radix select + one sort (few probes: the rounds are as quick)
            if (kpt <= 4) hipLaunchKernelGGL(coarse_select_radix_kernel<4>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
            else if (kpt <= 16) hipLaunchKernelGGL(coarse_select_radix_kernel<16>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
            else hipLaunchKernelGGL(coarse_select_radix_kernel<64>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
        }
        else if (kpt <= 4) hipLaunchKernelGGL(coarse_select_kernel<4>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
        else if (kpt <= 16) hipLaunchKernelGGL(coarse_select_kernel<16>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
        else hipLaunchKernelGGL(coarse_select_kernel<64>, dim3(nq), dim3(256), 0, stream, d_dist, K, ma, d_assign);
    } else if (nq >= 512) {
        if (kpt <= 4) QADC_CA(4, 4);
        else if (kpt <= 16) QADC_CA(16, 4);
        else if (kpt <= 32) QADC_CA(32, 2);
        else QADC_CA(0, 4);
    } else {
        if (kpt <= 4) QADC_CA(4, 1);
        else if (kpt <= 16) QADC_CA(16, 1);
        else if (kpt <= 32) QADC_CA(32, 1);
        else QADC_CA(0, 1);
    }
#undef QADC_CA
}

__global__ __launch_bounds__(256) void build_tables_kernel(const float* __restrict__ queries, const float* __restrict__ coarse,
                                                           const int32_t* __restrict__ assign,
                                                           const float* __restrict__ codebooks,
                                                           const float* __restrict__ rotation, int ma, int M, int dim,
                                                           int expansion, int sum_mode, float* __restrict__ ftables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* res = reinterpret_cast<float*>(dyn);               // [dim] residual of this (query, probe)
    float* tmp = res + dim;                                   // [dim] un-rotated residual (OPQ only)
    const int a = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x;
    const int ds = dim / M;
    const float* __restrict__ c = coarse ? coarse + (size_t)assign[(size_t)qi * ma + a] * dim : nullptr;
    float* first = rotation ? tmp : res;
    for (int d = tid; d < dim; d += 256) {
        const float x = queries[(size_t)qi * dim + d];
        first[d] = c ? x - c[d] : x;
    }
    __syncthreads();
    if (rotation) {
        // opq::rotate_multiple_vectors (quantizers.hpp:289-301): rotated[r] = sum_c x[c] * rotation[r][c]
        for (int r = tid; r < dim; r += 256) {
            const float* __restrict__ row = rotation + (size_t)r * dim;
            float acc = 0.0f;
            for (int cc = 0; cc < dim; ++cc) acc += tmp[cc] * row[cc];
            res[r] = acc;
        }
        __syncthreads();
    }
    float* __restrict__ out = ftables + ((size_t)qi * ma + a) * (M * 16);
    for (int e = tid; e < M * 16; e += 256) {
        const int m = e >> 4;
        const float* __restrict__ ce = codebooks + (size_t)e * ds;   // [m][c][ds] is contiguous in e
        float s = 0.0f;
        if (expansion) {
            // compute_cross_dists_blas (distances.hpp:151-183): ||v||^2 + ||c||^2, then sgemm(alpha = -2, beta = 1) adds
            // -2 v.c: the BLAS-expansion form nns_engine uses for ma > 1 and nns_engine_batch always.  It can come out
            // slightly NEGATIVE when v ~ c — the case query_scan clamps (db_query_4.cpp:258-269).  Norms as compiled,
            // sequential dot, no contraction (expansion_dist above); host twin: pq4::tables_blas.
            const float* __restrict__ v = res + m * ds;
            s = expansion_dist(v, ce, ds, expansion_sqnorm(v, ds, sum_mode), expansion_sqnorm(ce, ds, sum_mode));
        } else {
            s = direct_sqdist(res + m * ds, ce, ds, sum_mode);
        }
        out[e] = s;
    }
}

// The same tables for kBTProbes (16; 8 and 32 within 15 %) probes of a query per workgroup (no OPQ rotation, sub-vectors of <= 8 components): a thread
// keeps its table entry's codebook row in registers and walks the probes, whose residuals wait in LDS.  One workgroup per
// (query, probe) is 65 K workgroups of 512 results each at the C5 shape — the launch was bound by workgroup dispatch and by
// re-reading the codebook row per result (92 us alone on the GPU for 134 MB of tables; round 4: -> 8 K workgroups).
// Entry for entry the arithmetic of build_tables_kernel: the same residual, the same sums in the same grouping.
constexpr int kBTProbes = 16;
__global__ __launch_bounds__(256) void build_tables_multi_kernel(const float* __restrict__ queries, const float* __restrict__ coarse,
                                                                 const int32_t* __restrict__ assign,
                                                                 const float* __restrict__ codebooks, int ma, int M, int dim,
                                                                 int expansion, int sum_mode, float* __restrict__ ftables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* res = reinterpret_cast<float*>(dyn);               // [kBTProbes][dim] residuals of this query's probes a0 ..
    const int a0 = blockIdx.x * kBTProbes, qi = blockIdx.y, tid = threadIdx.x;
    const int na = min(kBTProbes, ma - a0);
    const int ds = dim / M;
    for (int i = tid; i < na * dim; i += 256) {
        const int a = i / dim, d = i - a * dim;
        const float x = queries[(size_t)qi * dim + d];
        res[i] = coarse ? x - coarse[(size_t)assign[(size_t)qi * ma + a0 + a] * dim + d] : x;
    }
    __syncthreads();
    for (int e = tid; e < M * 16; e += 256) {
        const int m = e >> 4;
        float ce[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ce[d] = d < ds ? codebooks[(size_t)e * ds + d] : 0.0f;
        // (expansion form: ||c||^2 does not depend on the probe; constant ds so that the lane loops unroll over the registers)
        const float cn = !expansion ? 0.0f : ds == 8 ? expansion_sqnorm(ce, 8, sum_mode) : ds == 4 ? expansion_sqnorm(ce, 4, sum_mode)
                                                                                          : expansion_sqnorm(ce, ds, 0);
        for (int a = 0; a < na; ++a) {
            const float* __restrict__ r = res + a * dim + m * ds;
            float s = 0.0f;
            if (expansion) {
                const float vn = ds == 8 ? expansion_sqnorm(r, 8, sum_mode) : ds == 4 ? expansion_sqnorm(r, 4, sum_mode) : expansion_sqnorm(r, ds, 0);
                float dot = 0.0f;
#pragma unroll
                for (int d = 0; d < 8; ++d)
                    if (d < ds) dot += r[d] * ce[d];
                s = (vn + cn) + (-2.0f * dot);
            } else if (ds == 8) {
                s = direct_sqdist(r, ce, 8, sum_mode);               // (constant ds: the block loop unrolls over the registers)
            } else if (ds == 4) {
                s = direct_sqdist(r, ce, 4, sum_mode);
            } else {
#pragma unroll
                for (int d = 0; d < 8; ++d)                          // other ds <= 8: no instance in the reference — sequential
                    if (d < ds) {
                        const float t_ = r[d] - ce[d];
                        s += t_ * t_;
                    }
            }
            ftables[((size_t)qi * ma + a0 + a) * (M * 16) + e] = s;
        }
    }
}

void launch_build_tables(const float* d_queries, const float* d_coarse, const int32_t* d_assign, const float* d_codebooks,
                         const float* d_rotation, int nq, int ma, int M, int dim, int expansion, int sum_mode, float* d_ftables,
                         hipStream_t stream) {
    if (!d_rotation && dim / M <= 8 && ma > 1) {
        hipLaunchKernelGGL(build_tables_multi_kernel, dim3((ma + kBTProbes - 1) / kBTProbes, nq), dim3(256),
                           (size_t)kBTProbes * dim * sizeof(float), stream, d_queries, d_coarse, d_assign, d_codebooks, ma, M, dim,
                           expansion, sum_mode, d_ftables);
        return;
    }
    hipLaunchKernelGGL(build_tables_kernel, dim3(ma, nq), dim3(256), 2 * dim * sizeof(float), stream, d_queries, d_coarse,
                       d_assign, d_codebooks, d_rotation, ma, M, dim, expansion, sum_mode, d_ftables);
}

// ---------------------------------------------------------------------------------------------
// PQ encoder (base_pq::encode_multiple_vectors, quantizers.hpp:222-245): per sub-quantizer
// find_k_neighbors(count, 16, sq_dim, k = 1, ...) (neighbors.cpp:30-76) — the BLAS-expansion distances of
// compute_cross_dists_blas (expansion_dist above) pushed in centroid order into a kv_binheap of capacity 1
// (add_candidates_heaps, 18-28; binheap.hpp:75-116): the first centroid is appended whatever its distance, a later one
// replaces it iff its distance is strictly smaller — the first strict minimum, centroid 0 when its distance is NaN.
// Two sub-quantizers per byte, the even one in the low nibble (multiple_set_bits_4, quantizers.hpp:49-68).
// form 1 = that (the reference's form; parity: the oracle's orc_pq_encode, tests/test_gpu_parity.py); form 0 = the direct
// sum (x - c)^2 in one sequential loop (this repository's encoder before round 6).  One thread per code byte.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* __restrict__ vectors, uint64_t n, int M, int dim,
                                                        const float* __restrict__ codebooks, int form, int sum_mode,
                                                        uint8_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* cb = reinterpret_cast<float*>(dyn);                // [M][16][ds]
    float* cnorm = cb + (size_t)M * 16 * (dim / M);           // [M][16] ||c||^2 (form 1)
    const int ds = dim / M, cs = M / 2;
    for (int i = threadIdx.x; i < M * 16 * ds; i += 256) cb[i] = codebooks[i];
    __syncthreads();
    for (int e = threadIdx.x; e < M * 16; e += 256) cnorm[e] = form ? expansion_sqnorm(cb + (size_t)e * ds, ds, sum_mode) : 0.0f;
    __syncthreads();
    const uint64_t total = n * (uint64_t)cs;
    for (uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (uint64_t)gridDim.x * 256) {
        const uint64_t vi = o / cs;
        const int b = (int)(o % cs);
        uint32_t packed = 0;
        for (int h = 0; h < 2; ++h) {
            const int m = 2 * b + h;
            const float* __restrict__ x = vectors + vi * dim + (uint64_t)m * ds;
            const float vn = form ? expansion_sqnorm(x, ds, sum_mode) : 0.0f;
            int best = 0;
            float bestd = 0.0f;
            for (int c = 0; c < 16; ++c) {
                const float* ce = cb + ((size_t)m * 16 + c) * ds;
                float s = 0.0f;
                if (form) {
                    s = expansion_dist(x, ce, ds, vn, cnorm[m * 16 + c]);
                } else {
                    for (int d = 0; d < ds; ++d) {
                        const float t = x[d] - ce[d];
                        s += t * t;
                    }
                }
                if (c == 0 || s < bestd) { bestd = s; best = c; }
            }
            packed |= (uint32_t)best << (4 * h);
        }
        codes[o] = (uint8_t)packed;
    }
}

void launch_pq_encode(const float* d_vectors, uint64_t n, int M, int dim, const float* d_codebooks, int form, int sum_mode,
                      uint8_t* d_codes, hipStream_t stream) {
    const uint64_t total = n * (uint64_t)(M / 2);
    const int grid = (int)std::min<uint64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(pq_encode_kernel, dim3(grid), dim3(256), ((size_t)M * 16 * (dim / M) + (size_t)M * 16) * sizeof(float), stream,
                       d_vectors, n, M, dim, d_codebooks, form, sum_mode, d_codes);
}

// ---------------------------------------------------------------------------------------------
// Database build.  index_db::add_vectors (databases.hpp:270-298): nearest coarse centroid
// (coarse_dist_kernel + coarse_select_kernel with ma = 1), then per vector the residual x - centroid and, for OPQ, the
// rotation rotated[r] = sum_c x[c] * rotation[r][c] (quantizers.hpp:289-301) in sequential float sums — the same loops as
// build_tables_kernel and host/query_driver.hpp — and pq_encode_kernel on the result.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void residual_rotate_kernel(const float* __restrict__ vectors, uint64_t n, int dim,
                                                              const float* __restrict__ coarse, const int32_t* __restrict__ assign,
                                                              const float* __restrict__ rotation, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    float* res = reinterpret_cast<float*>(dyn);                  // [dim] residual of this workgroup's vector
    for (uint64_t v = blockIdx.x; v < n; v += gridDim.x) {
        const float* __restrict__ x = vectors + v * dim;
        const float* __restrict__ c = coarse ? coarse + (size_t)assign[v] * dim : nullptr;
        __syncthreads();
        for (int d = threadIdx.x; d < dim; d += 256) res[d] = c ? x[d] - c[d] : x[d];
        __syncthreads();
        for (int r = threadIdx.x; r < dim; r += 256) {
            float o = res[r];
            if (rotation) {
                const float* __restrict__ row = rotation + (size_t)r * dim;
                o = 0.0f;
                for (int cc = 0; cc < dim; ++cc) o += res[cc] * row[cc];
            }
            out[v * dim + r] = o;
        }
    }
}

void launch_residual_rotate(const float* d_vectors, uint64_t n, int dim, const float* d_coarse, const int32_t* d_assign,
                            const float* d_rotation, float* d_out, hipStream_t stream) {
    if (!n) return;
    hipLaunchKernelGGL(residual_rotate_kernel, dim3((unsigned)std::min<uint64_t>(n, 65536)), dim3(256), (size_t)dim * sizeof(float),
                       stream, d_vectors, n, dim, d_coarse, d_assign, d_rotation, d_out);
}

// kmeans_fast_iterations_thread's centroid update (databases.cpp:67-88): centroid = (sum of its members, added in
// ascending vector order) / member count; an empty cluster divides 0 by 0 like the reference does.  One workgroup per
// centroid walks assign[] in order, compacts the members of every 256-vector chunk IN ORDER (ballot + prefix) and
// lets its lanes (one per component) add them sequentially: the float sums are those of the sequential host loop.
// div_mode 1 (default) — AS COMPILED: the reference is built with -ffast-math, and g++ turns the loop's division by the member
// count (databases.cpp:83-88) into one reciprocal 1.0f / (float)count and a multiplication per component (pinned to the
// reference's own loops compiled here: oracle/_ref, qadc_reff_kmeans_update); 0 — the source's division.
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float* __restrict__ vectors, uint64_t n, int dim,
                                                            const int32_t* __restrict__ assign, float* __restrict__ centroids,
                                                            int div_mode) {
    __shared__ uint32_t members[256];
    __shared__ uint32_t wcount[4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kMaxPer = 8;                                   // components per lane (dim <= 2048)
    float acc[kMaxPer];
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) acc[j] = 0.0f;
    uint32_t count = 0;
    for (uint64_t base = 0; base < n; base += 256) {
        const uint64_t i = base + tid;
        const bool mine = i < n && assign[i] == c;
        const uint64_t m = __builtin_amdgcn_ballot_w64(mine);
        if (lane == 0) wcount[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)), total = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) before += wcount[w];
            total += wcount[w];
        }
        if (mine) members[before] = (uint32_t)(i - base);
        __syncthreads();
        for (uint32_t k = 0; k < total; ++k) {
            const float* __restrict__ x = vectors + (base + members[k]) * dim;
#pragma unroll
            for (int j = 0; j < kMaxPer; ++j) {
                const int d = tid + 256 * j;
                if (d < dim) acc[j] += x[d];
            }
        }
        count += total;
    }
#pragma unroll
    for (int j = 0; j < kMaxPer; ++j) {
        const int d = tid + 256 * j;
        if (d < dim) centroids[(size_t)c * dim + d] = div_mode ? acc[j] * (1.0f / (float)(int)count) : acc[j] / (float)(int)count;
    }
}

void launch_kmeans_update(const float* d_vectors, uint64_t n, int dim, int K, const int32_t* d_assign, float* d_centroids,
                          int div_mode, hipStream_t stream) {
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(K), dim3(256), 0, stream, d_vectors, n, dim, d_assign, d_centroids, div_mode);
}

// ---------------------------------------------------------------------------------------------
// Synthetic codes: word w (8 code bytes) = splitmix64(seed ^ splitmix64(w)), little endian.
// Same function as orc_fill_codes in the oracle, so any sub-range is reproducible on the CPU.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void fill_codes_kernel(uint64_t* __restrict__ dst, uint64_t first_word, uint64_t nwords,
                                                         uint64_t seed) {
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * 256)
        dst[w] = splitmix64(seed ^ splitmix64(first_word + w));
}

void launch_fill_codes(uint8_t* d_dst, uint64_t first_word, uint64_t nwords, uint64_t seed, hipStream_t stream) {
    const int grid = (int)std::min<uint64_t>((nwords + 255) / 256, 8192);
    hipLaunchKernelGGL(fill_codes_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<uint64_t*>(d_dst), first_word,
                       nwords, seed);
}

// Reference block layout [B][cs][16] (simd_layout.hpp:41-65) -> device row-major [n][cs].
__global__ __launch_bounds__(256) void deinterleave_kernel(uint8_t* __restrict__ rowmajor, const uint8_t* __restrict__ inter,
                                                           uint32_t n, int cs) {
    const uint64_t total = (uint64_t)n * cs;
    for (uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (uint64_t)gridDim.x * 256) {
        const uint64_t ci = o / cs;
        const int b = (int)(o % cs);
        rowmajor[o] = inter[(ci / 16) * cs * 16 + (uint64_t)b * 16 + (ci % 16)];
    }
}

void launch_deinterleave(uint8_t* d_rowmajor, const uint8_t* d_inter, uint32_t n, int cs, hipStream_t stream) {
    const uint64_t total = (uint64_t)n * cs;
    const int grid = (int)std::min<uint64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(deinterleave_kernel, dim3(grid), dim3(256), 0, stream, d_rowmajor, d_inter, n, cs);
}

}  // namespace qadc
