/*
 * TEST INFRASTRUCTURE — CPU oracle for the Quick-ADC scan path.  NOT product code.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported baseline.  The product path
 * (quick-adc_amd/, libqadc_hip.so) never links, loads or calls anything in oracle/.
 *
 * Plain-C restatement of the reference's algorithm for the path, written from the
 * reference's behaviour (file:line cited per function, paths relative to
 * /root/reference).  Parity status:
 *   - heap, layout, int8 scan (orc_heap_*, orc_interleave, orc_scan_i8_*):
 *       PINNED — checked entry-by-entry against the reference's own kernels compiled
 *       from /root/reference into oracle/_ref/libqadc_ref.so (tests/test_oracle_vs_ref.py)
 *       and against the committed fixtures in tests/golden/ generated from that build.
 *   - float start scan, QuantizerMAX, query_scan glue, start sizes, the direct table form, scan_standard, the 4-bit
 *       packer (orc_scan4_start, orc_candidates_f32, orc_quantize_tables, orc_query_scan, orc_start_size,
 *       orc_tables_direct, orc_scan_standard_u8, orc_pack4):
 *       PINNED — checked bit for bit against the reference's own scanner_4 (whole), QuantizerMAX, scan_4,
 *       scan_standard, fmanorm / compute_dists_single_simd_cg and multiple_set_bits_4, compiled with the reference's
 *       flags from line ranges of its files into oracle/_ref/libqadc_ref_float.so (oracle/ref_extract.sh,
 *       oracle/ref_float_harness.cpp; tests/test_oracle_float_ref.py: 10^4 random queries, 1.6 M quantizer entries,
 *       every sq_dim of the reference's dispatch) and against tests/golden/ref_query_scan_cases.npz generated from
 *       that build (oracle/gen_golden_float.py).  The reference is built with -ffast-math, so "as the source reads"
 *       and "as the binary computes" differ in the grouping of float sums; where they do, mode 1 (the default) is the
 *       binary's and mode 0 the source's.
 *   - the BLAS-expansion distance form (orc_cross_dists, orc_tables_expansion; the encoder orc_pq_encode on top of it):
 *       its norm half (||v||^2 + ||c||^2 as compute_cross_dists_blas hands it to sgemm) is PINNED to the reference's text
 *       compiled up to the sgemm call (qadc_reff_cross_norms); the product half is cblas_sgemm of OpenBLAS, absent from
 *       this image: RESTATED as one sequential dot, unpinned.
 *   - NOT pinned (third-party arithmetic absent from this image): that sgemm (distances.hpp:178-182, quantizers.hpp:296),
 *       cv::kmeans — not restated here at all.
 *       (The SELECTION half of find_k_neighbors — add_candidates_heaps + kv_binheap::sort on given distances — IS pinned:
 *       orc_select_k_neighbors below against the reference's own text in oracle/_ref/libqadc_ref_float.so, exact ties
 *       included.  The .pq.data / .opq.data readers are in that library too; the tests use them directly.)
 *
 * Build: strict IEEE (no -ffast-math) so every float expression evaluates exactly as
 * written here.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------
 * kv_binheap<unsigned, T>::push  — binheap.hpp:75-116
 * Array max-heap on value, two parallel arrays.  Not full: append + sift-up with strict '>'.
 * Full: replace root iff value < root, sift-down taking the right child only if strictly
 * greater than the left, stop when child <= current.
 * ---------------------------------------------------------------------------------------- */
#define ORC_DEFINE_HEAP(SUFFIX, VT)                                                           \
    typedef struct { uint32_t* keys; VT* vals; int cap; int size; } orc_heap_##SUFFIX;        \
    static void orc_heap_##SUFFIX##_push(orc_heap_##SUFFIX* h, uint32_t key, VT value) {      \
        if (h->size != h->cap) {                                                              \
            int i = h->size++;                                                                \
            h->vals[i] = value; h->keys[i] = key;                                             \
            int p = (i - 1) / 2;                                                              \
            while (i != 0 && h->vals[i] > h->vals[p]) {                                       \
                VT tv = h->vals[i]; h->vals[i] = h->vals[p]; h->vals[p] = tv;                 \
                uint32_t tk = h->keys[i]; h->keys[i] = h->keys[p]; h->keys[p] = tk;           \
                i = p; p = (i - 1) / 2;                                                       \
            }                                                                                 \
            return;                                                                           \
        }                                                                                     \
        if (value < h->vals[0]) {                                                             \
            int i = 0;                                                                        \
            h->vals[0] = value; h->keys[0] = key;                                             \
            for (;;) {                                                                        \
                const int l = 2 * i + 1, r = 2 * i + 2;                                       \
                if (l >= h->size) break;                                                      \
                int c = l;                                                                    \
                if (r < h->size && h->vals[r] > h->vals[l]) c = r;                            \
                if (h->vals[c] <= h->vals[i]) break;                                          \
                VT tv = h->vals[i]; h->vals[i] = h->vals[c]; h->vals[c] = tv;                 \
                uint32_t tk = h->keys[i]; h->keys[i] = h->keys[c]; h->keys[c] = tk;           \
                i = c;                                                                        \
            }                                                                                 \
        }                                                                                     \
    }

ORC_DEFINE_HEAP(i8, int8_t)
ORC_DEFINE_HEAP(f32, float)

/* Exported replays: push (keys[i], vals[i]) in order into an empty heap of capacity R. */
void orc_heap_replay_i8(long n, const uint32_t* keys, const int8_t* vals, int R,
                        uint32_t* out_keys, int8_t* out_vals, int* out_size) {
    orc_heap_i8 h = { out_keys, out_vals, R, 0 };
    for (long i = 0; i < n; ++i) orc_heap_i8_push(&h, keys[i], vals[i]);
    *out_size = h.size;
}

void orc_heap_replay_f32(long n, const uint32_t* keys, const float* vals, int R,
                         uint32_t* out_keys, float* out_vals, int* out_size) {
    orc_heap_f32 h = { out_keys, out_vals, R, 0 };
    for (long i = 0; i < n; ++i) orc_heap_f32_push(&h, keys[i], vals[i]);
    *out_size = h.size;
}

/* ------------------------------------------------------------------------------------------
 * kv_binheap::sort_keys — binheap.hpp:129-137 (and sort, 118-127): std::sort of the index
 * permutation 0..size-1 with comparator values_[a] < values_[b]; keys are emitted in that order.
 * std::sort is not stable, so the order of tied values is whatever the library's algorithm yields
 * on the heap array.  The algorithm lives in a third-party dependency of the reference: libstdc++
 * (GCC 11.4 in this image; bits/stl_algo.h, bits/stl_heap.h — unchanged since GCC 4.x):
 * introsort = median-of-3 quicksort loop down to ranges of 16 with a 2*floor(log2 n) depth limit
 * (heapsort below it), then one final insertion sort.  Restated here step for step; pinned by the
 * reference build's own sort_keys output on every golden case (tests/test_oracle_golden.py).
 * ------------------------------------------------------------------------------------------ */
typedef struct { const int8_t* v; const float* f; } orc_sortctx;      /* int8 heap values, or float ones (f != NULL) */
#define ORC_LESS(c, a, b) ((c)->f ? (c)->f[(a)] < (c)->f[(b)] : (c)->v[(a)] < (c)->v[(b)])

static void orc_ss_swap(int* a, int* b) { int t = *a; *a = *b; *b = t; }

static void orc_ss_push_heap(const orc_sortctx* c, int* first, long hole, long top, int value) {
    long parent = (hole - 1) / 2;
    while (hole > top && ORC_LESS(c, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void orc_ss_adjust_heap(const orc_sortctx* c, int* first, long hole, long len, int value) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (ORC_LESS(c, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    orc_ss_push_heap(c, first, hole, top, value);
}

static void orc_ss_heapsort(const orc_sortctx* c, int* first, int* last) {   /* __partial_sort(first, last, last) */
    const long len = last - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; --parent) {                        /* __make_heap */
            orc_ss_adjust_heap(c, first, parent, len, first[parent]);
            if (parent == 0) break;
        }
    while (last - first > 1) {                                                /* __sort_heap / __pop_heap */
        --last;
        const int value = *last;
        *last = *first;
        orc_ss_adjust_heap(c, first, 0, last - first, value);
    }
}

static void orc_ss_median_to_first(const orc_sortctx* c, int* result, int* a, int* b, int* d) {
    if (ORC_LESS(c, *a, *b)) {
        if (ORC_LESS(c, *b, *d)) orc_ss_swap(result, b);
        else if (ORC_LESS(c, *a, *d)) orc_ss_swap(result, d);
        else orc_ss_swap(result, a);
    } else if (ORC_LESS(c, *a, *d)) orc_ss_swap(result, a);
    else if (ORC_LESS(c, *b, *d)) orc_ss_swap(result, d);
    else orc_ss_swap(result, b);
}

static int* orc_ss_partition(const orc_sortctx* c, int* first, int* last, int* pivot) {   /* __unguarded_partition */
    for (;;) {
        while (ORC_LESS(c, *first, *pivot)) ++first;
        --last;
        while (ORC_LESS(c, *pivot, *last)) --last;
        if (!(first < last)) return first;
        orc_ss_swap(first, last);
        ++first;
    }
}

static void orc_ss_introsort_loop(const orc_sortctx* c, int* first, int* last, long depth) {
    while (last - first > 16) {
        if (depth == 0) {
            orc_ss_heapsort(c, first, last);
            return;
        }
        --depth;
        int* mid = first + (last - first) / 2;
        orc_ss_median_to_first(c, first, first + 1, mid, last - 1);
        int* cut = orc_ss_partition(c, first + 1, last, first);
        orc_ss_introsort_loop(c, cut, last, depth);
        last = cut;
    }
}

static void orc_ss_linear_insert(const orc_sortctx* c, int* last) {          /* __unguarded_linear_insert */
    const int val = *last;
    int* next = last - 1;
    while (ORC_LESS(c, val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

static void orc_ss_insertion_sort(const orc_sortctx* c, int* first, int* last) {
    if (first == last) return;
    for (int* i = first + 1; i != last; ++i) {
        if (ORC_LESS(c, *i, *first)) {
            const int val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(int));
            *first = val;
        } else {
            orc_ss_linear_insert(c, i);
        }
    }
}

static void orc_ss_std_sort(const orc_sortctx* c, int* perm, int size) {       /* std::sort(perm, perm + size, comp) */
    if (size <= 0) return;
    long lg = 0;
    for (long n = size; n > 1; n >>= 1) ++lg;                                 /* std::__lg */
    orc_ss_introsort_loop(c, perm, perm + size, lg * 2);
    if (size > 16) {                                                          /* __final_insertion_sort */
        orc_ss_insertion_sort(c, perm, perm + 16);
        for (int* i = perm + 16; i != perm + size; ++i) orc_ss_linear_insert(c, i);
    } else {
        orc_ss_insertion_sort(c, perm, perm + size);
    }
}

/* out_keys[i] = heap_keys[perm[i]] for the std::sort-ed permutation of a heap array of `size` entries. */
void orc_sort_keys_i8(int size, const uint32_t* heap_keys, const int8_t* heap_vals, uint32_t* out_keys) {
    if (size <= 0) return;
    int* perm = (int*)malloc(sizeof(int) * (size_t)size);
    for (int i = 0; i < size; ++i) perm[i] = i;
    const orc_sortctx c = { heap_vals, NULL };
    orc_ss_std_sort(&c, perm, size);
    for (int i = 0; i < size; ++i) out_keys[i] = heap_keys[perm[i]];
    free(perm);
}

/* ------------------------------------------------------------------------------------------
 * The SELECTION half of find_k_neighbors — neighbors.cpp:18-28 (add_candidates_heaps), 47-71: per vector a
 * kv_binheap<int, float> of capacity k takes the neighbour distances in index order (binheap.hpp:75-116, the same push
 * as the int8 heap's), then kv_binheap::sort (118-127) writes keys and values in std::sort's order of the permutation.
 * The distances are GIVEN (the reference's come from cblas_sgemm).  Pinned to the reference's own text
 * (oracle/_ref/libqadc_ref_float.so: qadc_reff_select_k_neighbors) by tests/test_oracle_float_ref.py, exact ties included.
 * ---------------------------------------------------------------------------------------- */
void orc_select_k_neighbors(const float* dists, long count, int neighbor_count, int k, int32_t* assign, float* sorted) {
    float* hv = (float*)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    int* hk = (int*)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
    int* perm = (int*)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
    for (long v = 0; v < count; ++v) {
        const float* d = dists + (size_t)v * neighbor_count;
        int size = 0;
        for (int n = 0; n < neighbor_count; ++n) {
            const float value = d[n];
            if (size != k) {                                                  /* binheap.hpp:78-90 */
                int index = size++;
                hv[index] = value; hk[index] = n;
                int parent = (index - 1) / 2;
                while (index != 0 && hv[index] > hv[parent]) {
                    const float tv = hv[index]; hv[index] = hv[parent]; hv[parent] = tv;
                    const int tk = hk[index]; hk[index] = hk[parent]; hk[parent] = tk;
                    index = parent;
                    parent = (index - 1) / 2;
                }
            } else if (value < hv[0]) {                                       /* 93-115 */
                int index = 0;
                hv[0] = value; hk[0] = n;
                for (;;) {
                    const int left = 2 * index + 1, right = 2 * index + 2;
                    if (left >= size) break;
                    int largest = left;
                    if (right < size && hv[right] > hv[left]) largest = right;
                    if (hv[largest] <= hv[index]) break;
                    const float tv = hv[index]; hv[index] = hv[largest]; hv[largest] = tv;
                    const int tk = hk[index]; hk[index] = hk[largest]; hk[largest] = tk;
                    index = largest;
                }
            }
        }
        for (int i = 0; i < size; ++i) perm[i] = i;
        const orc_sortctx c = { NULL, hv };
        orc_ss_std_sort(&c, perm, size);
        for (int i = 0; i < size; ++i) {
            assign[(size_t)v * k + i] = hk[perm[i]];
            if (sorted) sorted[(size_t)v * k + i] = hv[perm[i]];
        }
    }
    free(hv); free(hk); free(perm);
}

/* ------------------------------------------------------------------------------------------
 * multiple_set_bits_4 — quantizers.hpp:49-68.  assign is [n][M] centroid ids (0..15);
 * byte b of a code: low nibble = sub-quantizer 2b, high nibble = 2b+1.
 * ---------------------------------------------------------------------------------------- */
void orc_pack4(const int32_t* assign, long n, int M, uint8_t* codes) {
    const int cs = M / 2;
    for (long i = 0; i < n; ++i)
        for (int m = 0; m < M; ++m) {
            const uint8_t a = (uint8_t)assign[i * M + m];
            uint8_t* c = codes + i * cs + m / 2;
            if (m % 2 == 1) *c = (uint8_t)(*c | (uint8_t)(a << 4));
            else            *c = a;
        }
}

/* ------------------------------------------------------------------------------------------
 * simd_layout.hpp:31-65 — row-major [n][cs] -> [B][cs][16]; lanes past n replicate code n-1.
 * ---------------------------------------------------------------------------------------- */
long orc_interleaved_size(uint32_t n, int cs) {
    return (long)((n + 15u) / 16u) * cs * 16;
}

void orc_interleave(uint8_t* dst, const uint8_t* rowmajor, uint32_t n, int cs) {
    const long blocks = (n + 15u) / 16u;
    for (long k = 0; k < blocks; ++k)
        for (int b = 0; b < cs; ++b)
            for (int j = 0; j < 16; ++j) {
                long ci = k * 16 + j;
                if (ci >= (long)n) ci = (long)n - 1;
                *dst++ = rowmajor[ci * cs + b];
            }
}

/* Inverse view used by tests: block layout -> row-major (first n codes). */
void orc_deinterleave(uint8_t* rowmajor, const uint8_t* inter, uint32_t n, int cs) {
    for (long ci = 0; ci < (long)n; ++ci)
        for (int b = 0; b < cs; ++b)
            rowmajor[ci * cs + b] = inter[(ci / 16) * cs * 16 + b * 16 + (ci % 16)];
}

static inline int8_t sat_add8(int8_t a, int8_t b) {    /* _mm256_adds_epi8 */
    int s = (int)a + (int)b;
    return (int8_t)(s > 127 ? 127 : (s < -128 ? -128 : s));
}

/* ------------------------------------------------------------------------------------------
 * scan_avx_4<M> + compare_extract_matches_sse + bh_push — simd_scan.hpp:63-187.
 * Works on the interleaved layout, block by block, with the reference's exact
 * saturating-add order:
 *   lo-lane chain: T[0][p0.lo] (+) T[1][p0.hi] (+) T[4][p2.lo] (+) T[5][p2.hi] ...   (138-174)
 *   hi-lane chain: T[2][p1.lo] (+) T[3][p1.hi] (+) T[6][p3.lo] (+) T[7][p3.hi] ...
 *   cand = hi (+) lo                                                                 (177-179)
 * Signed compare against the bound sampled at block start; matches pushed in ascending
 * lane order with index clamped to size-1 (padding quirk, 67); bound refreshed after a
 * block with at least one match (116).
 * qt: int8 [M][16].
 * ---------------------------------------------------------------------------------------- */
static void scan_i8_blocks(orc_heap_i8* h, int M, const uint8_t* part, const uint32_t* labels,
                           uint32_t size, const int8_t* qt) {
    const int cs = M / 2, rows = M / 4;
    int8_t bound = h->vals[0];
    const uint32_t max_scan = size - 1u;
    uint32_t scanned = 0;
    while (scanned <= max_scan) {
        int8_t cand[16];
        for (int j = 0; j < 16; ++j) {
            int8_t lo = 0, hi = 0;
            for (int r = 0; r < rows; ++r) {
                const uint8_t p0 = part[(2 * r) * 16 + j], p1 = part[(2 * r + 1) * 16 + j];
                const int8_t a = qt[(4 * r) * 16 + (p0 & 15)], b = qt[(4 * r + 1) * 16 + (p0 >> 4)];
                const int8_t c = qt[(4 * r + 2) * 16 + (p1 & 15)], d = qt[(4 * r + 3) * 16 + (p1 >> 4)];
                if (r == 0) { lo = sat_add8(b, a); hi = sat_add8(d, c); }
                else { lo = sat_add8(a, lo); lo = sat_add8(b, lo); hi = sat_add8(c, hi); hi = sat_add8(d, hi); }
            }
            cand[j] = sat_add8(hi, lo);
        }
        int any = 0;
        for (int j = 0; j < 16; ++j) {
            if (cand[j] < bound) {
                uint32_t ci = scanned + (uint32_t)j;
                if (ci > max_scan) ci = max_scan;
                orc_heap_i8_push(h, labels ? labels[ci] : ci, cand[j]);
                any = 1;
            }
        }
        if (any) bound = h->vals[0];
        scanned += 16;
        part += (long)cs * 16;
    }
}

/* Row-major statement of the same scan (SURVEY.md §8 A1 facts i-iii): for entries in
 * [0,127] cand == min(127, sum); per lane in block order push iff cand < current bound;
 * the last block's padding lanes replay code n-1.  Used for large inputs / CPU "port" timing.
 * Identical results to scan_i8_blocks whenever all table entries are in [0,127]. */
static void scan_i8_rowmajor(orc_heap_i8* h, int M, const uint8_t* codes, const uint32_t* labels,
                             uint32_t size, const int8_t* qt) {
    const int cs = M / 2;
    const uint32_t blocks = (size + 15u) / 16u;
    int8_t bound = h->vals[0];
    for (uint32_t k = 0; k < blocks; ++k) {
        int any = 0;
        const int8_t bound_blk = bound;
        for (int j = 0; j < 16; ++j) {
            uint32_t ci = k * 16u + (uint32_t)j;
            if (ci > size - 1u) ci = size - 1u;
            const uint8_t* c = codes + (long)ci * cs;
            int s = 0;
            for (int b = 0; b < cs; ++b)
                s += qt[(2 * b) * 16 + (c[b] & 15)] + qt[(2 * b + 1) * 16 + (c[b] >> 4)];
            const int8_t cand = (int8_t)(s > 127 ? 127 : s);
            if (cand < bound_blk) { orc_heap_i8_push(h, labels ? labels[ci] : ci, cand); any = 1; }
        }
        if (any) bound = h->vals[0];
    }
}

/* Integer half of scanner_4::query_scan (db_query_4.cpp:276, 287-308): optional (0,127)
 * sentinel, then every probed partition in order into one heap.
 * layout: 0 = parts[] are interleaved (reference layout), 1 = parts[] are row-major. */
int orc_scan_i8(int M, int nparts, const uint8_t* const* parts, const uint32_t* const* labels,
                const uint32_t* sizes, const int8_t* qtables, int R, int push_sentinel, int layout,
                uint32_t* out_keys, int8_t* out_vals, int* out_size) {
    if (M != 16 && M != 32) return -1;
    orc_heap_i8 h = { out_keys, out_vals, R, 0 };
    if (push_sentinel) orc_heap_i8_push(&h, 0, 127);
    for (int p = 0; p < nparts; ++p) {
        if (sizes[p] == 0) continue;
        const int8_t* qt = qtables + (long)p * M * 16;
        const uint32_t* lab = labels ? labels[p] : NULL;
        if (layout == 0) scan_i8_blocks(&h, M, parts[p], lab, sizes[p], qt);
        else             scan_i8_rowmajor(&h, M, parts[p], lab, sizes[p], qt);
    }
    *out_size = h.size;
    return 0;
}

/* Per-code candidate values min(127, sum) for row-major codes (test helper: what the
 * device must compute for every code; SURVEY.md §8 A1 fact i). */
void orc_candidates_i8(int M, const uint8_t* codes, long n, const int8_t* qt, int8_t* out) {
    const int cs = M / 2;
    for (long i = 0; i < n; ++i) {
        int s = 0;
        for (int b = 0; b < cs; ++b)
            s += qt[(2 * b) * 16 + (codes[i * cs + b] & 15)] + qt[(2 * b + 1) * 16 + (codes[i * cs + b] >> 4)];
        out[i] = (int8_t)(s > 127 ? 127 : s);
    }
}

/* ------------------------------------------------------------------------------------------
 * scan_4<NSQ> — query_common.hpp:59-90.  Float ADC on row-major 4-bit codes; push iff
 * cand < min, min refreshed after every push (81-88).  The SOURCE accumulates sequentially
 * from 0, byte by byte, low nibble then high nibble (72-80); the reference is built with
 * -ffast-math, which lets the compiler re-associate that sum, and it does.
 *
 * sum_mode 1 (default) = the association of the reference AS COMPILED HERE (g++ 11.4,
 *   -O3 -ffast-math; same with -march=native and with the explicit ISA list), read off the
 *   disassembly of the stand-alone scan_4<16> / scan_4<32> instances in
 *   oracle/_ref/libqadc_ref_float.so and pinned to that binary by tests/test_oracle_float_ref.py.
 *   With L_b = dists[(2b)*16 + lo(byte b)], H_b = dists[(2b+1)*16 + hi(byte b)]:
 *     A = (H2+L3)+(H3+L4)   B = (H0+L1)+(H1+L2)   C = (H5+L6)+(H4+L5)   D = (H6+L7)+(H7+L0)
 *     s = ((A+B)+C)+D                                   bytes 0..7 (all of NSQ=16)
 *     s = s + ((L_{b+1}+H_{b+1}) + (L_b+H_b))           b = 8, 10, 12, 14   (NSQ=32 only)
 *   (IEEE addition is commutative, so only the grouping matters.)
 * sum_mode 0 = source order.
 * ---------------------------------------------------------------------------------------- */
static inline float adc4_f32(int M, const uint8_t* c, const float* dists, int sum_mode) {
    const int cs = M / 2;
    if (sum_mode == 0) {
        float cand = 0;
        for (int b = 0; b < cs; ++b) {
            cand += dists[(2 * b) * 16 + (c[b] & 0xf)];
            cand += dists[(2 * b + 1) * 16 + ((c[b] & 0xf0) >> 4)];
        }
        return cand;
    }
    float L[16], H[16];
    for (int b = 0; b < cs; ++b) {
        L[b] = dists[(2 * b) * 16 + (c[b] & 0xf)];
        H[b] = dists[(2 * b + 1) * 16 + (c[b] >> 4)];
    }
    const float A = (H[2] + L[3]) + (H[3] + L[4]);
    const float B = (H[0] + L[1]) + (H[1] + L[2]);
    const float Cq = (H[5] + L[6]) + (H[4] + L[5]);
    const float D = (H[6] + L[7]) + (H[7] + L[0]);
    float s = ((A + B) + Cq) + D;
    for (int b = 8; b < cs; b += 2)
        s = s + ((L[b + 1] + H[b + 1]) + (L[b] + H[b]));
    return s;
}

static void scan4_f32(orc_heap_f32* h, int M, const uint8_t* codes, const uint32_t* labels,
                      uint32_t count, const float* dists, int sum_mode) {
    const int cs = M / 2;
    float min = h->vals[0];
    for (uint32_t i = 0; i < count; ++i) {
        const float cand = adc4_f32(M, codes + (long)i * cs, dists, sum_mode);
        if (cand < min) {
            orc_heap_f32_push(h, labels ? labels[i] : i, cand);
            min = h->vals[0];
        }
    }
}

/* Per-code float ADC values in either summation order (test helper). */
void orc_candidates_f32_mode(int M, const uint8_t* codes, long n, const float* dists, int sum_mode, float* out) {
    const int cs = M / 2;
    for (long i = 0; i < n; ++i) out[i] = adc4_f32(M, codes + i * cs, dists, sum_mode);
}
void orc_candidates_f32(int M, const uint8_t* codes, long n, const float* dists, float* out) {
    orc_candidates_f32_mode(M, codes, n, dists, 1, out);
}

/* query_scan_start alone (db_query_4.cpp:230-242): push (0, FLT_MAX), then scan_4 over each run with its table. */
int orc_scan4_start(int M, int nparts, const uint8_t* const* parts, const uint32_t* const* labels,
                    const uint32_t* sizes, const float* tables, int R, int sum_mode,
                    uint32_t* out_keys, float* out_vals, int* out_size) {
    orc_heap_f32 h = { out_keys, out_vals, R, 0 };
    orc_heap_f32_push(&h, 0, FLT_MAX);
    for (int p = 0; p < nparts; ++p)
        scan4_f32(&h, M, parts[p], labels ? labels[p] : NULL, sizes[p], tables + (long)p * M * 16, sum_mode);
    *out_size = h.size;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * scan_standard<uint8_t,NSQ> — query_common.hpp:92-118 (BASELINE config 1: PQ 8x8 float ADC),
 * preceded by scanner_simple::query_scan's R sentinel pushes (db_query.cpp:32-34).
 * codes row-major [n][NSQ] bytes, dists [NSQ][256].  The source adds t_m = dists[m*256 + code[m]]
 * sequentially from 0; sum_mode 1 = the grouping of the reference as compiled here (same build and
 * same pin as scan_4 above; instances NSQ = 4, 8, 16 exist in the reference, query_common.hpp:126-131):
 *   NSQ 4:  (t1+t2) + (t3+t0)
 *   NSQ 8:  ((t1+t2)+(t3+t4)) + ((t5+t6)+(t7+t0))
 *   NSQ 16: ((A+B)+C)+D,  A = (t5+t6)+(t7+t8)  B = (t1+t2)+(t3+t4)  C = (t11+t12)+(t9+t10)  D = (t13+t14)+(t15+t0)
 *           (the 16-term grouping of scan_4<16>, whose terms are L0,H0,L1,H1,...)
 * sum_mode 0 (or another NSQ) = source order.
 * ---------------------------------------------------------------------------------------- */
static inline float adc8_f32(int NSQ, const uint8_t* c, const float* dists, int sum_mode) {
    float t[16] = {0};
    if (sum_mode == 0 || !(NSQ == 4 || NSQ == 8 || NSQ == 16)) {
        float cand = 0;
        for (int m = 0; m < NSQ; ++m) cand += dists[m * 256 + c[m]];
        return cand;
    }
    for (int m = 0; m < NSQ; ++m) t[m] = dists[m * 256 + c[m]];
    if (NSQ == 4) return (t[1] + t[2]) + (t[3] + t[0]);
    if (NSQ == 8) return ((t[1] + t[2]) + (t[3] + t[4])) + ((t[5] + t[6]) + (t[7] + t[0]));
    const float A = (t[5] + t[6]) + (t[7] + t[8]);
    const float B = (t[1] + t[2]) + (t[3] + t[4]);
    const float Cq = (t[11] + t[12]) + (t[9] + t[10]);
    const float D = (t[13] + t[14]) + (t[15] + t[0]);
    return ((A + B) + Cq) + D;
}

int orc_scan_standard_u8(int NSQ, int nparts, const uint8_t* const* parts, const uint32_t* const* labels,
                         const uint32_t* sizes, const float* tables, int R, int sum_mode,
                         uint32_t* out_keys, float* out_vals, int* out_size) {
    orc_heap_f32 h = { out_keys, out_vals, R, 0 };
    for (int t = 0; t < R; ++t) orc_heap_f32_push(&h, 0, FLT_MAX - (float)t);
    for (int p = 0; p < nparts; ++p) {
        const float* dists = tables + (long)p * NSQ * 256;
        const uint32_t* lab = labels ? labels[p] : NULL;
        float min = h.vals[0];
        for (uint32_t i = 0; i < sizes[p]; ++i) {
            const float cand = adc8_f32(NSQ, parts[p] + (long)i * NSQ, dists, sum_mode);
            if (cand < min) {
                orc_heap_f32_push(&h, lab ? lab[i] : i, cand);
                min = h.vals[0];
            }
        }
    }
    *out_size = h.size;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * QuantizerMAX<int8_t> — db_query_4.cpp:37-71.  val >= max -> 127, else trunc toward zero.
 * mode 0: source-level  int8(((val - min) / delta)),  delta = (max - min) / 127   (44-55)
 * mode 1: as compiled by g++ 11.4 with the reference's flags (-O3 -ffast-math): one
 *         scale = 127.0f / (max - min), then int8((val - min) * scale)  (SURVEY.md §8 A5 [probe]).
 * The two can differ by 1 at bucket edges.  PARITY UNPINNED (see header).
 * ---------------------------------------------------------------------------------------- */
void orc_quantize_tables(const float* tables, long count, float qmin, float qmax, int mode, int8_t* out) {
    const float delta = (qmax - qmin) / 127;
    const float scale = 127.0f / (qmax - qmin);
    for (long i = 0; i < count; ++i) {
        const float v = tables[i];
        if (v >= qmax) { out[i] = 127; continue; }
        const float q = mode == 0 ? (v - qmin) / delta : (v - qmin) * scale;
        out[i] = (int8_t)(int)q;
    }
}

/* ------------------------------------------------------------------------------------------
 * Direct table form: compute_dists_single_simd_cg<DSQ> -> fmanorm<DSQ/8, DSQ%8>(subvector, centroid)
 * (distances.hpp:294-311, 60-76, 27-36).  centroids [M][16][dsq], vector [M*dsq], dists [M*16].
 *
 * sum_mode 1 = the reference AS COMPILED HERE (pinned to oracle/_ref/libqadc_ref_float.so for every
 *   sq_dim of the reference's dispatch, distances.cpp:50-84, by tests/test_oracle_float_ref.py):
 *     vector part  acc[j] = fma(d, d, acc[j]) per AVX lane j over the DSQ/8 blocks, d = x - c, acc from 0
 *                  (the first fma is r(d*d) exactly), then reduceadd's tree:
 *                  r[j] = acc[j] + acc[j+4] (j < 4);  (r[0] + r[2]) + (r[1] + r[3])
 *     remainder    d = c - x, g++ -ffast-math -mfma pairs the scalar loop `norm += d*d` as
 *                  p_k = fma(d_2k, d_2k, r(d_{2k+1} * d_{2k+1})) and groups
 *                    REM 4 (sq_dim 4, 60):  (p0 + p1) + vec          (vec = 0 for sq_dim 4)
 *                    REM 6 (sq_dim 30):     (vec + p2) + (p0 + p1)
 *   Other remainders have no instance in the reference (sq_dim 3 of BASELINE configs[4] is not in its
 *   dispatch): they take sum_mode 0's loop.  Returns 1 when the as-compiled form was used, else 0.
 * sum_mode 0 = one sequential float loop s += (x - c)^2 in ascending d, no fused multiply-add (this
 *   repository's own order from before the float half was pinned).
 * ---------------------------------------------------------------------------------------- */
static float tables_direct_seq(int dsq, const float* x, const float* c) {
    float s = 0;
    for (int d = 0; d < dsq; ++d) {
        const float t = x[d] - c[d];
        const float sq = t * t;
        s = s + sq;
    }
    return s;
}

static float tables_direct_compiled(int dsq, const float* x, const float* c) {
    const int blocks = dsq / 8, rem = dsq % 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int j = 0; j < 8; ++j) {
            const float d = x[b * 8 + j] - c[b * 8 + j];
            acc[j] = fmaf(d, d, acc[j]);
        }
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    const float vec = (r0 + r2) + (r1 + r3);
    if (rem == 0) return vec;
    float p[3] = {0, 0, 0};
    for (int k = 0; k < rem / 2; ++k) {
        const float d0 = c[blocks * 8 + 2 * k] - x[blocks * 8 + 2 * k];
        const float d1 = c[blocks * 8 + 2 * k + 1] - x[blocks * 8 + 2 * k + 1];
        const float sq1 = d1 * d1;
        p[k] = fmaf(d0, d0, sq1);
    }
    if (rem == 4) return (p[0] + p[1]) + vec;
    return (vec + p[2]) + (p[0] + p[1]);                      /* rem == 6 */
}

int orc_tables_direct(int dsq, int M, const float* centroids, const float* vector, int sum_mode, float* dists) {
    const int rem = dsq % 8;
    const int compiled = sum_mode != 0 && (rem == 0 || rem == 4 || rem == 6);
    for (int m = 0; m < M; ++m)
        for (int cent = 0; cent < 16; ++cent) {
            const float* x = vector + (long)m * dsq;
            const float* c = centroids + ((long)m * 16 + cent) * dsq;
            dists[m * 16 + cent] = compiled ? tables_direct_compiled(dsq, x, c) : tables_direct_seq(dsq, x, c);
        }
    return compiled;
}

/* ------------------------------------------------------------------------------------------
 * The BLAS-expansion distance form — compute_cross_dists_blas<DSQ>, distances.hpp:151-183 (and its <4> specialisation,
 * 185-215): dists[v][c] = ||v||^2 + ||c||^2 first (153-176), then cblas_sgemm(alpha = -2, beta = 1) adds -2 v.c (178-182).
 * It is what find_k_neighbors feeds its heaps with (neighbors.cpp:42, 58-59: the encoder, quantizers.hpp:222-245, and the
 * coarse assignment) and what compute_dists_multiple_blas_cg builds tables from (277-292: every ma > 1 query, every
 * batch).
 *   norms (PINNED to the reference's own text compiled up to the sgemm call: oracle/_ref qadc_reff_cross_norms,
 *     tests/test_oracle_float_ref.py): sum_mode 1 = fmanorm<DSQ/8, DSQ%8>(vec) / norm_4(vec) AS COMPILED — the grouping of
 *     tables_direct_compiled above with c = 0 (AVX lanes by fma, reduceadd's tree, remainder pairs
 *     p_k = fma(x_2k, x_2k, r(x_2k+1^2)); norm_4 = p0 + p1), for the reference's cent_count of 16 (with a centroid
 *     count that is not a multiple of 8 g++'s vectorised DSQ = 4 norm loop groups its tail centroids differently — the
 *     quantizers always have 16); sum_mode 0 or a remainder outside {0, 4, 6} = one sequential loop.
 *   product (RESTATED, unpinned: OpenBLAS is not in this image): one sequential dot in ascending d, then
 *     fma(-2, dot, norms) — one rounding, as a gemm micro-kernel's C += alpha * acc.
 * ---------------------------------------------------------------------------------------- */
float orc_sqnorm(int dsq, const float* x, int sum_mode) {
    const int blocks = dsq / 8, rem = dsq % 8;
    if (sum_mode == 0 || !(rem == 0 || rem == 4 || rem == 6)) {
        float s = 0;
        for (int d = 0; d < dsq; ++d) {
            const float sq = x[d] * x[d];
            s = s + sq;
        }
        return s;
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(x[b * 8 + j], x[b * 8 + j], acc[j]);
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    const float vec = (r0 + r2) + (r1 + r3);
    if (rem == 0) return vec;
    float p[3] = {0, 0, 0};
    for (int k = 0; k < rem / 2; ++k) {
        const float sq1 = x[blocks * 8 + 2 * k + 1] * x[blocks * 8 + 2 * k + 1];
        p[k] = fmaf(x[blocks * 8 + 2 * k], x[blocks * 8 + 2 * k], sq1);
    }
    if (rem == 4) return (p[0] + p[1]) + vec;
    return (vec + p[2]) + (p[0] + p[1]);                      /* rem == 6 */
}

/* dists[v * dists_dim + c], v < vec_count, c < cent_count; with_product 0 = the matrix as sgemm receives it (norms only). */
void orc_cross_dists(int dsq, const float* centroids, int cent_count, const float* vectors, long vec_count, long dists_dim,
                     int sum_mode, int with_product, float* dists) {
    float* cn = (float*)malloc(sizeof(float) * (size_t)(cent_count > 0 ? cent_count : 1));
    for (int c = 0; c < cent_count; ++c) cn[c] = orc_sqnorm(dsq, centroids + (size_t)c * dsq, sum_mode);
    for (long v = 0; v < vec_count; ++v) {
        const float* x = vectors + (size_t)v * dsq;
        const float vn = orc_sqnorm(dsq, x, sum_mode);
        for (int c = 0; c < cent_count; ++c) {
            const float base = vn + cn[c];
            float out = base;
            if (with_product) {
                const float* ce = centroids + (size_t)c * dsq;
                float dot = 0;
                for (int d = 0; d < dsq; ++d) {
                    const float pr = x[d] * ce[d];
                    dot = dot + pr;
                }
                out = fmaf(-2.0f, dot, base);
            }
            dists[(size_t)v * dists_dim + c] = out;
        }
    }
    free(cn);
}

/* compute_dists_multiple_blas_cg (distances.hpp:277-292): tables [count][M*16] of `count` vectors [count][M*dsq]. */
void orc_tables_expansion(int dsq, int M, const float* centroids, const float* vectors, long count, int sum_mode, float* dists) {
    float* sub = (float*)malloc(sizeof(float) * (size_t)(count > 0 ? count : 1) * dsq);
    for (int m = 0; m < M; ++m) {
        for (long v = 0; v < count; ++v)                                     /* extract_subvectors, quantizers.hpp:86-94 */
            memcpy(sub + (size_t)v * dsq, vectors + (size_t)v * M * dsq + (size_t)m * dsq, sizeof(float) * dsq);
        orc_cross_dists(dsq, centroids + (size_t)m * 16 * dsq, 16, sub, count, (long)M * 16, sum_mode, 1, dists + m * 16);
    }
    free(sub);
}

/* ------------------------------------------------------------------------------------------
 * base_pq::encode_multiple_vectors — quantizers.hpp:222-245 (opq: rotate_multiple_vectors first, 289-301): per
 * sub-quantizer extract the sub-vectors (86-94), find_k_neighbors(count, 16, sq_dim, k = 1, ...) (neighbors.cpp:30-76:
 * the expansion distances of orc_cross_dists, pushed in centroid order into a capacity-1 kv_binheap — the first strict
 * minimum, the first centroid when its distance is NaN), multiple_set_bits_4 (49-68).
 *   form 1 = that (the reference's); form 0 = the direct form sum (x - c)^2 in one sequential loop, first minimum (this
 *   repository's encoder before round 6, kept as an option).
 * rotation (nullable) [dim][dim]: rotated[r] = sum_c x[c] * rotation[r][c], one sequential float sum (the reference's is
 * a cblas_sgemm: restated, unpinned).  codebooks [M][16][dim/M]; vectors [n][dim]; codes [n][M/2].
 * ---------------------------------------------------------------------------------------- */
void orc_pq_encode(int M, int dim, const float* codebooks, const float* rotation, const float* vectors, long n, int form,
                   int sum_mode, uint8_t* codes) {
    const int dsq = dim / M;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    float* rot = NULL;
    if (rotation) {
        rot = (float*)malloc(sizeof(float) * nn * dim);
        for (long v = 0; v < n; ++v)
            for (int r = 0; r < dim; ++r) {
                float acc = 0;
                for (int c = 0; c < dim; ++c) {
                    const float pr = vectors[(size_t)v * dim + c] * rotation[(size_t)r * dim + c];
                    acc = acc + pr;
                }
                rot[(size_t)v * dim + r] = acc;
            }
        vectors = rot;
    }
    float* sub = (float*)malloc(sizeof(float) * nn * dsq);
    float* dists = (float*)malloc(sizeof(float) * nn * 16);
    int32_t* assign = (int32_t*)malloc(sizeof(int32_t) * nn * M);
    int32_t* a1 = (int32_t*)malloc(sizeof(int32_t) * nn);
    for (int m = 0; m < M; ++m) {
        const float* cb = codebooks + (size_t)m * 16 * dsq;
        for (long v = 0; v < n; ++v) memcpy(sub + (size_t)v * dsq, vectors + (size_t)v * dim + (size_t)m * dsq, sizeof(float) * dsq);
        if (form == 1) {
            orc_cross_dists(dsq, cb, 16, sub, n, 16, sum_mode, 1, dists);
        } else {
            for (long v = 0; v < n; ++v)
                for (int c = 0; c < 16; ++c) dists[(size_t)v * 16 + c] = tables_direct_seq(dsq, sub + (size_t)v * dsq, cb + (size_t)c * dsq);
        }
        orc_select_k_neighbors(dists, n, 16, 1, a1, NULL);
        for (long v = 0; v < n; ++v) assign[(size_t)v * M + m] = a1[v];
    }
    orc_pack4(assign, n, M, codes);
    free(sub); free(dists); free(assign); free(a1); free(rot);
}

/* starts size — db_query_4.cpp:125-126: max(1u, unsigned(size * keep)), product in float. */
uint32_t orc_start_size(uint32_t size, float keep) {
    if (size == 0) return 0;
    const uint32_t s = (uint32_t)((float)size * keep);
    return s > 1u ? s : 1u;
}

/* ------------------------------------------------------------------------------------------
 * scanner_4::query_scan — db_query_4.cpp:230-309, on row-major partitions of the database.
 *   all_parts / all_labels / all_sizes : the database's partitions (labels NULL = flat)
 *   assign[ma], tables[ma][M*16] (MUTABLE: negatives are clamped in place, 262-269)
 * Returns 0, or 1 when qmax > 1e30 (the reference prints a warning and exit(1)s, 271-274).
 * Outputs: qmin/qmax, the int8 tables [ma][M][16], and the final heap arrays.
 * ---------------------------------------------------------------------------------------- */
int orc_query_scan(int M, const uint8_t* const* all_parts, const uint32_t* const* all_labels,
                   const uint32_t* all_sizes, float keep, const int32_t* assign, int ma,
                   float* tables, int R, int quant_mode, int sum_mode,
                   float* out_qmin, float* out_qmax, int8_t* out_qtables,
                   uint32_t* out_keys, int8_t* out_vals, int* out_size) {
    const int table_dim = M * 16;
    /* query_scan_start (230-242) */
    uint32_t* tk = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)R);
    float* tv = (float*)malloc(sizeof(float) * (size_t)R);
    orc_heap_f32 th = { tk, tv, R, 0 };
    orc_heap_f32_push(&th, 0, FLT_MAX);
    for (int a = 0; a < ma; ++a) {
        const int p = assign[a];
        const uint32_t s = orc_start_size(all_sizes[p], keep);
        scan4_f32(&th, M, all_parts[p], all_labels ? all_labels[p] : NULL, s, tables + (long)a * table_dim, sum_mode);
    }
    float qmax = th.vals[0];
    free(tk); free(tv);
    /* qmin + clamp (258-269) */
    const long all = (long)ma * table_dim;
    float qmin = tables[0];
    for (long i = 1; i < all; ++i) if (tables[i] < qmin) qmin = tables[i];
    if (qmin < 0) {
        qmin = 0;
        for (long i = 0; i < all; ++i) if (tables[i] < 0) tables[i] = 0;
    }
    *out_qmin = qmin; *out_qmax = qmax;
    if ((double)qmax > 1e30) return 1;                           /* db_query_4.cpp:271 compares the float with a double literal */
    orc_quantize_tables(tables, all, qmin, qmax, quant_mode, out_qtables);
    /* sentinel + scan of every probed partition IN FULL, in assign order (276, 287-308) */
    orc_heap_i8 h = { out_keys, out_vals, R, 0 };
    orc_heap_i8_push(&h, 0, 127);
    for (int a = 0; a < ma; ++a) {
        const int p = assign[a];
        const uint32_t n = all_sizes[p];
        if (n == 0) continue;
        uint8_t* inter = (uint8_t*)malloc((size_t)orc_interleaved_size(n, M / 2));
        orc_interleave(inter, all_parts[p], n, M / 2);
        scan_i8_blocks(&h, M, inter, all_labels ? all_labels[p] : NULL, n, out_qtables + (long)a * table_dim);
        free(inter);
    }
    *out_size = h.size;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Counter-based synthetic code generator: byte stream of a splitmix64
 * hash of (seed, 64-bit word index).  Word w of partition stream = 8 consecutive code bytes.
 * Must match qadc_fill_codes in the HIP library bit for bit (tests compare them).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void orc_fill_codes(uint8_t* dst, uint64_t first_word, uint64_t nwords, uint64_t seed) {
    for (uint64_t w = 0; w < nwords; ++w) {
        const uint64_t v = splitmix64(seed ^ splitmix64(first_word + w));
        memcpy(dst + 8 * w, &v, 8);
    }
}

/* ------------------------------------------------------------------------------------------
 * Push stream of ONE SHARD of a partition (multi-GPU tests; SURVEY.md §8 A3 "S2"): the local range
 * [first_pos, first_pos + n) of a partition of global_n codes is scanned into a LOCAL heap seeded
 * with the (0,127) sentinel, and every push the block-stale local bound lets through is recorded
 * (key, value) in scan order — padding-lane replicas of the partition's last code included when
 * this shard holds it.  Concatenating the shard streams in shard order and replaying them through
 * one heap must reproduce the sequential scan of the whole partition.
 * Keys: labels[local pos] or first_pos + local pos.  Returns the stream length (may exceed cap).
 * ---------------------------------------------------------------------------------------- */
long orc_scan_i8_shard_stream(int M, const uint8_t* codes, const uint32_t* labels, uint32_t n, uint32_t global_n,
                              uint32_t first_pos, const int8_t* qt, int R, uint32_t* out_keys, int8_t* out_vals, long cap) {
    const int cs = M / 2;
    uint32_t* hk = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)R);
    int8_t* hv = (int8_t*)malloc((size_t)R);
    orc_heap_i8 h = { hk, hv, R, 0 };
    orc_heap_i8_push(&h, 0, 127);
    long count = 0;
    int8_t bound = h.vals[0];
    const uint32_t last_global = global_n - 1u;
    const uint32_t end = first_pos + n;                       /* exclusive, global */
    const uint32_t blk_end = (end == global_n) ? (global_n + 15u) / 16u * 16u : end;
    for (uint32_t g0 = first_pos; g0 < blk_end; g0 += 16) {  /* first_pos is a multiple of 16 */
        int any = 0;
        const int8_t bound_blk = bound;
        for (int j = 0; j < 16; ++j) {
            uint32_t gi = g0 + (uint32_t)j;
            if (gi > last_global) gi = last_global;
            const uint32_t li = gi - first_pos;
            const uint8_t* c = codes + (long)li * cs;
            int s = 0;
            for (int b = 0; b < cs; ++b)
                s += qt[(2 * b) * 16 + (c[b] & 15)] + qt[(2 * b + 1) * 16 + (c[b] >> 4)];
            const int8_t cand = (int8_t)(s > 127 ? 127 : s);
            if (cand < bound_blk) {
                const uint32_t key = labels ? labels[li] : gi;
                orc_heap_i8_push(&h, key, cand);
                if (count < cap) { out_keys[count] = key; out_vals[count] = cand; }
                ++count;
                any = 1;
            }
        }
        if (any) bound = h.vals[0];
    }
    free(hk); free(hv);
    return count;
}

/* Multi-partition form of orc_scan_i8_shard_stream (IVF on several GPUs): one rank holds the range
 * [first_pos[a], first_pos[a] + n[a]) of every probed partition a (assign order); its LOCAL heap (sentinel
 * first) is shared across its partition pieces exactly like scanner_4's heap across partitions.  Records every
 * attempted push with its assign slot.  qt: int8 [ma][M][16].  Returns the stream length. */
long orc_scan_i8_shards_stream(int M, int ma, const uint8_t* const* codes, const uint32_t* const* labels,
                               const uint32_t* n, const uint32_t* global_n, const uint32_t* first_pos, const int8_t* qt,
                               int R, uint32_t* out_keys, int8_t* out_vals, uint16_t* out_slots, long cap) {
    const int cs = M / 2;
    uint32_t* hk = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)R);
    int8_t* hv = (int8_t*)malloc((size_t)R);
    orc_heap_i8 h = { hk, hv, R, 0 };
    orc_heap_i8_push(&h, 0, 127);
    long count = 0;
    for (int a = 0; a < ma; ++a) {
        if (n[a] == 0) continue;
        const int8_t* t = qt + (long)a * M * 16;
        int8_t bound = h.vals[0];                             /* scan_avx_4 samples bh.max() on entry */
        const uint32_t last_global = global_n[a] - 1u;
        const uint32_t end = first_pos[a] + n[a];
        const uint32_t blk_end = (end == global_n[a]) ? (global_n[a] + 15u) / 16u * 16u : end;
        for (uint32_t g0 = first_pos[a]; g0 < blk_end; g0 += 16) {
            int any = 0;
            const int8_t bound_blk = bound;
            for (int j = 0; j < 16; ++j) {
                uint32_t gi = g0 + (uint32_t)j;
                if (gi > last_global) gi = last_global;
                const uint32_t li = gi - first_pos[a];
                const uint8_t* c = codes[a] + (long)li * cs;
                int s = 0;
                for (int b = 0; b < cs; ++b)
                    s += t[(2 * b) * 16 + (c[b] & 15)] + t[(2 * b + 1) * 16 + (c[b] >> 4)];
                const int8_t cand = (int8_t)(s > 127 ? 127 : s);
                if (cand < bound_blk) {
                    const uint32_t key = (labels && labels[a]) ? labels[a][li] : gi;
                    orc_heap_i8_push(&h, key, cand);
                    if (count < cap) { out_keys[count] = key; out_vals[count] = cand; out_slots[count] = (uint16_t)a; }
                    ++count;
                    any = 1;
                }
            }
            if (any) bound = h.vals[0];
        }
    }
    free(hk); free(hv);
    return count;
}
