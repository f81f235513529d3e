// Float ADC sum of one 4-bit code in the association of the reference's scan_4<NSQ> (query_common.hpp:59-90).
//
// The source adds the 2*CS looked-up entries sequentially from 0 (72-80), but the reference is built with
// -ffast-math (CMakeLists.txt:7), which lets g++ re-associate the sum — and it does.  The float pre-scan decides
// qmax = the R-th smallest sum of the starts (db_query_4.cpp:259), and qmax scales every int8 table, so the
// grouping is part of the result.  Two groupings are provided:
//
//   sum_mode 1 (default) — AS COMPILED: the grouping g++ 11.4 emits for the stand-alone scan_4<16> / scan_4<32>
//     instances at -O3 -ffast-math (identical with -march=native and with an explicit AVX2/FMA ISA list), read off
//     that build's disassembly and pinned to the binary by the oracle's tests.  With L_b / H_b the entries looked
//     up by the low / high nibble of code byte b:
//         A = (H2+L3)+(H3+L4)   B = (H0+L1)+(H1+L2)   C = (H5+L6)+(H4+L5)   D = (H6+L7)+(H7+L0)
//         s = ((A+B)+C)+D                                      bytes 0..7  (all of NSQ = 16)
//         s = s + ((L_{b+1}+H_{b+1}) + (L_b+H_b))              b = 8, 10, 12, 14  (NSQ = 32 only)
//     (IEEE addition is commutative: only the grouping matters.)  It is also the cheaper one here: the longest
//     dependent chain is 5 adds (NSQ 16) / 9 adds (NSQ 32) instead of 16 / 32.
//   sum_mode 0 — SOURCE ORDER: ((((0 + L0) + H0) + L1) + H1) ...
//
// T is float, or a float vector (one component per query of a multi-query pass: the component-wise adds round like
// the scalar ones).  v[2k] = L_{h+k}, v[2k+1] = H_{h+k} for the eight code bytes h .. h+7 of one call.
#pragma once

template <typename T>
__device__ __forceinline__ T adc_sum8_source(T s, const T* v) {
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    return s;
}

#define QADC_L(b) v[2 * (b)]
#define QADC_H(b) v[2 * (b) + 1]

// bytes 0..7: the result does not take an incoming sum (the compiled code has no "0 +")
template <typename T>
__device__ __forceinline__ T adc_sum8_compiled_first(const T* v) {
    const T a = (QADC_H(2) + QADC_L(3)) + (QADC_H(3) + QADC_L(4));
    const T b = (QADC_H(0) + QADC_L(1)) + (QADC_H(1) + QADC_L(2));
    const T c = (QADC_H(5) + QADC_L(6)) + (QADC_H(4) + QADC_L(5));
    const T d = (QADC_H(6) + QADC_L(7)) + (QADC_H(7) + QADC_L(0));
    return ((a + b) + c) + d;
}

// bytes 8..15 (NSQ = 32)
template <typename T>
__device__ __forceinline__ T adc_sum8_compiled_next(T s, const T* v) {
#pragma unroll
    for (int b = 0; b < 8; b += 2) s = s + ((QADC_L(b + 1) + QADC_H(b + 1)) + (QADC_L(b) + QADC_H(b)));
    return s;
}

#undef QADC_L
#undef QADC_H

// all CS = M/2 bytes: v[2b] = L_b, v[2b+1] = H_b
template <int M, typename T>
__device__ __forceinline__ T adc_sum_code(const T* v, int sum_mode, T zero) {
    if (sum_mode == 0) {
        T s = zero;
#pragma unroll
        for (int h = 0; h < M / 2; h += 8) s = adc_sum8_source(s, v + 2 * h);
        return s;
    }
    T s = adc_sum8_compiled_first(v);
    if constexpr (M == 32) s = adc_sum8_compiled_next(s, v + 16);
    return s;
}
