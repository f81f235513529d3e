// The grouping of the reference's float sums AS COMPILED — C++14, header only, host side.
//
// The reference is built with -O3 -ffast-math (CMakeLists.txt:7), which lets g++ re-associate float sums, and for
// the sums on the query path it does.  These helpers evaluate those sums in the grouping the reference binary uses
// (g++ 11.4; identical with -march=native and with an explicit AVX2/FMA ISA list), read off the disassembly of the
// reference's own functions and pinned to that build by the oracle's tests (tests/test_oracle_float_ref.py; the
// device twins are csrc/qadc_float_sum.h and direct_sqdist in csrc/qadc_kernels.hip).  This translation unit must be
// compiled WITHOUT -ffast-math and without FP contraction of a*b+c (the fused multiply-adds below are explicit).
//
//   adc_sum<N>(t)      N looked-up table entries of one code, t[0..N) in the source's order:
//                        scan_standard<T,N> (query_common.hpp:92-118): t[m] = dists[m*NCENT + code[m]]
//                        scan_4<N>          (query_common.hpp:59-90):  t[2b] = low-nibble entry, t[2b+1] = high-nibble entry
//                      N 4:  (t1+t2) + (t3+t0)
//                      N 8:  ((t1+t2)+(t3+t4)) + ((t5+t6)+(t7+t0))
//                      N 16: ((A+B)+C)+D   A=(t5+t6)+(t7+t8)  B=(t1+t2)+(t3+t4)  C=(t11+t12)+(t9+t10)  D=(t13+t14)+(t15+t0)
//                      N 32: the 16-term grouping of t[0..16), then s = s + ((t[j+2]+t[j+3]) + (t[j]+t[j+1])), j = 16, 20, 24, 28
//                      other N, and every 16-bit instance (scan_standard<uint16_t, N>: no such instance of the reference was read or
//                      pinned — adc_sum<N, false>): source order
//   sqdist(x, c, ds)   fmanorm<ds/8, ds%8>(x, c) (distances.hpp:60-76) as compute_dists_single_simd_cg calls it (294-311):
//                      per AVX lane j acc[j] = fma(d, d, acc[j]) over the blocks (d = x - c), reduceadd's tree
//                      (acc[j] + acc[j+4]; (r0+r2) + (r1+r3)); the scalar remainder is paired
//                      p_k = fma(d_2k, d_2k, r(d_2k+1 * d_2k+1)) (d = c - x): REM 4 -> (p0+p1) + vec, REM 6 -> (vec+p2) + (p0+p1).
//                      Other remainders (no instance in the reference's dispatch, distances.cpp:50-84): one sequential loop.
//   sqnorm(x, ds)      fmanorm<ds/8, ds%8>(x) / norm_4(x) as compute_cross_dists_blas calls them (distances.hpp:151-215) for the
//                      BLAS-expansion distances: the grouping of sqdist with c = 0 (norm_4: fma(x0,x0,r(x1*x1)) + fma(x2,x2,r(x3*x3))),
//                      pinned to the reference's text compiled up to its sgemm call (oracle/_ref qadc_reff_cross_norms).
//   expansion_dist(x, c, ds, vn, cn)   (vn + cn) - 2 x.c with ONE sequential dot: the sgemm(alpha = -2, beta = 1) that follows in
//                      the reference is OpenBLAS's (restated, unpinned); -2 dot is exact, so the sum rounds once like C += alpha*acc.
//
// float_sum_mode() = 1 (default) selects these groupings; 0 = the source-order / sequential loops.
#pragma once
#include <cmath>

namespace qadc {

inline int& float_sum_mode() {
    static int mode = 1;
    return mode;
}

template <int N, bool PINNED = true>
inline float adc_sum(const float* t) {
    if (!PINNED || float_sum_mode() == 0 || !(N == 4 || N == 8 || N == 16 || N == 32)) {
        float s = 0;
        for (int i = 0; i < N; ++i) s += t[i];
        return s;
    }
    if (N == 4) return (t[1] + t[2]) + (t[3] + t[0]);
    if (N == 8) return ((t[1] + t[2]) + (t[3] + t[4])) + ((t[5] + t[6]) + (t[7] + t[0]));
    const float a = (t[5] + t[6]) + (t[7] + t[8]);
    const float b = (t[1] + t[2]) + (t[3] + t[4]);
    const float c = (t[11] + t[12]) + (t[9] + t[10]);
    const float d = (t[13] + t[14]) + (t[15] + t[0]);
    float s = ((a + b) + c) + d;
    for (int j = 16; j + 3 < N; j += 4) s = s + ((t[j + 2] + t[j + 3]) + (t[j] + t[j + 1]));
    return s;
}

inline float sqdist(const float* x, const float* c, int ds) {
    const int blocks = ds / 8, rem = ds % 8;
    if (float_sum_mode() == 0 || !(rem == 0 || rem == 4 || rem == 6)) {
        float s = 0;
        for (int d = 0; d < ds; ++d) {
            const float t = x[d] - c[d];
            const float sq = t * t;
            s = s + sq;
        }
        return s;
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int j = 0; j < 8; ++j) {
            const float d = x[b * 8 + j] - c[b * 8 + j];
            acc[j] = std::fma(d, d, acc[j]);
        }
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    const float vec = (r0 + r2) + (r1 + r3);
    if (rem == 0) return vec;
    float p[3] = {0, 0, 0};
    for (int k = 0; k < rem / 2; ++k) {
        const float d0 = c[blocks * 8 + 2 * k] - x[blocks * 8 + 2 * k];
        const float d1 = c[blocks * 8 + 2 * k + 1] - x[blocks * 8 + 2 * k + 1];
        const float sq1 = d1 * d1;
        p[k] = std::fma(d0, d0, sq1);
    }
    if (rem == 4) return (p[0] + p[1]) + vec;
    return (vec + p[2]) + (p[0] + p[1]);
}

inline float sqnorm(const float* x, int ds) {
    const int blocks = ds / 8, rem = ds % 8;
    if (float_sum_mode() == 0 || !(rem == 0 || rem == 4 || rem == 6)) {
        float s = 0;
        for (int d = 0; d < ds; ++d) {
            const float sq = x[d] * x[d];
            s = s + sq;
        }
        return s;
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b)
        for (int j = 0; j < 8; ++j) acc[j] = std::fma(x[b * 8 + j], x[b * 8 + j], acc[j]);
    const float r0 = acc[0] + acc[4], r1 = acc[1] + acc[5], r2 = acc[2] + acc[6], r3 = acc[3] + acc[7];
    const float vec = (r0 + r2) + (r1 + r3);
    if (rem == 0) return vec;
    float p[3] = {0, 0, 0};
    for (int k = 0; k < rem / 2; ++k) {
        const float sq1 = x[blocks * 8 + 2 * k + 1] * x[blocks * 8 + 2 * k + 1];
        p[k] = std::fma(x[blocks * 8 + 2 * k], x[blocks * 8 + 2 * k], sq1);
    }
    if (rem == 4) return (p[0] + p[1]) + vec;
    return (vec + p[2]) + (p[0] + p[1]);
}

inline float expansion_dist(const float* x, const float* c, int ds, float vn, float cn) {
    float dot = 0;
    for (int d = 0; d < ds; ++d) {
        const float pr = x[d] * c[d];
        dot = dot + pr;
    }
    const float base = vn + cn;
    const float m2 = -2.0f * dot;
    return base + m2;
}

}  // namespace qadc
