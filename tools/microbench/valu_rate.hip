// Microbenchmark (round 5): issue rate of the adds the multi-query scans are made of — v_lshl_add_u64 (four u16 fields per
// instruction), v_add3_u32, v_add_u32, v_pk_add_u16 — as wave-instructions per SIMD cycle, 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kIters = 4096, kChains = 8;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, uint64_t seed) {
    uint64_t a[kChains];
    uint32_t b[kChains], c[kChains];
    for (int i = 0; i < kChains; ++i) { a[i] = seed + threadIdx.x * 977u + i; b[i] = (uint32_t)a[i] * 3u; c[i] = (uint32_t)a[i] * 5u; }
    const uint64_t inc = seed | 1u;
    const uint32_t inc32 = (uint32_t)seed | 1u, inc2 = (uint32_t)(seed >> 7) | 3u;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (OP == 0) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(inc));
            if (OP == 1) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(b[i]) : "v"(b[i]), "v"(inc32), "v"(inc2));
            if (OP == 2) asm volatile("v_add_u32 %0, %1, %2" : "=v"(b[i]) : "v"(b[i]), "v"(inc32));
            if (OP == 3) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(b[i]) : "v"(b[i]), "v"(inc32));
            if (OP == 4) asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(c[i]) : "v"(inc32), "v"(b[i]));
            if (OP == 5) asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(c[i]) : "v"(inc32), "v"(b[i]));
            if (OP == 6) asm volatile("v_lshrrev_b32_e32 %0, 4, %1" : "=v"(c[i]) : "v"(b[i]));
            if (OP == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(c[i]) : "v"(b[i]), "v"(inc32), "v"(inc2));
            if (OP == 8) asm volatile("v_bfe_u32 %0, %1, 4, 4" : "=v"(c[i]) : "v"(b[i]));
            if (OP == 9) asm volatile("v_add_u32_e64 %0, %1, %2" : "=v"(b[i]) : "v"(b[i]), "v"(inc32));
            if (OP == 10) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c[i]) : "v"(b[i]), "v"(inc32), "v"(inc2));
            if (OP == 11) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(c[i]) : "v"(inc32), "v"(b[i]));
            if (OP == 12) asm volatile("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(c[i]) : "v"(b[i]), "v"(inc32));
            if (OP == 13) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(c[i]) : "v"(b[i]), "v"(inc32), "v"(inc2));
            if (OP == 14) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c[i]) : "v"(b[i]), "v"(inc32), "v"(inc2));
            if (OP == 15) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(c[i]) : "v"(b[i]));
            if (OP == 16) asm volatile("v_and_b32_e32 %0, 0xf0, %1" : "=v"(c[i]) : "v"(b[i]));
            if (OP == 17) asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(c[i]) : "s"(inc32), "v"(b[i]));
            if (OP == 18) asm volatile("v_add_u32_e32 %0, %1, %2\n\tv_add_u32_e32 %3, %4, %5" : "=v"(b[i]), "=v"(c[i]) : "v"(b[i]), "v"(inc32), "0"(b[i]), "1"(c[i]));
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < kChains; ++i) s += a[i] + b[i] + c[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
int run(const char* name, uint64_t* d_out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 8;                                  // 8 workgroups (32 waves) per CU: every SIMD has 8 waves
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr = (double)blocks * 4 * kIters * kChains;     // wave-instructions
    const double per_simd = wave_instr / (256.0 * 4.0);
    printf("%-16s %.3f ms  -> %.2f ns per wave-instruction per SIMD  (at 2.4 GHz: %.2f cycles)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return 0;
}

int main() {
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * 256 * 8 * 256));
    if (run<0>("v_lshl_add_u64", d_out)) return 1;
    if (run<1>("v_add3_u32", d_out)) return 1;
    if (run<2>("v_add_u32", d_out)) return 1;
    if (run<3>("v_pk_add_u16", d_out)) return 1;
    if (run<4>("v_and_b32_sdwa", d_out)) return 1;
    if (run<5>("v_and_b32_e32", d_out)) return 1;
    if (run<6>("v_lshrrev_b32_e32", d_out)) return 1;
    if (run<7>("v_perm_b32", d_out)) return 1;
    if (run<8>("v_bfe_u32", d_out)) return 1;
    if (run<9>("v_add_u32_e64", d_out)) return 1;
    if (run<10>("v_fma_f32", d_out)) return 1;
    if (run<11>("v_mul_f32_e32", d_out)) return 1;
    if (run<12>("v_lshl_or_b32", d_out)) return 1;
    if (run<13>("v_and_or_b32", d_out)) return 1;
    if (run<14>("v_mad_u32_u24", d_out)) return 1;
    if (run<15>("v_mov_b32_sdwa", d_out)) return 1;
    if (run<16>("v_and_b32 literal", d_out)) return 1;
    if (run<17>("v_and_b32 sgpr", d_out)) return 1;
    return 0;
}
