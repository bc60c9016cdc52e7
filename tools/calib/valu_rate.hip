// Vector-instruction issue rate on gfx950 as a function of the waves per SIMD: every workgroup is one wave that runs ITER rounds of
// 8 independent chains of one instruction kind; W waves per SIMD are made resident (4 W workgroups per CU, static LDS keeps more
// from joining).  Prints SIMD cycles per wave-instruction (2.4 GHz assumed; the real clock is printed from wall_clock64 / s_memtime).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/calib/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KIND, int LDSB>
__global__ __launch_bounds__(64) void k(int iters, unsigned* out, unsigned long long* cyc) {
    __shared__ unsigned pad[LDSB / 4];
    unsigned a[8];
    unsigned long long w[4];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    for (int i = 0; i < 4; i++) w[i] = threadIdx.x * 0x9E3779B97F4A7C15ull + i;
    unsigned b = out[0] | 3u, c = out[1] | 5u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[i]));
                if (KIND == 8) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i & 3]) : "v"(b), "v"(c) : "s10", "s11");
                if (KIND == 9) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w[i & 3]) : "v"(b & 7));
                if (KIND == 10) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i & 3]) : "v"(w[(i + 1) & 3]));
                if (KIND == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(b) : "s10", "s11");
                if (KIND == 12) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(a[i]));
                if (KIND == 13) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 14) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 15) asm volatile("v_ffbl_b32_e32 %0, %0" : "+v"(a[i]));
                if (KIND == 16) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 17) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    for (int i = 0; i < 4; i++) s ^= (unsigned)w[i] ^ (unsigned)(w[i] >> 32);
    if (s == 0x12345u) pad[threadIdx.x] = s;
    out[2 + blockIdx.x % 64] = s + pad[0] * 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int LDSB> double run(int wgs, int iters, unsigned* d, unsigned long long* dc, unsigned long long* hc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, LDSB>), dim3(wgs), dim3(64), 0, 0, 10, d, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, LDSB>), dim3(wgs), dim3(64), 0, 0, iters, d, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost);
    return ms;
}
template <int KIND> void kind(const char* name, unsigned* d, unsigned long long* dc) {
    const int iters = 20000;
    const double n_inst = (double)iters * 64;                  // wave-instructions per wave
    unsigned long long hc = 0;
    // waves per SIMD 1, 2, 4, 8: LDS per one-wave workgroup 40 KB, 20 KB, 10 KB, 5 KB (160 KB per CU)
    const double m1 = run<KIND, 40 * 1024>(256 * 4, iters, d, dc, &hc); const double c1 = (double)hc;
    const double m2 = run<KIND, 20 * 1024>(256 * 8, iters, d, dc, &hc); const double c2 = (double)hc;
    const double m4 = run<KIND, 10 * 1024>(256 * 16, iters, d, dc, &hc); const double c4 = (double)hc;
    const double m8 = run<KIND, 5 * 1024>(256 * 32, iters, d, dc, &hc); const double c8 = (double)hc;
    // SIMD cycles per wave-instruction = time x clock / (instructions per wave x waves per SIMD); clock from the wave's own counter
    auto cpi = [&](double ms, double cyc, int w) { return ms * 1e-3 * 2.4e9 / (n_inst * w); };   // SIMD cycles per wave-instruction at 2.4 GHz
    printf("%-16s waves/SIMD 1: %.2f ms (%.2f cyc/inst, clock %.2f GHz)  2: %.2f ms (%.2f)  4: %.2f ms (%.2f)  8: %.2f ms (%.2f)   [SIMD cycles per wave-instruction]\n", name,
           m1, cpi(m1, c1, 1), c1 / (m1 * 1e6), m2, cpi(m2, c2, 2), m4, cpi(m4, c4, 4), m8, cpi(m8, c8, 8));
}
int main() {
    unsigned* d; unsigned long long* dc;
    hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&dc, 64);
    kind<0>("v_xor_b32", d, dc); kind<1>("v_add_u32", d, dc); kind<2>("v_alignbit_b32", d, dc); kind<3>("v_bitop3_b32", d, dc);
    kind<4>("v_mul_lo_u32", d, dc); kind<5>("v_mul_hi_u32", d, dc); kind<6>("v_lshl_add_u32", d, dc); kind<7>("v_bfe_u32", d, dc);
    kind<8>("v_mad_u64_u32", d, dc); kind<9>("v_lshlrev_b64", d, dc); kind<10>("v_lshl_add_u64", d, dc); kind<11>("v_cndmask_e64", d, dc);
    kind<12>("v_lshrrev_b32", d, dc); kind<13>("v_and_b32", d, dc); kind<14>("v_min3_u32", d, dc); kind<15>("v_ffbl_b32", d, dc);
    kind<16>("v_lshl_or_b32", d, dc); kind<17>("v_add3_u32", d, dc);
    return 0;
}
