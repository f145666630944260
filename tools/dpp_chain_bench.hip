// Microbenchmark: throughput of DEPENDENT chains of v_fmac_f64_dpp (the shape of the Riccati
// dot products: acc += own * row_newbcast<l>(src)) with K interleaved accumulators, one wave per
// SIMD.  Reports cycles per FMA instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_chain_bench tools/dpp_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMA(acc, l) "v_fmac_f64_dpp " acc ", %4, %5 row_newbcast:" #l " row_mask:0xf bank_mask:0xf\n\t"

template <int K>
__global__ __launch_bounds__(64) void k_chain(double* out, int iters, double a, double b, unsigned long long* cyc) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    double x = a + threadIdx.x * 1e-9, y = b;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (K == 1)
            asm volatile("s_nop 1\n\t" FMA("%0", 0) FMA("%0", 1) FMA("%0", 2) FMA("%0", 3) FMA("%0", 4) FMA("%0", 5) FMA("%0", 6) FMA("%0", 7)
                         FMA("%0", 8) FMA("%0", 9) FMA("%0", 10) FMA("%0", 11) FMA("%0", 12) FMA("%0", 13) FMA("%0", 14) FMA("%0", 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        if (K == 2)
            asm volatile("s_nop 1\n\t" FMA("%0", 0) FMA("%1", 1) FMA("%0", 2) FMA("%1", 3) FMA("%0", 4) FMA("%1", 5) FMA("%0", 6) FMA("%1", 7)
                         FMA("%0", 8) FMA("%1", 9) FMA("%0", 10) FMA("%1", 11) FMA("%0", 12) FMA("%1", 13) FMA("%0", 14) FMA("%1", 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        if (K == 3)
            asm volatile("s_nop 1\n\t" FMA("%0", 0) FMA("%1", 1) FMA("%2", 2) FMA("%0", 3) FMA("%1", 4) FMA("%2", 5) FMA("%0", 6) FMA("%1", 7)
                         FMA("%2", 8) FMA("%0", 9) FMA("%1", 10) FMA("%2", 11) FMA("%0", 12) FMA("%1", 13) FMA("%2", 14) FMA("%0", 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
        if (K == 4)
            asm volatile("s_nop 1\n\t" FMA("%0", 0) FMA("%1", 1) FMA("%2", 2) FMA("%3", 3) FMA("%0", 4) FMA("%1", 5) FMA("%2", 6) FMA("%3", 7)
                         FMA("%0", 8) FMA("%1", 9) FMA("%2", 10) FMA("%3", 11) FMA("%0", 12) FMA("%1", 13) FMA("%2", 14) FMA("%3", 15)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
// K = 8 accumulators; LANES8: only row_newbcast:0..7 (as in dpp_bench.hip)
template <bool LANES8>
__global__ __launch_bounds__(64) void k_chain8(double* out, int iters, double a, double b, unsigned long long* cyc) {
    double c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    double x = a + threadIdx.x * 1e-9, y = b;
#define FMA8(acc, l) "v_fmac_f64_dpp " acc ", %8, %9 row_newbcast:" #l " row_mask:0xf bank_mask:0xf\n\t"
    for (int it = 0; it < iters; it++) {
        if (LANES8)
            asm volatile("s_nop 1\n\t" FMA8("%0", 0) FMA8("%1", 1) FMA8("%2", 2) FMA8("%3", 3) FMA8("%4", 4) FMA8("%5", 5) FMA8("%6", 6) FMA8("%7", 7)
                         FMA8("%0", 0) FMA8("%1", 1) FMA8("%2", 2) FMA8("%3", 3) FMA8("%4", 4) FMA8("%5", 5) FMA8("%6", 6) FMA8("%7", 7)
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y));
        else
            asm volatile("s_nop 1\n\t" FMA8("%0", 0) FMA8("%1", 1) FMA8("%2", 2) FMA8("%3", 3) FMA8("%4", 4) FMA8("%5", 5) FMA8("%6", 6) FMA8("%7", 7)
                         FMA8("%0", 8) FMA8("%1", 9) FMA8("%2", 10) FMA8("%3", 11) FMA8("%4", 12) FMA8("%5", 13) FMA8("%6", 14) FMA8("%7", 15)
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y));
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = 0;
}
// same lane for all FMAs of a chain (one accumulator, row_newbcast:3 sixteen times), and the
// "two-source" form of the Riccati code: acc_j += own[l] * bcast_l(src) with 16 different `own`
__global__ __launch_bounds__(64) void k_chain_samelane(double* out, int iters, double a, double b, unsigned long long* cyc) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; it++)
        asm volatile("s_nop 1\n\t" FMA("%0", 3) FMA("%1", 3) FMA("%2", 3) FMA("%3", 3) FMA("%0", 3) FMA("%1", 3) FMA("%2", 3) FMA("%3", 3)
                     FMA("%0", 3) FMA("%1", 3) FMA("%2", 3) FMA("%3", 3) FMA("%0", 3) FMA("%1", 3) FMA("%2", 3) FMA("%3", 3)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = 0;
}
// v_mov_b64_dpp + plain FMA (two instructions per product)
__global__ __launch_bounds__(64) void k_mov_fma(double* out, int iters, double a, double b, unsigned long long* cyc) {
    double a0 = threadIdx.x, a1 = 1, t0 = 0, t1 = 0;
    double x = a + threadIdx.x * 1e-9, y = b;
#define MF(acc, tmp, l) "v_mov_b64_dpp " tmp ", %4 row_newbcast:" #l " row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64 " acc ", " tmp ", %5\n\t"
    for (int it = 0; it < iters; it++)
        asm volatile("s_nop 1\n\t" MF("%0", "%2", 0) MF("%1", "%3", 1) MF("%0", "%2", 2) MF("%1", "%3", 3) MF("%0", "%2", 4) MF("%1", "%3", 5) MF("%0", "%2", 6) MF("%1", "%3", 7)
                     MF("%0", "%2", 8) MF("%1", "%3", 9) MF("%0", "%2", 10) MF("%1", "%3", 11) MF("%0", "%2", 12) MF("%1", "%3", 13) MF("%0", "%2", 14) MF("%1", "%3", 15)
                     : "+v"(a0), "+v"(a1), "+v"(t0), "+v"(t1) : "v"(x), "v"(y));
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = 0;
}
// plain (non-DPP) dependent chain for comparison
template <int K>
__global__ __launch_bounds__(64) void k_plain(double* out, int iters, double a, double b, unsigned long long* cyc) {
    double acc[4] = {(double)threadIdx.x, 1, 2, 3};
    double x = a + threadIdx.x * 1e-9, y = b;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r % K] = __builtin_fma(x, y, acc[r % K]);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <class F>
static void run(const char* name, F launch, int iters, unsigned long long* dcyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %.3f ms  -> %.2f ns per FMA instr (wave 0: %.2f clock64 ticks per FMA)\n", name, ms, ms * 1e6 / ((double)iters * 16),
           (double)c / ((double)iters * 16));
}

int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 8 * 64 * 1024); hipMalloc(&cyc, 8);
    const int iters = 20000, blocks = 1024;  // one wave per SIMD
#define RUN(K) run("dpp chain K=" #K, [&] { hipLaunchKernelGGL(k_chain<K>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc); \
               run("plain chain K=" #K, [&] { hipLaunchKernelGGL(k_plain<K>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc);
    RUN(1) RUN(2) RUN(3) RUN(4)
    run("dpp K=8 lanes 0..7", [&] { hipLaunchKernelGGL(k_chain8<true>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc);
    run("dpp K=8 lanes 0..15", [&] { hipLaunchKernelGGL(k_chain8<false>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc);
    run("dpp K=4 same lane", [&] { hipLaunchKernelGGL(k_chain_samelane, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc);
    run("mov_dpp + fmac K=2", [&] { hipLaunchKernelGGL(k_mov_fma, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9, cyc); }, iters, cyc);
    return 0;
}
