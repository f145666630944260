// Microbenchmark behind the "MFMA only if a batched-stage dense GEMM formulation proves faster"
// clause of BASELINE.json's north_star: the Riccati stage's largest product,
//     [W | V] = P [A | B]      (13 x 13) . (13 x 17), FP64, one per instance and stage,
// formed (a) on the matrix cores -- v_mfma_f64_16x16x4_f64, the instance padded to 16 x 16 x 16 per
// 16-column tile, two tiles for the 17 columns: 8 MFMAs per instance -- and (b) with the engine's
// broadcast-FMA primitive -- 16-lane rows, v_fmac_f64_dpp row_newbcast, structural zeros of A skipped:
// 149 FMAs per FOUR instances (cfnmpc_kernels.hip: factor_stage steps (1), (2)).
// Reports instructions/s per variant, the products/s they amount to, and the useful FP64 rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_bench tools/mfma_f64_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// (a) 8 dependent-free MFMA accumulators per "instance": 2 column tiles x 4 k-steps (k = 4 each)
template <int NACC>
__global__ void k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0} + (double)(threadIdx.x * 1e-3 + i);
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    d4 s = acc[0];
    for (int i = 1; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// (b) the engine's primitive: chains of fused broadcast FMAs (13 per output column as in dotbc<13,0>)
__global__ void k_dpp(double* out, int iters, double a0, double b0) {
    double acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a0 + threadIdx.x * 1e-9, y = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile(
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                : "v"(x), "v"(y));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const int iters = 4000;
    double* out;
    (void)hipMalloc(&out, sizeof(double) * (size_t)simds * 8 * 64);
    std::printf("%s: %d CUs, %d SIMDs\n", prop.name, cus, simds);
    const double useful = 13.0 * 13.0 * 17.0 * 2.0;                 // flops of P [A | B], dense
    const double useful_sparse = (97.0 + 52.0) * 13.0 * 2.0 / 13.0; // (per product, counting only stored entries: 149 x 13 rows ... see note)
    (void)useful_sparse;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = simds * wps;
        // MFMA: 8 accumulators x 8 x iters instructions per wave, 8 per padded instance product
        const double ms_m = time_ms([&] { hipLaunchKernelGGL((k_mfma<8>), dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9); });
        const double n_mfma = (double)blocks * iters * 64.0;
        const double mfma_per_s = n_mfma / (ms_m * 1e-3);
        const double prod_m = mfma_per_s / 8.0;                       // one instance per wave: 8 MFMAs per product
        // DPP: 64 x iters instructions per wave; 149 instructions form the product for the 4 instances of a wave
        const double ms_d = time_ms([&] { hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9); });
        const double n_dpp = (double)blocks * iters * 64.0;
        const double dpp_per_s = n_dpp / (ms_d * 1e-3);
        const double prod_d = dpp_per_s / 149.0 * 4.0;
        std::printf("waves/SIMD %d | v_mfma_f64_16x16x4: %.3e instr/s = %.1f TFLOP/s issued (2*16*16*4 each), %.3e products/s, %.1f TFLOP/s useful"
                    " | v_fmac_f64_dpp: %.3e instr/s = %.1f TFLOP/s issued (2*64 each), %.3e products/s, %.1f TFLOP/s useful\n",
                    wps, mfma_per_s, mfma_per_s * 2048.0 / 1e12, prod_m, prod_m * useful / 1e12, dpp_per_s, dpp_per_s * 128.0 / 1e12, prod_d,
                    prod_d * useful / 1e12);
    }
    (void)hipFree(out);
    return 0;
}
