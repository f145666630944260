// Microbenchmark: FP64 FMA issue rate on gfx950 with (a) plain v_fma_f64, (b) v_fmac_f64_dpp
// row_newbcast (operand broadcast inside each 16-lane row, inline asm), (c) v_mov_b64_dpp +
// v_fma_f64 (compiler builtin path).  Also checks the broadcast semantics.
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_bench tools/dpp_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NACC 8
#define INNER 64

template <int L>
__device__ __forceinline__ double bcast_builtin(double v) {
    long long x = __builtin_bit_cast(long long, v);
    long long r = __builtin_amdgcn_update_dpp((long long)0, x, 0x150 + L, 0xf, 0xf, true);
    return __builtin_bit_cast(double, r);
}

__global__ void k_plain(double* out, int iters, double a, double b) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < INNER / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(x, y, acc[i]);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_dpp_asm(double* out, int iters, double a, double b) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < INNER / NACC; r++) {
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
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_dpp_builtin(double* out, int iters, double a, double b) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < INNER / NACC; r++) {
            acc[0] = __builtin_fma(bcast_builtin<0>(x), y, acc[0]);
            acc[1] = __builtin_fma(bcast_builtin<1>(x), y, acc[1]);
            acc[2] = __builtin_fma(bcast_builtin<2>(x), y, acc[2]);
            acc[3] = __builtin_fma(bcast_builtin<3>(x), y, acc[3]);
            acc[4] = __builtin_fma(bcast_builtin<4>(x), y, acc[4]);
            acc[5] = __builtin_fma(bcast_builtin<5>(x), y, acc[5]);
            acc[6] = __builtin_fma(bcast_builtin<6>(x), y, acc[6]);
            acc[7] = __builtin_fma(bcast_builtin<7>(x), y, acc[7]);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// semantics: out[lane] = bcast_L(x) for L = 5 (asm: acc=0 + x_bcast*1)
__global__ void k_sem(double* out) {
    double x = 100.0 + threadIdx.x, one = 1.0, acc = 0.0, acc2 = 0.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(one));
    acc2 = bcast_builtin<5>(x);
    out[threadIdx.x] = acc;
    out[64 + threadIdx.x] = acc2;
}

template <typename F>
static double time_kernel(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    double* out;
    hipMalloc(&out, sizeof(double) * 1024 * 1024 * 8);
    std::vector<double> h(128);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h.data(), out, 128 * 8, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; l++) {
        const double want = 100.0 + (l / 16) * 16 + 5;
        if (h[l] != want || h[64 + l] != want) ok = false;
    }
    printf("row_newbcast semantics (asm & builtin): %s  [lane0=%g lane17=%g lane63=%g]\n", ok ? "OK" : "MISMATCH", h[0], h[17], h[63]);
    const int iters = 2000;
    for (int wpb : {1, 2, 4, 8}) {  // waves per SIMD
        const int blocks = 256 * 4 * wpb, threads = 64;  // 1 wave per block
        const double fmas = (double)blocks * threads * iters * INNER;
        double t0 = time_kernel([&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, 5);
        double t1 = time_kernel([&] { hipLaunchKernelGGL(k_dpp_asm, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, 5);
        double t2 = time_kernel([&] { hipLaunchKernelGGL(k_dpp_builtin, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1e-9); }, 5);
        printf("waves/SIMD=%d  plain %.3f ms %.1f TFLOP/s | dpp-asm %.3f ms %.1f TFLOP/s | dpp-builtin %.3f ms %.1f TFLOP/s\n", wpb,
               t0, 2 * fmas / t0 * 1e-9, t1, 2 * fmas / t1 * 1e-9, t2, 2 * fmas / t2 * 1e-9);
    }
    return 0;
}
