// FP64 VALU issue rate, dependent against independent chains, one and two waves per SIMD (development microbenchmark;
// hipcc -O3 --offload-arch=gfx950 -o fma_latency_bench tools/fma_latency_bench.hip; results: profiles/r03_fma_latency_microbench.log)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC, int KIND>
__global__ __launch_bounds__(64) void k(double* out, int reps, double a, double b) {
    double acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < 64 / NACC; j++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
                if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "s"(a), "v"(b));
                if (KIND == 4) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
        }
    }
    double s = 0; for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC, int KIND> void run(const char* name, double* d, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20000;
    k<NACC, KIND><<<grid, 64>>>(d, 100, 1.0000001, 1e-9); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC, KIND><<<grid, 64>>>(d, reps, 1.0000001, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %5d: %.3f ns per instruction per wave\n", name, grid, ms * 1e6 / (reps * 64.0));
}
int main() {
    double* d; hipMalloc(&d, 8 * 64 * 4096);
    for (int grid : {1024, 2048}) {
        run<1, 0>("fma dependent (1 chain)", d, grid); run<2, 0>("fma 2 chains", d, grid); run<4, 0>("fma 4 chains", d, grid); run<8, 0>("fma 8 chains", d, grid);
        run<1, 1>("mul dependent", d, grid); run<4, 1>("mul 4 chains", d, grid);
        run<1, 2>("add dependent", d, grid); run<4, 2>("add 4 chains", d, grid);
        run<1, 3>("fma sgpr-operand dependent", d, grid); run<4, 3>("fma sgpr-operand 4 chains", d, grid);
        run<1, 4>("fmac dependent", d, grid); run<4, 4>("fmac 4 chains", d, grid);
    }
    return 0;
}
