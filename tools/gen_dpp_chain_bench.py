"""Generates tools/dev/dpp_chain_gen.hip: variants of 16-FMA blocks of v_fmac_f64_dpp with a
given (accumulator, lane) sequence -- to find what limits the issue rate of broadcast FMAs on
gfx950 with one wave per SIMD (development aid)."""
import sys
variants = {
    "K8_lane3":      [(i % 8, 3) for i in range(16)],
    "K8_lanes8to15": [(i % 8, 8 + i % 8) for i in range(16)],
    "K8_lanes0to7":  [(i % 8, i % 8) for i in range(16)],
    "K4_lanes0to3":  [(i % 4, i % 4) for i in range(16)],
    "K4_lanes0to7":  [(i % 4, i % 8) for i in range(16)],
    "K2_lanes0to7":  [(i % 2, i % 8) for i in range(16)],
    "K1_lanes0to7":  [(0, i % 8) for i in range(16)],
    "K1_lanes0to15": [(0, i) for i in range(16)],
    "K8_lanes0to15": [(i % 8, i) for i in range(16)],
    "K8_lanes_0_8_alt": [(i % 8, (i % 2) * 8) for i in range(16)],
    "K8_lanes0to12": [(i % 8, i % 13) for i in range(16)],
}
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>']
for name, seq in variants.items():
    body = '"s_nop 1\\n\\t"' + "".join(f' "v_fmac_f64_dpp %{a}, %8, %9 row_newbcast:{l} row_mask:0xf bank_mask:0xf\\n\\t"' for a, l in seq)
    out.append(f'''__global__ __launch_bounds__(64) void k_{name}(double* out, int iters, double a, double b) {{
    double c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; it++)
        asm volatile({body}
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y));
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}}''')
out.append('''template <class F> static void run(const char* name, F launch, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s %.3f ms -> %.2f ns per FMA instruction\\n", name, ms, ms * 1e6 / ((double)iters * 16));
}
int main() { double* out; hipMalloc(&out, 8 * 64 * 4096); const int iters = 20000;
  for (int blocks : {1024, 2048}) { printf("-- %d waves per SIMD\\n", blocks / 1024);''')
for name in variants:
    out.append(f'  run("{name}", [&] {{ hipLaunchKernelGGL(k_{name}, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9); }}, iters);')
out.append('  }\n  return 0; }')
open(sys.argv[1], "w").write("\n".join(out) + "\n")
