"""Generates tools/dev/fetch_bench.hip: is a single wave per SIMD limited by instruction fetch?
Same 16-FMA loop body with (a) different code alignment (leading s_nop padding), (b) 4-byte VOP2
vs 8-byte VOP3 encodings of the plain FP64 FMA, (c) DPP (8 bytes).  Development aid."""
import sys
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>']
names = []
NF = 512  # FMAs per loop iteration: long straight-line body, branch / refetch effects negligible
def kern(name, pad, ins):
    names.append(name)
    body = '".p2align 8\\n\\t"' + ''.join(' "s_nop 0\\n\\t"' for _ in range(pad)) + ' "1:\\n\\t"' + \
           ''.join(f' "{ins(i)}\\n\\t"' for i in range(NF)) + \
           ' "s_add_i32 %10, %10, -1\\n\\t" "s_cmp_lg_u32 %10, 0\\n\\t" "s_cbranch_scc1 1b\\n\\t"'
    out.append(f'''__global__ __launch_bounds__(64) void k_{name}(double* out, int iters, double a, double b) {{
    double c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    double x = a + threadIdx.x * 1e-9, y = b;
    int n = iters;
    asm volatile({body}
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x), "v"(y), "s"(n) : "scc");
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}}''')
for K in (1, 2, 4, 8):
    kern(f"dpp_K{K}", 0, lambda i, K=K: f"v_fmac_f64_dpp %{i % K}, %8, %9 row_newbcast:{i % 13} row_mask:0xf bank_mask:0xf")
    kern(f"vop2_K{K}", 0, lambda i, K=K: f"v_fmac_f64_e32 %{i % K}, %8, %9")
kern("vop3_K8", 0, lambda i: f"v_fma_f64 %{i % 8}, %8, %9, %{i % 8}")
kern("dpp_K8_pad5", 5, lambda i: f"v_fmac_f64_dpp %{i % 8}, %8, %9 row_newbcast:{i % 13} row_mask:0xf bank_mask:0xf")
out.append('''template <class F> static void run(const char* name, F launch, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s %.3f ms -> %.2f ns per FMA instruction\\n", name, ms, ms * 1e6 / ((double)iters * 512));
}
int main() { double* out; hipMalloc(&out, 8 * 64 * 4096); const int iters = 1000;
  for (int blocks : {1024, 2048, 4096}) { printf("-- %d waves per SIMD\\n", blocks / 1024);''')
for n in names:
    out.append(f'  run("{n}", [&] {{ hipLaunchKernelGGL(k_{n}, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 1e-9); }}, iters);')
out.append('  }\n  return 0; }')
open(sys.argv[1], "w").write("\n".join(out) + "\n")
