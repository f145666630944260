// layout_bench.hip -- does the ORDER of the (wave, stage) blocks in HBM matter for a kernel in which every wavefront
// walks its own blocks stage by stage (k_factor's access pattern)?   wave-major  [wave][stage][S]  (the product's layout:
// concurrent wavefronts touch blocks 50 x S apart) against stage-major  [stage][wave][S]  (concurrent wavefronts touch
// one contiguous region per stage).  Reads S doubles per (wave, stage) as 8-byte-per-lane loads with the next stage
// prefetched, optionally writes W doubles; 16 384 wavefronts x 50 stages, 64 lanes, two wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/layout_bench tools/layout_bench.hip && tools/layout_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NLD, int NST>
__global__ __launch_bounds__(64, 2) void walk(const double* __restrict__ src, double* __restrict__ dst, int NW, int N, int S, int W,
                                              int stage_major, double* __restrict__ sink) {
    const int w = blockIdx.x, l = threadIdx.x;
    double acc = 0.0;
    double cur[NLD], nxt[NLD];
    auto base = [&](int k, int sz) -> size_t { return stage_major ? ((size_t)k * NW + w) * sz : ((size_t)w * N + k) * sz; };
    for (int j = 0; j < NLD; j++) cur[j] = src[base(N - 1, S) + j * 64 + l];
    for (int k = N - 1; k >= 0; k--) {
        const int kn = k > 0 ? k - 1 : 0;
        for (int j = 0; j < NLD; j++) nxt[j] = src[base(kn, S) + j * 64 + l];
        for (int j = 0; j < NLD; j++) acc = acc * 1.0000001 + cur[j];
        for (int r = 0; r < 200; r++) acc = acc * 1.0000001 + 1e-9;    // stand-in for the stage's arithmetic (~200 dependent FMAs)
        for (int j = 0; j < NST; j++) dst[base(k, W) + j * 64 + l] = acc + j;
        for (int j = 0; j < NLD; j++) cur[j] = nxt[j];
    }
    if (acc == 123.456) sink[0] = acc;
}
int main() {
    const int NW = 16384, N = 50;
    constexpr int NLD = 13, NST = 4;           // 13 x 512 B read (~ A, B, b, ... of four instances), 4 x 512 B written (~ K, d)
    const int S = NLD * 64, W = NST * 64;
    double *src, *dst, *sink;
    (void)hipMalloc(&src, (size_t)NW * N * S * 8); (void)hipMalloc(&dst, (size_t)NW * N * W * 8); (void)hipMalloc(&sink, 8);
    (void)hipMemset(src, 0, (size_t)NW * N * S * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
        for (int sm = 0; sm < 2; sm++) {
            hipLaunchKernelGGL((walk<NLD, NST>), dim3(NW), dim3(64), 0, 0, src, dst, NW, N, S, W, sm, sink);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL((walk<NLD, NST>), dim3(NW), dim3(64), 0, 0, src, dst, NW, N, S, W, sm, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double gb = (double)NW * N * (S + W) * 8 / 1e9;
            printf("%s: %.3f ms, %.2f GB (%.1f read + %.1f written) -> %.2f TB/s\n", sm ? "stage-major [stage][wave]" : "wave-major  [wave][stage]",
                   ms, gb, (double)NW * N * S * 8 / 1e9, (double)NW * N * W * 8 / 1e9, gb / ms);
        }
    return 0;
}
