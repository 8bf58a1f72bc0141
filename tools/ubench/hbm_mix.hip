// hbm_mix.hip — what HBM rate a plain streaming kernel reaches on this GPU for the read : write mixes of the hot-path kernels
// (the practical roof beside the nominal 8 TB/s): pure read, pure write, copy, and the transform pass's 4 bytes read : 14 bytes
// written per sample (residual + prediction in; coefficients, quantised, dequantised, reconstruction out).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_mix.hip -o tools/ubench/hbm_mix.bin && tools/ubench/hbm_mix.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
// per work item: R 16-byte loads, W 16-byte stores (all streams disjoint, coalesced, non-temporal)
template <int R, int W>
__global__ __launch_bounds__(256) void k(const v4i *__restrict__ in, v4i *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; r++) acc += __builtin_nontemporal_load(in + (size_t)r * n + i);
        if (R == 0)
            acc = v4i{(int)i, 1, 2, 3};
#pragma unroll
        for (int w = 0; w < W; w++) __builtin_nontemporal_store(acc + w, out + (size_t)w * n + i);
        if (W == 0 && acc.x == 0x7fffffff)
            out[i] = acc;
    }
}
template <int R, int W> void run(const char *name, v4i *in, v4i *out, size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 16;
    hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e0);
    for (int it = 0; it < 5; it++) hipLaunchKernelGGL((k<R, W>), dim3(blocks), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double bytes = (double)(R + W) * n * 16;
    printf("%-28s %8.3f ms  %7.1f MB  -> %7.1f GB/s (%.2f of 8 TB/s)\n", name, ms, bytes / 1e6, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
}
int main() {
    const size_t n = (size_t)6 << 20;  // 16-byte items per stream: 96 MiB per stream, the working set is far beyond the 256 MiB Infinity Cache
    v4i *in, *out;
    hipMalloc(&in, 8 * n * 16), hipMalloc(&out, 8 * n * 16);
    hipMemset(in, 1, 8 * n * 16), hipMemset(out, 0, 8 * n * 16);
    run<8, 0>("read only", in, out, n);
    run<0, 8>("write only", in, out, n);
    run<4, 4>("copy 1:1", in, out, n);
    run<2, 7>("read 4 : write 14 (txfm)", in, out, n);
    run<7, 2>("read 14 : write 4", in, out, n);
    return 0;
}
