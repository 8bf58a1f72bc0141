// int_rates.hip — issue-rate probe for the 32 / 64-bit integer forms that show up in address arithmetic (gfx950; harness of valu_rates.hip)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/int_rates.hip -o tools/ubench/int_rates.bin && tools/ubench/int_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int CH = 8, IT = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    uint64_t a[CH];
    uint32_t b[CH];
    uint32_t x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;
    for (int c = 0; c < CH; c++) a[c] = c, b[c] = c + 1;
    for (int i = 0; i < IT; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) b[c] = b[c] * (y + c);                                            // v_mul_lo_u32
            if (OP == 1) b[c] = __umulhi(b[c], y + c) + 1;                                 // v_mul_hi_u32 (+ add)
            if (OP == 2) a[c] = (uint64_t)(uint32_t)a[c] * (uint64_t)(y + c) + a[c];       // v_mad_u64_u32
            if (OP == 3) a[c] = (a[c] << 1) + (uint64_t)(x + c);                           // v_lshl_add_u64
            if (OP == 4) a[c] = a[c] + (((uint64_t)x << 32) | (y + c));                    // 64-bit add (add_co + addc)
            if (OP == 5) b[c] = __builtin_amdgcn_ubfe(b[c] + x, 3, 9) + c;                 // v_bfe_u32 (+ add)
            if (OP == 6) a[c] = (uint64_t)((int64_t)(int32_t)(uint32_t)a[c] * (int64_t)(int32_t)(y + c)) + a[c];  // v_mad_i64_i32
        }
        x += 0x10001;
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += (uint32_t)a[c] + (uint32_t)(a[c] >> 32) + b[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *d, double per) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 8.0 * CH * IT;
    printf("%-34s %8.3f ms  -> %.2f ns per source operation per SIMD (%g instruction(s) each)\n", name, ms, ms * 1e6 / inst_per_simd, per);
}
int main() {
    unsigned *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_mul_lo_u32", d, 1);
    run<1>("v_mul_hi_u32 + v_add", d, 2);
    run<2>("v_mad_u64_u32", d, 1);
    run<6>("v_mad_i64_i32", d, 1);
    run<3>("v_lshl_add_u64", d, 1);
    run<4>("64-bit add", d, 2);
    run<5>("v_add + v_bfe_u32 + v_add", d, 3);
    return 0;
}
