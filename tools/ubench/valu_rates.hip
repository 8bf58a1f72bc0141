// valu_rates.hip — issue-rate probe for the integer multiply-accumulate forms the statistics kernels can use (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates.bin && tools/ubench/valu_rates.bin
// Every kernel runs N dependent-free chains of one instruction per lane; the reported figure is wave-instructions per
// SIMD-cycle equivalents: time per instruction per wave in ns with 8 waves per SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short          i16x2 __attribute__((ext_vector_type(2)));
constexpr int CH = 16, IT = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    unsigned a[CH], x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;
    for (int c = 0; c < CH; c++) a[c] = c;
    for (int i = 0; i < IT; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) a[c] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y + c), a[c], false);
            if (OP == 1) a[c] = __umul24(x, y + c) + a[c];
            if (OP == 2) a[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, y + c), (int)a[c], false);
            if (OP == 3) a[c] = __builtin_amdgcn_udot4(x, y + c, a[c], false);
            if (OP == 4) a[c] = x * (y + c) + a[c];
            if (OP == 5) a[c] = __builtin_amdgcn_sad_u8(x, y + c, a[c]);
            if (OP == 6) { i16x2 r = __builtin_bit_cast(i16x2, x) * __builtin_bit_cast(i16x2, y + c) + __builtin_bit_cast(i16x2, a[c]); a[c] = __builtin_bit_cast(unsigned, r); }
        }
        x += 0x10001;
    }
    unsigned s = 0;
    for (int c = 0; c < CH; c++) s += a[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 workgroups (32 waves) per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 8.0 * CH * IT;  // wave-instructions each SIMD has to issue
    printf("%-22s %8.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / inst_per_simd);
}
int main() {
    unsigned *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1>("v_mad_u32_u24", d);
    run<0>("v_dot2_u32_u16", d);
    run<2>("v_dot2_i32_i16", d);
    run<3>("v_dot4_u32_u8", d);
    run<4>("v_mad_u32 (mul_lo+add)", d);
    run<5>("v_sad_u8", d);
    run<6>("v_pk_mad_i16", d);
    return 0;
}
