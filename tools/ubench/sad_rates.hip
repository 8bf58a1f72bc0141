// sad_rates.hip — issue-rate probe for the SAD instruction forms of gfx950 (same harness as valu_rates.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sad_rates.hip -o tools/ubench/sad_rates.bin && tools/ubench/sad_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int CH = 8, IT = 4096;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
    uint64_t a[CH];
    uint32_t b[CH];
    u32x4    q[CH];
    uint64_t x = ((uint64_t)(seed + threadIdx.x) << 32) | (seed * 7 + threadIdx.x);
    uint32_t y = seed * 3 + threadIdx.x;
    for (int c = 0; c < CH; c++) a[c] = c, b[c] = c, q[c] = u32x4{(uint32_t)c, 0, 0, 0};
    for (int i = 0; i < IT; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) a[c] = __builtin_amdgcn_qsad_pk_u16_u8(x, y + c, a[c]);
            if (OP == 1) a[c] = __builtin_amdgcn_mqsad_pk_u16_u8(x, y + c, a[c]);
            if (OP == 2) q[c] = __builtin_amdgcn_mqsad_u32_u8(x, y + c, q[c]);
            if (OP == 3) b[c] = __builtin_amdgcn_sad_u8((uint32_t)x, y + c, b[c]);
            if (OP == 4) b[c] = __builtin_amdgcn_msad_u8((uint32_t)x, y + c, b[c]);
            if (OP == 5) b[c] = __builtin_amdgcn_sad_u16((uint32_t)x, y + c, b[c]);
            if (OP == 6) b[c] = __builtin_amdgcn_alignbyte((uint32_t)x, b[c], 1u) + c;
            if (OP == 7) b[c] = __builtin_amdgcn_sad_hi_u8((uint32_t)x, y + c, b[c]);
        }
        x += 0x100000001ull;
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s += (uint32_t)a[c] + (uint32_t)(a[c] >> 32) + b[c] + q[c].x + q[c].y + q[c].z + q[c].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *d) {
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
    printf("%-22s %8.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / inst_per_simd);
}
int main() {
    unsigned *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("v_sad_u8", d);
    run<0>("v_qsad_pk_u16_u8", d);
    run<1>("v_mqsad_pk_u16_u8", d);
    run<2>("v_mqsad_u32_u8", d);
    run<4>("v_msad_u8", d);
    run<5>("v_sad_u16", d);
    run<7>("v_sad_hi_u8", d);
    run<6>("v_alignbyte+add", d);
    return 0;
}
