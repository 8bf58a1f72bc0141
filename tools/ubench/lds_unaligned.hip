// lds_unaligned.hip — what an LDS read costs on gfx950 by width and alignment, when every lane reads a window that overlaps its
// neighbour's (lane c reads bytes [c * STEP + OFF, + WIDTH)): the access pattern of a per-lane FIR filter over a row in LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_unaligned.hip -o tools/ubench/lds_unaligned.bin && tools/ubench/lds_unaligned.bin
// Prints ns per wave-instruction per CU with 4 waves per SIMD, four accesses in flight per wave, nothing else issued (the LDS
// pipe is the only busy unit).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int IT = 2048;
typedef unsigned v4 __attribute__((ext_vector_type(4)));
typedef unsigned v2 __attribute__((ext_vector_type(2)));
template <int WIDTH, int STEP, int OFF, int ACTIVE = 64>
__global__ __launch_bounds__(256) void k(unsigned *out, int rows) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = (uint8_t)(i * 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned  acc = 0;
    int       off = lane * STEP + OFF;
    if (lane >= ACTIVE)
        return;
    for (int i = 0; i < IT; i += 4) {  // four accesses in flight per wave: the LDS pipe, not its latency, is what is measured
        const unsigned a0 = (unsigned)(uintptr_t)(buf + off), a1 = (unsigned)(uintptr_t)(buf + ((off + rows) & 8191)),
                       a2 = (unsigned)(uintptr_t)(buf + ((off + 2 * rows) & 8191)), a3 = (unsigned)(uintptr_t)(buf + ((off + 3 * rows) & 8191));
        if (WIDTH == -16) {
            v4 v = {acc, acc, acc, acc};
            asm volatile("ds_write_b128 %0, %4\n ds_write_b128 %1, %4\n ds_write_b128 %2, %4\n ds_write_b128 %3, %4\n s_waitcnt lgkmcnt(0)"
                         :
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v)
                         : "memory");
            acc += i;
        } else if (WIDTH == -8) {
            v2 v = {acc, acc};
            asm volatile("ds_write_b64 %0, %4\n ds_write_b64 %1, %4\n ds_write_b64 %2, %4\n ds_write_b64 %3, %4\n s_waitcnt lgkmcnt(0)"
                         :
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v)
                         : "memory");
            acc += i;
        } else if (WIDTH == 16) {
            v4 v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                         : "memory");
            acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
        } else if (WIDTH == 8) {
            v2 v0, v1, v2, v3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                         : "memory");
            acc += v0.x ^ v1.y ^ v2.x ^ v3.y;
        } else if (WIDTH == 4) {
            unsigned v0, v1, v2, v3;
            asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                         : "memory");
            acc += v0 ^ v1 ^ v2 ^ v3;
        } else {
            unsigned v0, v1, v2, v3;
            asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %5\n ds_read_u16 %2, %6\n ds_read_u16 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                         : "memory");
            acc += v0 ^ v1 ^ v2 ^ v3;
        }
        off = (off + 4 * rows) & 8191;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int WIDTH, int STEP, int OFF, int ACTIVE = 64> void run(const char *name, unsigned *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    const int rows = 256;        // keeps the alignment class of every lane
    hipLaunchKernelGGL((k<WIDTH, STEP, OFF, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, rows);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WIDTH, STEP, OFF, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = 16.0 * IT;  // wave-instructions per CU
    printf("%-52s %7.3f ms -> %6.2f ns per wave-instruction per CU = %5.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
}
int main() {
    unsigned *d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    run<16, 16, 0>("b128, lane stride 16 B, 16 B aligned", d);
    run<16, 2, 0>("b128, lane stride 2 B (overlapping), even lanes 4 B aligned", d);
    run<16, 4, 0>("b128, lane stride 4 B (overlapping), 4 B aligned", d);
    run<16, 4, 2>("b128, lane stride 4 B (overlapping), 2 B aligned", d);
    run<8, 8, 0>("b64, lane stride 8 B, aligned", d);
    run<8, 1, 0>("b64, lane stride 1 B (overlapping), any alignment", d);
    run<8, 2, 0>("b64, lane stride 2 B (overlapping), 2 B aligned", d);
    run<8, 4, 0>("b64, lane stride 4 B (overlapping), 4 B aligned", d);
    run<8, 4, 1>("b64, lane stride 4 B (overlapping), 1 B off", d);
    run<4, 4, 0>("b32, lane stride 4 B, aligned", d);
    run<4, 2, 0>("b32, lane stride 2 B (overlapping), 2 B aligned", d);
    run<4, 1, 0>("b32, lane stride 1 B (overlapping), any alignment", d);
    run<16, 16, 4>("b128, lane stride 16 B (disjoint), 4 B aligned", d);
    run<16, 16, 8>("b128, lane stride 16 B (disjoint), 8 B aligned", d);
    run<8, 8, 4>("b64, lane stride 8 B (disjoint), 4 B aligned", d);
    run<16, 16, 4, 1>("b128, 4 B aligned, ONE active lane", d);
    run<16, 16, 4, 8>("b128, 4 B aligned, 8 active lanes", d);
    run<-16, 16, 0>("write b128, lane stride 16 B, aligned", d);
    run<-16, 16, 4>("write b128, lane stride 16 B, 4 B aligned", d);
    run<-16, 16, 8>("write b128, lane stride 16 B, 8 B aligned", d);
    run<-8, 8, 0>("write b64, lane stride 8 B, aligned", d);
    run<-8, 8, 4>("write b64, lane stride 8 B, 4 B aligned", d);
    run<-16, 16, 4, 4>("write b128, 4 B aligned, 4 active lanes", d);
    run<2, 2, 0>("u16, lane stride 2 B", d);
    run<2, 1, 0>("u16, lane stride 1 B, any alignment", d);
    return 0;
}
