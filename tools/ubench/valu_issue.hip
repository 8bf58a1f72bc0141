// valu_issue.hip — vector-instruction ISSUE rate per instruction class at 1, 2, 4 and 8 waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue.bin && tools/ubench/valu_issue.bin
// bench.py prices a kernel's vector instructions against these figures (VALU_SLOT_NS); the printed table is committed under
// profiles/ (r03_ubench_valu_issue.txt).  Every kernel runs CH independent chains of ONE instruction (inline asm, so the
// instruction timed is the instruction named); a workgroup is 256 threads = one wave on each of the CU's four SIMDs, and the
// dynamic LDS size pins the number of workgroups per CU, i.e. the waves per SIMD.  Reported: ns per wave-instruction per SIMD
// (= elapsed / instructions each SIMD issued; the best of three launches of ~0.5 ms or more).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
constexpr int CH = 16, IT = 2048;

#define CHAIN3(name, ASM)                                                                                     \
    struct name {                                                                                             \
        static constexpr const char *label = #name;                                                           \
        __device__ static __forceinline__ void op(unsigned &a, unsigned x, unsigned y, unsigned long long m) { \
            asm volatile(ASM : "+v"(a) : "v"(x), "v"(y), "s"(m));                                             \
        }                                                                                                     \
    };
// %0 = accumulator (in/out), %1 = x, %2 = y, %3 = 64-bit scalar mask
CHAIN3(v_add_u32, "v_add_u32 %0, %1, %0")
CHAIN3(v_sub_u32, "v_sub_u32 %0, %1, %0")
CHAIN3(v_and_b32, "v_and_b32 %0, %1, %0")
CHAIN3(v_xor_b32, "v_xor_b32 %0, %1, %0")
CHAIN3(v_lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
CHAIN3(v_ashrrev_i32, "v_ashrrev_i32 %0, 1, %0")
CHAIN3(v_max_i32, "v_max_i32 %0, %1, %0")
CHAIN3(v_min_u32, "v_min_u32 %0, %1, %0")
CHAIN3(v_mov_b32, "v_mov_b32 %0, %1")
CHAIN3(v_cndmask_b32, "v_cndmask_b32_e64 %0, %0, %1, %3")
CHAIN3(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1")
CHAIN3(v_add3_u32, "v_add3_u32 %0, %0, %1, %2")
CHAIN3(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 1, %1")
CHAIN3(v_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
CHAIN3(v_bfe_u32, "v_bfe_u32 %0, %0, 3, 9")
CHAIN3(v_bfi_b32, "v_bfi_b32 %0, %1, %2, %0")
CHAIN3(v_med3_i32, "v_med3_i32 %0, %0, %1, %2")
CHAIN3(v_min3_u32, "v_min3_u32 %0, %0, %1, %2")
CHAIN3(v_mul_i32_i24, "v_mul_i32_i24 %0, %1, %0")
CHAIN3(v_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")
CHAIN3(v_mad_u32_u24, "v_mad_u32_u24 %0, %1, %2, %0")
CHAIN3(v_mul_lo_u32, "v_mul_lo_u32 %0, %1, %0")
CHAIN3(v_mul_hi_u32, "v_mul_hi_u32 %0, %1, %0")
CHAIN3(v_fma_f32, "v_fma_f32 %0, %1, %2, %0")
CHAIN3(v_add_f32, "v_add_f32 %0, %1, %0")
CHAIN3(v_pk_fma_f32_lo, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,0,0]")  /* operands are register PAIRS in the real form: see v_pk_fma_f32 below */
CHAIN3(v_sad_u8, "v_sad_u8 %0, %1, %2, %0")
CHAIN3(v_sad_u16, "v_sad_u16 %0, %1, %2, %0")
CHAIN3(v_alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
CHAIN3(v_perm_b32, "v_perm_b32 %0, %0, %1, %2")
CHAIN3(v_pk_add_i16, "v_pk_add_i16 %0, %1, %0")
CHAIN3(v_pk_sub_i16, "v_pk_sub_i16 %0, %1, %0")
CHAIN3(v_pk_max_i16, "v_pk_max_i16 %0, %1, %0")
CHAIN3(v_pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
CHAIN3(v_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %1, %0")
CHAIN3(v_pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
CHAIN3(v_dot2_u32_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
CHAIN3(v_dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
CHAIN3(v_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
CHAIN3(v_dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
CHAIN3(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
CHAIN3(v_mov_dpp_row_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
CHAIN3(v_add_dpp_row_shr, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
CHAIN3(v_readlane_to_s, "v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0")  /* two instructions: counted as two below */

// 64-bit accumulator forms
#define CHAIN64(name, ASM)                                                                                          \
    struct name {                                                                                                   \
        static constexpr const char *label = #name;                                                                 \
        __device__ static __forceinline__ void op(unsigned long long &a, unsigned x, unsigned y, unsigned long long m) { \
            asm volatile(ASM : "+v"(a) : "v"(x), "v"(y), "s"(m));                                                   \
        }                                                                                                           \
    };
CHAIN64(v_qsad_pk_u16_u8, "v_qsad_pk_u16_u8 %0, %0, %1, %0")
CHAIN64(v_mqsad_pk_u16_u8, "v_mqsad_pk_u16_u8 %0, %0, %1, %0")
CHAIN64(v_mad_u64_u32, "v_mad_u64_u32 %0, s[22:23], %1, %2, %0")
CHAIN64(v_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
CHAIN64(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
CHAIN64(v_pk_add_f32, "v_pk_add_f32 %0, %0, %0")
CHAIN64(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %0")
CHAIN64(v_add_f64, "v_add_f64 %0, %0, %0")
CHAIN64(v_fma_f64, "v_fma_f64 %0, %0, %0, %0")

template <class OP, class ACC>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters) {
    extern __shared__ unsigned char pin[];  // only its size matters (workgroups per CU)
    ACC      a[CH];
    unsigned x = seed + threadIdx.x, y = seed * 3 + threadIdx.x;
    const unsigned long long m = 0x5555555555555555ull ^ seed;
    for (int c = 0; c < CH; c++) a[c] = c + x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) OP::op(a[c], x, y, m);
    }
    ACC s = 0;
    for (int c = 0; c < CH; c++) s += a[c];
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s + pin[threadIdx.x & 3] * 0;
}

static unsigned           *d_out;
static int                 n_cu = 256;
static const size_t        LDS_CU = 160 * 1024;

template <class OP, class ACC>
void run(int per_inst = 1) {
    printf("%-22s", OP::label);
    for (int w : {1, 2, 4, 8}) {
        // w workgroups per CU: each takes 1/w of the LDS (minus a little), so exactly w fit
        const size_t lds = LDS_CU / w - (w == 1 ? 1024 : 512);
        hipFuncSetAttribute((const void *)k<OP, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int  blocks = n_cu * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        const int iters = IT * 8 / w;  // the same number of wave-instructions per SIMD at every occupancy: ~0.5 ms per launch or more
        hipLaunchKernelGGL((k<OP, ACC>), dim3(blocks), dim3(256), lds, 0, d_out, 1u, iters);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<OP, ACC>), dim3(blocks), dim3(256), lds, 0, d_out, 2u + rep, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double inst_per_simd = (double)w * CH * iters * per_inst;  // wave-instructions each SIMD issues
        printf("  w=%d %6.3f", w, best * 1e6 / inst_per_simd);
        hipEventDestroy(e0), hipEventDestroy(e1);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    n_cu = p.multiProcessorCount;
    printf("# %s, %d CUs, clockRate %d kHz; %d chains x %d iterations per lane; ns per wave-instruction per SIMD at w waves per SIMD\n", p.gcnArchName,
           n_cu, p.clockRate, CH, IT);
    hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4);
#define R(op) run<op, unsigned>()
#define R64(op) run<op, unsigned long long>()
    R(v_add_u32); R(v_sub_u32); R(v_and_b32); R(v_xor_b32); R(v_lshlrev_b32); R(v_ashrrev_i32); R(v_max_i32); R(v_min_u32); R(v_mov_b32);
    R(v_cndmask_b32); R(v_lshl_add_u32); R(v_add3_u32); R(v_lshl_or_b32); R(v_and_or_b32); R(v_bfe_u32); R(v_bfi_b32); R(v_med3_i32); R(v_min3_u32);
    R(v_mul_i32_i24); R(v_mad_i32_i24); R(v_mad_u32_u24); R(v_mul_lo_u32); R(v_mul_hi_u32);
    R(v_fma_f32); R(v_add_f32); R(v_cvt_f32_u32);
    R(v_sad_u8); R(v_sad_u16); R(v_alignbyte_b32); R(v_perm_b32);
    R(v_pk_add_i16); R(v_pk_sub_i16); R(v_pk_max_i16); R(v_pk_mad_i16); R(v_pk_mul_lo_u16); R(v_pk_lshlrev_b16);
    R(v_dot2_u32_u16); R(v_dot2_i32_i16); R(v_dot4_u32_u8); R(v_dot4_i32_i8);
    R(v_mov_dpp_row_shr); R(v_add_dpp_row_shr);
    run<v_readlane_to_s, unsigned>(2);
    R64(v_qsad_pk_u16_u8); R64(v_mqsad_pk_u16_u8); R64(v_mad_u64_u32); R64(v_lshlrev_b64);
    R64(v_pk_fma_f32); R64(v_pk_add_f32); R64(v_pk_mul_f32); R64(v_add_f64); R64(v_fma_f64);
    return 0;
}
