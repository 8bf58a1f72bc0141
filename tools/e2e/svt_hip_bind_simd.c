/*
 * svt_hip_bind_simd.c — BENCH / TEST INFRASTRUCTURE of the e2e build only (oracle/Makefile `e2e`): installs the reference's own x86
 * C-intrinsics kernels (SSE2 ... AVX2, the ladder oracle/build_simd.py extracted from aom_dsp_rtcd.c / common_dsp_rtcd.c into
 * oracle/_ref/simd_table.inc and oracle/_ref/libsvtsimd.a) into the RTCD pointers of the patched encoder, so that bench.py has an
 * encoder-level CPU baseline that is more than `--asm c` (the build container has no NASM: the reference's real `--asm avx2` build is
 * not available; pointers whose AVX2 form is NASM keep their best intrinsics form).  svt_hip_bind_install() (svt_hip_bind.c) calls this
 * through a weak reference when SVTAV1_E2E_SIMD=1, instead of loading the HIP library; a maintainer's build of tools/reference_hip.patch
 * does not contain this file.
 */
#include <stdio.h>

#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"

typedef struct SimdRow {
    void **slot;
    void  *fn;
} SimdRow;
#define SIMD_ROW(p, f, isa, want) {(void **)&p, (void *)f},
static const SimdRow simd_rows[] = {
#include "simd_table.inc"
};
#undef SIMD_ROW

int svt_hip_bind_simd_install(void) {
    const int n = (int)(sizeof(simd_rows) / sizeof(simd_rows[0]));
    for (int i = 0; i < n; i++) *simd_rows[i].slot = simd_rows[i].fn;
    return n;
}
