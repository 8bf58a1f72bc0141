// lr_device.hpp — device code shared by the loop-restoration kernels (loopfilter_sgr.hip, loopfilter_wiener.hip,
// loopfilter_lr_frame.hip): the self-guided filter and the Wiener filter of ONE processing unit whose samples (with their
// 3-sample border) already sit in LDS.  Who fills the tile decides what the filter sees: straight picture samples for the
// per-unit entry points, the stripe-boundary rows of svt_aom_setup_processing_stripe_boundary for the frame-level pass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svthip {
namespace lr {

constexpr int RST_BITS = 4, PRJ_BITS = 7, SGR_BITS = 8, MTABLE_BITS = 20, RECIP_BITS = 12;
constexpr int TP = 72;   // LDS pitch of the sample tile (64 + 2*3 = 70 used)
constexpr int AP = 67;   // LDS pitch of the A / B maps (66 used, odd)
constexpr int PRJ_MIN0 = -96, PRJ_MAX0 = 31, PRJ_MIN1 = -32, PRJ_MAX1 = 95;  // restoration.h:101-104

static __device__ const int32_t SGR_PRM[16][4] = {/* r0, r1, s0, s1: svt_aom_eb_sgr_params (restoration.c:85-103) */
                                           {2, 1, 140, 3236}, {2, 1, 112, 2158}, {2, 1, 93, 1618}, {2, 1, 80, 1438},
                                           {2, 1, 70, 1295},  {2, 1, 58, 1177},  {2, 1, 47, 1079}, {2, 1, 37, 996},
                                           {2, 1, 30, 925},   {2, 1, 25, 863},   {0, 1, -1, 2589}, {0, 1, -1, 1618},
                                           {0, 1, -1, 1177},  {0, 1, -1, 925},   {2, 0, 56, -1},   {2, 0, 22, -1}};

__device__ __forceinline__ int32_t rnd(int32_t v, int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ int32_t ldpx(const void *p, size_t idx, int is16) {
    return is16 ? ((const __attribute__((address_space(1))) uint16_t *)p)[idx] : ((const __attribute__((address_space(1))) uint8_t *)p)[idx];  // pictures are global memory: no flat loads
}


// A / B of one map position from the LDS tile (restoration.c:709-770 / 842-903); (i, j) relative to the unit, r = 1 | 2
template <int R> __device__ __forceinline__ void ab_at(const uint16_t *tile, int i, int j, uint32_t s, int bd, int32_t &A, int32_t &B) {
    // box sums over (2R + 1)^2 samples.  A lane's 2R + 1 samples of a row start at a 2-byte aligned address; read as adjacent 16-bit
    // values they become one unaligned ds_read_b64, which gfx950 executes one lane per cycle (tools/ubench/lds_unaligned.hip).  So: the
    // R + 1 aligned dwords that hold them, funnel-shifted by the lane's parity (the pitch is even: one parity for all rows), the odd
    // sample masked; sum and sum of squares by v_dot2_u32_u16.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const int      e0 = (i + 3 - R) * TP + j + 3 - R;
    const uint32_t sh = (uint32_t)(e0 & 1) * 16;
    const u16x2    ones = {1, 1};
    uint32_t       sum = 0, ssq = 0;
#pragma unroll
    for (int dy = 0; dy <= 2 * R; dy++) {
        const uint32_t *q = (const uint32_t *)tile + ((e0 + dy * TP) >> 1);
        uint32_t        d[R + 1];
#pragma unroll
        for (int k = 0; k <= R; k++) d[k] = q[k];
#pragma unroll
        for (int k = 0; k <= R; k++) {
            uint32_t pr = __builtin_amdgcn_alignbit(d[k < R ? k + 1 : k], d[k], sh);
            if (k == R)
                pr &= 0xffffu;
            const u16x2 v = __builtin_bit_cast(u16x2, pr);
            sum = __builtin_amdgcn_udot2(v, ones, sum, false), ssq = __builtin_amdgcn_udot2(v, v, ssq, false);
        }
    }
    constexpr uint32_t n = (2 * R + 1) * (2 * R + 1), one_by_n = (4096 + n / 2) / n;  // svt_aom_eb_one_by_x[n - 1]
    const uint32_t a = (ssq + ((1u << (2 * (bd - 8))) >> 1)) >> (2 * (bd - 8)), b = (sum + ((1u << (bd - 8)) >> 1)) >> (bd - 8);
    const uint32_t p = (a * n < b * b) ? 0u : a * n - b * b;
    uint32_t       z = (p * s + (1u << (MTABLE_BITS - 1))) >> MTABLE_BITS;
    z                = z > 255 ? 255 : z;
    A = z == 0 ? 1 : (z == 255 ? 256 : (int32_t)((256 * z + (z + 1) / 2) / (z + 1)));  // svt_aom_eb_x_by_xplus1[z]
    B = (int32_t)(((uint32_t)(256 - A) * sum * one_by_n + (1u << (RECIP_BITS - 1))) >> RECIP_BITS);
}

// Filter of one processing unit of w x h samples (<= 64 x 64) from the LDS tile (pitch TP, sample (0,0) at [3][3]); Am / Bm are
// 66 x AP int32 scratch maps in LDS.  MODE 0: write flt0 / flt1 (at [i0 + i][j0 + j]).  MODE 1: fused
// svt_apply_selfguided_restoration (projection with xq, clip, store samples to dst).  Called by all NT threads of the workgroup
// after a barrier that made the tile visible; contains barriers.
template <int MODE, int NT>
__device__ __forceinline__ void sgr_tile_filter(const uint16_t *tile, int32_t *Am, int32_t *Bm, int tid, int w, int h, int i0, int j0, int ep, int bd,
                                                int is16, int32_t *__restrict__ flt0, int32_t *__restrict__ flt1, uint32_t flt_stride,
                                                void *__restrict__ dst, uint32_t dst_stride, int xq0, int xq1) {
    const int r0 = SGR_PRM[ep][0], r1 = SGR_PRM[ep][1];
    constexpr int PER = 64 * 64 / NT;  // samples per thread of a 64x64 unit
    int32_t       f0[PER];
    if (r0 > 0) {  // selfguided_restoration_fast_internal: maps on rows -1, 1, 3, ...
        const int nrows = (h + 2 + 1) / 2, W2 = w + 2;
        for (int idx = tid; idx < nrows * W2; idx += NT) {
            const int ii = idx / W2, j = idx - ii * W2 - 1, i = 2 * ii - 1;
            ab_at<2>(tile, i, j, (uint32_t)SGR_PRM[ep][2], bd, Am[(i + 1) * AP + j + 1], Bm[(i + 1) * AP + j + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int idx = tid + k * NT;
            f0[k]         = 0;
            if (idx < w * h) {
                const int  i = idx / w, j = idx - i * w;
                const int *A = &Am[(i + 1) * AP + j + 1], *B = &Bm[(i + 1) * AP + j + 1];
                int32_t    a, b, nb;
                if (!(i & 1)) {
                    nb = 5;
                    a  = (A[-AP] + A[AP]) * 6 + (A[-AP - 1] + A[AP - 1] + A[-AP + 1] + A[AP + 1]) * 5;
                    b  = (B[-AP] + B[AP]) * 6 + (B[-AP - 1] + B[AP - 1] + B[-AP + 1] + B[AP + 1]) * 5;
                } else {
                    nb = 4;
                    a  = A[0] * 6 + (A[-1] + A[1]) * 5;
                    b  = B[0] * 6 + (B[-1] + B[1]) * 5;
                }
                const int32_t v = a * (int32_t)tile[(i + 3) * TP + j + 3] + b;
                f0[k]           = rnd(v, SGR_BITS + nb - RST_BITS);
                if (MODE == 0)
                    flt0[(size_t)(i0 + i) * flt_stride + j0 + j] = f0[k];
            }
        }
        __syncthreads();
    }
    if (r1 > 0) {  // selfguided_restoration_internal (r = 1): maps on every row
        const int W2 = w + 2;
        for (int idx = tid; idx < (h + 2) * W2; idx += NT) {
            const int ii = idx / W2, j = idx - ii * W2 - 1, i = ii - 1;
            ab_at<1>(tile, i, j, (uint32_t)SGR_PRM[ep][3], bd, Am[(i + 1) * AP + j + 1], Bm[(i + 1) * AP + j + 1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int idx = tid + k * NT;
        if (idx >= w * h)
            continue;
        const int i = idx / w, j = idx - i * w;
        int32_t   f1 = 0;
        if (r1 > 0) {
            const int    *A = &Am[(i + 1) * AP + j + 1], *B = &Bm[(i + 1) * AP + j + 1];
            const int32_t a = (A[0] + A[-1] + A[1] + A[-AP] + A[AP]) * 4 + (A[-AP - 1] + A[AP - 1] + A[-AP + 1] + A[AP + 1]) * 3;
            const int32_t b = (B[0] + B[-1] + B[1] + B[-AP] + B[AP]) * 4 + (B[-AP - 1] + B[AP - 1] + B[-AP + 1] + B[AP + 1]) * 3;
            f1              = rnd(a * (int32_t)tile[(i + 3) * TP + j + 3] + b, SGR_BITS + 5 - RST_BITS);
            if (MODE == 0)
                flt1[(size_t)(i0 + i) * flt_stride + j0 + j] = f1;
        }
        if (MODE == 1) {
            const int32_t u = (int32_t)tile[(i + 3) * TP + j + 3] << RST_BITS;
            int32_t       v = u << PRJ_BITS;
            if (r0 > 0)
                v += xq0 * (f0[k] - u);
            if (r1 > 0)
                v += xq1 * (f1 - u);
            const int16_t wv = (int16_t)rnd(v, PRJ_BITS + RST_BITS);
            const int32_t hi = (1 << bd) - 1, o = wv < 0 ? 0 : (wv > hi ? hi : wv);
            const size_t  di = (size_t)(i0 + i) * dst_stride + j0 + j;
            if (is16)
                ((uint16_t *)dst)[di] = (uint16_t)o;
            else
                ((uint8_t *)dst)[di] = (uint8_t)o;
        }
    }
}

// Separable 7-tap Wiener filter of one tw x th (<= 64 x 64) unit: `in` = LDS tile of (th + 7) x (tw + 7) samples (4-byte aligned, WIENER_IN_SLACK readable elements behind it), pitch IP, sample
// (0,0) at [3][3]; `tmp` = (th + 7) x 64 uint16 LDS scratch.  svt_av1_(highbd_)wiener_convolve_add_src (convolve.c:57-200).
constexpr int WIENER_IP = 64 + 8, WIENER_IN_SLACK = 2;  // the horizontal pass reads one dword past the last sample of a row
typedef short lr_i16x2 __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ void wiener_tile_filter(const uint16_t *in, uint16_t *tmp, int tid, int tw, int th, int x0, int y0, const int16_t *fx,
                                                   const int16_t *fy, int bd, int r0, int r1, int is16, void *__restrict__ dst, uint32_t dst_stride) {
    constexpr int IP = WIENER_IP;
    const int limit = (1 << (bd + 1 + 7 - r0)) - 1;
    // the seven horizontal taps (+ 128 on the centre one: the "add_src" term) as four packed pairs, the 8th coefficient is zero by
    // construction.  The seven samples of a lane start at a 2-byte aligned address: adjacent 16-bit reads become ONE unaligned
    // ds_read_b64 / b128, which gfx950 executes one lane per cycle (tools/ubench/lds_unaligned.hip) — read the aligned dwords around
    // them and funnel-shift by the lane's parity instead; v_dot2_i32_i16 does two taps per instruction.
    uint32_t fxp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t lo = fx[2 * k] + (2 * k == 3 ? 128 : 0), hi2 = 2 * k + 1 < 7 ? fx[2 * k + 1] + (2 * k + 1 == 3 ? 128 : 0) : 0;
        fxp[k] = __builtin_amdgcn_readfirstlane((int32_t)(((uint32_t)lo & 0xffffu) | ((uint32_t)hi2 << 16)));
    }
    for (int idx = tid; idx < (th + 7) * tw; idx += NT) {
        const int r = idx / tw, c = idx - r * tw, e = r * IP + c;
        const uint32_t *q  = (const uint32_t *)in + (e >> 1);
        const uint32_t  sh = (uint32_t)(e & 1) * 16;
        uint32_t        d[5];
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = q[k];
        int32_t sum = 1 << (bd + 7 - 1);
#pragma unroll
        for (int k = 0; k < 4; k++)
            sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(lr_i16x2, __builtin_amdgcn_alignbit(d[k + 1], d[k], sh)), __builtin_bit_cast(lr_i16x2, fxp[k]), sum, false);
        const int32_t v = (sum + ((1 << r0) >> 1)) >> r0;
        tmp[r * 64 + c] = (uint16_t)(v < 0 ? 0 : (v > limit ? limit : v));
    }
    __syncthreads();
    const int hi = (1 << bd) - 1;
    for (int idx = tid; idx < th * tw; idx += NT) {
        const int r = idx / tw, c = idx - r * tw;
        int32_t   sum = ((int32_t)tmp[(r + 3) * 64 + c] << 7) - (1 << (bd + r1 - 1));
#pragma unroll
        for (int k = 0; k < 7; k++) sum += (int32_t)tmp[(r + k) * 64 + c] * fy[k];
        int32_t v = (sum + ((1 << r1) >> 1)) >> r1;
        v         = v < 0 ? 0 : (v > hi ? hi : v);
        const size_t o = (size_t)(y0 + r) * dst_stride + x0 + c;
        if (is16)
            ((uint16_t *)dst)[o] = (uint16_t)v;
        else
            ((uint8_t *)dst)[o] = (uint8_t)v;
    }
}

}  // namespace lr
}  // namespace svthip
