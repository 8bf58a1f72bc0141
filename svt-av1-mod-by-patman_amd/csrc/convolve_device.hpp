// convolve_device.hpp — one 64 x 64 tile of one predicted block (the body of convolve_sr_kernel, inter_convolve.hip) as a device
// function, shared with the temporal filter's prediction kernel (tf_picture.hip).  All 256 threads of the workgroup call it;
// `in` / `im` are the workgroup's LDS buffers ((TILE + 7) * IP + CONV_IN_SLACK uint16, 4-byte aligned, and (TILE + 7) * TILE int16).
#pragma once
#include <cstdint>

#include "../../include/svt_hip_inter.h"

namespace svthip {
namespace conv {


constexpr int FILTER_BITS = 7, TILE = 64, IP = TILE + 8;

__device__ __forceinline__ int32_t ldpx(const void *p, ptrdiff_t idx, int is16) {
    return is16 ? ((const __attribute__((address_space(1))) uint16_t *)p)[idx] : ((const __attribute__((address_space(1))) uint8_t *)p)[idx];  // pictures are global memory: no flat loads
}
__device__ __forceinline__ int32_t rnd(int32_t v, int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ void stpx(void *p, size_t idx, int is16, int32_t v, int bd) {
    const int32_t hi = (1 << bd) - 1;
    v                = v < 0 ? 0 : (v > hi ? hi : v);
    if (is16)
        ((__attribute__((address_space(1))) uint16_t *)p)[idx] = (uint16_t)v;
    else
        ((__attribute__((address_space(1))) uint8_t *)p)[idx] = (uint8_t)v;
}

// compound epilogue (inter_prediction.c:531-543 and its siblings): store the offset intermediate, or average with the stored
// one and write the pixel
__device__ __forceinline__ void comp_out(const SvtHipConvolveDesc &d, int y, int x, int32_t res, int32_t round_offset, int round_bits) {
    uint16_t *cb = d.cbuf + (size_t)y * d.cbuf_stride + x;
    if (d.compound == 1) {
        *cb = (uint16_t)res;
    } else {
        int32_t tmp = *cb;
        tmp         = d.compound == 3 ? (tmp * (int32_t)d.fwd_offset + res * (int32_t)d.bck_offset) >> 4 : (tmp + res) >> 1;
        tmp -= round_offset;
        stpx(d.dst, (size_t)y * d.dst_stride + x, d.is_16bit, rnd(tmp, round_bits), d.bit_depth);
    }
}

// acc + sum of f[k] * in[e + k], k = 0 .. 7, the taps as four packed pairs.  The eight samples start at a 2-byte aligned address, and a
// DS access that is not naturally aligned is executed one lane per cycle on gfx950 (tools/ubench/lds_unaligned.hip: 65 cycles per
// wave-instruction) — which is what the compiler makes of eight adjacent 16-bit reads (one ds_read_b128), since the target allows
// unaligned DS access.  So: the five aligned dwords around the samples, funnel-shifted by the parity of e, and four v_dot2_i32_i16
// (samples < 2^15, taps are int16: the sum is the same integer).  Reads one dword past in[e + 7] when e is even: `in` is declared with
// CONV_IN_SLACK spare elements.
constexpr int CONV_IN_SLACK = 2;
typedef short conv_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t hsum8(const uint16_t *in, int e, const uint32_t (&fp)[4], int32_t acc) {
    const uint32_t *q  = (const uint32_t *)in + (e >> 1);
    const uint32_t  sh = (uint32_t)(e & 1) * 16;
    uint32_t        d[5];
#pragma unroll
    for (int k = 0; k < 5; k++) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < 4; k++)
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(conv_i16x2, __builtin_amdgcn_alignbit(d[k + 1], d[k], sh)), __builtin_bit_cast(conv_i16x2, fp[k]), acc, false);
    return acc;
}

// tile: index of the T x T tile inside the block (row-major); a tile beyond the block returns at once.  NT threads call it together
// (the whole workgroup); in / im: (T + 7) * P uint16 and (T + 7) * T int16 of LDS.
template <int NT, int T, int P>
__device__ __forceinline__ void convolve_tile_t(const SvtHipConvolveDesc &d, const int tile, uint16_t *__restrict__ in, int16_t *__restrict__ im) {
    if (d.w == 0 || d.h == 0)  // an unused slot of a fixed-size descriptor array (tf_picture.hip)
        return;
    const int tiles_x = (d.w + T - 1) / T;
    const int x0 = (tile % tiles_x) * T, y0 = (tile / tiles_x) * T;
    if (y0 >= d.h)
        return;
    const int tw = min(T, d.w - x0), th = min(T, d.h - y0);
    const int tx = d.taps_x, ty = d.taps_y, is16 = d.is_16bit, bd = d.bit_depth, r0 = d.round_0, r1 = d.round_1;
    const int fo_h = tx ? tx / 2 - 1 : 0, fo_v = ty ? ty / 2 - 1 : 0;
    // the taps are uniform: once into scalar registers, padded with zeros to eight so that the tap loops unroll (indexing the
    // descriptor's arrays with a run-time tap counter put them in scratch memory: a memory access per tap and sample); the samples
    // a zero tap multiplies lie inside the LDS buffers (rows of T + 8, T + 7 rows)
    int32_t fx[8], fy[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        fx[k] = __builtin_amdgcn_readfirstlane(k < tx ? (int)d.filter_x[k] : 0);
        fy[k] = __builtin_amdgcn_readfirstlane(k < ty ? (int)d.filter_y[k] : 0);
    }
    uint32_t fxp[4];  // the horizontal taps as packed pairs
#pragma unroll
    for (int k = 0; k < 4; k++) fxp[k] = ((uint32_t)fx[2 * k] & 0xffffu) | ((uint32_t)fx[2 * k + 1] << 16);
    const int ew = tw + (tx ? tx - 1 : 0), eh = th + (ty ? ty - 1 : 0);  // staged extent
    for (int idx = threadIdx.x; idx < eh * ew; idx += NT) {
        const int r = idx / ew, c = idx - r * ew;
        in[r * P + c] = (uint16_t)ldpx(d.src, (ptrdiff_t)(y0 + r - fo_v) * d.src_stride + (x0 + c - fo_h), is16);
    }
    __syncthreads();
    if (d.compound) {  // jnt_convolve_{2d_copy, x, y, 2d}
        const int     offset_bits = bd + 2 * FILTER_BITS - r0, round_bits = 2 * FILTER_BITS - r0 - r1;
        const int32_t round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
        if (tx && ty) {
            for (int idx = threadIdx.x; idx < eh * tw; idx += NT) {
                const int r = idx / tw, c = idx - r * tw;
                int32_t   sum = 1 << (bd + FILTER_BITS - 1);
                sum = hsum8(in, r * P + c, fxp, sum);
                im[r * T + c] = (int16_t)(uint16_t)rnd(sum, r0);
            }
            __syncthreads();
        }
        for (int idx = threadIdx.x; idx < th * tw; idx += NT) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res;
            if (tx && ty) {
                int32_t sum = 1 << offset_bits;
                _Pragma("unroll") for (int k = 0; k < 8; k++) sum += fy[k] * (int32_t)im[(r + k) * T + c];
                res = (uint16_t)rnd(sum, r1);
            } else if (ty) {
                res = 0;
                _Pragma("unroll") for (int k = 0; k < 8; k++) res += fy[k] * (int32_t)in[(r + k) * P + c];
                res *= 1 << (FILTER_BITS - r0);
                res = rnd(res, r1) + round_offset;
            } else if (tx) {
                res = 0;
                res = hsum8(in, r * P + c, fxp, res);
                res = (1 << (FILTER_BITS - r1)) * rnd(res, r0) + round_offset;
            } else {
                res = (uint16_t)((uint16_t)((int32_t)in[r * P + c] << round_bits) + (uint16_t)round_offset);
            }
            comp_out(d, y0 + r, x0 + c, res, round_offset, round_bits);
        }
        return;
    }
    if (!tx && !ty) {  // 2d_copy_sr
        for (int idx = threadIdx.x; idx < th * tw; idx += NT) {
            const int r = idx / tw, c = idx - r * tw;
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, in[r * P + c], 16);
        }
        return;
    }
    if (!ty) {  // x_sr
        const int bits = FILTER_BITS - r0;
        for (int idx = threadIdx.x; idx < th * tw; idx += NT) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res = 0;
            res = hsum8(in, r * P + c, fxp, res);
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(rnd(res, r0), bits), bd);
        }
        return;
    }
    if (!tx) {  // y_sr
        for (int idx = threadIdx.x; idx < th * tw; idx += NT) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res = 0;
            _Pragma("unroll") for (int k = 0; k < 8; k++) res += fy[k] * (int32_t)in[(r + k) * P + c];
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(res, FILTER_BITS), bd);
        }
        return;
    }
    // 2d_sr
    for (int idx = threadIdx.x; idx < eh * tw; idx += NT) {
        const int r = idx / tw, c = idx - r * tw;
        int32_t   sum = 1 << (bd + FILTER_BITS - 1);
        sum = hsum8(in, r * P + c, fxp, sum);
        im[r * T + c] = (int16_t)(uint16_t)rnd(sum, r0);
    }
    __syncthreads();
    const int bits = 2 * FILTER_BITS - r0 - r1, offset_bits = bd + 2 * FILTER_BITS - r0;
    for (int idx = threadIdx.x; idx < th * tw; idx += NT) {
        const int r = idx / tw, c = idx - r * tw;
        int32_t   sum = 1 << offset_bits;
        _Pragma("unroll") for (int k = 0; k < 8; k++) sum += fy[k] * (int32_t)im[(r + k) * T + c];
        int32_t res = rnd(sum, r1) - ((1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)));
        if (!is16)
            res = (int16_t)res;  // the 8-bit function narrows to int16 first (inter_prediction.c:343-345)
        stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(res, bits), bd);
    }

}

__device__ __forceinline__ void convolve_tile(const SvtHipConvolveDesc &d, const int tile, uint16_t *__restrict__ in, int16_t *__restrict__ im) {
    convolve_tile_t<256, TILE, IP>(d, tile, in, im);
}

}  // namespace conv
}  // namespace svthip
