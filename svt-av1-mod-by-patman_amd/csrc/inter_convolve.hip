// inter_convolve.hip — inter-prediction interpolation on gfx950: single-reference and compound.
// Replaces svt_av1_convolve_{2d_sr,x_sr,y_sr,2d_copy_sr}_c, svt_av1_jnt_convolve_{2d,x,y,2d_copy}_c and the highbd sets
// (inter_prediction.c:311-668, 670-1035).
// One workgroup per 64 x 64 tile of one predicted block: the tile and its filter margin are staged in LDS, the
// horizontal pass writes the int16 intermediate to LDS, the vertical pass writes the prediction.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_inter.h"
#include "common.hpp"

using namespace svthip;

namespace {

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

__global__ __launch_bounds__(256) void convolve_sr_kernel(const SvtHipConvolveDesc *__restrict__ descs) {
    __shared__ uint16_t in[(TILE + 7) * IP];
    __shared__ int16_t  im[(TILE + 7) * TILE];
    const SvtHipConvolveDesc d = descs[blockIdx.x];
    if (d.w == 0 || d.h == 0)  // an unused slot of a fixed-size descriptor array (tf_picture.hip)
        return;
    const int tiles_x = (d.w + TILE - 1) / TILE;
    const int x0 = (blockIdx.y % tiles_x) * TILE, y0 = (blockIdx.y / tiles_x) * TILE;
    if (y0 >= d.h)
        return;
    const int tw = min(TILE, d.w - x0), th = min(TILE, d.h - y0);
    const int tx = d.taps_x, ty = d.taps_y, is16 = d.is_16bit, bd = d.bit_depth, r0 = d.round_0, r1 = d.round_1;
    const int fo_h = tx ? tx / 2 - 1 : 0, fo_v = ty ? ty / 2 - 1 : 0;
    const int ew = tw + (tx ? tx - 1 : 0), eh = th + (ty ? ty - 1 : 0);  // staged extent
    for (int idx = threadIdx.x; idx < eh * ew; idx += 256) {
        const int r = idx / ew, c = idx - r * ew;
        in[r * IP + c] = (uint16_t)ldpx(d.src, (ptrdiff_t)(y0 + r - fo_v) * d.src_stride + (x0 + c - fo_h), is16);
    }
    __syncthreads();
    if (d.compound) {  // jnt_convolve_{2d_copy, x, y, 2d}
        const int     offset_bits = bd + 2 * FILTER_BITS - r0, round_bits = 2 * FILTER_BITS - r0 - r1;
        const int32_t round_offset = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
        if (tx && ty) {
            for (int idx = threadIdx.x; idx < eh * tw; idx += 256) {
                const int r = idx / tw, c = idx - r * tw;
                int32_t   sum = 1 << (bd + FILTER_BITS - 1);
                for (int k = 0; k < tx; k++) sum += d.filter_x[k] * (int32_t)in[r * IP + c + k];
                im[r * TILE + c] = (int16_t)(uint16_t)rnd(sum, r0);
            }
            __syncthreads();
        }
        for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res;
            if (tx && ty) {
                int32_t sum = 1 << offset_bits;
                for (int k = 0; k < ty; k++) sum += d.filter_y[k] * (int32_t)im[(r + k) * TILE + c];
                res = (uint16_t)rnd(sum, r1);
            } else if (ty) {
                res = 0;
                for (int k = 0; k < ty; k++) res += d.filter_y[k] * (int32_t)in[(r + k) * IP + c];
                res *= 1 << (FILTER_BITS - r0);
                res = rnd(res, r1) + round_offset;
            } else if (tx) {
                res = 0;
                for (int k = 0; k < tx; k++) res += d.filter_x[k] * (int32_t)in[r * IP + c + k];
                res = (1 << (FILTER_BITS - r1)) * rnd(res, r0) + round_offset;
            } else {
                res = (uint16_t)((uint16_t)((int32_t)in[r * IP + c] << round_bits) + (uint16_t)round_offset);
            }
            comp_out(d, y0 + r, x0 + c, res, round_offset, round_bits);
        }
        return;
    }
    if (!tx && !ty) {  // 2d_copy_sr
        for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
            const int r = idx / tw, c = idx - r * tw;
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, in[r * IP + c], 16);
        }
        return;
    }
    if (!ty) {  // x_sr
        const int bits = FILTER_BITS - r0;
        for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res = 0;
            for (int k = 0; k < tx; k++) res += d.filter_x[k] * (int32_t)in[r * IP + c + k];
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(rnd(res, r0), bits), bd);
        }
        return;
    }
    if (!tx) {  // y_sr
        for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
            const int r = idx / tw, c = idx - r * tw;
            int32_t   res = 0;
            for (int k = 0; k < ty; k++) res += d.filter_y[k] * (int32_t)in[(r + k) * IP + c];
            stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(res, FILTER_BITS), bd);
        }
        return;
    }
    // 2d_sr
    for (int idx = threadIdx.x; idx < eh * tw; idx += 256) {
        const int r = idx / tw, c = idx - r * tw;
        int32_t   sum = 1 << (bd + FILTER_BITS - 1);
        for (int k = 0; k < tx; k++) sum += d.filter_x[k] * (int32_t)in[r * IP + c + k];
        im[r * TILE + c] = (int16_t)(uint16_t)rnd(sum, r0);
    }
    __syncthreads();
    const int bits = 2 * FILTER_BITS - r0 - r1, offset_bits = bd + 2 * FILTER_BITS - r0;
    for (int idx = threadIdx.x; idx < th * tw; idx += 256) {
        const int r = idx / tw, c = idx - r * tw;
        int32_t   sum = 1 << offset_bits;
        for (int k = 0; k < ty; k++) sum += d.filter_y[k] * (int32_t)im[(r + k) * TILE + c];
        int32_t res = rnd(sum, r1) - ((1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)));
        if (!is16)
            res = (int16_t)res;  // the 8-bit function narrows to int16 first (inter_prediction.c:343-345)
        stpx(d.dst, (size_t)(y0 + r) * d.dst_stride + x0 + c, is16, rnd(res, bits), bd);
    }
}

[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// Tier A: stage the block and its margins, run the batch kernel on one descriptor, copy the prediction back.
void conv_tier_a(const void *src, int32_t src_stride, void *dst, int32_t dst_stride, int32_t w, int32_t h, const SvtHipInterpFilterParams *fx,
                 const SvtHipInterpFilterParams *fy, int32_t sx, int32_t sy, const SvtHipConvolveParams *cp, int use_x, int use_y, int is16, int bd,
                 int compound = 0) {
    if (!ensure_init())
        fatal("convolve_sr");
    if (w < 1 || h < 1 || w > 128 || h > 128 || (use_x && (!fx || fx->taps > 8 || (fx->taps & 1))) || (use_y && (!fy || fy->taps > 8 || (fy->taps & 1)))) {
        set_error("convolve_sr: unsupported block %dx%d / filter", w, h);
        fatal("convolve_sr");
    }
    const int    px = is16 ? 2 : 1, tx = use_x ? fx->taps : 0, ty = use_y ? fy->taps : 0;
    const int    fo_h = tx ? tx / 2 - 1 : 0, fo_v = ty ? ty / 2 - 1 : 0, ew = w + (tx ? tx - 1 : 0), eh = h + (ty ? ty - 1 : 0);
    const size_t ibytes = up256((size_t)ew * eh * px), obytes = up256((size_t)w * h * px), cbytes = up256((size_t)w * h * 2);
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    uint8_t     *d = sc.device(ibytes + obytes + 512 + cbytes), *hh = sc.host(ibytes + obytes + 512 + cbytes);
    for (int r = 0; r < eh; r++)
        memcpy(hh + (size_t)r * ew * px, (const uint8_t *)src + ((ptrdiff_t)(r - fo_v) * src_stride - fo_h) * px, (size_t)ew * px);
    SvtHipConvolveDesc ds{};
    ds.src = d + ((size_t)fo_v * ew + fo_h) * px, ds.dst = d + ibytes, ds.src_stride = (uint32_t)ew, ds.dst_stride = (uint32_t)w;
    ds.w = (uint16_t)w, ds.h = (uint16_t)h, ds.taps_x = (uint8_t)tx, ds.taps_y = (uint8_t)ty;
    if (tx)
        memcpy(ds.filter_x, fx->filter_ptr + (size_t)fx->taps * (sx & 15), sizeof(int16_t) * tx);  // av1_get_interp_filter_subpel_kernel
    if (ty)
        memcpy(ds.filter_y, fy->filter_ptr + (size_t)fy->taps * (sy & 15), sizeof(int16_t) * ty);
    ds.round_0 = (uint8_t)cp->round_0, ds.round_1 = (uint8_t)cp->round_1, ds.bit_depth = (uint8_t)bd, ds.is_16bit = (uint8_t)is16;
    // compound: the ConvBufType block sits behind the descriptor in the same staging buffers
    uint16_t    *cb_host = compound ? (uint16_t *)cp->dst : nullptr;
    const size_t cboff = ibytes + obytes + 256;
    if (compound) {
        if (!cb_host) {
            set_error("jnt_convolve: conv_params->dst is NULL");
            fatal("jnt_convolve");
        }
        ds.compound    = cp->do_average ? (cp->use_jnt_comp_avg ? 3 : 2) : 1;
        ds.fwd_offset  = (uint8_t)cp->fwd_offset, ds.bck_offset = (uint8_t)cp->bck_offset;
        ds.cbuf        = (uint16_t *)(d + cboff), ds.cbuf_stride = (uint32_t)w;
        if (cp->do_average)
            for (int r = 0; r < h; r++) memcpy(hh + cboff + (size_t)r * w * 2, cb_host + (size_t)r * cp->dst_stride, (size_t)w * 2);
    }
    memcpy(hh + ibytes + obytes, &ds, sizeof(ds));
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, hh, ibytes, hipMemcpyHostToDevice, st));
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d + ibytes + obytes, hh + ibytes + obytes, 256 + (compound && cp->do_average ? cbytes : 0),
                                       hipMemcpyHostToDevice, st));
    if (svt_hip_convolve_batch((const SvtHipConvolveDesc *)(d + ibytes + obytes), 1, st) != SVT_HIP_OK)
        fatal("convolve");
    if (compound && !cp->do_average) {
        SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + cboff, d + cboff, (size_t)w * h * 2, hipMemcpyDeviceToHost, st));
        SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
        for (int r = 0; r < h; r++) memcpy(cb_host + (size_t)r * cp->dst_stride, hh + cboff + (size_t)r * w * 2, (size_t)w * 2);
        return;
    }
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + ibytes, d + ibytes, obytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    for (int r = 0; r < h; r++) memcpy((uint8_t *)dst + (size_t)r * dst_stride * px, hh + ibytes + (size_t)r * w * px, (size_t)w * px);
}

}  // namespace

extern "C" int32_t svt_hip_convolve_sr_batch(const SvtHipConvolveDesc *d_desc, uint32_t n, void *stream) {
    return svt_hip_convolve_batch(d_desc, n, stream);
}
extern "C" int32_t svt_hip_convolve_batch(const SvtHipConvolveDesc *d_desc, uint32_t n, void *stream) {
    if (!d_desc || n == 0) {
        set_error("svt_hip_convolve_batch: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipLaunchKernelGGL(convolve_sr_kernel, dim3(n, 4), dim3(256), 0, resolve_stream(stream), d_desc);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

#define SVT_HIP_DEF_CONV(mode, UX, UY)                                                                                          \
    extern "C" void svt_av1_convolve_##mode##_hip(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride,      \
                                                  int32_t w, int32_t h, SvtHipInterpFilterParams *fx, SvtHipInterpFilterParams *fy, \
                                                  const int32_t sx, const int32_t sy, SvtHipConvolveParams *cp) {               \
        TIER_A_CALL(svt_av1_convolve_##mode, conv_tier_a(src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, UX, UY, 0, 8), (src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp)); \
    }                                                                                                                            \
    extern "C" void svt_av1_highbd_convolve_##mode##_hip(const uint16_t *src, int32_t src_stride, uint16_t *dst, int32_t dst_stride, \
                                                         int32_t w, int32_t h, const SvtHipInterpFilterParams *fx,              \
                                                         const SvtHipInterpFilterParams *fy, const int32_t sx, const int32_t sy, \
                                                         SvtHipConvolveParams *cp, int32_t bd) {                                \
        TIER_A_CALL(svt_av1_highbd_convolve_##mode, conv_tier_a(src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, UX, UY, 1, bd), (src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, bd)); \
    }
SVT_HIP_DEF_CONV(2d_sr, 1, 1)
SVT_HIP_DEF_CONV(x_sr, 1, 0)
SVT_HIP_DEF_CONV(y_sr, 0, 1)
SVT_HIP_DEF_CONV(2d_copy_sr, 0, 0)


#define SVT_HIP_DEF_JNT(mode, UX, UY)                                                                                           \
    extern "C" void svt_av1_jnt_convolve_##mode##_hip(const uint8_t *src, int32_t src_stride, uint8_t *dst, int32_t dst_stride,  \
                                                      int32_t w, int32_t h, SvtHipInterpFilterParams *fx, SvtHipInterpFilterParams *fy, \
                                                      const int32_t sx, const int32_t sy, SvtHipConvolveParams *cp) {           \
        TIER_A_CALL(svt_av1_jnt_convolve_##mode, conv_tier_a(src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, UX, UY, 0, 8, 1), (src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp)); \
    }                                                                                                                            \
    extern "C" void svt_av1_highbd_jnt_convolve_##mode##_hip(const uint16_t *src, int32_t src_stride, uint16_t *dst,            \
                                                             int32_t dst_stride, int32_t w, int32_t h,                          \
                                                             const SvtHipInterpFilterParams *fx, const SvtHipInterpFilterParams *fy, \
                                                             const int32_t sx, const int32_t sy, SvtHipConvolveParams *cp, int32_t bd) { \
        TIER_A_CALL(svt_av1_highbd_jnt_convolve_##mode, conv_tier_a(src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, UX, UY, 1, bd, 1), (src, src_stride, dst, dst_stride, w, h, fx, fy, sx, sy, cp, bd)); \
    }
SVT_HIP_DEF_JNT(2d, 1, 1)
SVT_HIP_DEF_JNT(x, 1, 0)
SVT_HIP_DEF_JNT(y, 0, 1)
SVT_HIP_DEF_JNT(2d_copy, 0, 0)
