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
#include "convolve_device.hpp"

using namespace svthip;
using namespace svthip::conv;

namespace {

__global__ __launch_bounds__(256) void convolve_sr_kernel(const SvtHipConvolveDesc *__restrict__ descs) {
    __shared__ alignas(4) uint16_t in[(TILE + 7) * IP + CONV_IN_SLACK];
    __shared__ int16_t  im[(TILE + 7) * TILE];
    const SvtHipConvolveDesc d = descs[blockIdx.x];
    convolve_tile(d, (int)blockIdx.y, in, im);
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

SVT_HIP_MODULE_WARMUP(inter_convolve)
