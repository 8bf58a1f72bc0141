// loopfilter_lr_frame.hip — frame-level loop restoration on gfx950: svt_av1_loop_restoration_filter_frame
// (restoration.c:1179-1248) for up to three planes in ONE launch.
//
// One workgroup = one processing unit ((64 >> ss_x) columns) of one 64-row processing stripe of one plane.  The reference
// patches the three rows above and below a stripe into the picture before it filters the stripe and patches them back
// afterwards (svt_aom_setup_/restore_processing_stripe_boundary, :288-435); here the same rows are CHOSEN while the tile is
// assembled in LDS — first / last stripe and the picture's sides: edge replication (svt_extend_frame, :197-203); other
// stripes: the two saved deblocked rows of stripe_boundary_above / _below used 0,0,1 / 0,1,1, or for optimized_lr the
// picture's own rows with the outermost duplicated — so the source plane is never written and all stripes run concurrently.
// The stripe filters are the shared tile filters of lr_device.hpp (self-guided: fused projection; Wiener: separable 7 tap),
// selected per restoration unit; RESTORE_NONE units are copied.
#include "../../include/svt_hip_lf.h"
#include "common.hpp"
#include "lr_device.hpp"

using namespace svthip;
using namespace svthip::lr;

namespace {

constexpr int LR_NT = 512;
struct LrFrame {
    SvtHipLrPlane pl[3];
    uint32_t      n_planes;
};

__device__ __forceinline__ int32_t stripe_px(const SvtHipLrPlane &pl, int x, int y, int ys, int h, int stripe, bool first, bool last) {
    const int W = (int)pl.width, H = (int)pl.height, is16 = pl.is_16bit;
    const int xc = x < 0 ? 0 : (x >= W ? W - 1 : x);
    if (y < ys && !first) {  // copy_above (svt_aom_get_stripe_boundary_info, :257-276)
        const int i = y - ys;  // -3 .. -1
        if (!pl.optimized_lr)
            return ldpx(pl.boundary_above, (size_t)(2 * stripe + (i + 2 > 0 ? i + 2 : 0)) * pl.boundary_stride + (x + SVT_HIP_LR_EXTRA_HORZ), is16);
        const int ya = i == -3 ? ys - 2 : y;
        return ldpx(pl.src, (size_t)((ptrdiff_t)(ya > 0 ? ya : 0) * pl.src_stride + xc), is16);
    }
    if (y >= ys + h && !last) {  // copy_below
        const int i = y - (ys + h);  // 0 .. 2
        if (!pl.optimized_lr)
            return ldpx(pl.boundary_below, (size_t)(2 * stripe + (i < 1 ? i : 1)) * pl.boundary_stride + (x + SVT_HIP_LR_EXTRA_HORZ), is16);
        // the picture's own rows: below the last picture row they are its replica (svt_extend_frame), e.g. H = 64 k + 57
        const int yb = i == 2 ? ys + h + 1 : y;
        return ldpx(pl.src, (size_t)((ptrdiff_t)(yb < H ? yb : H - 1) * pl.src_stride + xc), is16);
    }
    const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return ldpx(pl.src, (size_t)((ptrdiff_t)yc * pl.src_stride + xc), is16);
}

__global__ __launch_bounds__(LR_NT) void lr_frame_kernel(LrFrame f) {
    // one LDS area, carved per filter: self-guided = tile (70 x TP u16) + A / B maps; Wiener = tile (71 x WIENER_IP u16) + tmp
    __shared__ __attribute__((aligned(16))) uint8_t smem[70 * TP * 2 + 2 * 66 * AP * 4];
    static_assert(sizeof(smem) >= (64 + 7) * WIENER_IP * 2 + (64 + 7) * 64 * 2, "Wiener tiles fit the self-guided area");
    const SvtHipLrPlane &pl = f.pl[blockIdx.z];
    const int W = (int)pl.width, H = (int)pl.height, is16 = pl.is_16bit, bd = pl.bit_depth, tid = threadIdx.x;
    const int full = 64 >> pl.ss_y, off = 8 >> pl.ss_y, pw = 64 >> pl.ss_x, us = (int)pl.unit_size;
    const int stripe = blockIdx.y, x0 = blockIdx.x * pw;
    const int ys = stripe == 0 ? 0 : stripe * full - off;
    if (x0 >= W || ys >= H)
        return;
    const bool first = stripe == 0;
    const int  nominal = full - (first ? off : 0);
    const int  h = min(nominal, H - ys), w = min(pw, W - x0);
    const bool last = ys + nominal >= H;
    int        ur = (ys + off) / us, uc = x0 / us;
    ur = min(ur, (int)pl.vert_units - 1), uc = min(uc, (int)pl.horz_units - 1);
    const SvtHipLrUnit u = pl.units[ur * pl.horz_units + uc];
    if (u.restoration_type == 0) {  // RESTORE_NONE: svt_aom_copy_tile
        for (int idx = tid; idx < w * h; idx += LR_NT) {
            const int    r = idx / w, c = idx - r * w;
            const size_t o = (size_t)(ys + r) * pl.dst_stride + x0 + c;
            const int32_t v = ldpx(pl.src, (size_t)(ys + r) * pl.src_stride + x0 + c, is16);
            if (is16)
                ((uint16_t *)pl.dst)[o] = (uint16_t)v;
            else
                ((uint8_t *)pl.dst)[o] = (uint8_t)v;
        }
        return;
    }
    if (u.restoration_type == 2) {  // RESTORE_SGRPROJ: svt_aom_sgrproj_filter_stripe(_highbd)
        uint16_t *tile = (uint16_t *)smem;
        int32_t  *Am = (int32_t *)(smem + 70 * TP * 2), *Bm = Am + 66 * AP;
        for (int idx = tid; idx < (h + 6) * (w + 6); idx += LR_NT) {
            const int r = idx / (w + 6), c = idx - r * (w + 6);
            tile[r * TP + c] = (uint16_t)stripe_px(pl, x0 + c - 3, ys + r - 3, ys, h, stripe, first, last);
        }
        __syncthreads();
        const int ep = u.ep, r0 = SGR_PRM[ep][0], r1 = SGR_PRM[ep][1];
        int       xq0, xq1;  // svt_decode_xq (restoration.c:634-645)
        if (r0 == 0)
            xq0 = 0, xq1 = (1 << PRJ_BITS) - u.xqd[1];
        else if (r1 == 0)
            xq0 = u.xqd[0], xq1 = 0;
        else
            xq0 = u.xqd[0], xq1 = (1 << PRJ_BITS) - xq0 - u.xqd[1];
        sgr_tile_filter<1, LR_NT>(tile, Am, Bm, tid, w, h, ys, x0, ep, bd, is16, nullptr, nullptr, 0u, pl.dst, pl.dst_stride, xq0, xq1);
        return;
    }
    // RESTORE_WIENER: svt_aom_wiener_filter_stripe(_highbd); get_conv_params_wiener (restoration.c:49-73)
    uint16_t *in = (uint16_t *)smem, *tmp = in + (64 + 7) * WIENER_IP;
    for (int idx = tid; idx < (h + 7) * (w + 7); idx += LR_NT) {
        const int r = idx / (w + 7), c = idx - r * (w + 7);
        in[r * WIENER_IP + c] = (uint16_t)stripe_px(pl, x0 + c - 3, ys + r - 3, ys, h, stripe, first, last);
    }
    __syncthreads();
    int r0 = 3, r1 = 11;
    if (bd + 7 - r0 + 2 > 16) {
        const int d = bd + 7 - r0 + 2 - 16;
        r0 += d, r1 -= d;
    }
    wiener_tile_filter<LR_NT>(in, tmp, tid, w, h, x0, ys, u.hfilter, u.vfilter, bd, r0, r1, is16, pl.dst, pl.dst_stride);
}

}  // namespace

extern "C" int32_t svt_hip_restoration_filter_frame(const SvtHipLrPlane *planes, uint32_t n_planes, void *stream) {
    if (!planes || n_planes == 0 || n_planes > 3) {
        set_error("svt_hip_restoration_filter_frame: 1..3 planes");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    LrFrame  f;
    uint32_t gx = 0, gy = 0;
    f.n_planes = n_planes;
    for (uint32_t p = 0; p < n_planes; p++) {
        const SvtHipLrPlane &pl = planes[p];
        const bool           bd_ok = pl.bit_depth == 8 || pl.bit_depth == 10 || pl.bit_depth == 12;
        if (!pl.src || !pl.dst || pl.src == pl.dst || !pl.units || !pl.width || !pl.height || pl.src_stride < pl.width || pl.dst_stride < pl.width ||
            pl.ss_x > 1 || pl.ss_y > 1 || !bd_ok || (pl.bit_depth > 8 && !pl.is_16bit) || pl.unit_size < 32 || pl.unit_size > 256 ||
            (pl.unit_size & (pl.unit_size - 1)) || !pl.horz_units || !pl.vert_units ||
            (!pl.optimized_lr && (!pl.boundary_above || !pl.boundary_below || pl.boundary_stride < pl.width + 2 * SVT_HIP_LR_EXTRA_HORZ))) {
            set_error("svt_hip_restoration_filter_frame: plane %u: bad argument", p);
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
        f.pl[p] = pl;
        const uint32_t pw = 64u >> pl.ss_x, full = 64u >> pl.ss_y, off = 8u >> pl.ss_y;
        const uint32_t nx = (pl.width + pw - 1) / pw, ny = (pl.height + off + full - 1) / full;
        gx = nx > gx ? nx : gx, gy = ny > gy ? ny : gy;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipLaunchKernelGGL(lr_frame_kernel, dim3(gx, gy, n_planes), dim3(LR_NT), 0, resolve_stream(stream), f);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

SVT_HIP_MODULE_WARMUP(loopfilter_lr_frame)
