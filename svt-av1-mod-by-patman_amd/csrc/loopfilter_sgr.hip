// loopfilter_sgr.hip — AV1 self-guided restoration on gfx950.
// Replaces svt_av1_selfguided_restoration_c and its two internal filters (restoration.c:468-955),
// svt_apply_selfguided_restoration_c (:957-992), svt_av1_{lowbd,highbd}_pixel_proj_error_c and
// svt_get_proj_subspace_c (restoration_pick.c:167-303, 417-506) and, as one device-side pipeline, apply_sgr +
// search_selfguided_restoration + finer_search_pixel_proj_error (restoration_pick.c:320-411, 523-652).
//
// Filter: one workgroup per 64x64 (or 32x32) processing unit.  The unit + its 3-sample border is staged once in LDS
// as uint16; the (2r+1)^2 box sums are taken straight from that tile for every position of the (w+2) x (h+2) A/B maps
// (odd rows only for the r = 2 "fast" filter), the maps stay in LDS, and every output sample reads its 6 or 9
// neighbours from there.  Nothing but the final flt0 / flt1 (or the projected samples, in the fused apply) goes back
// to memory.  Search: one workgroup per epsilon candidate owns the whole decision chain of that candidate — the five
// second-moment sums (exact in int64; the reference's double sums are exact too), the 2x2 solve in IEEE double
// (-ffp-contract=off), encode_xq and the data-dependent finer search, each error evaluation being a block reduction
// over the unit — so the host sees one readback per restoration unit.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_lf.h"
#include "common.hpp"
#include "lr_device.hpp"

using namespace svthip;
using namespace svthip::lr;

namespace {

static const int32_t SGR_PRM_H[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1},
                                         {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}};

struct SgrGeom {
    const void *dat;
    uint32_t    dat_stride, width, height;
    uint8_t     is16, bit_depth, pu_w, pu_h;
};

// MODE 0: write flt0 / flt1.  MODE 1: fused svt_apply_selfguided_restoration (projection with xq, clip, store samples).
// SGR_NT threads share one 64x64 unit's 45 KB of LDS: three workgroups fit a CU, so the thread count sets the number of
// waves that can cover each other's barrier phases
constexpr int SGR_NT = 512;
template <int MODE>
__global__ __launch_bounds__(SGR_NT) void sgr_filter_kernel(SgrGeom g, int ep, int32_t *__restrict__ flt0, int32_t *__restrict__ flt1,
                                                         uint32_t flt_stride, void *__restrict__ dst, uint32_t dst_stride, int xq0, int xq1) {
    __shared__ alignas(4) uint16_t tile[70 * TP];
    __shared__ int32_t  Am[66 * AP], Bm[66 * AP];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * g.pu_w, i0 = blockIdx.y * g.pu_h;
    const int w = min((int)g.pu_w, (int)g.width - j0), h = min((int)g.pu_h, (int)g.height - i0);
    for (int idx = tid; idx < (h + 6) * (w + 6); idx += SGR_NT) {
        const int r = idx / (w + 6), c = idx - r * (w + 6);
        tile[r * TP + c] = (uint16_t)ldpx(g.dat, (size_t)((ptrdiff_t)(i0 + r - 3) * g.dat_stride + (j0 + c - 3)), g.is16);
    }
    __syncthreads();
    sgr_tile_filter<MODE, SGR_NT>(tile, Am, Bm, tid, w, h, i0, j0, ep, g.bit_depth, g.is16, flt0, flt1, flt_stride, dst, dst_stride, xq0, xq1);
}

// ---- block-wide int64 reductions -------------------------------------------------------------------------------
__device__ __forceinline__ long long wave_sum64(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// sum over the block; result valid in every thread.  scratch: one long long per wave (+1)
__device__ long long block_sum64(long long v, long long *scratch) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum64(v);
    __syncthreads();  // scratch free
    if (lane == 0)
        scratch[wv] = v;
    __syncthreads();
    long long t = 0;
    for (int i = 0; i < nw; i++) t += scratch[i];
    return t;
}

struct UnitData {
    const void    *src, *dat;
    const int32_t *flt0, *flt1;
    uint32_t       src_stride, dat_stride, flt0_stride, flt1_stride, width, height;
    int            is16, r0, r1;
};

// svt_av1_{lowbd,highbd}_pixel_proj_error_c over the unit (block-wide)
__device__ long long unit_error(const UnitData &u, int xq0, int xq1, long long *scratch) {
    long long      acc = 0;
    const uint32_t n   = u.width * u.height;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint32_t i = idx / u.width, j = idx - i * u.width;
        const int32_t  d = ldpx(u.dat, (size_t)i * u.dat_stride + j, u.is16), s = ldpx(u.src, (size_t)i * u.src_stride + j, u.is16);
        const int32_t  uu = d << RST_BITS;
        int32_t        v  = 1 << (RST_BITS + PRJ_BITS - 1);
        if (u.r0 > 0)
            v += xq0 * (u.flt0[(size_t)i * u.flt0_stride + j] - uu);
        if (u.r1 > 0)
            v += xq1 * (u.flt1[(size_t)i * u.flt1_stride + j] - uu);
        const int32_t e = (u.r0 > 0 || u.r1 > 0) ? (v >> (RST_BITS + PRJ_BITS)) + d - s : d - s;
        acc += (long long)e * e;
    }
    return block_sum64(acc, scratch);
}

// the five sums of svt_get_proj_subspace_c (block-wide)
__device__ void unit_sums(const UnitData &u, long long out[5], long long *scratch) {
    long long      a[5] = {0, 0, 0, 0, 0};
    const uint32_t n    = u.width * u.height;
    for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint32_t  i = idx / u.width, j = idx - i * u.width;
        const long long uu = (long long)ldpx(u.dat, (size_t)i * u.dat_stride + j, u.is16) << RST_BITS;
        const long long s  = ((long long)ldpx(u.src, (size_t)i * u.src_stride + j, u.is16) << RST_BITS) - uu;
        const long long f1 = u.r0 > 0 ? u.flt0[(size_t)i * u.flt0_stride + j] - uu : 0;
        const long long f2 = u.r1 > 0 ? u.flt1[(size_t)i * u.flt1_stride + j] - uu : 0;
        a[0] += f1 * f1, a[1] += f2 * f2, a[2] += f1 * f2, a[3] += f1 * s, a[4] += f2 * s;
    }
    for (int k = 0; k < 5; k++) out[k] = block_sum64(a[k], scratch);
}

// closed-form part of svt_get_proj_subspace_c (restoration_pick.c:471-506), IEEE double without contraction
__device__ void solve_subspace(const long long sums[5], int size, int r0, int r1, int xq[2]) {
    double H00 = (double)sums[0], H11 = (double)sums[1], H01 = (double)sums[2], C0 = (double)sums[3], C1 = (double)sums[4];
    xq[0] = xq[1] = 0;
    H00 /= size, H01 /= size, H11 /= size, C0 /= size, C1 /= size;
    const double H10 = H01;
    if (r0 == 0) {
        if (H11 < 1e-8)
            return;
        xq[1] = (int)rint(C1 / H11 * (1 << PRJ_BITS));
    } else if (r1 == 0) {
        if (H00 < 1e-8)
            return;
        xq[0] = (int)rint(C0 / H00 * (1 << PRJ_BITS));
    } else {
        const double det = H00 * H11 - H01 * H10;
        if (det < 1e-8)
            return;
        const double x0 = (H11 * C0 - H01 * C1) / det, x1 = (H00 * C1 - H10 * C0) / det;
        xq[0] = (int)rint(x0 * (1 << PRJ_BITS)), xq[1] = (int)rint(x1 * (1 << PRJ_BITS));
    }
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ void decode_xq(const int xqd[2], int xq[2], int r0, int r1) {
    if (r0 == 0)
        xq[0] = 0, xq[1] = (1 << PRJ_BITS) - xqd[1];
    else if (r1 == 0)
        xq[0] = xqd[0], xq[1] = 0;
    else
        xq[0] = xqd[0], xq[1] = (1 << PRJ_BITS) - xq[0] - xqd[1];
}
__device__ void encode_xq(const int xq[2], int xqd[2], int r0, int r1) {
    if (r0 == 0) {
        xqd[0] = 0;
        xqd[1] = clampi((1 << PRJ_BITS) - xq[1], PRJ_MIN1, PRJ_MAX1);
    } else if (r1 == 0) {
        xqd[0] = clampi(xq[0], PRJ_MIN0, PRJ_MAX0);
        xqd[1] = clampi((1 << PRJ_BITS) - xqd[0], PRJ_MIN1, PRJ_MAX1);
    } else {
        xqd[0] = clampi(xq[0], PRJ_MIN0, PRJ_MAX0);
        xqd[1] = clampi((1 << PRJ_BITS) - xqd[0] - xq[1], PRJ_MIN1, PRJ_MAX1);
    }
}
__device__ long long err_of(const UnitData &u, const int xqd[2], long long *scratch) {
    int xq[2];
    decode_xq(xqd, xq, u.r0, u.r1);
    return unit_error(u, xq[0], xq[1], scratch);
}

struct EpResult {
    long long err;
    int32_t   ep, xqd[2], pad;
};

// One workgroup per epsilon candidate: get_proj_subspace + encode_xq + finer_search_pixel_proj_error.  All threads run
// the (uniform) control flow; every err_of() is a block reduction whose result all threads see.
__global__ __launch_bounds__(1024) void sgr_search_ep_kernel(SvtHipSgrUnit un, const int32_t *__restrict__ work, uint32_t flt_stride,
                                                             int start_ep, int ep_inc, int do_refine, EpResult *__restrict__ res) {
    __shared__ long long scratch[17];
    const int      ep = start_ep + blockIdx.x * ep_inc;
    const size_t   plane = (size_t)flt_stride * un.height;
    const int32_t *f0 = work + (size_t)blockIdx.x * 2 * plane, *f1 = f0 + plane;
    UnitData       u{un.src, un.dat, f0, f1, un.src_stride, un.dat_stride, flt_stride, flt_stride, un.width, un.height, un.is_16bit,
               SGR_PRM[ep][0], SGR_PRM[ep][1]};
    long long sums[5];
    unit_sums(u, sums, scratch);
    int exq[2], xqd[2];
    solve_subspace(sums, (int)(un.width * un.height), u.r0, u.r1, exq);
    encode_xq(exq, xqd, u.r0, u.r1);
    long long err = err_of(u, xqd, scratch), err2;
    if (do_refine) {
        const int tap_min[2] = {PRJ_MIN0, PRJ_MIN1}, tap_max[2] = {PRJ_MAX0, PRJ_MAX1};
        const int start_step = 2;
        for (int s = start_step; s >= 1; s >>= 1)
            for (int p = 0; p < 2; ++p) {
                if ((u.r0 == 0 && p == 0) || (u.r1 == 0 && p == 1))
                    continue;
                int skip = 0;
                do {
                    if (xqd[p] - s >= tap_min[p]) {
                        xqd[p] -= s;
                        err2 = err_of(u, xqd, scratch);
                        if (err2 > err)
                            xqd[p] += s;
                        else {
                            err = err2, skip = 1;
                            if (s == start_step)
                                continue;
                        }
                    }
                    break;
                } while (1);
                if (skip)
                    break;
                do {
                    if (xqd[p] + s <= tap_max[p]) {
                        xqd[p] += s;
                        err2 = err_of(u, xqd, scratch);
                        if (err2 > err)
                            xqd[p] -= s;
                        else {
                            err = err2;
                            if (s == start_step)
                                continue;
                        }
                    }
                    break;
                } while (1);
            }
    }
    if (threadIdx.x == 0)
        res[blockIdx.x] = EpResult{err, ep, {xqd[0], xqd[1]}, 0};
}
// first strict minimum in candidate order (restoration_pick.c:638-643)
__global__ void sgr_pick_kernel(const EpResult *res, int n, EpResult *best) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        EpResult b = res[0];
        for (int i = 1; i < n; i++)
            if (res[i].err < b.err)
                b = res[i];
        *best = b;
    }
}

// Tier A reductions over host-staged data: out[0..4] = subspace sums, out[5] = projection error for (xq0, xq1)
__global__ __launch_bounds__(1024) void sgr_stats_kernel(UnitData u, int xq0, int xq1, int want_sums, long long *out) {
    __shared__ long long scratch[17];
    if (want_sums) {
        long long s[5];
        unit_sums(u, s, scratch);
        if (threadIdx.x == 0)
            for (int k = 0; k < 5; k++) out[k] = s[k];
    } else {
        const long long e = unit_error(u, xq0, xq1, scratch);
        if (threadIdx.x == 0)
            out[5] = e;
    }
}
__global__ void sgr_solve_kernel(const long long *sums, int size, int r0, int r1, int32_t *xq) {
    if (threadIdx.x == 0) {
        int x[2];
        solve_subspace(sums, size, r0, r1, x);
        xq[0] = x[0], xq[1] = x[1];
    }
}

// The search works on one restoration unit (<= 384 x 384, restoration.h:80-84); filter / apply accept any region whose
// origin lies on the processing-unit grid, e.g. a whole plane in one launch.
bool unit_ok(const SvtHipSgrUnit *u, bool need_src) {
    const uint32_t lim = need_src ? 384u : 16384u;
    return u && u->dat && (!need_src || u->src) && u->width && u->height && u->width <= lim && u->height <= lim &&
        (u->pu_w == 64 || u->pu_w == 32) && (u->pu_h == 64 || u->pu_h == 32) && (u->bit_depth == 8 || u->bit_depth == 10 || u->bit_depth == 12) &&
        (u->bit_depth == 8 || u->is_16bit);
}
SgrGeom geom_of(const SvtHipSgrUnit *u) {
    return SgrGeom{u->dat, u->dat_stride, u->width, u->height, u->is_16bit, u->bit_depth, u->pu_w, u->pu_h};
}
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
template <typename T> const T *decode_ptr(const uint8_t *p, int highbd) {
    return highbd ? (const T *)((uintptr_t)p << 1) : (const T *)p;  // CONVERT_TO_SHORTPTR (definitions.h:953)
}

}  // namespace

// ------------------------------------------------------------------------------------------------ Tier B
extern "C" int32_t svt_hip_sgr_filter_unit(const SvtHipSgrUnit *unit, int32_t ep, int32_t *d_flt0, int32_t *d_flt1, uint32_t flt_stride,
                                           void *stream) {
    if (!unit_ok(unit, false) || ep < 0 || ep > 15 || !d_flt0 || !d_flt1 || flt_stride < unit->width) {
        set_error("svt_hip_sgr_filter_unit: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const dim3 grid((unit->width + unit->pu_w - 1) / unit->pu_w, (unit->height + unit->pu_h - 1) / unit->pu_h);
    hipLaunchKernelGGL(sgr_filter_kernel<0>, grid, dim3(SGR_NT), 0, resolve_stream(stream), geom_of(unit), ep, d_flt0, d_flt1, flt_stride,
                       (void *)nullptr, 0u, 0, 0);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_sgr_apply_unit(const SvtHipSgrUnit *unit, int32_t ep, const int32_t xqd[2], void *d_dst, uint32_t dst_stride,
                                          void *stream) {
    if (!unit_ok(unit, false) || ep < 0 || ep > 15 || !xqd || !d_dst || dst_stride < unit->width || d_dst == unit->dat) {
        set_error("svt_hip_sgr_apply_unit: bad argument (output must not alias the input)");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    int xq[2];  // svt_decode_xq (restoration.c:634-645)
    if (SGR_PRM_H[ep][0] == 0)
        xq[0] = 0, xq[1] = (1 << PRJ_BITS) - xqd[1];
    else if (SGR_PRM_H[ep][1] == 0)
        xq[0] = xqd[0], xq[1] = 0;
    else
        xq[0] = xqd[0], xq[1] = (1 << PRJ_BITS) - xq[0] - xqd[1];
    const dim3 grid((unit->width + unit->pu_w - 1) / unit->pu_w, (unit->height + unit->pu_h - 1) / unit->pu_h);
    hipLaunchKernelGGL(sgr_filter_kernel<1>, grid, dim3(SGR_NT), 0, resolve_stream(stream), geom_of(unit), ep, (int32_t *)nullptr,
                       (int32_t *)nullptr, 0u, d_dst, dst_stride, xq[0], xq[1]);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

static uint32_t search_flt_stride(uint32_t width) { return ((width + 7) & ~7u) + 8; }  // restoration_pick.c:561
extern "C" size_t svt_hip_sgr_search_work_bytes(uint32_t width, uint32_t height, int32_t n_ep) {
    if (n_ep < 1)
        n_ep = 1;
    return (size_t)n_ep * 2 * search_flt_stride(width) * height * sizeof(int32_t) + (size_t)(n_ep + 1) * sizeof(EpResult) + 256;
}

extern "C" int32_t svt_hip_sgr_search_unit(const SvtHipSgrUnit *unit, int32_t start_ep, int32_t end_ep, int32_t ep_inc, int32_t do_refine,
                                           void *d_work, int32_t out[3], int64_t *best_err, void *stream) {
    if (!unit_ok(unit, true) || start_ep < 0 || end_ep > 16 || ep_inc < 1 || start_ep >= end_ep || !d_work || !out) {
        set_error("svt_hip_sgr_search_unit: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t    st    = resolve_stream(stream);
    const int      n_ep  = (end_ep - start_ep + ep_inc - 1) / ep_inc;
    const uint32_t fs    = search_flt_stride(unit->width);
    const size_t   plane = (size_t)fs * unit->height;
    int32_t       *work  = (int32_t *)d_work;
    EpResult      *res   = (EpResult *)((uint8_t *)d_work + up256((size_t)n_ep * 2 * plane * sizeof(int32_t)));
    const dim3     grid((unit->width + unit->pu_w - 1) / unit->pu_w, (unit->height + unit->pu_h - 1) / unit->pu_h);
    for (int k = 0; k < n_ep; k++)
        hipLaunchKernelGGL(sgr_filter_kernel<0>, grid, dim3(SGR_NT), 0, st, geom_of(unit), start_ep + k * ep_inc, work + (size_t)k * 2 * plane,
                           work + (size_t)k * 2 * plane + plane, fs, (void *)nullptr, 0u, 0, 0);
    hipLaunchKernelGGL(sgr_search_ep_kernel, dim3(n_ep), dim3(1024), 0, st, *unit, (const int32_t *)work, fs, start_ep, ep_inc, do_refine,
                       res);
    hipLaunchKernelGGL(sgr_pick_kernel, dim3(1), dim3(64), 0, st, (const EpResult *)res, n_ep, res + n_ep);
    SVT_HIP_CHECK(hipGetLastError());
    EpResult best;
    SVT_HIP_CHECK(hipMemcpyAsync(&best, res + n_ep, sizeof(best), hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK(hipStreamSynchronize(st));
    out[0] = best.ep, out[1] = best.xqd[0], out[2] = best.xqd[1];
    if (best_err)
        *best_err = best.err;
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------ Tier A
// Stage a w x h region (+border) of host samples as a packed device plane; returns the device pointer of sample (0,0).
template <typename T> static T *stage_region(uint8_t *d, uint8_t *h, size_t &off, const T *p, int w, int hh, int stride, int border, size_t &pitch_out) {
    const size_t pitch = (size_t)w + 2 * border, rows = (size_t)hh + 2 * border;
    T           *hp    = (T *)(h + off);
    for (size_t r = 0; r < rows; r++) memcpy(hp + r * pitch, p + ((ptrdiff_t)r - border) * stride - border, pitch * sizeof(T));
    T *dp = (T *)(d + off) + (size_t)border * pitch + border;
    off += up256(pitch * rows * sizeof(T));
    pitch_out = pitch;
    return dp;
}

static void svt_av1_selfguided_restoration_hip_impl(const uint8_t *dgd8, int32_t width, int32_t height, int32_t dgd_stride, int32_t *flt0, int32_t *flt1, int32_t flt_stride, int32_t ep, int32_t bit_depth, int32_t highbd);
extern "C" void svt_av1_selfguided_restoration_hip(const uint8_t *dgd8, int32_t width, int32_t height, int32_t dgd_stride, int32_t *flt0, int32_t *flt1, int32_t flt_stride, int32_t ep, int32_t bit_depth, int32_t highbd) { TIER_A_CALL(svt_av1_selfguided_restoration, svt_av1_selfguided_restoration_hip_impl(dgd8, width, height, dgd_stride, flt0, flt1, flt_stride, ep, bit_depth, highbd), (dgd8, width, height, dgd_stride, flt0, flt1, flt_stride, ep, bit_depth, highbd)); }
static void svt_av1_selfguided_restoration_hip_impl(const uint8_t *dgd8, int32_t width, int32_t height, int32_t dgd_stride, int32_t *flt0, int32_t *flt1, int32_t flt_stride, int32_t ep, int32_t bit_depth, int32_t highbd) {
    if (!ensure_init())
        fatal("selfguided_restoration");
    if (width <= 0 || height <= 0 || width > 384 || height > 384 || ep < 0 || ep > 15) {
        set_error("svt_av1_selfguided_restoration_hip: unsupported size %dx%d / ep %d", width, height, ep);
        fatal("selfguided_restoration");
    }
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t px = highbd ? 2 : 1, in_bytes = up256((size_t)(width + 6) * (height + 6) * px), fl_bytes = up256((size_t)width * height * 4);
    uint8_t     *d = sc.device(in_bytes + 2 * fl_bytes + 256), *h = sc.host(in_bytes + 2 * fl_bytes + 256);
    size_t       off = 0, pitch = 0;
    const void  *dp = highbd ? (const void *)stage_region<uint16_t>(d, h, off, decode_ptr<uint16_t>(dgd8, 1), width, height, dgd_stride, 3, pitch)
                             : (const void *)stage_region<uint8_t>(d, h, off, dgd8, width, height, dgd_stride, 3, pitch);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    SvtHipSgrUnit u{dp, nullptr, (uint32_t)pitch, 0, (uint32_t)width, (uint32_t)height, (uint8_t)(highbd != 0), (uint8_t)bit_depth, 64, 64};
    int32_t      *df0 = (int32_t *)(d + in_bytes), *df1 = (int32_t *)(d + in_bytes + fl_bytes);
    const dim3    grid((width + 63) / 64, (height + 63) / 64);
    hipLaunchKernelGGL(sgr_filter_kernel<0>, grid, dim3(SGR_NT), 0, st, geom_of(&u), ep, df0, df1, (uint32_t)width, (void *)nullptr, 0u, 0, 0);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + in_bytes, d + in_bytes, 2 * fl_bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const int32_t *hf0 = (const int32_t *)(h + in_bytes), *hf1 = (const int32_t *)(h + in_bytes + fl_bytes);
    for (int r = 0; r < height; r++) {
        if (SGR_PRM_H[ep][0] > 0)
            memcpy(flt0 + (size_t)r * flt_stride, hf0 + (size_t)r * width, (size_t)width * 4);
        if (SGR_PRM_H[ep][1] > 0)
            memcpy(flt1 + (size_t)r * flt_stride, hf1 + (size_t)r * width, (size_t)width * 4);
    }
}

static void svt_apply_selfguided_restoration_hip_impl(const uint8_t *dat, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t *xqd, uint8_t *dst, int32_t dst_stride, int32_t *tmpbuf, int32_t bit_depth, int32_t highbd);
extern "C" void svt_apply_selfguided_restoration_hip(const uint8_t *dat, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t *xqd, uint8_t *dst, int32_t dst_stride, int32_t *tmpbuf, int32_t bit_depth, int32_t highbd) { TIER_A_CALL(svt_apply_selfguided_restoration, svt_apply_selfguided_restoration_hip_impl(dat, width, height, stride, eps, xqd, dst, dst_stride, tmpbuf, bit_depth, highbd), (dat, width, height, stride, eps, xqd, dst, dst_stride, tmpbuf, bit_depth, highbd)); }
static void svt_apply_selfguided_restoration_hip_impl(const uint8_t *dat, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t *xqd, uint8_t *dst, int32_t dst_stride, int32_t *tmpbuf, int32_t bit_depth, int32_t highbd) {
    (void)tmpbuf;
    if (!ensure_init())
        fatal("apply_selfguided_restoration");
    if (width <= 0 || height <= 0 || width > 384 || height > 384 || eps < 0 || eps > 15) {
        set_error("svt_apply_selfguided_restoration_hip: unsupported size %dx%d / eps %d", width, height, eps);
        fatal("apply_selfguided_restoration");
    }
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t px = highbd ? 2 : 1, in_bytes = up256((size_t)(width + 6) * (height + 6) * px), out_bytes = up256((size_t)width * height * px);
    uint8_t     *d = sc.device(in_bytes + out_bytes + 256), *h = sc.host(in_bytes + out_bytes + 256);
    size_t       off = 0, pitch = 0;
    const void  *dp = highbd ? (const void *)stage_region<uint16_t>(d, h, off, decode_ptr<uint16_t>(dat, 1), width, height, stride, 3, pitch)
                             : (const void *)stage_region<uint8_t>(d, h, off, dat, width, height, stride, 3, pitch);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    SvtHipSgrUnit u{dp, nullptr, (uint32_t)pitch, 0, (uint32_t)width, (uint32_t)height, (uint8_t)(highbd != 0), (uint8_t)bit_depth, 64, 64};
    if (svt_hip_sgr_apply_unit(&u, eps, xqd, d + in_bytes, (uint32_t)width, st) != SVT_HIP_OK)
        fatal("apply_selfguided_restoration");
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + in_bytes, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    uint8_t *dst_b = highbd ? (uint8_t *)((uintptr_t)dst << 1) : dst;
    for (int r = 0; r < height; r++) memcpy(dst_b + (size_t)r * dst_stride * px, h + in_bytes + (size_t)r * width * px, (size_t)width * px);
}

// Shared staging of (src, dat, flt0, flt1) for the two reductions
static void proj_tier_a(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8, int dat_stride, int highbd,
                        const int32_t *flt0, int flt0_stride, const int32_t *flt1, int flt1_stride, const SvtHipSgrParams *params, int xq0,
                        int xq1, int want_sums, long long out[6], int32_t *xq_solved) {
    if (!ensure_init())
        fatal("sgr projection");
    if (width <= 0 || height <= 0 || (size_t)width * height > (size_t)1 << 20 || !params) {
        set_error("sgr projection: unsupported size %dx%d", width, height);
        fatal("sgr projection");
    }
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t px = highbd ? 2 : 1, pl = up256((size_t)width * height * px), fl = up256((size_t)width * height * 4);
    const size_t total = 2 * pl + 2 * fl + 256;
    uint8_t     *d = sc.device(total + 256), *h = sc.host(total + 256);
    const uint8_t *sb = highbd ? (const uint8_t *)((uintptr_t)src8 << 1) : src8, *db = highbd ? (const uint8_t *)((uintptr_t)dat8 << 1) : dat8;
    const int      r0 = params->r[0], r1 = params->r[1];
    for (int r = 0; r < height; r++) {
        memcpy(h + (size_t)r * width * px, sb + (size_t)r * src_stride * px, (size_t)width * px);
        memcpy(h + pl + (size_t)r * width * px, db + (size_t)r * dat_stride * px, (size_t)width * px);
        if (r0 > 0)
            memcpy(h + 2 * pl + (size_t)r * width * 4, flt0 + (size_t)r * flt0_stride, (size_t)width * 4);
        if (r1 > 0)
            memcpy(h + 2 * pl + fl + (size_t)r * width * 4, flt1 + (size_t)r * flt1_stride, (size_t)width * 4);
    }
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, 2 * pl + 2 * fl, hipMemcpyHostToDevice, st));
    UnitData   u{d, d + pl, (const int32_t *)(d + 2 * pl), (const int32_t *)(d + 2 * pl + fl), (uint32_t)width, (uint32_t)width, (uint32_t)width,
               (uint32_t)width, (uint32_t)width, (uint32_t)height, highbd != 0, r0, r1};
    long long *dres = (long long *)(d + 2 * pl + 2 * fl);
    hipLaunchKernelGGL(sgr_stats_kernel, dim3(1), dim3(1024), 0, st, u, xq0, xq1, want_sums, dres);
    if (want_sums && xq_solved)
        hipLaunchKernelGGL(sgr_solve_kernel, dim3(1), dim3(64), 0, st, (const long long *)dres, width * height, r0, r1, (int32_t *)(dres + 8));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + 2 * pl + 2 * fl, dres, 80, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(out, h + 2 * pl + 2 * fl, 6 * sizeof(long long));
    if (xq_solved)
        memcpy(xq_solved, h + 2 * pl + 2 * fl + 64, 8);
}

static int64_t svt_av1_lowbd_pixel_proj_error_hip_impl(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params);
extern "C" int64_t svt_av1_lowbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params) { TIER_A_CALL(svt_av1_lowbd_pixel_proj_error, svt_av1_lowbd_pixel_proj_error_hip_impl(src8, width, height, src_stride, dat8, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, params), (src8, width, height, src_stride, dat8, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, params)); }
static int64_t svt_av1_lowbd_pixel_proj_error_hip_impl(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params) {
    long long out[6];
    proj_tier_a(src8, width, height, src_stride, dat8, dat_stride, 0, flt0, flt0_stride, flt1, flt1_stride, params, xq[0], xq[1], 0, out, nullptr);
    return out[5];
}
static int64_t svt_av1_highbd_pixel_proj_error_hip_impl(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params);
extern "C" int64_t svt_av1_highbd_pixel_proj_error_hip(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params) { TIER_A_CALL(svt_av1_highbd_pixel_proj_error, svt_av1_highbd_pixel_proj_error_hip_impl(src8, width, height, src_stride, dat8, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, params), (src8, width, height, src_stride, dat8, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, params)); }
static int64_t svt_av1_highbd_pixel_proj_error_hip_impl(const uint8_t *src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t *dat8, int32_t dat_stride, int32_t *flt0, int32_t flt0_stride, int32_t *flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams *params) {
    long long out[6];
    proj_tier_a(src8, width, height, src_stride, dat8, dat_stride, 1, flt0, flt0_stride, flt1, flt1_stride, params, xq[0], xq[1], 0, out, nullptr);
    return out[5];
}
static void svt_get_proj_subspace_hip_impl(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8, int dat_stride, int use_highbitdepth, int32_t *flt0, int flt0_stride, int32_t *flt1, int flt1_stride, int *xq, const SvtHipSgrParams *params);
extern "C" void svt_get_proj_subspace_hip(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8, int dat_stride, int use_highbitdepth, int32_t *flt0, int flt0_stride, int32_t *flt1, int flt1_stride, int *xq, const SvtHipSgrParams *params) { TIER_A_CALL(svt_get_proj_subspace, svt_get_proj_subspace_hip_impl(src8, width, height, src_stride, dat8, dat_stride, use_highbitdepth, flt0, flt0_stride, flt1, flt1_stride, xq, params), (src8, width, height, src_stride, dat8, dat_stride, use_highbitdepth, flt0, flt0_stride, flt1, flt1_stride, xq, params)); }
static void svt_get_proj_subspace_hip_impl(const uint8_t *src8, int width, int height, int src_stride, const uint8_t *dat8, int dat_stride, int use_highbitdepth, int32_t *flt0, int flt0_stride, int32_t *flt1, int flt1_stride, int *xq, const SvtHipSgrParams *params) {
    long long out[6];
    int32_t   solved[2] = {0, 0};
    proj_tier_a(src8, width, height, src_stride, dat8, dat_stride, use_highbitdepth, flt0, flt0_stride, flt1, flt1_stride, params, 0, 0, 1, out,
                solved);
    xq[0] = solved[0], xq[1] = solved[1];
}

SVT_HIP_MODULE_WARMUP(loopfilter_sgr)
