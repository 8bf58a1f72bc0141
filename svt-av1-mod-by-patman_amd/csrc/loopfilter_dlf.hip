// loopfilter_dlf.hip — AV1 deblocking on gfx950.
// Replaces svt_av1_loop_filter_frame / svt_aom_loop_filter_sb / svt_av1_filter_block_plane_vert|horz /
// set_lpf_parameters (deblocking_filter.c:162-653) and the sixteen svt_aom_[highbd_]lpf_* leaves
// (deblocking_common.c:141-865).
//
// AV1 sizes every edge filter so that the samples one edge modifies are never read by another edge of the same
// direction, so a pass is embarrassingly parallel: one thread owns one 4-sample edge segment (one 4x4 unit),
// derives the edge decision from the two mode-info records either side of it and filters its four lines in
// registers.  The frame is two passes per plane — all vertical edges, then all horizontal edges — which is what
// the reference's superblock schedule (vert SB n, horz SB n-1) is equivalent to.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_lf.h"
#include "common.hpp"

using namespace svthip;

namespace {

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Bd {  // per-bit-depth constants of signed_char_clamp_high / the <<(bd-8) threshold scaling
    int shift, lo, hi, off;
    __device__ explicit Bd(int bd) : shift(bd - 8), lo(-(128 << (bd - 8))), hi((128 << (bd - 8)) - 1), off(0x80 << (bd - 8)) {}
    __device__ int sc(int t) const { return clampi(t, lo, hi); }
};

__device__ __forceinline__ void filter4(bool mask, int thresh, int &p1, int &p0, int &q0, int &q1, const Bd &b) {
    const int  t16 = thresh << b.shift;
    const int  ps1 = p1 - b.off, ps0 = p0 - b.off, qs0 = q0 - b.off, qs1 = q1 - b.off;
    const bool hev = iabs(p1 - p0) > t16 || iabs(q1 - q0) > t16;
    int        filter = hev ? b.sc(ps1 - qs1) : 0;
    filter            = mask ? b.sc(filter + 3 * (qs0 - ps0)) : 0;
    const int f1 = b.sc(filter + 4) >> 3, f2 = b.sc(filter + 3) >> 3;
    q0     = b.sc(qs0 - f1) + b.off;
    p0     = b.sc(ps0 + f2) + b.off;
    filter = hev ? 0 : ((f1 + 1) >> 1);
    q1     = b.sc(qs1 - filter) + b.off;
    p1     = b.sc(ps1 + filter) + b.off;
}

// One line across an edge; v[0..13] = p6..p0,q0..q6 (only the +-reach middle part is valid / used).
__device__ __forceinline__ void filter_line(int (&v)[14], int len, int blimit, int limit, int thresh, const Bd &b) {
    const int l16 = limit << b.shift, b16 = blimit << b.shift, one = 1 << b.shift;
    int &p0 = v[6], &p1 = v[5], &p2 = v[4], &p3 = v[3], &q0 = v[7], &q1 = v[8], &q2 = v[9], &q3 = v[10];
    bool mask = !(iabs(p1 - p0) > l16 || iabs(q1 - q0) > l16 || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > b16);
    if (len == 4) {
        filter4(mask, thresh, p1, p0, q0, q1, b);
        return;
    }
    if (len == 6) {
        mask = mask && !(iabs(p2 - p1) > l16 || iabs(q2 - q1) > l16);
        const bool flat = !(iabs(p1 - p0) > one || iabs(q1 - q0) > one || iabs(p2 - p0) > one || iabs(q2 - q0) > one);
        if (flat && mask) {
            const int a2 = p2, a1 = p1, a0 = p0, c0 = q0, c1 = q1, c2 = q2;
            p1 = (a2 * 3 + a1 * 2 + a0 * 2 + c0 + 4) >> 3;
            p0 = (a2 + a1 * 2 + a0 * 2 + c0 * 2 + c1 + 4) >> 3;
            q0 = (a1 + a0 * 2 + c0 * 2 + c1 * 2 + c2 + 4) >> 3;
            q1 = (a0 + c0 * 2 + c1 * 2 + c2 * 3 + 4) >> 3;
        } else
            filter4(mask, thresh, p1, p0, q0, q1, b);
        return;
    }
    mask = mask && !(iabs(p3 - p2) > l16 || iabs(p2 - p1) > l16 || iabs(q2 - q1) > l16 || iabs(q3 - q2) > l16);
    const bool flat = !(iabs(p1 - p0) > one || iabs(q1 - q0) > one || iabs(p2 - p0) > one || iabs(q2 - q0) > one ||
                        iabs(p3 - p0) > one || iabs(q3 - q0) > one);
    const int a3 = p3, a2 = p2, a1 = p1, a0 = p0, c0 = q0, c1 = q1, c2 = q2, c3 = q3;
    if (len == 14) {
        const int  a6 = v[0], a5 = v[1], a4 = v[2], c4 = v[11], c5 = v[12], c6 = v[13];
        const bool flat2 = !(iabs(a4 - a0) > one || iabs(c4 - c0) > one || iabs(a5 - a0) > one || iabs(c5 - c0) > one ||
                             iabs(a6 - a0) > one || iabs(c6 - c0) > one);
        if (flat2 && flat && mask) {
            v[1]  = (a6 * 7 + a5 * 2 + a4 * 2 + a3 + a2 + a1 + a0 + c0 + 8) >> 4;
            v[2]  = (a6 * 5 + a5 * 2 + a4 * 2 + a3 * 2 + a2 + a1 + a0 + c0 + c1 + 8) >> 4;
            v[3]  = (a6 * 4 + a5 + a4 * 2 + a3 * 2 + a2 * 2 + a1 + a0 + c0 + c1 + c2 + 8) >> 4;
            v[4]  = (a6 * 3 + a5 + a4 + a3 * 2 + a2 * 2 + a1 * 2 + a0 + c0 + c1 + c2 + c3 + 8) >> 4;
            v[5]  = (a6 * 2 + a5 + a4 + a3 + a2 * 2 + a1 * 2 + a0 * 2 + c0 + c1 + c2 + c3 + c4 + 8) >> 4;
            v[6]  = (a6 + a5 + a4 + a3 + a2 + a1 * 2 + a0 * 2 + c0 * 2 + c1 + c2 + c3 + c4 + c5 + 8) >> 4;
            v[7]  = (a5 + a4 + a3 + a2 + a1 + a0 * 2 + c0 * 2 + c1 * 2 + c2 + c3 + c4 + c5 + c6 + 8) >> 4;
            v[8]  = (a4 + a3 + a2 + a1 + a0 + c0 * 2 + c1 * 2 + c2 * 2 + c3 + c4 + c5 + c6 * 2 + 8) >> 4;
            v[9]  = (a3 + a2 + a1 + a0 + c0 + c1 * 2 + c2 * 2 + c3 * 2 + c4 + c5 + c6 * 3 + 8) >> 4;
            v[10] = (a2 + a1 + a0 + c0 + c1 + c2 * 2 + c3 * 2 + c4 * 2 + c5 + c6 * 4 + 8) >> 4;
            v[11] = (a1 + a0 + c0 + c1 + c2 + c3 * 2 + c4 * 2 + c5 * 2 + c6 * 5 + 8) >> 4;
            v[12] = (a0 + c0 + c1 + c2 + c3 + c4 * 2 + c5 * 2 + c6 * 7 + 8) >> 4;
            return;
        }
    }
    if (flat && mask) {
        p2 = (a3 + a3 + a3 + 2 * a2 + a1 + a0 + c0 + 4) >> 3;
        p1 = (a3 + a3 + a2 + 2 * a1 + a0 + c0 + c1 + 4) >> 3;
        p0 = (a3 + a2 + a1 + 2 * a0 + c0 + c1 + c2 + 4) >> 3;
        q0 = (a2 + a1 + a0 + 2 * c0 + c1 + c2 + c3 + 4) >> 3;
        q1 = (a1 + a0 + c0 + 2 * c1 + c2 + c3 + c3 + 4) >> 3;
        q2 = (a0 + c0 + c1 + 2 * c2 + c3 + c3 + c3 + 4) >> 3;
    } else
        filter4(mask, thresh, p1, p0, q0, q1, b);
}

// Filter the 4 lines of one edge segment in memory.  s = first q0 sample; tap / along in samples.
template <typename T>
__device__ __forceinline__ void filter_segment(T *s, ptrdiff_t tap, ptrdiff_t along, int len, int blimit, int limit, int thresh, int bd) {
    const int reach = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const Bd  b(bd);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int v[14];
#pragma unroll
        for (int k = -7; k < 7; k++) v[7 + k] = (k >= -reach && k < reach) ? (int)s[i * along + k * tap] : 0;
        filter_line(v, len, blimit, limit, thresh, b);
#pragma unroll
        for (int k = -6; k < 6; k++)  // filters longer than 4 only read their outermost tap
            if (k >= -reach + (len != 4) && k < reach - (len != 4))
                s[i * along + k * tap] = (T)v[7 + k];
    }
}

__device__ const uint8_t TX_W_LOG2[19]  = {2, 3, 4, 5, 6, 2, 3, 3, 4, 4, 5, 5, 6, 2, 4, 3, 5, 4, 6};
__device__ const uint8_t TX_H_LOG2[19]  = {2, 3, 4, 5, 6, 3, 2, 4, 3, 5, 4, 6, 5, 4, 2, 5, 3, 6, 4};
__device__ const uint8_t BLK_W_LOG2[22] = {2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 2, 4, 3, 5, 4, 6};
__device__ const uint8_t BLK_H_LOG2[22] = {2, 3, 2, 3, 4, 3, 4, 5, 4, 5, 6, 5, 6, 7, 6, 7, 4, 2, 5, 3, 6, 4};

struct LfGeom {  // what the pass kernel needs besides the level table
    void             *plane;
    uint32_t          stride, width, height;  // plane size (unpadded), in samples
    const SvtHipLfMi *mi;
    uint32_t          mi_stride, units_x, units_y;
    uint8_t           ss, plane_id, sharpness, bit_depth, is_16bit;
};
struct LfLevels {
    uint8_t lvl[8][2][8][2];  // this plane's slice of LoopFilterInfoN.lvl
};

// set_lpf_parameters (deblocking_filter.c:162-282) for the unit at plane position (x, y).
template <int DIR>
__device__ __forceinline__ int edge_decision(const LfGeom &g, const LfLevels &L, uint32_t x, uint32_t y, int &level) {
    const int         ss = g.ss;
    const uint32_t    mi_row = ss | ((y << ss) >> 2), mi_col = ss | ((x << ss) >> 2);
    const SvtHipLfMi *mp = g.mi + (size_t)mi_row * g.mi_stride + mi_col;
    const SvtHipLfMi  mi = *mp;
    const int         tsz = g.plane_id ? mi.tx_size_uv : mi.tx_size_y;
    const int         ts  = DIR == 0 ? TX_W_LOG2[tsz] : TX_H_LOG2[tsz];
    const uint32_t    coord = DIR == 0 ? x : y;
    if (coord == 0 || (coord & ((1u << ts) - 1)))
        return 0;
    const int        curr = L.lvl[mi.segment_id][DIR][mi.ref_frame0][mi.mode_lf];
    const SvtHipLfMi pv   = DIR == 0 ? mp[-(1 << ss)] : *(mp - ((size_t)g.mi_stride << ss));
    const int        ptsz = g.plane_id ? pv.tx_size_uv : pv.tx_size_y;
    const int        pv_ts = DIR == 0 ? TX_W_LOG2[ptsz] : TX_H_LOG2[ptsz];
    const int        pv_lvl = L.lvl[pv.segment_id][DIR][pv.ref_frame0][pv.mode_lf];
    int              bdim = DIR == 0 ? BLK_W_LOG2[mi.bsize] : BLK_H_LOG2[mi.bsize];
    if (ss)
        bdim = bdim - 1 < 2 ? 2 : bdim - 1;
    const bool pu_edge = !(coord & ((1u << bdim) - 1));
    if ((curr || pv_lvl) && (!pv.skip_inter || !mi.skip_inter || pu_edge)) {
        const int min_ts = ts < pv_ts ? ts : pv_ts;
        level            = curr ? curr : pv_lvl;
        return min_ts <= 2 ? 4 : (g.plane_id ? 6 : (min_ts == 3 ? 8 : 14));
    }
    return 0;
}

// All planes of a pass in one launch (blockIdx.z = plane): a pass over one 4K plane is a 10 - 15 us kernel, so six separate
// launches spend a good part of their time ramping up and draining.
struct LfPassArgs {
    LfGeom   g[3];
    LfLevels L[3];
};
template <int DIR>
__global__ __launch_bounds__(256) void dlf_pass_kernel(LfPassArgs a) {
    const LfGeom   &g = a.g[blockIdx.z];
    const LfLevels &L = a.L[blockIdx.z];
    // 64x4 thread tiles: consecutive lanes walk along x so that the horizontal-edge pass is fully coalesced
    const uint32_t ux = blockIdx.x * 64 + (threadIdx.x & 63), uy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ux >= g.units_x || uy >= g.units_y)
        return;
    const uint32_t x = ux * 4, y = uy * 4;
    if (x >= g.width || y >= g.height)
        return;
    int       level = 0;
    const int len   = edge_decision<DIR>(g, L, x, y, level);
    if (!len)
        return;
    int inside = level >> ((g.sharpness > 0) + (g.sharpness > 4));  // svt_aom_update_sharpness
    if (g.sharpness > 0 && inside > 9 - g.sharpness)
        inside = 9 - g.sharpness;
    if (inside < 1)
        inside = 1;
    const int       mblim = 2 * (level + 2) + inside, hev = level >> 4;
    const ptrdiff_t tap = DIR == 0 ? 1 : (ptrdiff_t)g.stride, along = DIR == 0 ? (ptrdiff_t)g.stride : 1;
    const size_t    o   = (size_t)y * g.stride + x;
    if (g.is_16bit)
        filter_segment((uint16_t *)g.plane + o, tap, along, len, mblim, inside, hev, g.bit_depth);
    else
        filter_segment((uint8_t *)g.plane + o, tap, along, len, mblim, inside, hev, 8);
}

template <typename T>
__global__ void lpf_leaf_kernel(T *s, int pitch, int vertical, int len, int blimit, int limit, int thresh, int bd) {
    if (threadIdx.x == 0)
        filter_segment(s, vertical ? (ptrdiff_t)1 : (ptrdiff_t)pitch, vertical ? (ptrdiff_t)pitch : (ptrdiff_t)1, len, blimit, limit, thresh, bd);
}

// Tier A: stage the samples the reference call touches (4 lines x 2*reach taps), run the leaf, copy them back.
template <typename T> void lpf_tier_a(T *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd, int len, int vertical) {
    if (!ensure_init()) {
        svthip::tier_a_throw("lpf: %s", svt_hip_last_error());
    }
    const int       reach = len == 4 ? 2 : (len == 6 ? 3 : (len == 8 ? 4 : 7));
    const ptrdiff_t first = vertical ? -reach : -(ptrdiff_t)reach * pitch;
    const ptrdiff_t last  = vertical ? 3 * (ptrdiff_t)pitch + reach : (ptrdiff_t)(reach - 1) * pitch + 4;  // one past
    const size_t    n     = (size_t)(last - first), bytes = n * sizeof(T);
    hipStream_t     st    = resolve_stream(nullptr);
    Scratch        &sc    = tls_scratch();
    uint8_t        *d = sc.device(bytes + 256), *h = sc.host(bytes + 256);
    memcpy(h, s + first, bytes);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lpf_leaf_kernel<T>, dim3(1), dim3(64), 0, st, (T *)d - first, (int)pitch, vertical, len, (int)*blimit, (int)*limit,
                       (int)*thresh, bd);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    // write back only the taps of the four lines (the span in between belongs to the caller's other samples)
    for (int i = 0; i < 4; i++)
        for (int k = -reach; k < reach; k++) {
            const ptrdiff_t o = vertical ? i * (ptrdiff_t)pitch + k : k * (ptrdiff_t)pitch + i;
            s[o]              = ((const T *)h)[o - first];
        }
}

}  // namespace

#define SVT_HIP_DEF_LPF(dir, n, vert)                                                                                              \
    extern "C" void svt_aom_lpf_##dir##_##n##_hip(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit,          \
                                                  const uint8_t *thresh) {                                                         \
        TIER_A_CALL(svt_aom_lpf_##dir##_##n, lpf_tier_a<uint8_t>(s, pitch, blimit, limit, thresh, 8, n, vert), (s, pitch, blimit, limit, thresh)); \
    }                                                                                                                              \
    extern "C" void svt_aom_highbd_lpf_##dir##_##n##_hip(uint16_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit,  \
                                                         const uint8_t *thresh, int32_t bd) {                                      \
        TIER_A_CALL(svt_aom_highbd_lpf_##dir##_##n, lpf_tier_a<uint16_t>(s, pitch, blimit, limit, thresh, bd, n, vert), (s, pitch, blimit, limit, thresh, bd)); \
    }
SVT_HIP_DEF_LPF(horizontal, 4, 0)
SVT_HIP_DEF_LPF(horizontal, 6, 0)
SVT_HIP_DEF_LPF(horizontal, 8, 0)
SVT_HIP_DEF_LPF(horizontal, 14, 0)
SVT_HIP_DEF_LPF(vertical, 4, 1)
SVT_HIP_DEF_LPF(vertical, 6, 1)
SVT_HIP_DEF_LPF(vertical, 8, 1)
SVT_HIP_DEF_LPF(vertical, 14, 1)

extern "C" int32_t svt_hip_loop_filter_frame(const SvtHipLfFrame *f, void *stream) {
    if (!f || !f->mi || !f->width || !f->height || f->plane_start > f->plane_end || f->plane_end > 3 || f->mi_stride < f->mi_cols ||
        (f->bit_depth != 8 && f->bit_depth != 10 && f->bit_depth != 12) || (f->bit_depth > 8 && !f->is_16bit) ||
        f->width > f->mi_cols * 4 || f->height > f->mi_rows * 4 || (f->mi_cols & 1) || (f->mi_rows & 1) || f->sharpness_level > 7) {
        set_error("svt_hip_loop_filter_frame: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    for (int p = f->plane_start; p < f->plane_end; p++)
        if (!f->plane[p] || !f->stride[p]) {
            set_error("svt_hip_loop_filter_frame: missing plane");
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t st = resolve_stream(stream);
    LfPassArgs  a{};
    uint32_t    np = 0, gx = 0, gy = 0;
    for (int p = f->plane_start; p < f->plane_end; p++) {
        if (p == 0 && !f->filter_level[0] && !f->filter_level[1])
            break;  // deblocking_filter.c:570-572
        if ((p == 1 && !f->filter_level_u) || (p == 2 && !f->filter_level_v))
            continue;
        const int ss = p > 0;
        LfGeom   &g = a.g[np];
        g.plane = f->plane[p], g.stride = f->stride[p], g.width = f->width >> ss, g.height = f->height >> ss;
        g.mi = f->mi, g.mi_stride = f->mi_stride;
        g.units_x = (f->mi_cols >> ss), g.units_y = (f->mi_rows >> ss);
        g.ss = (uint8_t)ss, g.plane_id = (uint8_t)p, g.sharpness = f->sharpness_level, g.bit_depth = f->bit_depth, g.is_16bit = f->is_16bit;
        memcpy(a.L[np].lvl, f->lvl[p], sizeof(a.L[np].lvl));
        gx = (g.units_x + 63) / 64 > gx ? (g.units_x + 63) / 64 : gx;
        gy = (g.units_y + 3) / 4 > gy ? (g.units_y + 3) / 4 : gy;
        np++;
    }
    if (np) {  // vertical edges of every plane, then horizontal edges of every plane (the planes are independent)
        hipLaunchKernelGGL(dlf_pass_kernel<0>, dim3(gx, gy, np), dim3(256), 0, st, a);
        hipLaunchKernelGGL(dlf_pass_kernel<1>, dim3(gx, gy, np), dim3(256), 0, st, a);
    }
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

SVT_HIP_MODULE_WARMUP(loopfilter_dlf)
