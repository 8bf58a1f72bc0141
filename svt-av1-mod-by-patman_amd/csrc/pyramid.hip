// pyramid.hip — 1/4 and 1/16 decimation with fused edge padding, and per-64x64 block mean / variance.
// Replaces svt_aom_downsample_filtering_input_picture (pic_analysis_process.c:1922-1979 of the reference,
// leaf svt_aom_downsample_2d_c :131-161 + svt_aom_generate_padding pic_operators.c:338-383) and
// compute_picture_spatial_statistics (:1533-1553, leaf compute_block_mean_compute_variance :307-1382).
// Both are pure HBM-bound streaming kernels (1.5625*P and P + 170*B bytes).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

using namespace svthip;

namespace {

// One thread = one output dword (4 horizontally adjacent samples) of the PADDED destination plane.
// Output sample (X, Y) of the padded buffer = decimated sample (clamp(X-pad), clamp(Y-pad)): this is
// exactly what downsample followed by svt_aom_generate_padding produces.
// STEP = 2: mean of in(2i..2i+1, 2j..2j+1).  STEP = 4: mean of in(4i+1..4i+2, 4j+1..4j+2).
template <int STEP>
__device__ __forceinline__ void downsample_pad_body(const uint8_t *__restrict__ in, uint32_t in_stride, uint8_t *__restrict__ out,
                                                    uint32_t out_stride, uint32_t out_w, uint32_t out_h, uint32_t pad, uint32_t Y) {
    const uint32_t X4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;  // first padded column of this dword
    if (X4 >= out_stride || Y >= out_h + 2 * pad)
        return;
    int yi = (int)Y - (int)pad;
    yi     = yi < 0 ? 0 : (yi > (int)out_h - 1 ? (int)out_h - 1 : yi);
    constexpr int  OFF = STEP == 2 ? 0 : 1;
    const uint8_t *r0  = in + (size_t)(STEP * yi + OFF) * in_stride;
    const uint8_t *r1  = r0 + in_stride;
    uint32_t       res = 0;
    const int      x0  = (int)X4 - (int)pad;
    if (STEP == 2 && x0 >= 0 && x0 + 3 < (int)out_w && ((((uintptr_t)r0 | in_stride) & 3u) == 0) && ((x0 & 1) == 0)) {
        // interior fast path: 8 input bytes per row as two aligned dwords
        const uint32_t *pa = (const uint32_t *)(r0 + 2 * x0), *pb = (const uint32_t *)(r1 + 2 * x0);
        const uint2     a = make_uint2(pa[0], pa[1]), b = make_uint2(pb[0], pb[1]);
        // per 16-bit pair: bytes (lo, hi) of both rows
        auto avg2 = [](uint32_t ta, uint32_t tb, int sh) -> uint32_t {
            const uint32_t s = ((ta >> sh) & 0xff) + ((ta >> (sh + 8)) & 0xff) + ((tb >> sh) & 0xff) + ((tb >> (sh + 8)) & 0xff);
            return (s + 2) >> 2;
        };
        res = avg2(a.x, b.x, 0) | (avg2(a.x, b.x, 16) << 8) | (avg2(a.y, b.y, 0) << 16) | (avg2(a.y, b.y, 16) << 24);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (X4 + k >= out_stride)
                break;
            int xi = x0 + k;
            xi     = xi < 0 ? 0 : (xi > (int)out_w - 1 ? (int)out_w - 1 : xi);
            const int      xs = STEP * xi + OFF;
            const uint32_t s  = (uint32_t)r0[xs] + r0[xs + 1] + r1[xs] + r1[xs + 1];
            res |= ((s + 2) >> 2) << (8 * k);
        }
    }
    uint8_t *o = out + (size_t)Y * out_stride + X4;
    if (X4 + 3 < out_stride && (((uintptr_t)o) & 3u) == 0) {
        *(uint32_t *)o = res;
    } else {
        for (int k = 0; k < 4 && X4 + k < out_stride; k++) o[k] = (uint8_t)(res >> (8 * k));
    }
}

template <int STEP>
__global__ __launch_bounds__(256) void downsample_pad_kernel(const uint8_t *__restrict__ in, uint32_t in_stride,
                                                             uint8_t *__restrict__ out, uint32_t out_stride,
                                                             uint32_t out_w, uint32_t out_h, uint32_t pad) {
    downsample_pad_body<STEP>(in, in_stride, out, out_stride, out_w, out_h, pad, blockIdx.y);
}

constexpr uint32_t BATCH_ROWS = 4;
// Batched form: blockIdx.z = picture.  level 0: full -> quarter (STEP 2), 1: quarter -> sixteenth (STEP 2),
// 2: full -> sixteenth (STEP 4, HME level 1 off).
__global__ __launch_bounds__(256) void downsample_pad_batch_kernel(const SvtHipAnalysisJob *__restrict__ jobs, int level) {
    const SvtHipPyramid8 &y = jobs[blockIdx.z].pyr;
    const SvtHipPlane8   &i = level == 1 ? y.quarter : y.full, &o = level == 0 ? y.quarter : y.sixteenth;
    const uint8_t        *in = i.buf + i.org_x + (size_t)i.org_y * i.stride;
    // BATCH_ROWS output rows per workgroup: a row of one dword per thread is too little work per launch slot
    for (uint32_t r = 0; r < BATCH_ROWS; r++) {
        if (level == 2)
            downsample_pad_body<4>(in, i.stride, o.buf, o.stride, o.width, o.height, o.org_x, blockIdx.y * BATCH_ROWS + r);
        else
            downsample_pad_body<2>(in, i.stride, o.buf, o.stride, o.width, o.height, o.org_x, blockIdx.y * BATCH_ROWS + r);
    }
}

// Edge replication of a plane in place (svt_aom_generate_padding); interior untouched.
__global__ __launch_bounds__(256) void pad_plane_kernel(uint8_t *__restrict__ buf, uint32_t stride, uint32_t w,
                                                        uint32_t h, uint32_t pad_x, uint32_t pad_y) {
    const uint32_t X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= stride)
        return;
    const bool inside = X >= pad_x && X < pad_x + w && Y >= pad_y && Y < pad_y + h;
    if (inside)
        return;
    int xi = (int)X - (int)pad_x, yi = (int)Y - (int)pad_y;
    xi     = xi < 0 ? 0 : (xi > (int)w - 1 ? (int)w - 1 : xi);
    yi     = yi < 0 ? 0 : (yi > (int)h - 1 ? (int)h - 1 : yi);
    buf[(size_t)Y * stride + X] = buf[(size_t)(pad_y + yi) * stride + pad_x + xi];
}

// One wave64 per 64x64 block, one lane per 8x8 sub-block; 4 blocks per workgroup.
__device__ __forceinline__ void variance_body(const uint8_t *__restrict__ pic /* at (org_x, org_y) */, uint32_t stride, uint32_t b64_w,
                                              uint32_t n_b64, uint16_t *__restrict__ variance, uint64_t *__restrict__ mean,
                                              int full_precision) {
    __shared__ uint64_t m8[4][64], q8[4][64], m16[4][16], q16[4][16], m32[4][4], q32[4][4];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b  = blockIdx.x * 4 + wv;
    const bool     on = b < n_b64;
    if (on) {
        const uint32_t bx = b % b64_w, by = b / b64_w;
        const uint32_t sx = lane & 7, sy = lane >> 3;
        const uint8_t *p  = pic + (size_t)(64 * by + 8 * sy) * stride + 64 * bx + 8 * sx;
        uint32_t       s = 0, ss = 0;
        const int      step = full_precision ? 1 : 2;
        for (int r = 0; r < 8; r += step) {
            const uint8_t *row = p + (size_t)r * stride;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const uint32_t v = row[c];
                s += v, ss += v * v;
            }
        }
        // sub-sampled: sum<<3, sumsq<<11 (pic_analysis_process.c:234-272); full: (sum<<8)/64, (sumsq<<16)/64
        m8[wv][lane] = full_precision ? ((uint64_t)s << 2) : ((uint64_t)s << 3);
        q8[wv][lane] = full_precision ? ((uint64_t)ss << 10) : ((uint64_t)ss << 11);
    }
    __syncthreads();
    if (on && lane < 16) {
        const uint32_t a = 16 * (lane >> 2) + 2 * (lane & 3);
        m16[wv][lane]    = (m8[wv][a] + m8[wv][a + 1] + m8[wv][a + 8] + m8[wv][a + 9]) >> 2;
        q16[wv][lane]    = (q8[wv][a] + q8[wv][a + 1] + q8[wv][a + 8] + q8[wv][a + 9]) >> 2;
    }
    __syncthreads();
    if (on && lane < 4) {
        const uint32_t a = 8 * (lane >> 1) + 2 * (lane & 1);
        m32[wv][lane]    = (m16[wv][a] + m16[wv][a + 1] + m16[wv][a + 4] + m16[wv][a + 5]) >> 2;
        q32[wv][lane]    = (q16[wv][a] + q16[wv][a + 1] + q16[wv][a + 4] + q16[wv][a + 5]) >> 2;
    }
    __syncthreads();
    if (!on)
        return;
    uint16_t *vo = variance + (size_t)85 * b;
    uint64_t *mo = mean ? mean + (size_t)85 * b : nullptr;
    for (uint32_t i = lane; i < 85; i += 64) {
        uint64_t m, q;
        if (i == 0) {
            m = (m32[wv][0] + m32[wv][1] + m32[wv][2] + m32[wv][3]) >> 2;
            q = (q32[wv][0] + q32[wv][1] + q32[wv][2] + q32[wv][3]) >> 2;
        } else if (i < 5) {
            m = m32[wv][i - 1], q = q32[wv][i - 1];
        } else if (i < 21) {
            m = m16[wv][i - 5], q = q16[wv][i - 5];
        } else {
            m = m8[wv][i - 21], q = q8[wv][i - 21];
        }
        vo[i] = (uint16_t)((q - m * m) >> 16);
        if (mo)
            mo[i] = m;
    }
}

__global__ __launch_bounds__(256) void variance_kernel(const uint8_t *__restrict__ pic, uint32_t stride, uint32_t b64_w, uint32_t n_b64,
                                                       uint16_t *__restrict__ variance, uint64_t *__restrict__ mean, int full_precision) {
    variance_body(pic, stride, b64_w, n_b64, variance, mean, full_precision);
}
__global__ __launch_bounds__(256) void variance_batch_kernel(const SvtHipAnalysisJob *__restrict__ jobs, int full_precision) {
    const SvtHipAnalysisJob &j = jobs[blockIdx.y];
    const SvtHipPlane8      &f = j.pyr.full;
    const uint32_t           bw = (f.width + 63u) / 64u, bh = (f.height + 63u) / 64u;
    variance_body(f.buf + f.org_x + (size_t)f.org_y * f.stride, f.stride, bw, bw * bh, j.variance, j.mean, full_precision);
}

bool plane_ok(const SvtHipPlane8 *p) {
    return p && p->buf && p->width && p->height && p->stride >= (uint32_t)p->width + 2u * p->org_x;
}

}  // namespace

extern "C" int32_t svt_hip_pyramid_frame(const SvtHipPlane8 *full, const SvtHipPlane8 *quarter,
                                         const SvtHipPlane8 *sixteenth, int32_t hme_level1_enabled, void *stream) {
    if (!plane_ok(full) || !plane_ok(sixteenth) || (hme_level1_enabled && !plane_ok(quarter))) {
        set_error("svt_hip_pyramid_frame: bad plane descriptor");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    // The reference writes the decimated picture at org_x + org_x*stride (pic_analysis_process.c:1934,1953):
    // it only works for org_x == org_y, and so do we.
    if (sixteenth->org_x != sixteenth->org_y || (hme_level1_enabled && quarter->org_x != quarter->org_y)) {
        set_error("svt_hip_pyramid_frame: decimated planes need org_x == org_y (reference quirk)");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t    st   = resolve_stream(stream);
    const uint8_t *fsrc = full->buf + full->org_x + (size_t)full->org_y * full->stride;
    auto launch = [&](const uint8_t *in, uint32_t in_stride, const SvtHipPlane8 *o, int step) {
        const uint32_t rows = o->height + 2u * o->org_y;
        dim3           grid((o->stride / 4 + 1 + 255) / 256, rows);
        if (step == 2)
            hipLaunchKernelGGL(downsample_pad_kernel<2>, grid, dim3(256), 0, st, in, in_stride, o->buf, o->stride,
                               (uint32_t)o->width, (uint32_t)o->height, (uint32_t)o->org_x);
        else
            hipLaunchKernelGGL(downsample_pad_kernel<4>, grid, dim3(256), 0, st, in, in_stride, o->buf, o->stride,
                               (uint32_t)o->width, (uint32_t)o->height, (uint32_t)o->org_x);
    };
    if (hme_level1_enabled) {
        launch(fsrc, full->stride, quarter, 2);
        launch(quarter->buf + quarter->org_x + (size_t)quarter->org_y * quarter->stride, quarter->stride, sixteenth, 2);
    } else {
        launch(fsrc, full->stride, sixteenth, 4);
    }
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_pad_plane(const SvtHipPlane8 *plane, void *stream) {
    if (!plane_ok(plane)) {
        set_error("svt_hip_pad_plane: bad plane descriptor");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    dim3 grid((plane->stride + 255) / 256, plane->height + 2u * plane->org_y);
    hipLaunchKernelGGL(pad_plane_kernel, grid, dim3(256), 0, resolve_stream(stream), plane->buf, plane->stride,
                       (uint32_t)plane->width, (uint32_t)plane->height, (uint32_t)plane->org_x, (uint32_t)plane->org_y);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_variance_frame(const SvtHipPlane8 *full, uint16_t *d_variance, uint64_t *d_mean,
                                          int32_t full_precision, void *stream) {
    if (!plane_ok(full) || !d_variance) {
        set_error("svt_hip_variance_frame: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const uint32_t bw = (full->width + 63u) / 64u, bh = (full->height + 63u) / 64u, nb = bw * bh;
    hipLaunchKernelGGL(variance_kernel, dim3((nb + 3) / 4), dim3(256), 0, resolve_stream(stream),
                       (const uint8_t *)(full->buf + full->org_x + (size_t)full->org_y * full->stride), full->stride, bw,
                       nb, d_variance, d_mean, (int)full_precision);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// Batched picture analysis: pyramids + variances of n pictures in three launches (blockIdx.z / .y = picture).
extern "C" int32_t svt_hip_analysis_frames(const SvtHipAnalysisJob *jobs, uint32_t n_jobs, int32_t hme_level1_enabled,
                                           int32_t full_precision, void *stream) {
    if (!jobs || n_jobs == 0 || n_jobs > 65535) {
        set_error("svt_hip_analysis_frames: bad job count");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    uint32_t max_q_stride = 0, max_q_rows = 0, max_s_stride = 0, max_s_rows = 0, max_nb = 0;
    for (uint32_t i = 0; i < n_jobs; i++) {
        const SvtHipPyramid8 &y = jobs[i].pyr;
        if (!plane_ok(&y.full) || !plane_ok(&y.sixteenth) || (hme_level1_enabled && !plane_ok(&y.quarter)) || !jobs[i].variance ||
            y.sixteenth.org_x != y.sixteenth.org_y || (hme_level1_enabled && y.quarter.org_x != y.quarter.org_y)) {
            set_error("svt_hip_analysis_frames: job %u: bad plane descriptor / missing output", i);
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
        max_q_stride = y.quarter.stride > max_q_stride ? y.quarter.stride : max_q_stride;
        max_q_rows   = y.quarter.height + 2u * y.quarter.org_y > max_q_rows ? y.quarter.height + 2u * y.quarter.org_y : max_q_rows;
        max_s_stride = y.sixteenth.stride > max_s_stride ? y.sixteenth.stride : max_s_stride;
        max_s_rows   = y.sixteenth.height + 2u * y.sixteenth.org_y > max_s_rows ? y.sixteenth.height + 2u * y.sixteenth.org_y : max_s_rows;
        const uint32_t nb = ((y.full.width + 63u) / 64u) * ((y.full.height + 63u) / 64u);
        max_nb            = nb > max_nb ? nb : max_nb;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t st = resolve_stream(stream);
    const SvtHipAnalysisJob *d_jobs = (const SvtHipAnalysisJob *)stage_descriptors(jobs, (size_t)n_jobs * sizeof(SvtHipAnalysisJob), st);
    if (!d_jobs)
        return SVT_HIP_ERR_RUNTIME;
    if (hme_level1_enabled) {
        hipLaunchKernelGGL(downsample_pad_batch_kernel, dim3((max_q_stride / 4 + 1 + 255) / 256, (max_q_rows + BATCH_ROWS - 1) / BATCH_ROWS, n_jobs), dim3(256), 0, st, d_jobs, 0);
        hipLaunchKernelGGL(downsample_pad_batch_kernel, dim3((max_s_stride / 4 + 1 + 255) / 256, (max_s_rows + BATCH_ROWS - 1) / BATCH_ROWS, n_jobs), dim3(256), 0, st, d_jobs, 1);
    } else {
        hipLaunchKernelGGL(downsample_pad_batch_kernel, dim3((max_s_stride / 4 + 1 + 255) / 256, (max_s_rows + BATCH_ROWS - 1) / BATCH_ROWS, n_jobs), dim3(256), 0, st, d_jobs, 2);
    }
    hipLaunchKernelGGL(variance_batch_kernel, dim3((max_nb + 3) / 4, n_jobs), dim3(256), 0, st, d_jobs, (int)full_precision);
    stage_commit(st);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Tier A (host pointers): stage the touched bytes, run the same kernels, copy back.
// ------------------------------------------------------------------------------------------------
static void fatal_if(bool bad, const char *what) {
    if (bad) {
        svthip::tier_a_throw("%s: %s", what, svt_hip_last_error());
    }
}

// Generic strided decimation used by the per-call entry point (no padding involved).
__global__ __launch_bounds__(256) static void downsample_plain_kernel(const uint8_t *__restrict__ in, uint32_t in_stride,
                                                                       uint32_t in_w, uint32_t in_h,
                                                                       uint8_t *__restrict__ out, uint32_t out_stride,
                                                                       uint32_t step) {
    const uint32_t ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
    const uint32_t half = step >> 1, x = half + ox * step, y = half + oy * step;
    if (x >= in_w || y >= in_h)
        return;
    const uint8_t *p = in + (size_t)y * in_stride + x;
    const uint32_t s = (uint32_t)p[-(ptrdiff_t)in_stride - 1] + p[-(ptrdiff_t)in_stride] + p[-1] + p[0];
    out[(size_t)oy * out_stride + ox] = (uint8_t)((s + 2) >> 2);
}

static void svt_aom_downsample_2d_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height, uint8_t *decim_samples, uint32_t decim_stride, uint32_t decim_step);
extern "C" void svt_aom_downsample_2d_hip(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height, uint8_t *decim_samples, uint32_t decim_stride, uint32_t decim_step) { TIER_A_CALL(svt_aom_downsample_2d, svt_aom_downsample_2d_hip_impl(input_samples, input_stride, input_area_width, input_area_height, decim_samples, decim_stride, decim_step), (input_samples, input_stride, input_area_width, input_area_height, decim_samples, decim_stride, decim_step)); }
static void svt_aom_downsample_2d_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height, uint8_t *decim_samples, uint32_t decim_stride, uint32_t decim_step) {
    const uint32_t half = decim_step >> 1;
    if (decim_step == 0 || input_area_width <= half || input_area_height <= half)
        return;
    fatal_if(!ensure_init(), "svt_aom_downsample_2d_hip");
    const uint32_t ow = (input_area_width - half + decim_step - 1) / decim_step;
    const uint32_t oh = (input_area_height - half + decim_step - 1) / decim_step;
    // reads start one row / one column before (half,half): rows half-1 .., columns half-1 ..
    const size_t   in_first = (size_t)(half - 1) * input_stride + (half - 1);
    const size_t   in_span  = (size_t)(input_area_height - half) * input_stride + (input_area_width - half + 1);
    hipStream_t    st = resolve_stream(nullptr);
    Scratch       &sc = tls_scratch();
    const size_t   off_out = (in_span + 511) / 256 * 256, out_span = (size_t)(oh - 1) * decim_stride + ow;
    uint8_t       *d = sc.device(off_out + out_span + 256), *h = sc.host(off_out + out_span + 256);
    memcpy(h, input_samples + in_first, in_span);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, in_span, hipMemcpyHostToDevice, st));
    // device pointer equivalent of `input_samples`: d - in_first (never dereferenced below in_first)
    const uint8_t *din = d - in_first;
    hipLaunchKernelGGL(downsample_plain_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, st, din, input_stride,
                       input_area_width, input_area_height, d + off_out, decim_stride, decim_step);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_out, d + off_out, out_span, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    for (uint32_t y = 0; y < oh; y++) memcpy(decim_samples + (size_t)y * decim_stride, h + off_out + (size_t)y * decim_stride, ow);
}

// 8x8 block statistics: n8 side-by-side 8x8 blocks, (sum, sum of squares) over rows 0..7 step `rstep`.
__global__ __launch_bounds__(64) static void block8_stats_kernel(const uint8_t *__restrict__ in, uint32_t stride,
                                                                 uint32_t n8, uint32_t w, uint32_t hrows, uint32_t rstep,
                                                                 uint32_t *__restrict__ out) {
    const uint32_t k = threadIdx.x;
    if (k >= n8)
        return;
    uint32_t s = 0, ss = 0;
    for (uint32_t r = 0; r < hrows; r += rstep)
        for (uint32_t c = 0; c < w; c++) {
            const uint32_t v = in[(size_t)r * stride + w * k + c];
            s += v, ss += v * v;
        }
    out[2 * k] = s, out[2 * k + 1] = ss;
}

static void block_stats_host(const uint8_t *in, uint32_t stride, uint32_t n8, uint32_t w, uint32_t hrows, uint32_t rstep,
                             uint32_t *res /* 2*n8 */) {
    fatal_if(!ensure_init(), "block statistics");
    hipStream_t  st   = resolve_stream(nullptr);
    Scratch     &sc   = tls_scratch();
    const size_t span = (size_t)(hrows - 1) * stride + (size_t)w * n8, off = (span + 511) / 256 * 256;
    uint8_t     *d = sc.device(off + 256), *h = sc.host(off + 256);
    memcpy(h, in, span);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, span, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(block8_stats_kernel, dim3(1), dim3(64), 0, st, d, stride, n8, w, hrows, rstep, (uint32_t *)(d + off));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off, d + off, 8 * n8, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(res, h + off, 8 * n8);
}

static void svt_compute_interm_var_four8x8_hip_impl(uint8_t *input_samples, uint16_t input_stride, uint64_t *mean_of8x8_blocks, uint64_t *mean_of_squared8x8_blocks);
extern "C" void svt_compute_interm_var_four8x8_hip(uint8_t *input_samples, uint16_t input_stride, uint64_t *mean_of8x8_blocks, uint64_t *mean_of_squared8x8_blocks) { TIER_A_CALL(svt_compute_interm_var_four8x8, svt_compute_interm_var_four8x8_hip_impl(input_samples, input_stride, mean_of8x8_blocks, mean_of_squared8x8_blocks), (input_samples, input_stride, mean_of8x8_blocks, mean_of_squared8x8_blocks)); }
static void svt_compute_interm_var_four8x8_hip_impl(uint8_t *input_samples, uint16_t input_stride, uint64_t *mean_of8x8_blocks, uint64_t *mean_of_squared8x8_blocks) {
    uint32_t r[8];
    block_stats_host(input_samples, input_stride, 4, 8, 8, 2, r);
    for (int k = 0; k < 4; k++) {
        mean_of8x8_blocks[k]         = (uint64_t)r[2 * k] << 3;
        mean_of_squared8x8_blocks[k] = (uint64_t)r[2 * k + 1] << 11;
    }
}
static uint64_t svt_compute_sub_mean_8x8_hip_impl(uint8_t *input_samples, uint16_t input_stride);
extern "C" uint64_t svt_compute_sub_mean_8x8_hip(uint8_t *input_samples, uint16_t input_stride) { TIER_A_CALL(svt_compute_sub_mean_8x8, svt_compute_sub_mean_8x8_hip_impl(input_samples, input_stride), (input_samples, input_stride)); }
static uint64_t svt_compute_sub_mean_8x8_hip_impl(uint8_t *input_samples, uint16_t input_stride) {
    uint32_t r[2];
    block_stats_host(input_samples, input_stride, 1, 8, 8, 2, r);
    return (uint64_t)r[0] << 3;
}
static uint64_t svt_compute_mean_8x8_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height);
extern "C" uint64_t svt_compute_mean_8x8_hip(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height) { TIER_A_CALL(svt_compute_mean_8x8, svt_compute_mean_8x8_hip_impl(input_samples, input_stride, input_area_width, input_area_height), (input_samples, input_stride, input_area_width, input_area_height)); }
static uint64_t svt_compute_mean_8x8_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height) {
    uint32_t r[2];
    block_stats_host(input_samples, input_stride, 1, input_area_width, input_area_height, 1, r);
    return ((uint64_t)r[0] << 8) / (input_area_width * input_area_height);
}
static uint64_t svt_compute_mean_square_values_8x8_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height);
extern "C" uint64_t svt_compute_mean_square_values_8x8_hip(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height) { TIER_A_CALL(svt_compute_mean_square_values_8x8, svt_compute_mean_square_values_8x8_hip_impl(input_samples, input_stride, input_area_width, input_area_height), (input_samples, input_stride, input_area_width, input_area_height)); }
static uint64_t svt_compute_mean_square_values_8x8_hip_impl(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height) {
    uint32_t r[2];
    block_stats_host(input_samples, input_stride, 1, input_area_width, input_area_height, 1, r);
    return ((uint64_t)r[1] << 16) / (input_area_width * input_area_height);
}

SVT_HIP_MODULE_WARMUP(pyramid)
