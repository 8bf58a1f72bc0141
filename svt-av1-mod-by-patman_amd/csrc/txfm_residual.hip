// txfm_residual.hip — the per-call (Tier A) forms of the residual producer and the transform-domain cost that sit either
// side of the forward transform in the TPL dispenser and in mode decision:
//   svt_aom_subtract_block / svt_aom_highbd_subtract_block   (reference: inter_prediction.c:35-60)
//   svt_aom_satd                                              (reference: common_dsp_rtcd.c:71-78)
// Host pointers in, host pointers out, ABI of the reference's RTCD slots (common_dsp_rtcd.h:234-237, aom_dsp_rtcd.h:206-207).
// The batched path never calls these: txfm_kernel forms the residual from source and prediction itself
// (SVT_HIP_TX_SRC_PRED) and returns the cost in its result record (SVT_HIP_TX_SATD).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/svt_hip_txfm.h"
#include "common.hpp"

using namespace svthip;

namespace {

// diff[r][c] = src[r][c] - pred[r][c], stored as int16 (wraps like the reference's assignment)
template <class PIX>
__global__ __launch_bounds__(256) void subtract_kernel(int rows, int cols, int16_t *__restrict__ diff, int diff_stride,
                                                       const PIX *__restrict__ src, int src_stride, const PIX *__restrict__ pred,
                                                       int pred_stride) {
    const int n = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        diff[(size_t)r * diff_stride + c] = (int16_t)((int)src[(size_t)r * src_stride + c] - (int)pred[(size_t)r * pred_stride + c]);
    }
}

// sum of |coeff[i]| in int arithmetic (wraps like the reference's int accumulator)
__global__ __launch_bounds__(256) void satd_kernel(const int32_t *__restrict__ coeff, int length, int32_t *__restrict__ out) {
    __shared__ uint32_t part[4];
    uint32_t            acc = 0;
    for (int i = threadIdx.x; i < length; i += 256) {
        const int32_t v = coeff[i];
        acc += (uint32_t)(v < 0 ? 0u - (uint32_t)v : (uint32_t)v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        *out = (int32_t)(part[0] + part[1] + part[2] + part[3]);
}

// {sum (coeff - recon)^2, sum coeff^2} over an area (recon == nullptr: the cbf-zero form, both entries = sum coeff^2)
__global__ __launch_bounds__(256) void full_distortion_kernel(const int32_t *__restrict__ coeff, int cs, const int32_t *__restrict__ recon,
                                                              int rs, int w, int h, uint64_t *__restrict__ out) {
    __shared__ uint64_t part[2][4];
    uint64_t            a = 0, b = 0;
    for (int i = threadIdx.x; i < w * h; i += 256) {
        const int     r = i / w, c = i - r * w;
        const int64_t v = coeff[(size_t)r * cs + c], e = v - (recon ? (int64_t)recon[(size_t)r * rs + c] : 0);
        a += (uint64_t)(e * e), b += (uint64_t)(v * v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64), b += __shfl_xor(b, off, 64);
    if ((threadIdx.x & 63) == 0)
        part[0][threadIdx.x >> 6] = a, part[1][threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0)
        out[0] = part[0][0] + part[0][1] + part[0][2] + part[0][3], out[1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
}

// one wave per transform block: distortion of the stored coefficient / de-quantised arrays (row pitch = retained width)
__global__ __launch_bounds__(256) void distortion_batch_kernel(const uint8_t *__restrict__ base, const SvtHipTxfmDesc *__restrict__ descs,
                                                               uint64_t (*__restrict__ results)[2], uint32_t n, int iw, int ih) {
    const uint32_t tb = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tb >= n)
        return;
    const SvtHipTxfmDesc &d = descs[tb];
    uint64_t              a = 0, b = 0;
    if (d.coeff_off != SVT_HIP_NO_OFFSET && d.dqcoeff_off != SVT_HIP_NO_OFFSET && !(d.flags & SVT_HIP_TX_FULLCOEFF)) {
        const int32_t *co = (const int32_t *)(base + d.coeff_off), *dq = (const int32_t *)(base + d.dqcoeff_off);
        const int      dw = d.dist_w ? (d.dist_w < iw ? d.dist_w : iw) : iw, dh = d.dist_h ? (d.dist_h < ih ? d.dist_h : ih) : ih;
        for (int i = lane; i < dw * dh; i += 64) {
            const int     r = i / dw, c = i - r * dw;
            const int64_t v = co[r * iw + c], e = v - (int64_t)dq[r * iw + c];
            a += (uint64_t)(e * e), b += (uint64_t)(v * v);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64), b += __shfl_xor(b, off, 64);
    if (lane == 0)
        results[tb][0] = a, results[tb][1] = b;
}

[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
// sum (a - b)^2 over a w x h area of pixels (svt_spatial_full_distortion_kernel / svt_full_distortion_kernel16_bits)
template <class PIX>
__global__ __launch_bounds__(256) void spatial_sse_kernel(const PIX *__restrict__ a, int as, const PIX *__restrict__ b, int bs, int w, int h,
                                                          unsigned long long *__restrict__ out) {
    uint64_t acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < w * h; i += gridDim.x * 256) {
        const int     r = i / w, c = i - r * w;
        const int64_t d = (int64_t)a[(size_t)r * as + c] - (int64_t)b[(size_t)r * bs + c];
        acc += (uint64_t)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, (unsigned long long)acc);
}

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

template <class PIX>
uint64_t spatial_sse_tier_a(const PIX *a, uint32_t as, const PIX *b, uint32_t bs, uint32_t w, uint32_t h, const char *what) {
    if (!w || !h)
        return 0;
    if (!ensure_init())
        fatal(what);
    const size_t pix = (size_t)w * h, pb = up256(pix * sizeof(PIX));
    Scratch     &sc = tls_scratch();
    uint8_t     *hh = sc.host(2 * pb + 256), *d = sc.device(2 * pb + 256);
    for (uint32_t r = 0; r < h; r++) {
        memcpy(hh + (size_t)r * w * sizeof(PIX), a + (size_t)r * as, (size_t)w * sizeof(PIX));
        memcpy(hh + pb + (size_t)r * w * sizeof(PIX), b + (size_t)r * bs, (size_t)w * sizeof(PIX));
    }
    memset(hh + 2 * pb, 0, 8);
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, hh, 2 * pb + 8, hipMemcpyHostToDevice, st));
    const int blocks = (int)((pix + 255) / 256);
    hipLaunchKernelGGL((spatial_sse_kernel<PIX>), dim3(blocks < 256 ? blocks : 256), dim3(256), 0, st, (const PIX *)d, (int)w,
                       (const PIX *)(d + pb), (int)w, (int)w, (int)h, (unsigned long long *)(d + 2 * pb));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + 2 * pb, d + 2 * pb, 8, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return *(const uint64_t *)(hh + 2 * pb);
}

template <class PIX>
void subtract_tier_a(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const PIX *src, ptrdiff_t src_stride, const PIX *pred,
                     ptrdiff_t pred_stride) {
    if (rows <= 0 || cols <= 0)
        return;
    if (!ensure_init())
        fatal("subtract_block");
    // dense staging: rows x cols of each operand
    const size_t pix = (size_t)rows * cols, pb = up256(pix * sizeof(PIX)), db = up256(pix * 2);
    Scratch     &sc  = tls_scratch();
    uint8_t     *h = sc.host(2 * pb + db), *d = sc.device(2 * pb + db);
    for (int r = 0; r < rows; r++) {
        memcpy(h + (size_t)r * cols * sizeof(PIX), src + (ptrdiff_t)r * src_stride, (size_t)cols * sizeof(PIX));
        memcpy(h + pb + (size_t)r * cols * sizeof(PIX), pred + (ptrdiff_t)r * pred_stride, (size_t)cols * sizeof(PIX));
    }
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, 2 * pb, hipMemcpyHostToDevice, st));
    const int blocks = (int)((pix + 255) / 256);
    hipLaunchKernelGGL((subtract_kernel<PIX>), dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, st, rows, cols, (int16_t *)(d + 2 * pb),
                       cols, (const PIX *)d, cols, (const PIX *)(d + pb), cols);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + 2 * pb, d + 2 * pb, pix * 2, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    for (int r = 0; r < rows; r++) memcpy(diff + (ptrdiff_t)r * diff_stride, h + 2 * pb + (size_t)r * cols * 2, (size_t)cols * 2);
}

}  // namespace

static void svt_aom_subtract_block_hip_impl(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride);
extern "C" void svt_aom_subtract_block_hip(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride) { TIER_A_CALL(svt_aom_subtract_block, svt_aom_subtract_block_hip_impl(rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride), (rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride)); }
static void svt_aom_subtract_block_hip_impl(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride) {
    subtract_tier_a<uint8_t>(rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride);
}
static void svt_aom_highbd_subtract_block_hip_impl(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride, int bd);
extern "C" void svt_aom_highbd_subtract_block_hip(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride, int bd) { TIER_A_CALL(svt_aom_highbd_subtract_block, svt_aom_highbd_subtract_block_hip_impl(rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride, bd), (rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride, bd)); }
static void svt_aom_highbd_subtract_block_hip_impl(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride, const uint8_t *pred_ptr, ptrdiff_t pred_stride, int bd) {
    (void)bd;
    subtract_tier_a<uint16_t>(rows, cols, diff_ptr, diff_stride, (const uint16_t *)src_ptr, src_stride, (const uint16_t *)pred_ptr,
                              pred_stride);
}
// svt_residual_kernel8bit / 16bit (common_dsp_rtcd.h:163,174; pic_operators.c:101-143): the same difference with the
// operands in another order
static void svt_residual_kernel8bit_hip_impl(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
extern "C" void svt_residual_kernel8bit_hip(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_residual_kernel8bit, svt_residual_kernel8bit_hip_impl(input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height), (input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height)); }
static void svt_residual_kernel8bit_hip_impl(uint8_t *input, uint32_t input_stride, uint8_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height) {
    subtract_tier_a<uint8_t>((int)area_height, (int)area_width, residual, residual_stride, input, input_stride, pred, pred_stride);
}
static void svt_residual_kernel16bit_hip_impl(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height);
extern "C" void svt_residual_kernel16bit_hip(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_residual_kernel16bit, svt_residual_kernel16bit_hip_impl(input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height), (input, input_stride, pred, pred_stride, residual, residual_stride, area_width, area_height)); }
static void svt_residual_kernel16bit_hip_impl(uint16_t *input, uint32_t input_stride, uint16_t *pred, uint32_t pred_stride, int16_t *residual, uint32_t residual_stride, uint32_t area_width, uint32_t area_height) {
    subtract_tier_a<uint16_t>((int)area_height, (int)area_width, residual, residual_stride, input, input_stride, pred, pred_stride);
}
// Tier B: the same sum over two DEVICE planes (what picture_sse_calculations, deblocking_filter.c:716-834, asks of the two leaves after
// every trial of the deblocking level search): *d_out (device, 8 bytes) is zeroed by the call and holds the sum when the stream has run
extern "C" int32_t svt_hip_plane_sse(const void *d_a, uint32_t a_stride, const void *d_b, uint32_t b_stride, uint32_t width, uint32_t height, int32_t is_16bit,
                                     uint64_t *d_out, void *stream) {
    if (!d_a || !d_b || !d_out || !width || !height) {
        set_error("svt_hip_plane_sse: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t st = resolve_stream(stream);
    SVT_HIP_CHECK(hipMemsetAsync(d_out, 0, 8, st));
    const size_t pix = (size_t)width * height;
    const int    blocks = (int)((pix + 255) / 256 < 1024 ? (pix + 255) / 256 : 1024);
    if (is_16bit)
        hipLaunchKernelGGL((spatial_sse_kernel<uint16_t>), dim3(blocks), dim3(256), 0, st, (const uint16_t *)d_a, (int)a_stride, (const uint16_t *)d_b, (int)b_stride, (int)width,
                           (int)height, (unsigned long long *)d_out);
    else
        hipLaunchKernelGGL((spatial_sse_kernel<uint8_t>), dim3(blocks), dim3(256), 0, st, (const uint8_t *)d_a, (int)a_stride, (const uint8_t *)d_b, (int)b_stride, (int)width,
                           (int)height, (unsigned long long *)d_out);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// svt_spatial_full_distortion_kernel (common_dsp_rtcd.h:171; picture_operators_c.c:62-78) and svt_full_distortion_kernel16_bits
// (common_dsp_rtcd.h:173; pic_operators.c:174-196: byte pointers reinterpreted as 16-bit samples, offsets in samples)
static uint64_t svt_spatial_full_distortion_kernel_hip_impl(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset, uint32_t recon_stride, uint32_t area_width, uint32_t area_height);
extern "C" uint64_t svt_spatial_full_distortion_kernel_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset, uint32_t recon_stride, uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_spatial_full_distortion_kernel, svt_spatial_full_distortion_kernel_hip_impl(input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height), (input, input_offset, input_stride, recon, recon_offset, recon_stride, area_width, area_height)); }
static uint64_t svt_spatial_full_distortion_kernel_hip_impl(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *recon, int32_t recon_offset, uint32_t recon_stride, uint32_t area_width, uint32_t area_height) {
    return spatial_sse_tier_a<uint8_t>(input + input_offset, input_stride, recon + recon_offset, recon_stride, area_width, area_height,
                                       "svt_spatial_full_distortion_kernel");
}
static uint64_t svt_full_distortion_kernel16_bits_hip_impl(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *pred, int32_t pred_offset, uint32_t pred_stride, uint32_t area_width, uint32_t area_height);
extern "C" uint64_t svt_full_distortion_kernel16_bits_hip(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *pred, int32_t pred_offset, uint32_t pred_stride, uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_full_distortion_kernel16_bits, svt_full_distortion_kernel16_bits_hip_impl(input, input_offset, input_stride, pred, pred_offset, pred_stride, area_width, area_height), (input, input_offset, input_stride, pred, pred_offset, pred_stride, area_width, area_height)); }
static uint64_t svt_full_distortion_kernel16_bits_hip_impl(uint8_t *input, uint32_t input_offset, uint32_t input_stride, uint8_t *pred, int32_t pred_offset, uint32_t pred_stride, uint32_t area_width, uint32_t area_height) {
    return spatial_sse_tier_a<uint16_t>((const uint16_t *)input + input_offset, input_stride, (const uint16_t *)pred + pred_offset,
                                        pred_stride, area_width, area_height, "svt_full_distortion_kernel16_bits");
}
static int svt_aom_satd_hip_impl(const int32_t *coeff, int length);
extern "C" int svt_aom_satd_hip(const int32_t *coeff, int length) { TIER_A_CALL(svt_aom_satd, svt_aom_satd_hip_impl(coeff, length), (coeff, length)); }
static int svt_aom_satd_hip_impl(const int32_t *coeff, int length) {
    if (length <= 0)
        return 0;
    if (!ensure_init())
        fatal("satd");
    const size_t cb = up256((size_t)length * 4);
    Scratch     &sc = tls_scratch();
    uint8_t     *h = sc.host(cb + 256), *d = sc.device(cb + 256);
    memcpy(h, coeff, (size_t)length * 4);
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, (size_t)length * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(satd_kernel, dim3(1), dim3(256), 0, st, (const int32_t *)d, length, (int32_t *)(d + cb));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + cb, d + cb, 4, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return *(const int32_t *)(h + cb);
}

namespace {
void distortion_tier_a(const int32_t *coeff, uint32_t cs, const int32_t *recon, uint32_t rs, uint64_t out[2], uint32_t w, uint32_t h) {
    out[0] = out[1] = 0;
    if (!w || !h)
        return;
    if (!ensure_init())
        fatal("full_distortion");
    const size_t n = (size_t)w * h, cb = up256(n * 4);
    Scratch     &sc = tls_scratch();
    uint8_t     *hh = sc.host(2 * cb + 256), *d = sc.device(2 * cb + 256);
    for (uint32_t r = 0; r < h; r++) {
        memcpy(hh + (size_t)r * w * 4, coeff + (size_t)r * cs, (size_t)w * 4);
        if (recon)
            memcpy(hh + cb + (size_t)r * w * 4, recon + (size_t)r * rs, (size_t)w * 4);
    }
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, hh, recon ? 2 * cb : cb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(full_distortion_kernel, dim3(1), dim3(256), 0, st, (const int32_t *)d, (int)w, recon ? (const int32_t *)(d + cb) : nullptr,
                       (int)w, (int)w, (int)h, (uint64_t *)(d + 2 * cb));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + 2 * cb, d + 2 * cb, 16, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(out, hh + 2 * cb, 16);
}
}  // namespace

static void svt_full_distortion_kernel32_bits_hip_impl(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height);
extern "C" void svt_full_distortion_kernel32_bits_hip(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_full_distortion_kernel32_bits, svt_full_distortion_kernel32_bits_hip_impl(coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height), (coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height)); }
static void svt_full_distortion_kernel32_bits_hip_impl(int32_t *coeff, uint32_t coeff_stride, int32_t *recon_coeff, uint32_t recon_coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height) {
    distortion_tier_a(coeff, coeff_stride, recon_coeff, recon_coeff_stride, distortion_result, area_width, area_height);
}
static void svt_full_distortion_kernel_cbf_zero32_bits_hip_impl(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height);
extern "C" void svt_full_distortion_kernel_cbf_zero32_bits_hip(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height) { TIER_A_CALL(svt_full_distortion_kernel_cbf_zero32_bits, svt_full_distortion_kernel_cbf_zero32_bits_hip_impl(coeff, coeff_stride, distortion_result, area_width, area_height), (coeff, coeff_stride, distortion_result, area_width, area_height)); }
static void svt_full_distortion_kernel_cbf_zero32_bits_hip_impl(int32_t *coeff, uint32_t coeff_stride, uint64_t distortion_result[2], uint32_t area_width, uint32_t area_height) {
    distortion_tier_a(coeff, coeff_stride, nullptr, 0, distortion_result, area_width, area_height);
}

extern "C" int32_t svt_hip_txfm_distortion_batch(const uint8_t *d_base, const SvtHipTxfmDesc *d_desc, uint64_t (*d_result)[2],
                                                 uint32_t n_blocks, uint32_t w, uint32_t h, void *stream) {
    const bool ok_w = w == 4 || w == 8 || w == 16 || w == 32 || w == 64, ok_h = h == 4 || h == 8 || h == 16 || h == 32 || h == 64;
    if (!d_base || !d_desc || !d_result || !ok_w || !ok_h) {
        set_error("svt_hip_txfm_distortion_batch: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    if (n_blocks == 0)
        return SVT_HIP_OK;
    hipLaunchKernelGGL(distortion_batch_kernel, dim3((n_blocks + 3) / 4), dim3(256), 0, resolve_stream(stream), d_base, d_desc, d_result,
                       n_blocks, (int)(w < 32 ? w : 32), (int)(h < 32 ? h : 32));
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

SVT_HIP_MODULE_WARMUP(txfm_residual)
