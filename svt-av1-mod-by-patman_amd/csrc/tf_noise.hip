// tf_noise.hip — the temporal filter's noise estimate on gfx950: svt_estimate_noise_fp16 / svt_estimate_noise_highbd_fp16
// (aom_dsp_rtcd.h:874-877; temporal_filtering.c:3668-3736).  A 3x3 stencil over the interior of one plane: Sobel gradients
// pick the smooth samples, the mean absolute Laplacian of those gives the noise level in 16.16 fixed point.  HBM-bound:
// every sample is read once from memory (the 3x3 neighbourhood of a thread's 4 x ROWS strip comes from its own loads and
// the caches), the two sums are reduced by wave shuffles and one 64-bit atomic per wave.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_tf.h"
#include "common.hpp"

using namespace svthip;

namespace {

constexpr int ROWS = 4;                   // interior rows per thread strip
constexpr int COLS = 4;                   // interior columns per thread
constexpr int EDGE_THRESHOLD = 50, SMOOTH_THRESHOLD = 16, SQRT_PI_BY_2_FP16 = 82137;   // temporal_filtering.h:26-29

// All (ROWS + 2) x (COLS + 2) samples of a thread's strip are requested before the first one is used (row and column
// indices clamped into the plane instead of branching), so one round of memory latency covers the whole strip.
template <class PIX>
__global__ __launch_bounds__(256) void noise_sums_kernel(const PIX *__restrict__ src_, int width, int height, int stride, int shift,
                                                         SvtHipTfNoise *__restrict__ out) {
    const __attribute__((address_space(1))) PIX *src = (const __attribute__((address_space(1))) PIX *)src_;
    const int tiles_x = (width - 2 + 64 * COLS - 1) / (64 * COLS), tiles_y = (height - 2 + 4 * ROWS - 1) / (4 * ROWS);
    uint32_t  sum = 0, num = 0;   // 16 samples per strip, each term <= 8 * 255 after the depth shift: 32 bits hold a wave's total
    for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int j0 = 1 + (tx * 64 + (threadIdx.x & 63)) * COLS;               // first interior column of this thread
        const int i0 = 1 + (ty * 4 + (threadIdx.x >> 6)) * ROWS;                // first interior row of its wave
        if (j0 < width - 1 && i0 < height - 1) {
            int v[ROWS + 2][COLS + 2];
#pragma unroll
            for (int y = 0; y < ROWS + 2; y++) {
                const size_t row = (size_t)min(i0 - 1 + y, height - 1) * stride;
#pragma unroll
                for (int x = 0; x < COLS + 2; x++) v[y][x] = src[row + min(j0 - 1 + x, width - 1)];
            }
            const int rnd = shift ? 1 << (shift - 1) : 0;
#pragma unroll
            for (int y = 1; y <= ROWS; y++)
#pragma unroll
                for (int x = 1; x <= COLS; x++) {
                    const int gx = (v[y - 1][x - 1] - v[y - 1][x + 1]) + (v[y + 1][x - 1] - v[y + 1][x + 1]) + 2 * (v[y][x - 1] - v[y][x + 1]);
                    const int gy = (v[y - 1][x - 1] - v[y + 1][x - 1]) + (v[y - 1][x + 1] - v[y + 1][x + 1]) + 2 * (v[y - 1][x] - v[y + 1][x]);
                    const int ga = (abs(gx) + abs(gy) + rnd) >> shift;
                    const int lap = 4 * v[y][x] - 2 * (v[y][x - 1] + v[y][x + 1] + v[y - 1][x] + v[y + 1][x]) +
                        (v[y - 1][x - 1] + v[y - 1][x + 1] + v[y + 1][x - 1] + v[y + 1][x + 1]);
                    const bool in = i0 + y - 1 < height - 1 && j0 + x - 1 < width - 1 && ga < EDGE_THRESHOLD;
                    sum += in ? (uint32_t)((abs(lap) + rnd) >> shift) : 0u, num += in ? 1u : 0u;
                }
        }
    }
    // one pair of atomics per workgroup: waves by shuffles, the four waves through LDS
    __shared__ uint32_t part[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64), num += __shfl_xor(num, off, 64);
    if ((threadIdx.x & 63) == 0)
        part[0][threadIdx.x >> 6] = sum, part[1][threadIdx.x >> 6] = num;
    __syncthreads();
    if (threadIdx.x == 0 && (part[1][0] | part[1][1] | part[1][2] | part[1][3])) {
        atomicAdd((unsigned long long *)&out->sum, (unsigned long long)part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd((unsigned long long *)&out->num, (unsigned long long)part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

__global__ void noise_finish_kernel(SvtHipTfNoise *out) {
    const int64_t sum = (int64_t)out->sum, num = (int64_t)out->num;
    out->noise_fp16   = num < SMOOTH_THRESHOLD ? -65536 : (int32_t)((sum * SQRT_PI_BY_2_FP16) / (6 * num));
}

int32_t launch(const void *d_src, uint32_t width, uint32_t height, uint32_t stride, int is16, int bd, SvtHipTfNoise *d_out, hipStream_t st) {
    SVT_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(SvtHipTfNoise), st));
    if (width > 2 && height > 2) {
        const uint32_t tiles = ((width - 2 + 64 * COLS - 1) / (64 * COLS)) * ((height - 2 + 4 * ROWS - 1) / (4 * ROWS));
        const dim3     grid(tiles < 1024 ? tiles : 1024);
        if (is16)
            hipLaunchKernelGGL(noise_sums_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t *)d_src, (int)width, (int)height, (int)stride,
                               bd - 8, d_out);
        else
            hipLaunchKernelGGL(noise_sums_kernel<uint8_t>, grid, dim3(256), 0, st, (const uint8_t *)d_src, (int)width, (int)height, (int)stride, 0,
                               d_out);
        SVT_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(noise_finish_kernel, dim3(1), dim3(1), 0, st, d_out);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

int32_t tier_a(const void *src, int width, int height, int stride, int is16, int bd, const char *what) {
    if (width < 1 || height < 1 || stride < width || (is16 && (bd < 8 || bd > 16))) {
        set_error("%s: bad plane geometry %dx%d stride %d depth %d", what, width, height, stride, bd);
        svthip::tier_a_throw("%s", svt_hip_last_error());
    }
    if (!ensure_init()) {
        svthip::tier_a_throw("%s: %s", what, svt_hip_last_error());
    }
    const size_t px = is16 ? 2 : 1, span = ((size_t)(height - 1) * stride + width) * px, off_out = (span + 255) / 256 * 256;
    Scratch     &sc = tls_scratch();
    uint8_t     *h = sc.host(off_out + 256), *d = sc.device(off_out + 256);
    memcpy(h, src, span);
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, span, hipMemcpyHostToDevice, st));
    if (launch(d, (uint32_t)width, (uint32_t)height, (uint32_t)stride, is16, bd, (SvtHipTfNoise *)(d + off_out), st) != SVT_HIP_OK) {
        svthip::tier_a_throw("%s: %s", what, svt_hip_last_error());
    }
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_out, d + off_out, sizeof(SvtHipTfNoise), hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return ((const SvtHipTfNoise *)(h + off_out))->noise_fp16;
}

}  // namespace

extern "C" int32_t svt_hip_tf_estimate_noise(const void *d_src, uint32_t width, uint32_t height, uint32_t stride, int32_t is_16bit,
                                             int32_t bit_depth, SvtHipTfNoise *d_out, void *stream) {
    if (!d_src || !d_out || width == 0 || height == 0 || stride < width || (is_16bit && (bit_depth < 8 || bit_depth > 16))) {
        set_error("svt_hip_tf_estimate_noise: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    return launch(d_src, width, height, stride, is_16bit, bit_depth, d_out, resolve_stream(stream));
}

static int32_t svt_estimate_noise_fp16_hip_impl(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y);
extern "C" int32_t svt_estimate_noise_fp16_hip(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y) { TIER_A_CALL(svt_estimate_noise_fp16, svt_estimate_noise_fp16_hip_impl(src, width, height, stride_y), (src, width, height, stride_y)); }
static int32_t svt_estimate_noise_fp16_hip_impl(const uint8_t *src, uint16_t width, uint16_t height, uint16_t stride_y) {
    return tier_a(src, width, height, stride_y, 0, 8, "svt_estimate_noise_fp16");
}
static int32_t svt_estimate_noise_highbd_fp16_hip_impl(const uint16_t *src, int width, int height, int stride, int bd);
extern "C" int32_t svt_estimate_noise_highbd_fp16_hip(const uint16_t *src, int width, int height, int stride, int bd) { TIER_A_CALL(svt_estimate_noise_highbd_fp16, svt_estimate_noise_highbd_fp16_hip_impl(src, width, height, stride, bd), (src, width, height, stride, bd)); }
static int32_t svt_estimate_noise_highbd_fp16_hip_impl(const uint16_t *src, int width, int height, int stride, int bd) {
    return tier_a(src, width, height, stride, 1, bd, "svt_estimate_noise_highbd_fp16");
}

SVT_HIP_MODULE_WARMUP(tf_noise)
