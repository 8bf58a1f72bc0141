// me_kernels.hip — leaf SAD kernels: batched svt_sad_loop_kernel (Tier B) and the ABI-identical
// per-call entry points (Tier A) of include/svt_hip_me.h.  gfx950 only.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "sad_device.hpp"

using namespace svthip;
using namespace svthip::dev;

namespace {

constexpr uint32_t BATCH_WIN_DW = 12288;  // 48 KiB window buffer
constexpr uint32_t BATCH_SRC_DW = 4096;   // 16 KiB: up to 128x128 source block

// One workgroup per descriptor.
__global__ __launch_bounds__(WG_THREADS) void sad_loop_batch_kernel(const uint8_t *__restrict__ base,
                                                                    const SvtHipSadLoopDesc *__restrict__ descs,
                                                                    SvtHipSadLoopResult *__restrict__ results,
                                                                    uint32_t n_desc) {
    __shared__ SearchShared sh;
    __shared__ uint32_t     win[BATCH_WIN_DW];
    __shared__ uint32_t     src[BATCH_SRC_DW];
    for (uint32_t di = blockIdx.x; di < n_desc; di += gridDim.x) {
        const SvtHipSadLoopDesc d  = descs[di];
        const uint32_t          bw = d.block_width, bh = d.block_height;
        const uint32_t          row_dw = (bw + 3) >> 2;
        const bool src_fits = row_dw * bh <= BATCH_SRC_DW && bw > 0 && bh > 0;
        if (threadIdx.x == 0) {
            SearchDesc &s = sh.desc[0];
            s.ref         = base + d.ref_off;
            s.ref_stride  = d.ref_stride;
            s.raw_stride  = d.src_stride_raw;
            s.sa_w        = src_fits ? d.search_area_width : 0;
            s.sa_h        = src_fits ? d.search_area_height : 0;
            s.skip        = d.skip_search_line && bw == 16 && bh <= 16;
        }
        if (src_fits)
            wg_stage_block(src, row_dw, base + d.src_off, d.src_stride, bw, bh);
        __syncthreads();
        wg_multi_search(sh, 1, src, row_dw, bw, bh, win, BATCH_WIN_DW);
        if (threadIdx.x == 0) {
            const uint64_t      key = sh.best[0];
            SvtHipSadLoopResult r;
            r.best_sad = key >> 32;
            r.pad_     = 0;
            if (key == KEY_NONE) {
                r.x = r.y = (int16_t)0x7fff;  // "not found": Tier A leaves the caller's x/y untouched
            } else {
                const uint32_t idx = (uint32_t)key;
                r.x                = (int16_t)(idx & 0xffffu);  // row << 16 | column
                r.y                = (int16_t)(idx >> 16);
            }
            results[di] = r;
        }
        __syncthreads();
    }
}

// Plain N x M SAD of one block (svt_nxm_sad_kernel): one workgroup, dword rows from global.
__global__ __launch_bounds__(WG_THREADS) void nxm_sad_kernel(const uint8_t *__restrict__ src, uint32_t ss,
                                                             const uint8_t *__restrict__ ref, uint32_t rs, uint32_t h,
                                                             uint32_t w, uint32_t *__restrict__ out) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0)
        total = 0;
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t idx = threadIdx.x; idx < w * h; idx += WG_THREADS) {
        const uint32_t r = idx / w, c = idx - r * w;
        const int      d = (int)src[(size_t)r * ss + c] - (int)ref[(size_t)r * rs + c];
        acc += (uint32_t)(d < 0 ? -d : d);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&total, acc);
    __syncthreads();
    if (threadIdx.x == 0)
        *out = total;
}

// svt_ext_all_sad_calculation_8x8_16x16 + svt_ext_eight_sad_calculation_32x32_64x64 for one call:
// 64 8x8 SADs at 8 x-positions, then the sequential best updates exactly in the reference's order
// (motion_estimation.c:210-425).  blk layout: p_best_* arrays in z-order as passed by the caller.
struct ExtAllIo {
    uint32_t best8[64], best16[16], mv8[64], mv16[16];
    uint32_t eight16[16][8];
};
__global__ __launch_bounds__(WG_THREADS) void ext_all_sad_kernel(const uint8_t *__restrict__ src, uint32_t ss,
                                                                 const uint8_t *__restrict__ ref, uint32_t rs,
                                                                 uint32_t mv, uint32_t sub_sad, ExtAllIo *__restrict__ io) {
    __shared__ uint32_t sad8[8][64];  // [position][z-order 8x8]
    const uint8_t z16[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
    for (uint32_t it = threadIdx.x; it < 512; it += WG_THREADS) {
        const uint32_t p = it & 7, b = it >> 3;  // b: raster 8x8 index (by*8+bx)
        const uint32_t by = b >> 3, bx = b & 7;
        const uint8_t *s = src + (size_t)(8 * by) * ss + 8 * bx;
        const uint8_t *r = ref + (size_t)(8 * by) * rs + 8 * bx + p;
        uint32_t       acc = 0;
        const uint32_t step = sub_sad ? 2 : 1;
        for (uint32_t y = 0; y < 8; y += step)
            for (uint32_t x = 0; x < 8; x++) {
                const int d = (int)s[(size_t)y * ss + x] - (int)r[(size_t)y * rs + x];
                acc += (uint32_t)(d < 0 ? -d : d);
            }
        if (sub_sad)
            acc <<= 1;
        const uint32_t z = 4 * z16[4 * (by >> 1) + (bx >> 1)] + 2 * (by & 1) + (bx & 1);
        sad8[p][z]       = acc;
    }
    __syncthreads();
    const int16_t mvx = (int16_t)(mv & 0xffff), mvy = (int16_t)(mv >> 16);
    if (threadIdx.x < 64) {  // 8x8 PUs
        const uint32_t z = threadIdx.x;
        uint32_t       b = io->best8[z], m = io->mv8[z];
        for (uint32_t p = 0; p < 8; p++)
            if (sad8[p][z] < b) {
                b = sad8[p][z];
                m = ((uint32_t)(uint16_t)mvy << 16) | (uint16_t)(int16_t)(mvx + (int16_t)p);
            }
        io->best8[z] = b, io->mv8[z] = m;
    } else if (threadIdx.x < 80) {  // 16x16 PUs
        const uint32_t z = threadIdx.x - 64;
        uint32_t       b = io->best16[z], m = io->mv16[z];
        for (uint32_t p = 0; p < 8; p++) {
            const uint32_t v = sad8[p][4 * z] + sad8[p][4 * z + 1] + sad8[p][4 * z + 2] + sad8[p][4 * z + 3];
            io->eight16[z][p] = v;
            if (v < b) {
                b = v;
                m = ((uint32_t)(uint16_t)mvy << 16) | (uint16_t)(int16_t)(mvx + (int16_t)p);
            }
        }
        io->best16[z] = b, io->mv16[z] = m;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Tier B
// ------------------------------------------------------------------------------------------------
extern "C" int32_t svt_hip_sad_loop_batch(const uint8_t *d_base, const SvtHipSadLoopDesc *d_desc,
                                          SvtHipSadLoopResult *d_result, uint32_t n_desc, void *stream) {
    if (!d_base || !d_desc || !d_result) {
        set_error("svt_hip_sad_loop_batch: NULL pointer");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    if (n_desc == 0)
        return SVT_HIP_OK;
    const uint32_t grid = n_desc < 65535u * 16u ? n_desc : 65535u * 16u;
    hipLaunchKernelGGL(sad_loop_batch_kernel, dim3(grid), dim3(WG_THREADS), 0, resolve_stream(stream), d_base, d_desc,
                       d_result, n_desc);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Tier A — host pointers in/out, one block per call.  Each call stages the byte span it touches.
// ------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void svt_sad_loop_kernel_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height);
extern "C" void svt_sad_loop_kernel_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) { TIER_A_CALL(svt_sad_loop_kernel, svt_sad_loop_kernel_hip_impl(src, src_stride, ref, ref_stride, block_height, block_width, best_sad, x_search_center, y_search_center, src_stride_raw, skip_search_line, search_area_width, search_area_height), (src, src_stride, ref, ref_stride, block_height, block_width, best_sad, x_search_center, y_search_center, src_stride_raw, skip_search_line, search_area_width, search_area_height)); }
static void svt_sad_loop_kernel_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center, uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) {
    *best_sad = 0xffffff;
    if (search_area_width <= 0 || search_area_height <= 0 || block_width == 0 || block_height == 0)
        return;
    if (!ensure_init()) 
        svthip::tier_a_throw("%s", svt_hip_last_error());
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t src_span = (size_t)(block_height - 1) * src_stride + block_width;
    const size_t ref_span = (size_t)(search_area_height - 1) * src_stride_raw + (size_t)(block_height - 1) * ref_stride +
        (size_t)search_area_width + block_width - 1;
    const size_t off_src = 0, off_ref = align_up(src_span + 64, 256);
    const size_t off_desc = off_ref + align_up(ref_span + 64, 256), off_res = off_desc + 256;
    const size_t total = off_res + 256;
    uint8_t     *d = sc.device(total), *h = sc.host(total);
    memcpy(h + off_src, src, src_span);
    memcpy(h + off_ref, ref, ref_span);
    SvtHipSadLoopDesc *dd = (SvtHipSadLoopDesc *)(h + off_desc);
    memset(dd, 0, sizeof(*dd));
    dd->src_off = off_src, dd->ref_off = off_ref;
    dd->src_stride = src_stride, dd->ref_stride = ref_stride, dd->src_stride_raw = src_stride_raw;
    dd->block_width = (uint16_t)block_width, dd->block_height = (uint16_t)block_height;
    dd->search_area_width = search_area_width, dd->search_area_height = search_area_height;
    dd->skip_search_line = skip_search_line;
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, off_res, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sad_loop_batch_kernel, dim3(1), dim3(WG_THREADS), 0, st, d, (const SvtHipSadLoopDesc *)(d + off_desc),
                       (SvtHipSadLoopResult *)(d + off_res), 1u);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_res, d + off_res, sizeof(SvtHipSadLoopResult), hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const SvtHipSadLoopResult *r = (const SvtHipSadLoopResult *)(h + off_res);
    *best_sad                    = r->best_sad;
    if (!(r->best_sad == 0xffffff && r->x == 0x7fff)) {
        *x_search_center = r->x;
        *y_search_center = r->y;
    }
}

static uint32_t svt_nxm_sad_kernel_hip_impl(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
extern "C" uint32_t svt_nxm_sad_kernel_hip(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) { TIER_A_CALL(svt_nxm_sad_kernel, svt_nxm_sad_kernel_hip_impl(src, src_stride, ref, ref_stride, height, width), (src, src_stride, ref, ref_stride, height, width)); }
static uint32_t svt_nxm_sad_kernel_hip_impl(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) {
    if (height == 0 || width == 0)
        return 0;
    if (!ensure_init()) 
        svthip::tier_a_throw("%s", svt_hip_last_error());
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t src_span = (size_t)(height - 1) * src_stride + width, ref_span = (size_t)(height - 1) * ref_stride + width;
    const size_t off_ref = align_up(src_span, 256), off_res = off_ref + align_up(ref_span, 256);
    uint8_t     *d = sc.device(off_res + 256), *h = sc.host(off_res + 256);
    memcpy(h, src, src_span);
    memcpy(h + off_ref, ref, ref_span);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, off_res, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(nxm_sad_kernel, dim3(1), dim3(WG_THREADS), 0, st, d, src_stride, d + off_ref, ref_stride, height,
                       width, (uint32_t *)(d + off_res));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_res, d + off_res, 4, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return *(uint32_t *)(h + off_res);
}

static void svt_ext_all_sad_calculation_8x8_16x16_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad);
extern "C" void svt_ext_all_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad) { TIER_A_CALL(svt_ext_all_sad_calculation_8x8_16x16, svt_ext_all_sad_calculation_8x8_16x16_hip_impl(src, src_stride, ref, ref_stride, mv, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, p_eight_sad16x16, p_eight_sad8x8, sub_sad), (src, src_stride, ref, ref_stride, mv, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, p_eight_sad16x16, p_eight_sad8x8, sub_sad)); }
static void svt_ext_all_sad_calculation_8x8_16x16_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t mv, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8], uint32_t p_eight_sad8x8[64][8], uint8_t sub_sad) {
    (void)p_eight_sad8x8;  // not written by the reference either (motion_estimation.c:218)
    if (!ensure_init()) 
        svthip::tier_a_throw("%s", svt_hip_last_error());
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    const size_t src_span = (size_t)63 * src_stride + 64, ref_span = (size_t)63 * ref_stride + 64 + 7;
    const size_t off_ref = align_up(src_span, 256), off_io = off_ref + align_up(ref_span, 256);
    const size_t total = off_io + align_up(sizeof(ExtAllIo), 256);
    uint8_t     *d = sc.device(total), *h = sc.host(total);
    memcpy(h, src, src_span);
    memcpy(h + off_ref, ref, ref_span);
    ExtAllIo *io = (ExtAllIo *)(h + off_io);
    memcpy(io->best8, p_best_sad_8x8, sizeof(io->best8));
    memcpy(io->best16, p_best_sad_16x16, sizeof(io->best16));
    memcpy(io->mv8, p_best_mv8x8, sizeof(io->mv8));
    memcpy(io->mv16, p_best_mv16x16, sizeof(io->mv16));
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ext_all_sad_kernel, dim3(1), dim3(WG_THREADS), 0, st, d, src_stride, d + off_ref, ref_stride, mv,
                       (uint32_t)(sub_sad != 0), (ExtAllIo *)(d + off_io));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(io, d + off_io, sizeof(ExtAllIo), hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(p_best_sad_8x8, io->best8, sizeof(io->best8));
    memcpy(p_best_sad_16x16, io->best16, sizeof(io->best16));
    memcpy(p_best_mv8x8, io->mv8, sizeof(io->mv8));
    memcpy(p_best_mv16x16, io->mv16, sizeof(io->mv16));
    memcpy(p_eight_sad16x16, io->eight16, sizeof(io->eight16));
}

// The three remaining pointers of the group only combine a handful of 32-bit integers that already live in
// host memory (no pixels): offloading them would cost a PCIe round trip per call for ~100 integer adds, so the
// Tier A entry points evaluate them on the calling thread.  (The device-resident equivalent is part of the
// frame-level ME kernel, me_frame.hip.)
static inline uint32_t mv_add_x(uint32_t mv, uint32_t p) {
    const int16_t x = (int16_t)((int16_t)(mv & 0xffff) + (int16_t)p), y = (int16_t)(mv >> 16);
    return ((uint32_t)(uint16_t)y << 16) | (uint16_t)x;
}
static void svt_ext_eight_sad_calculation_32x32_64x64_hip_impl(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]);
extern "C" void svt_ext_eight_sad_calculation_32x32_64x64_hip(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]) { TIER_A_CALL(svt_ext_eight_sad_calculation_32x32_64x64, svt_ext_eight_sad_calculation_32x32_64x64_hip_impl(p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32), (p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32)); }
static void svt_ext_eight_sad_calculation_32x32_64x64_hip_impl(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t p_sad32x32[4][8]) {
    for (uint32_t p = 0; p < 8; p++) {
        uint32_t t = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t s = p_sad16x16[4 * k][p] + p_sad16x16[4 * k + 1][p] + p_sad16x16[4 * k + 2][p] +
                p_sad16x16[4 * k + 3][p];
            p_sad32x32[k][p] = s;
            if (s < p_best_sad_32x32[k])
                p_best_sad_32x32[k] = s, p_best_mv32x32[k] = mv_add_x(mv, p);
            t += s;
        }
        if (t < p_best_sad_64x64[0])
            p_best_sad_64x64[0] = t, p_best_mv64x64[0] = mv_add_x(mv, p);
    }
}
static void svt_ext_sad_calculation_32x32_64x64_hip_impl(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32);
extern "C" void svt_ext_sad_calculation_32x32_64x64_hip(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32) { TIER_A_CALL(svt_ext_sad_calculation_32x32_64x64, svt_ext_sad_calculation_32x32_64x64_hip_impl(p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32), (p_sad16x16, p_best_sad_32x32, p_best_sad_64x64, p_best_mv32x32, p_best_mv64x64, mv, p_sad32x32)); }
static void svt_ext_sad_calculation_32x32_64x64_hip_impl(uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64, uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32) {
    uint32_t t = 0;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t s = p_sad16x16[4 * k] + p_sad16x16[4 * k + 1] + p_sad16x16[4 * k + 2] + p_sad16x16[4 * k + 3];
        p_sad32x32[k]    = s;
        if (s < p_best_sad_32x32[k])
            p_best_sad_32x32[k] = s, p_best_mv32x32[k] = mv;
        t += s;
    }
    if (t < p_best_sad_64x64[0])
        p_best_sad_64x64[0] = t, p_best_mv64x64[0] = mv;
}

static void svt_ext_sad_calculation_8x8_16x16_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8, uint8_t sub_sad);
extern "C" void svt_ext_sad_calculation_8x8_16x16_hip(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8, uint8_t sub_sad) { TIER_A_CALL(svt_ext_sad_calculation_8x8_16x16, svt_ext_sad_calculation_8x8_16x16_hip_impl(src, src_stride, ref, ref_stride, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, mv, p_sad16x16, p_sad8x8, sub_sad), (src, src_stride, ref, ref_stride, p_best_sad_8x8, p_best_sad_16x16, p_best_mv8x8, p_best_mv16x16, mv, p_sad16x16, p_sad8x8, sub_sad)); }
static void svt_ext_sad_calculation_8x8_16x16_hip_impl(uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8, uint8_t sub_sad) {
    // four 8x8 SADs of one 16x16 block at one position: four device N x M SADs
    uint32_t total = 0;
    for (uint32_t q = 0; q < 4; q++) {
        const uint8_t *s = src + (size_t)(q >> 1) * 8 * src_stride + (q & 1) * 8;
        const uint8_t *r = ref + (size_t)(q >> 1) * 8 * ref_stride + (q & 1) * 8;
        uint32_t       v = sub_sad ? (svt_nxm_sad_kernel_hip(s, 2 * src_stride, r, 2 * ref_stride, 4, 8) << 1)
                                   : svt_nxm_sad_kernel_hip(s, src_stride, r, ref_stride, 8, 8);
        p_sad8x8[q]      = v;
        if (v < p_best_sad_8x8[q])
            p_best_sad_8x8[q] = v, p_best_mv8x8[q] = mv;
        total += v;
    }
    if (total < p_best_sad_16x16[0])
        p_best_sad_16x16[0] = total, p_best_mv16x16[0] = mv;
    *p_sad16x16 = total;
}

SVT_HIP_MODULE_WARMUP(me_kernels)
