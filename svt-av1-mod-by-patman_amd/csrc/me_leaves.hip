// me_leaves.hip — the remaining per-call (Tier A) leaves of the ME group of the reference's RTCD table
// (aom_dsp_rtcd.h:852-866): host pointers in, host pointers out, the pointer's exact signature.
//   svt_nxm_sad_kernel_sub_sampled   (generic-C row of the table binds it to the plain N x M SAD: aom_dsp_rtcd.c:1213)
//   sad_16b_kernel                   (svt_aom_sad_16b_kernel_c, C_DEFAULT/compute_sad_c.c:39-56)
//   svt_initialize_buffer_32bits     (me_sad_calculation.c:14-17)
//   svt_pme_sad_loop_kernel          (product_coding_loop.c:1781-1828: SAD + motion-vector cost over a sparse search grid)
// The whole-frame ME path (me_frame.hip) never calls these.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <vector>

#include "common.hpp"

using namespace svthip;

namespace {

constexpr int WG = 256;

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }

__device__ __forceinline__ uint32_t wave_add(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(WG) void sad16_kernel(const uint16_t *__restrict__ src, uint32_t ss, const uint16_t *__restrict__ ref,
                                                   uint32_t rs, uint32_t h, uint32_t w, uint32_t *__restrict__ out) {
    __shared__ uint32_t part[WG / 64];
    uint32_t            acc = 0;
    for (uint32_t i = threadIdx.x; i < w * h; i += WG) {
        const uint32_t r = i / w, c = i - r * w;
        const int      d = (int)src[(size_t)r * ss + c] - (int)ref[(size_t)r * rs + c];
        acc += (uint32_t)(d < 0 ? -d : d);
    }
    acc = wave_add(acc);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        *out = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(WG) void fill32_kernel(uint32_t *__restrict__ p, uint32_t n, uint32_t value) {
    for (uint32_t i = blockIdx.x * WG + threadIdx.x; i < n; i += gridDim.x * WG) p[i] = value;
}

// One visited search position per wave.  A position carries its scan-order number (1-based): the winner is the smallest
// (cost, order) pair, and the caller's incoming best enters as (best_cost, 0), so a position only wins with a strictly
// smaller cost and the first one in scan order wins a tie — the reference's sequential `cost < *best_cost` updates.
struct PmeArgs {
    const uint8_t  *src, *ref;
    const uint32_t *pos;        // (y << 16) | x of every visited position, in scan order
    const int32_t  *row_rate;   // ENTROPY: mvcost[0][clamped row difference] per search row
    const int32_t  *col_rate;   //          mvcost[1][clamped column difference] per search column
    unsigned long long *best;   // (cost << 32) | order
    uint32_t src_stride, ref_stride, bw, bh, n_pos;
    int32_t  start_x, start_y, mvx, mvy, ref_row, ref_col, type, error_per_bit;
    int32_t  joint_rate[4];
};
__global__ __launch_bounds__(WG) void pme_sad_kernel(PmeArgs a) {
    const uint32_t lane = threadIdx.x & 63, p = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (p >= a.n_pos)
        return;
    const uint32_t xs = a.pos[p] & 0xffff, ys = a.pos[p] >> 16;
    const uint8_t *r  = a.ref + (size_t)ys * a.ref_stride + xs;
    uint32_t       acc = 0;
    for (uint32_t i = lane; i < a.bw * a.bh; i += 64) {
        const uint32_t y = i / a.bw, x = i - y * a.bw;
        const int      d = (int)a.src[(size_t)y * a.src_stride + x] - (int)r[(size_t)y * a.ref_stride + x];
        acc += (uint32_t)(d < 0 ? -d : d);
    }
    acc = wave_add(acc);
    if (lane)
        return;
    // motion-vector cost (mcomp.c:44-68); vector components and their differences are 16-bit fields
    const int16_t col = (int16_t)((uint32_t)a.mvx + (uint32_t)(a.start_x + (int)xs) * 8u);
    const int16_t row = (int16_t)((uint32_t)a.mvy + (uint32_t)(a.start_y + (int)ys) * 8u);
    const int16_t dr = (int16_t)(row - a.ref_row), dc = (int16_t)(col - a.ref_col);
    const int16_t ar = (int16_t)(dr < 0 ? -dr : dr), ac = (int16_t)(dc < 0 ? -dc : dc);
    int32_t       cost = 0;
    switch (a.type) {
    case 0: {  // MV_COST_ENTROPY
        const int     joint = dr == 0 ? (dc == 0 ? 0 : 1) : (dc == 0 ? 2 : 3);
        const int32_t rate  = a.joint_rate[joint] + a.row_rate[ys] + a.col_rate[xs];
        cost                = (int32_t)(((int64_t)rate * a.error_per_bit + (1ll << 13)) >> 14);
        break;
    }
    case 1: cost = (2 * (ar + ac)) >> 3; break;  // MV_COST_L1_LOWRES
    case 3: cost = (ar + ac) >> 3; break;        // MV_COST_L1_HDRES
    case 4: cost = (int32_t)(((int64_t)((ar + ac) << 8) * a.error_per_bit + (1ll << 13)) >> 14); break;  // MV_COST_OPT
    default: break;                              // MV_COST_L1_MIDRES (lambda 0), MV_COST_NONE
    }
    acc += (uint32_t)cost;
    atomicMin(a.best, ((unsigned long long)acc << 32) | (p + 1));
}

}  // namespace

static uint32_t svt_nxm_sad_kernel_sub_sampled_hip_impl(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
extern "C" uint32_t svt_nxm_sad_kernel_sub_sampled_hip(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) { TIER_A_CALL(svt_nxm_sad_kernel_sub_sampled, svt_nxm_sad_kernel_sub_sampled_hip_impl(src, src_stride, ref, ref_stride, height, width), (src, src_stride, ref, ref_stride, height, width)); }
static uint32_t svt_nxm_sad_kernel_sub_sampled_hip_impl(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) {
    return svt_nxm_sad_kernel_hip(src, src_stride, ref, ref_stride, height, width);
}

static uint32_t svt_aom_sad_16b_kernel_hip_impl(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width);
extern "C" uint32_t svt_aom_sad_16b_kernel_hip(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) { TIER_A_CALL(svt_aom_sad_16b_kernel, svt_aom_sad_16b_kernel_hip_impl(src, src_stride, ref, ref_stride, height, width), (src, src_stride, ref, ref_stride, height, width)); }
static uint32_t svt_aom_sad_16b_kernel_hip_impl(uint16_t *src, uint32_t src_stride, uint16_t *ref, uint32_t ref_stride, uint32_t height, uint32_t width) {
    if (!height || !width)
        return 0;
    if (!ensure_init())
        fatal("sad_16b_kernel");
    const size_t src_span = ((size_t)(height - 1) * src_stride + width) * 2, ref_span = ((size_t)(height - 1) * ref_stride + width) * 2;
    const size_t off_ref = up256(src_span), off_res = off_ref + up256(ref_span);
    Scratch     &sc = tls_scratch();
    uint8_t     *d = sc.device(off_res + 256), *h = sc.host(off_res + 256);
    memcpy(h, src, src_span);
    memcpy(h + off_ref, ref, ref_span);
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, off_res, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(sad16_kernel, dim3(1), dim3(WG), 0, st, (const uint16_t *)d, src_stride, (const uint16_t *)(d + off_ref), ref_stride,
                       height, width, (uint32_t *)(d + off_res));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_res, d + off_res, 4, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    return *(const uint32_t *)(h + off_res);
}

static void svt_initialize_buffer_32bits_hip_impl(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value);
extern "C" void svt_initialize_buffer_32bits_hip(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value) { TIER_A_CALL(svt_initialize_buffer_32bits, svt_initialize_buffer_32bits_hip_impl(pointer, count128, count32, value), (pointer, count128, count32, value)); }
static void svt_initialize_buffer_32bits_hip_impl(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value) {
    const uint32_t n = count128 * 4 + count32;
    if (!n)
        return;
    if (!ensure_init())
        fatal("svt_initialize_buffer_32bits");
    Scratch    &sc = tls_scratch();
    uint8_t    *d = sc.device((size_t)n * 4), *h = sc.host((size_t)n * 4);
    hipStream_t st = resolve_stream(nullptr);
    const uint32_t blocks = (n + WG - 1) / WG;
    hipLaunchKernelGGL(fill32_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(WG), 0, st, (uint32_t *)d, n, value);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h, d, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(pointer, h, (size_t)n * 4);
}

static void svt_pme_sad_loop_kernel_hip_impl(const SvtHipMvCostParam *mv_cost_params, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t search_position_start_x, int16_t search_position_start_y, int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy);
extern "C" void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParam *mv_cost_params, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t search_position_start_x, int16_t search_position_start_y, int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy) { TIER_A_CALL(svt_pme_sad_loop_kernel, svt_pme_sad_loop_kernel_hip_impl(mv_cost_params, src, src_stride, ref, ref_stride, block_height, block_width, best_cost, best_mvx, best_mvy, search_position_start_x, search_position_start_y, search_area_width, search_area_height, search_step, mvx, mvy), (mv_cost_params, src, src_stride, ref, ref_stride, block_height, block_width, best_cost, best_mvx, best_mvy, search_position_start_x, search_position_start_y, search_area_width, search_area_height, search_step, mvx, mvy)); }
static void svt_pme_sad_loop_kernel_hip_impl(const SvtHipMvCostParam *mv_cost_params, uint8_t *src, uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, uint32_t block_height, uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t search_position_start_x, int16_t search_position_start_y, int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy) {
    if (search_step <= 0) {
        set_error("svt_pme_sad_loop_kernel: search_step %d would never terminate", (int)search_step);
        fatal("svt_pme_sad_loop_kernel");
    }
    // the visiting order of the reference (product_coding_loop.c:1796-1806): eight consecutive columns, then a jump of
    // search_step; the column counter carries over from one search row to the next
    std::vector<uint32_t> pos;
    int16_t               col_num = 0, step_x = 1;
    int                   max_x = 0, max_y = 0;
    for (int16_t ys = 0; ys < search_area_height; ys = (int16_t)(ys + search_step))
        for (int16_t xs = 0; xs < search_area_width; xs = (int16_t)(xs + step_x)) {
            if ((search_area_width - xs) < 8 && col_num == 0)
                continue;
            if (col_num == 7)
                col_num = 0, step_x = search_step;
            else
                col_num++, step_x = 1;
            pos.push_back(((uint32_t)ys << 16) | (uint32_t)xs);
            max_x = xs > max_x ? xs : max_x, max_y = ys;
        }
    if (pos.empty() || !block_height || !block_width)
        return;
    if (!ensure_init())
        fatal("svt_pme_sad_loop_kernel");
    const int    type = mv_cost_params->mv_cost_type;
    const size_t src_span = (size_t)(block_height - 1) * src_stride + block_width;
    const size_t ref_span = (size_t)(max_y + block_height - 1) * ref_stride + max_x + block_width;
    const size_t n_pos = pos.size(), rows = (size_t)max_y + 1, cols = (size_t)max_x + 1;
    const size_t off_ref = up256(src_span), off_pos = off_ref + up256(ref_span), off_row = off_pos + up256(n_pos * 4),
                 off_col = off_row + up256(rows * 4), off_best = off_col + up256(cols * 4), total = off_best + 256;
    Scratch &sc = tls_scratch();
    uint8_t *d = sc.device(total), *h = sc.host(total);
    memcpy(h, src, src_span);
    memcpy(h + off_ref, ref, ref_span);
    memcpy(h + off_pos, pos.data(), n_pos * 4);
    PmeArgs a{};
    if (type == 0) {  // gather the table entries the search can reach (svt_mv_cost, mcomp.h:136-139)
        int32_t *rr = (int32_t *)(h + off_row), *cr = (int32_t *)(h + off_col);
        for (size_t y = 0; y < rows; y++) {
            const int16_t row = (int16_t)((uint32_t)mvy + (uint32_t)(search_position_start_y + (int)y) * 8u);
            const int16_t dr  = (int16_t)(row - mv_cost_params->ref_mv->row);
            rr[y]             = mv_cost_params->mvcost[0][dr < -16384 ? -16384 : (dr > 16384 ? 16384 : dr)];
        }
        for (size_t x = 0; x < cols; x++) {
            const int16_t col = (int16_t)((uint32_t)mvx + (uint32_t)(search_position_start_x + (int)x) * 8u);
            const int16_t dc  = (int16_t)(col - mv_cost_params->ref_mv->col);
            cr[x]             = mv_cost_params->mvcost[1][dc < -16384 ? -16384 : (dc > 16384 ? 16384 : dc)];
        }
        for (int j = 0; j < 4; j++) a.joint_rate[j] = mv_cost_params->mvjcost[j];
    }
    *(unsigned long long *)(h + off_best) = (unsigned long long)*best_cost << 32;
    a.src = d, a.ref = d + off_ref, a.pos = (const uint32_t *)(d + off_pos);
    a.row_rate = (const int32_t *)(d + off_row), a.col_rate = (const int32_t *)(d + off_col);
    a.best = (unsigned long long *)(d + off_best);
    a.src_stride = src_stride, a.ref_stride = ref_stride, a.bw = block_width, a.bh = block_height, a.n_pos = (uint32_t)n_pos;
    a.start_x = search_position_start_x, a.start_y = search_position_start_y, a.mvx = mvx, a.mvy = mvy;
    a.ref_row = mv_cost_params->ref_mv->row, a.ref_col = mv_cost_params->ref_mv->col, a.type = type;
    a.error_per_bit = mv_cost_params->error_per_bit;
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pme_sad_kernel, dim3((uint32_t)((n_pos + 3) / 4)), dim3(WG), 0, st, a);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(h + off_best, d + off_best, 8, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const unsigned long long key = *(const unsigned long long *)(h + off_best);
    if ((uint32_t)key) {  // a visited position beat the incoming cost
        const uint32_t w = pos[(uint32_t)key - 1], xs = w & 0xffff, ys = w >> 16;
        *best_mvx  = (int16_t)((uint32_t)mvx + (uint32_t)(search_position_start_x + (int)xs) * 8u);
        *best_mvy  = (int16_t)((uint32_t)mvy + (uint32_t)(search_position_start_y + (int)ys) * 8u);
        *best_cost = (uint32_t)(key >> 32);
    }
}

SVT_HIP_MODULE_WARMUP(me_leaves)
