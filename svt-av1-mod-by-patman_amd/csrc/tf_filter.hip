// tf_filter.hip — the temporal filter's accumulate / normalise stage on gfx950.
// Replaces svt_av1_apply_temporal_filter_planewise_medium_c / _hbd_c (temporal_filtering.c:999-1330),
// svt_aom_apply_filtering_central_c / _highbd_c (:349-420) and svt_aom_get_final_filtered_pixels_c (:2578-2650).
// One workgroup per 32x32 luma block (+ its chroma): the four quadrant squared-error sums by wave reductions, the four
// fixed-point weights on one lane (sqrt_fast, the exp(-x/16) table), then every lane adds weight * prediction into the
// accumulator and the weight into the counter.  Everything is 32-bit integer arithmetic as in the reference.
#include <cmath>
#include <cstring>
#include <mutex>

#include "../../include/svt_hip_tf.h"
#include "common.hpp"

using namespace svthip;

namespace {

__constant__ uint32_t d_exp_fp16[113];  // 65536 * exp(-i / 16), truncated (the reference's expf_tab_fp16)
__device__ const uint32_t SQRT_FP16[16] = {0,      65536,  92681,  113511, 131072, 146542, 160529, 173391,
                                           185363, 196608, 207243, 217358, 227023, 236293, 245213, 253819};  // (uint32)(sqrt(i) * 65536)

__device__ __forceinline__ uint32_t ldpx(const void *p, size_t i, int is16) {
    return is16 ? ((const __attribute__((address_space(1))) uint16_t *)p)[i] : ((const __attribute__((address_space(1))) uint8_t *)p)[i];  // pictures are global memory: no flat loads
}
__device__ __forceinline__ uint32_t sqrt_fast(uint32_t x) {  // temporal_filtering.c:705-714
    if (x > 15) {
        const int log2_half = (31 - __clz((int)x)) >> 1, mul2 = log2_half << 1;
        return SQRT_FP16[x >> (mul2 - 2)] >> (17 - log2_half);
    }
    return SQRT_FP16[x] >> 16;
}

__global__ __launch_bounds__(256) void tf_accumulate_kernel(const SvtHipTfBlock *__restrict__ blocks) {
    __shared__ SvtHipTfBlock b;
    __shared__ uint32_t      qsum[4], luma_err[4], weight[4];
    for (uint32_t i = threadIdx.x; i < sizeof(SvtHipTfBlock) / 4; i += 256) ((uint32_t *)&b)[i] = ((const uint32_t *)&blocks[blockIdx.x])[i];
    __syncthreads();
    const int is16 = b.is_16bit, shift = is16 ? (b.bit_depth - 8) * 2 : 0;
    for (int pl = 0; pl < (b.chroma ? 3 : 1); pl++) {
        const uint32_t bw = pl ? 32u >> b.ss_x : 32u, bh = pl ? 32u >> b.ss_y : 32u, hw = bw >> 1, hh = bh >> 1;
        if (threadIdx.x < 4)
            qsum[threadIdx.x] = 0;
        __syncthreads();
        // squared-error sums of the four quadrants (calculate_squared_errors_sum[_highbd]); the zero-motion variant does
        // not look at the picture
        uint32_t part[4] = {0, 0, 0, 0};
        for (uint32_t i = threadIdx.x; i < (b.zz_based ? 0u : bw * bh); i += 256) {
            const uint32_t r = i / bw, c = i - r * bw;
            const int32_t  d = (int32_t)ldpx(b.src[pl], (size_t)r * b.src_stride[pl] + c, is16) -
                (int32_t)ldpx(b.pred[pl], (size_t)r * b.pred_stride[pl] + c, is16);
            const uint32_t q = (r >= hh ? 2u : 0u) + (c >= hw ? 1u : 0u);
            const uint32_t e = (uint32_t)(d * d);
            part[0] += q == 0 ? e : 0, part[1] += q == 1 ? e : 0, part[2] += q == 2 ? e : 0, part[3] += q == 3 ? e : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t v = part[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0)
                atomicAdd(&qsum[q], v);
        }
        __syncthreads();
        if (threadIdx.x < 4) {  // one lane per quadrant: the weight (temporal_filtering.c:1009-1099)
            const int      q = threadIdx.x, k = b.split ? q : 0;
            const uint32_t th0 = (uint32_t)(((int)b.mv_dist_th << 16) / 10), dist_th = th0 > (1u << 16) ? th0 : (1u << 16);
            const int32_t  col = b.mv_x[k], row = b.mv_y[k];
            const uint32_t dist = sqrt_fast(((uint32_t)(col * col + row * row)) << 8);
            uint32_t       d_factor = (dist << 12) / (dist_th >> 8);
            d_factor                = d_factor > (1u << 8) ? d_factor : (1u << 8);
            const uint32_t blk_err = b.split ? (uint32_t)(is16 ? b.block_error[q] >> 4 : b.block_error[q])
                                             : (uint32_t)(b.block_error[0] >> (is16 ? 6 : 2));
            if (b.zz_based) {  // svt_av1_apply_zz_based_temporal_filter_planewise_medium_partial_c (:789-835, 890-940)
                const uint32_t den = (b.decay_factor_fp16[pl] >> 10) > 1 ? (b.decay_factor_fp16[pl] >> 10) : 1;
                uint32_t       sd  = (blk_err << 2) / den;
                sd                 = sd < 7 * 16 ? sd : 7 * 16;
                weight[q]          = (d_exp_fp16[sd] * 1000u) >> 17;
            } else {
            const uint32_t decay = b.split ? b.decay_factor_fp16[pl] : b.decay_factor_fp16[pl] << 1;
            uint32_t       win = ((((qsum[q] >> shift) << 4) / hw) << 4) / hh;
            if (pl)
                win = (win * 5 + luma_err[q]) / 6;
            else
                luma_err[q] = win;
            const uint32_t combined = (win * 5 + blk_err) / 6;  // TF_WINDOW_BLOCK_BALANCE_WEIGHT = 5
            const uint64_t avg_err  = (uint64_t)((combined >> 3) * (d_factor >> 3));
            const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
            uint32_t       sd  = (uint32_t)(avg_err / den);
            sd                 = sd < 7 * 16 ? sd : 7 * 16;
            weight[q]          = (d_exp_fp16[sd] * 1000u) >> 16;  // TF_WEIGHT_SCALE
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < bw * bh; i += 256) {
            const uint32_t r = i / bw, c = i - r * bw;
            const uint32_t w = weight[(r >= bh / 2 ? 2u : 0u) + (c >= bw / 2 ? 1u : 0u)];
            const size_t   k = (size_t)r * b.pred_stride[pl] + c;
            b.count[pl][k]   = (uint16_t)(b.count[pl][k] + w);
            b.accum[pl][k] += w * ldpx(b.pred[pl], k, is16);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void tf_central_kernel(const SvtHipTfBlock *__restrict__ blocks) {
    const SvtHipTfBlock &b = blocks[blockIdx.x];
    for (int pl = 0; pl < (b.chroma ? 3 : 1); pl++) {
        const uint32_t bw = pl ? 32u >> b.ss_x : 32u, bh = pl ? 32u >> b.ss_y : 32u;
        for (uint32_t i = threadIdx.x; i < bw * bh; i += 256) {
            const uint32_t r = i / bw, c = i - r * bw;
            const size_t   k = (size_t)r * b.pred_stride[pl] + c;
            b.accum[pl][k]   = 1000u * ldpx(b.src[pl], (size_t)r * b.src_stride[pl] + c, b.is_16bit);  // TF_PLANEWISE_FILTER_WEIGHT_SCALE
            b.count[pl][k]   = 1000;
        }
    }
}

__global__ __launch_bounds__(256) void tf_normalise_kernel(const SvtHipTfBlock *__restrict__ blocks, const SvtHipTfOut *__restrict__ outs) {
    const SvtHipTfBlock &b = blocks[blockIdx.x];
    const SvtHipTfOut   &o = outs[blockIdx.x];
    for (int pl = 0; pl < (b.chroma ? 3 : 1); pl++) {
        const uint32_t bw = pl ? 32u >> b.ss_x : 32u, bh = pl ? 32u >> b.ss_y : 32u;
        for (uint32_t i = threadIdx.x; i < bw * bh; i += 256) {
            const uint32_t r = i / bw, c = i - r * bw;
            const size_t   k = (size_t)r * b.pred_stride[pl] + c;
            const uint32_t cnt = b.count[pl][k], v = cnt ? (b.accum[pl][k] + (cnt >> 1)) / cnt : 0;
            if (b.is_16bit)
                ((uint16_t *)o.dst[pl])[(size_t)r * o.dst_stride[pl] + c] = (uint16_t)v;
            else
                ((uint8_t *)o.dst[pl])[(size_t)r * o.dst_stride[pl] + c] = (uint8_t)v;
        }
    }
}

// central + accumulate over every reference picture + normalise of one 32x32 block in ONE pass: the accumulators live in registers
// (a thread owns at most four samples of a plane), the source block is read once and each prediction once, and accum[] / count[] never
// go through memory — the three-kernel sequence above reads and writes them once per reference picture (12 bytes per sample and
// reference).  Same arithmetic in the same order per sample: sums of uint32 / uint16 terms, so the order over references is free.
struct TfRefLists {
    const SvtHipTfBlock *p[SVT_HIP_TF_MAX_REFS];
};
// KC: samples of a chroma plane per thread (1 for 4:2:0 — 16 x 16 = one per thread —, 4 for anything else)
template <int KC>
__global__ __launch_bounds__(256) void tf_filter_blocks_kernel(TfRefLists lists, uint32_t n_refs, const SvtHipTfBlock *__restrict__ statics,
                                                               const SvtHipTfOut *__restrict__ outs, uint32_t ss_x, uint32_t ss_y) {
    __shared__ SvtHipTfBlock b, sb;  // the current reference's record; the reference-independent one
    __shared__ uint32_t      qsum[3][4], weight[3][4];
    for (uint32_t i = threadIdx.x; i < sizeof(SvtHipTfBlock) / 4; i += 256) ((uint32_t *)&sb)[i] = ((const uint32_t *)&statics[blockIdx.x])[i];
    __syncthreads();
    const int      is16 = sb.is_16bit, shift = is16 ? (sb.bit_depth - 8) * 2 : 0, npl = sb.chroma ? 3 : 1;
    const uint32_t cw = 32u >> ss_x, ch = 32u >> ss_y;
    uint32_t       acc[3][4], sv[3][4], cnt[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const uint32_t bw = pl ? cw : 32u, bh = pl ? ch : 32u;
#pragma unroll
        for (int k = 0; k < (pl ? KC : 4); k++) {
            const uint32_t i = threadIdx.x + k * 256;
            acc[pl][k] = sv[pl][k] = cnt[pl][k] = 0;
            if (pl < npl && i < bw * bh) {
                const uint32_t r = i / bw, c = i - r * bw;
                sv[pl][k]  = ldpx(sb.src[pl], (size_t)r * sb.src_stride[pl] + c, is16);
                acc[pl][k] = 1000u * sv[pl][k], cnt[pl][k] = 1000;  // TF_PLANEWISE_FILTER_WEIGHT_SCALE (tf_central_kernel)
            }
        }
    }
    for (uint32_t ref = 0; ref < n_refs; ref++) {
        __syncthreads();  // the previous reference's record and weights have been used
        for (uint32_t i = threadIdx.x; i < sizeof(SvtHipTfBlock) / 4; i += 256) ((uint32_t *)&b)[i] = ((const uint32_t *)&lists.p[ref][blockIdx.x])[i];
        if (threadIdx.x < 12)
            qsum[threadIdx.x >> 2][threadIdx.x & 3] = 0;
        __syncthreads();
        uint32_t pv[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const uint32_t bw = pl ? cw : 32u, bh = pl ? ch : 32u, hw = bw >> 1, hh = bh >> 1;
            uint32_t       part[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < (pl ? KC : 4); k++) {
                const uint32_t i = threadIdx.x + k * 256;
                pv[pl][k]        = 0;
                if (pl < npl && i < bw * bh) {
                    const uint32_t r = i / bw, c = i - r * bw;
                    pv[pl][k]        = ldpx(b.pred[pl], (size_t)r * b.pred_stride[pl] + c, is16);
                    const int32_t  d = (int32_t)sv[pl][k] - (int32_t)pv[pl][k];
                    const uint32_t q = (r >= hh ? 2u : 0u) + (c >= hw ? 1u : 0u), e = (uint32_t)(d * d);
                    part[0] += q == 0 ? e : 0, part[1] += q == 1 ? e : 0, part[2] += q == 2 ? e : 0, part[3] += q == 3 ? e : 0;
                }
            }
            if (pl < npl && !b.zz_based) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t v = part[q];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                    if ((threadIdx.x & 63) == 0)
                        atomicAdd(&qsum[pl][q], v);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 12 && (int)(threadIdx.x >> 2) < npl) {  // one lane per (plane, quadrant): the weights (temporal_filtering.c:1009-1099), as
                                                                    // in tf_accumulate_kernel; a chroma lane derives the luma window error itself
            const int      q = threadIdx.x & 3, pl = threadIdx.x >> 2, k = b.split ? q : 0;
            const uint32_t th0 = (uint32_t)(((int)b.mv_dist_th << 16) / 10), dist_th = th0 > (1u << 16) ? th0 : (1u << 16);
            const int32_t  col = b.mv_x[k], row = b.mv_y[k];
            const uint32_t dist = sqrt_fast(((uint32_t)(col * col + row * row)) << 8);
            uint32_t       d_factor = (dist << 12) / (dist_th >> 8);
            d_factor                = d_factor > (1u << 8) ? d_factor : (1u << 8);
            const uint32_t blk_err = b.split ? (uint32_t)(is16 ? b.block_error[q] >> 4 : b.block_error[q])
                                             : (uint32_t)(b.block_error[0] >> (is16 ? 6 : 2));
            const uint32_t hw = (pl ? cw : 32u) >> 1, hh = (pl ? ch : 32u) >> 1;
            if (b.zz_based) {
                const uint32_t den = (b.decay_factor_fp16[pl] >> 10) > 1 ? (b.decay_factor_fp16[pl] >> 10) : 1;
                uint32_t       sd  = (blk_err << 2) / den;
                sd                 = sd < 7 * 16 ? sd : 7 * 16;
                weight[pl][q]      = (d_exp_fp16[sd] * 1000u) >> 17;
            } else {
                const uint32_t decay = b.split ? b.decay_factor_fp16[pl] : b.decay_factor_fp16[pl] << 1;
                uint32_t       win   = ((((qsum[pl][q] >> shift) << 4) / hw) << 4) / hh;
                if (pl)
                    win = (win * 5 + ((((qsum[0][q] >> shift) << 4) / 16u) << 4) / 16u) / 6;  // the luma quadrant is 16 x 16
                const uint32_t combined = (win * 5 + blk_err) / 6;
                const uint64_t avg_err  = (uint64_t)((combined >> 3) * (d_factor >> 3));
                const uint32_t den      = (decay >> 10) > 1 ? (decay >> 10) : 1;
                uint32_t       sd       = (uint32_t)(avg_err / den);
                sd                      = sd < 7 * 16 ? sd : 7 * 16;
                weight[pl][q]           = (d_exp_fp16[sd] * 1000u) >> 16;
            }
        }
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const uint32_t bw = pl ? cw : 32u, bh = pl ? ch : 32u;
#pragma unroll
            for (int k = 0; k < (pl ? KC : 4); k++) {
                const uint32_t i = threadIdx.x + k * 256;
                if (pl < npl && i < bw * bh) {
                    const uint32_t r = i / bw, c = i - r * bw, w = weight[pl][(r >= bh / 2 ? 2u : 0u) + (c >= bw / 2 ? 1u : 0u)];
                    cnt[pl][k] = (uint16_t)(cnt[pl][k] + w), acc[pl][k] += w * pv[pl][k];
                }
            }
        }
    }
    const SvtHipTfOut &o = outs[blockIdx.x];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const uint32_t bw = pl ? cw : 32u, bh = pl ? ch : 32u;
#pragma unroll
        for (int k = 0; k < (pl ? KC : 4); k++) {
            const uint32_t i = threadIdx.x + k * 256;
            if (pl < npl && i < bw * bh) {
                const uint32_t r = i / bw, c = i - r * bw, n = cnt[pl][k], v = n ? (acc[pl][k] + (n >> 1)) / n : 0;
                if (is16)
                    ((uint16_t *)o.dst[pl])[(size_t)r * o.dst_stride[pl] + c] = (uint16_t)v;
                else
                    ((uint8_t *)o.dst[pl])[(size_t)r * o.dst_stride[pl] + c] = (uint8_t)v;
            }
        }
    }
}

std::once_flag g_once;
int32_t        g_rc = SVT_HIP_OK;
void           upload() {
    uint32_t tab[113];
    for (int k = 0; k < 113; k++) tab[k] = (uint32_t)(65536.0 * exp(-k / 16.0));
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_exp_fp16), tab, sizeof(tab)) != hipSuccess) {
        set_error("uploading the temporal filter's weight table failed");
        g_rc = SVT_HIP_ERR_RUNTIME;
    }
}
int32_t ready(const void *p, uint32_t n, const char *who) {
    if (!p || n == 0) {
        set_error("%s: bad argument", who);
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    std::call_once(g_once, upload);
    return g_rc;
}

}  // namespace

extern "C" int32_t svt_hip_tf_accumulate_batch(const SvtHipTfBlock *d_blocks, uint32_t n_blocks, void *stream) {
    const int32_t rc = ready(d_blocks, n_blocks, "svt_hip_tf_accumulate_batch");
    if (rc != SVT_HIP_OK)
        return rc;
    hipLaunchKernelGGL(tf_accumulate_kernel, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), d_blocks);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_tf_central_batch(const SvtHipTfBlock *d_blocks, uint32_t n_blocks, void *stream) {
    const int32_t rc = ready(d_blocks, n_blocks, "svt_hip_tf_central_batch");
    if (rc != SVT_HIP_OK)
        return rc;
    hipLaunchKernelGGL(tf_central_kernel, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), d_blocks);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}
extern "C" int32_t svt_hip_tf_normalise_batch(const SvtHipTfBlock *d_blocks, const SvtHipTfOut *d_out, uint32_t n_blocks, void *stream) {
    const int32_t rc = ready(d_blocks, n_blocks, "svt_hip_tf_normalise_batch");
    if (rc != SVT_HIP_OK)
        return rc;
    if (!d_out) {
        set_error("svt_hip_tf_normalise_batch: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    hipLaunchKernelGGL(tf_normalise_kernel, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), d_blocks, d_out);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_tf_filter_blocks(const SvtHipTfBlock *const *d_ref_blocks, uint32_t n_refs, const SvtHipTfBlock *d_static_blocks,
                                            const SvtHipTfOut *d_out, uint32_t n_blocks, uint32_t ss_x, uint32_t ss_y, void *stream) {
    const int32_t rc = ready(d_static_blocks, n_blocks, "svt_hip_tf_filter_blocks");
    if (rc != SVT_HIP_OK)
        return rc;
    if (!d_out || n_refs > SVT_HIP_TF_MAX_REFS || (n_refs && !d_ref_blocks) || ss_x > 1 || ss_y > 1) {
        set_error("svt_hip_tf_filter_blocks: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    TfRefLists lists;
    memset(&lists, 0, sizeof(lists));
    for (uint32_t r = 0; r < n_refs; r++) {
        if (!d_ref_blocks[r]) {
            set_error("svt_hip_tf_filter_blocks: reference %u has no block list", r);
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
        lists.p[r] = d_ref_blocks[r];
    }
    if (ss_x && ss_y)
        hipLaunchKernelGGL(tf_filter_blocks_kernel<1>, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), lists, n_refs, d_static_blocks, d_out, ss_x, ss_y);
    else
        hipLaunchKernelGGL(tf_filter_blocks_kernel<4>, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), lists, n_refs, d_static_blocks, d_out, ss_x, ss_y);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

SVT_HIP_MODULE_WARMUP(tf_filter)
