// tpl.hip — the TPL dispenser of one picture on gfx950: tpl_mc_flow_dispenser_sb_generic
// (src_ops_process.c:519-1207) for the DC / SAD configurations described in include/svt_hip_tpl.h: 16x16 blocks with full-pel (tpl level 4)
// or quarter-pel refined vectors (level 3), or 32x32 blocks whose transform is TX_32X8 on every 4th row (level 5).
//
// One wavefront per block, all blocks of the picture in ONE launch:
//   source-based path (no dependencies): the block's source samples stay in registers (one or four dwords per lane); DC prediction
//     from the source neighbours, `v_sad_u8` against it and against every single-reference ME candidate; the inter winner's
//     residual goes through the fused transform block of txfm_block.hpp (DCT 16x16 with the pf_shape zero-out, quantize_fp)
//     and svt_av1_block_error is summed from the coefficient arrays it wrote;
//   reconstruction path: inter blocks copy the reference's reconstruction, intra blocks build the DC prediction from their
//     reconstructed neighbours — the only dependency inside the picture: an intra block first waits for the done-flags of its
//     left / top / top-left neighbours (acquire), every block sets its own flag when its reconstruction is in memory (release);
//     a workgroup takes its block index from a ticket counter when it STARTS, so every lower index belongs to a workgroup that
//     is already running or done whatever order the hardware dispatches in (no assumption about dispatch order or placement),
//     and the wait is bounded;
//   then residual -> transform -> quantise -> error -> inverse + reconstruction by the same transform block, TplStats.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_tpl.h"
#include "common.hpp"
#include "txfm_block.hpp"

using namespace svthip;
using namespace svthip::txb;

namespace {

constexpr int      TPL_PAD = 32, NEWMV_MODE = 16;
constexpr uint32_t SPIN_LIMIT = 1u << 22;  // x ~1 us: a neighbour that never finishes ends the wait instead of hanging the GPU
__device__ const uint16_t TPL_ISCAN[256] = {0};  // the scan only orders the end-of-block position, which nothing here depends on

struct TplArgs {
    SvtHipTplFrameJob j;
    const uint8_t    *src0;  // sample (0,0)
    uint8_t          *rec0;
    uint32_t          W, H, a16, rows16;
    uint32_t          coherent_rows;  // reconstruction rows are 4-byte aligned: blocks are published by write-through stores
    uint32_t         *flags;    // [blocks]
    uint32_t         *error;    // [0]: set when a dependency wait ran into SPIN_LIMIT; [1]: the ticket counter
};

__device__ __forceinline__ uint32_t ld8(const uint8_t *p) { return *(const __attribute__((address_space(1))) uint8_t *)p; }
// four samples of a row that need not be aligned
__device__ __forceinline__ uint32_t ld4(const uint8_t *p) {
    typedef uint32_t __attribute__((aligned(1))) u32u;
    return *(const __attribute__((address_space(1))) u32u *)p;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// svt_aom_update_neighbor_samples_array_open_loop_mb[_recon] (enc_intra_prediction.c:1127-1300) for a BS x BS block with
// use_top_right_bottom_left = update_top_neighbor = 1; above_ref / left_ref point at the [-1] entries.  One lane.
template <uint32_t BS>
__device__ void neighbours(uint8_t *above_ref, uint8_t *left_ref, const uint8_t *pic0, uint32_t stride, uint32_t x, uint32_t y, uint32_t width,
                           uint32_t height) {
    const uint32_t bw = BS, bh = BS, n = 2 * BS;
    const uint8_t *src = pic0 + (size_t)y * stride + x;
    for (uint32_t i = 0; i <= n; i++) above_ref[i] = 127, left_ref[i] = 129;
    uint8_t *a = above_ref, *l = left_ref;
    if (x != 0 && y != 0)
        *a = *l = (uint8_t)ld8(src - stride - 1);
    else
        *a = *l = 128;
    a++, l++;
    uint32_t count = n;
    if (x != 0) {
        const uint8_t *rp = src - 1;
        if (y == 0)
            l[-1] = (uint8_t)ld8(rp);
        count = (y + count > height) ? count - (y + count - height) : count;
        for (uint32_t i = 0; i < count; i++, rp += stride) *l++ = (uint8_t)ld8(rp);
        l += n - count;
        for (uint32_t i = 0; i < bh; i++) l[-(int)bh + (int)i] = l[-(int)bh - 1];
    } else if (y != 0) {
        count = (y + count > height) ? count - (y + count - height) : count;
        const uint8_t v = (uint8_t)ld8(src - stride);
        for (uint32_t i = 0; i <= count; i++) l[(int)i - 1] = v;
        a[-1] = v;
    } else
        l += count;
    count = n;
    if (y != 0) {
        count = (x + count > width) ? count - (x + count - width) : count;
        for (uint32_t i = 0; i < count; i++) a[i] = (uint8_t)ld8(src - stride + i);
        if (x != 0)
            for (uint32_t i = 0; i < bw; i++) a[bw + i] = a[bw - 1];
    } else if (x != 0) {
        count = (x + count > width) ? count - (x + count - width) : count;
        const uint8_t v = *(l - count);
        for (uint32_t i = 0; i <= count; i++) a[(int)i - 1] = v;
    }
}

// DC_PRED value of the block at (x, y) of plane pic0 (svt_aom_dc_pred[x > 0][y > 0][TX_16X16 / TX_32X32]); uniform over the wave
constexpr int NB_LEFT = 8 + 88;  // offsets of the neighbour arrays in their LDS buffer: above[-1 .. 64], left[-1 .. 64]
template <uint32_t BS>
__device__ uint32_t dc_value(const uint8_t *pic0, uint32_t stride, uint32_t x, uint32_t y, uint32_t W, uint32_t H, uint8_t *lds_nb) {
    const uint32_t lane = threadIdx.x;
    const bool     inside = x + BS <= W && y + BS <= H;
    uint32_t       sa, sl;
    if (x > 0 && y > 0 && inside) {  // get_neighbor_samples_dc
        const uint8_t *src = pic0 + (size_t)y * stride + x;
        uint32_t       v = 0;
        if (lane < BS)
            v = ld8(src - stride + lane);
        else if (lane < 2 * BS)
            v = ld8(src + (size_t)(lane - BS) * stride - 1);
        sa = wave_sum(lane < BS ? v : 0), sl = wave_sum(lane >= BS && lane < 2 * BS ? v : 0);
    } else {
        uint8_t *above = lds_nb + 8, *left = lds_nb + NB_LEFT;
        if (lane == 0)
            neighbours<BS>(above - 1, left - 1, pic0, stride, x, y, W, H);
        __syncthreads();
        sa = wave_sum(lane < BS ? above[lane] : 0), sl = wave_sum(lane < BS ? left[lane] : 0);
        __syncthreads();
    }
    if (x > 0 && y > 0)
        return (sa + sl + BS) / (2 * BS);
    if (x > 0)
        return (sl + BS / 2) / BS;
    if (y > 0)
        return (sa + BS / 2) / BS;
    return 128;
}

// residual (source - prediction) -> DCT BS x (BS >> SUB) on every (1 << SUB)-th row (pf_shape) -> quantize_fp -> svt_av1_block_error >> 2,
// max 1 (get_quantize_error: neither transform size here is TX_32X32); with `inv` the inverse transform reconstructs onto recon
// (prediction and reconstruction may be the same samples).  *nonzero = the block has a non-zero quantised coefficient (eob != 0).
template <int BS, int SUB>
__device__ int64_t quantize_error(const TplArgs &a, const uint8_t *src, uint32_t src_stride, const uint8_t *pred, uint32_t pred_stride,
                                  uint8_t *recon, uint32_t recon_stride, bool inv, int32_t *lds_tile, bool *nonzero) {
    constexpr int  TH = BS >> SUB;
    const int      lane = threadIdx.x;
    SvtHipTxfmDesc d;
    memset(&d, 0, sizeof(d));
    d.residual_off = (uint64_t)(uintptr_t)src, d.residual_stride = src_stride << SUB;
    d.pred_off = (uint64_t)(uintptr_t)pred, d.pred_stride = pred_stride << SUB;
    d.recon_off = (uint64_t)(uintptr_t)recon, d.recon_stride = recon_stride << SUB;
    d.coeff_off = d.dqcoeff_off = SVT_HIP_NO_OFFSET;  // the block error comes out of the transform block itself (BLKERR)
    d.qcoeff_off = SVT_HIP_NO_OFFSET, d.qm_off = d.iqm_off = SVT_HIP_NO_OFFSET;
    d.iscan_off = (uint64_t)(uintptr_t)TPL_ISCAN;
    for (int i = 0; i < 2; i++) d.round[i] = a.j.round_fp[i], d.quant[i] = a.j.quant_fp[i], d.dequant[i] = a.j.dequant[i];
    d.tx_type = 0, d.shape = a.j.pf_shape, d.bit_depth = 8, d.quant_mode = SVT_HIP_QUANT_FP, d.log_scale = 0;
    d.flags = (uint8_t)(SVT_HIP_TX_FWD | SVT_HIP_TX_SRC_PRED | (inv ? SVT_HIP_TX_INV : 0));
    SvtHipTxfmResult res;
    memset(&res, 0, sizeof(res));
    txfm_block<BS, TH, true>((uint8_t *)nullptr, d, &res, lane < BS, lane & (BS - 1), lds_tile + (lane / BS) * (TH * (BS + 1)));
    *nonzero = __shfl((int)res.eob, 0, 64) != 0;  // lane 0 holds the block's result (the all-zero scan table makes eob 0 / 1)
    const uint64_t e2 = ((uint64_t)(uint32_t)__shfl((int)(res.three_quad_energy >> 32), 0, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)res.three_quad_energy, 0, 64);
    if (inv) {
        // the reconstruction was stored by the lanes of the transform block: a workgroup-scope fence (wait for the stores, same
        // CU) makes it readable by this wave's other lanes — no agent-scope cache write-back / invalidate here
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    }
    const int64_t err = (int64_t)(e2 >> 2);
    return err > 1 ? err : 1;
}

// ---- tpl level 3 (tpl_ctrls.subpel_depth == QUARTER_PEL): tpl_subpel_search and the compensation of a fractional vector.
// One wave = one 16x16 block: lane (lr, lc) owns the four samples of row lr from column lc on.
__device__ const int16_t TPL_REGULAR[16][8] = {  // sub_pel_filters_8 (EIGHTTAP_REGULAR, inter_prediction.c:223-240)
    {0, 0, 0, 128, 0, 0, 0, 0},      {0, 2, -6, 126, 8, -2, 0, 0},    {0, 2, -10, 122, 18, -4, 0, 0},  {0, 2, -12, 116, 28, -8, 2, 0},
    {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0},  {0, 2, -14, 84, 66, -12, 2, 0},
    {0, 2, -14, 76, 76, -14, 2, 0},  {0, 2, -12, 66, 84, -14, 2, 0},  {0, 2, -12, 58, 94, -16, 2, 0},  {0, 2, -12, 48, 102, -14, 2, 0},
    {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0},  {0, 0, -4, 18, 122, -10, 2, 0},  {0, 0, -2, 8, 126, -6, 2, 0}};
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t byte_of(uint32_t w, int i) { return (w >> (8 * i)) & 0xffu; }
// svt_aom_sub_pixel_variance16x16_c (variance.c:28-68, 303-318) of the 16x16 block at `a` (this lane's first sample at the integer
// position) with the bilinear phase (xo, yo) in eighths, against the lane's source samples; phase (0, 0) is svt_aom_variance16x16_c
__device__ __forceinline__ uint32_t subpel_var16(const uint8_t *a, ptrdiff_t stride, int xo, int yo, uint32_t spx) {
    const int32_t  f0 = 128 - 16 * xo, f1 = 16 * xo, g0 = 128 - 16 * yo, g1 = 16 * yo;
    const uint32_t r0 = ld4(a), e0 = ld8(a + 4), r1 = ld4(a + stride), e1 = ld8(a + stride + 4);
    int32_t        sum = 0;
    uint32_t       sse = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int32_t a0 = (int32_t)byte_of(r0, k), a1 = k < 3 ? (int32_t)byte_of(r0, k + 1) : (int32_t)e0;
        const int32_t b0 = (int32_t)byte_of(r1, k), b1 = k < 3 ? (int32_t)byte_of(r1, k + 1) : (int32_t)e1;
        const int32_t h0 = (a0 * f0 + a1 * f1 + 64) >> 7, h1 = (b0 * f0 + b1 * f1 + 64) >> 7;
        const int32_t p  = ((h0 * g0 + h1 * g1 + 64) >> 7) & 0xff;
        const int32_t d  = p - (int32_t)byte_of(spx, k);
        sum += d, sse += (uint32_t)(d * d);
    }
    sum = wave_sum_i32(sum), sse = wave_sum(sse);
    return sse - (uint32_t)(((int64_t)sum * sum) >> 8);
}
// tpl_subpel_search (src_ops_process.c:418-517) = svt_av1_find_best_sub_pixel_tree_pruned (mcomp.c:609-695) as the dispenser configures
// it: two rounds (half, quarter) of the four cardinal neighbours of the round's start vector, no vector cost, no diagonal / second
// level (skip_diag_refinement 4), candidates outside the sub-pel limits skipped.  mx / my: in the clipped full-pel vector (1/8 units),
// out the refined one; uniform over the wave.  ref_blk = this lane's first sample of the co-located block in the reference.
__device__ void tpl_subpel(const uint8_t *ref_blk, ptrdiff_t stride, int x, int y, int mi_rows, int mi_cols, uint32_t spx, int &mx, int &my) {
    const int mi_row = y >> 2, mi_col = x >> 2, mi = 4;
    int row_min = -(((mi_row + mi) * 4) + 4), col_min = -(((mi_col + mi) * 4) + 4), row_max = (mi_rows - mi_row) * 4 + 4, col_max = (mi_cols - mi_col) * 4 + 4;
    col_min = max(col_min, -1023), row_min = max(row_min, -1023), col_max = min(col_max, 1023), row_max = min(row_max, 1023);  // MAX_FULL_PEL_VAL; MV_LOW / MV_UPP lie outside
    const int sc_min = col_min * 8, sc_max = col_max * 8, sr_min = row_min * 8, sr_max = row_max * 8;
    int       best_r = (my >> 3) * 8, best_c = (mx >> 3) * 8;
    uint32_t  besterr = subpel_var16(ref_blk + (ptrdiff_t)(best_r >> 3) * stride + (best_c >> 3), stride, 0, 0, spx);
    int       start_r = best_r, start_c = best_c;
    for (int iter = 0, hstep = 4; iter < 2; iter++, hstep >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {  // left, right, up, down
            const int cr = start_r + (k == 2 ? -hstep : (k == 3 ? hstep : 0)), cc = start_c + (k == 0 ? -hstep : (k == 1 ? hstep : 0));
            if (cc < sc_min || cc > sc_max || cr < sr_min || cr > sr_max)
                continue;
            const uint32_t cost = subpel_var16(ref_blk + (ptrdiff_t)(cr >> 3) * stride + (cc >> 3), stride, cc & 7, cr & 7, spx);
            if (cost < besterr)
                besterr = cost, best_r = cr, best_c = cc;
        }
        start_r = best_r, start_c = best_c;
    }
    mx = best_c, my = best_r;
}
// svt_aom_enc_make_inter_predictor of the 16x16 luma block (:814-850): regular 8-tap kernels, the vector clamped as
// clamp_mv_to_umv_border_sb does with the xd of init_xd_tpl.  The 2-D form (round_0 = 3, round_1 = 11) is used for every fractional
// vector: with the unit kernel of phase 0 on one axis it returns what svt_av1_convolve_x_sr / _y_sr return (the offsets cancel).
// Returns this lane's four samples.  Every lane filters the eight rows its samples need in registers: a version that shared the
// horizontal pass through LDS (23 x 16 intermediates) cost fewer instructions but is not needed at this level's speed.
__device__ uint32_t tpl_compensate(const uint8_t *plane0, ptrdiff_t stride, int x, int y, int mvx, int mvy, int mi_rows, int mi_cols) {
    const int     lane = threadIdx.x, lr = lane >> 2, lc = (lane & 3) * 4;
    const int     mirow = y >> 2, micol = x >> 2, bmi = 4;
    const int32_t to_top = -((mirow * 4) * 8), to_bottom = ((mi_rows - bmi - mirow) * 4) * 8, to_left = -((micol * 4) * 8), to_right = ((mi_cols - bmi - micol) * 4) * 8;
    const int32_t spel_left = (4 + 16) << 4, spel_right = spel_left - 16, spel_top = (4 + 16) << 4, spel_bottom = spel_top - 16;
    int           col = (int16_t)(mvx * 2), row = (int16_t)(mvy * 2);
    col = max(to_left * 2 - spel_left, min(col, to_right * 2 + spel_right)), row = max(to_top * 2 - spel_top, min(row, to_bottom * 2 + spel_bottom));
    const int      sx = col & 15, sy = row & 15;
    const uint8_t *p = plane0 + (ptrdiff_t)(y + (row >> 4)) * stride + x + (col >> 4);
    if (!sx && !sy)
        return ld4(p + (ptrdiff_t)lr * stride + lc);
    int32_t fx[8], fy[8];
#pragma unroll
    for (int t = 0; t < 8; t++) fx[t] = __builtin_amdgcn_readfirstlane((int)TPL_REGULAR[sx][t]), fy[t] = __builtin_amdgcn_readfirstlane((int)TPL_REGULAR[sy][t]);
    // per lane: the eight rows its four samples need, each filtered horizontally in registers (three dwords of a row), then the vertical taps
    int32_t v4[4] = {(1 << 19) + 1024 - ((((1 << 8) + (1 << 7))) << 11), 0, 0, 0};
    v4[1] = v4[2] = v4[3] = v4[0];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint8_t *q = p + (ptrdiff_t)(lr - 3 + t) * stride + lc - 3;
        const uint32_t w0 = ld4(q), w1 = ld4(q + 4), w2 = ld4(q + 8);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int32_t acc = (1 << 14) + 4;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = k + u;
                acc += fx[u] * (int32_t)byte_of(i < 4 ? w0 : (i < 8 ? w1 : w2), i & 3);
            }
            v4[k] += fy[t] * (int32_t)(int16_t)(acc >> 3);
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int32_t v = v4[k] >> 11;
        v         = v < 0 ? 0 : (v > 255 ? 255 : v);
        out |= (uint32_t)v << (8 * k);
    }
    return out;
}

__device__ __forceinline__ int64_t max64(int64_t x, int64_t y) { return x > y ? x : y; }

// One block: (x, y) its origin.  Dependency flags live on the 16x16-cell grid (a block of 32x32 covers four cells).
template <int BS, int SUB, bool QPEL = false>
__device__ __forceinline__ void tpl_block(const TplArgs &a, const uint32_t x, const uint32_t y, int32_t *tile, uint8_t *nb) {
    constexpr int NDW = BS * BS / 256;     // dwords of the block a lane holds: 1 (16x16) or 4 (32x32)
    constexpr int LPR = BS / (4 * NDW);    // lanes per block row: 4 or 2
    static_assert((64 / BS) * (BS >> SUB) * (BS + 1) <= 4 * 16 * 17, "transform tile");
    static_assert(!QPEL || (BS == 16 && SUB == 0), "quarter-pel refinement comes with 16x16 blocks only (set_tpl_params)");
    const int mi_rows = (int)(((a.H + 7) & ~7u) >> 2), mi_cols = (int)(((a.W + 7) & ~7u) >> 2);  // Av1Common of the picture
    const SvtHipTplFrameJob &j = a.j;
    const int      lane = threadIdx.x;
    const uint32_t cx = x >> 4, cy = y >> 4, cell0 = cy * a.a16 + cx;
    if (x + BS / 2 > a.W || y + BS / 2 > a.H)  // at least half of the block inside
        return;
    const uint32_t ss = j.src.stride, rs = j.recon.stride;
    const uint8_t *src = a.src0 + (size_t)y * ss + x;
    uint8_t       *dst = a.rec0 + (size_t)y * rs + x;
    SvtHipTplSrcStats *sst = &j.src_stats[cell0];
    const int      lr = lane / LPR, lc = (lane % LPR) * 4 * NDW;  // this lane's samples: row lr, columns lc .. lc + 4 NDW - 1
    uint32_t       spx[NDW];
#pragma unroll
    for (int k = 0; k < NDW; k++) spx[k] = ld4(src + (size_t)lr * ss + lc + 4 * k);

    int64_t  srcrf_dist = 0, recon_error = 1;
    uint64_t best_ref_poc = 0;
    int32_t  best_rf_idx = -1;
    int      mv_row = 0, mv_col = 0;
    uint32_t best_mode = 0;
    bool     nonzero = false;
    if (!j.src_data_ready) {
        int64_t best_inter = INT64_MAX, best_intra = INT64_MAX;
        if (!j.disable_intra_pred) {
            const uint32_t dc = dc_value<BS>(a.src0, ss, x, y, a.W, a.H, nb);
            uint32_t       sad = 0;
#pragma unroll
            for (int k = 0; k < NDW; k++) sad = __builtin_amdgcn_sad_u8(spx[k], dc * 0x01010101u, sad);
            best_intra = wave_sum(sad);
        }
        const uint32_t sb = (y >> 6) * ((((a.W + 7) & ~7u) + 63) >> 6) + (x >> 6);
        // tpl_blk_idx_tab[1]: the 16x16 / 32x32 PU of the open-loop ME results
        uint32_t me_off = BS == 32 ? 1 + ((y >> 5) & 1) * 2 + ((x >> 5) & 1) : 5 + ((y >> 4) & 3) * 4 + ((x >> 4) & 3);
        if (!j.enable_me_16x16)
            me_off = (me_off - 1) / 4;
        const size_t   pu = (size_t)sb * j.stored_pus + me_off;
        const uint8_t *cands = j.me_candidate_array + pu * j.max_cand;
        const uint32_t n_cand = j.i_slice ? 0 : j.total_me_candidate_index[pu];
        // the candidates are prepared side by side, one per lane (candidate byte -> reference -> vector -> clip: three dependent
        // loads that cost a round trip each when taken one candidate after the other); only the SADs run in sequence
        for (uint32_t base = 0; base < n_cand; base += 64) {
            const uint32_t ci = base + lane;
            bool           ok = false;
            int            mx = 0, my = 0;
            uint32_t       rfi = 0;
            if (ci < n_cand) {
                const uint32_t cb = cands[ci], dir = cb & 3;
                if (dir <= 1) {  // single-reference candidates only
                    const uint32_t      ri = dir == 0 ? (cb >> 2) & 3 : (cb >> 4) & 3;
                    const SvtHipTplRef &rf = j.ref[dir][ri];
                    if (rf.usable) {
                        const uint32_t mv = j.me_mv_array[pu * j.max_refs + (dir ? j.max_l0 : 0) + ri];
                        mx = (int16_t)((int16_t)(mv & 0xffff) << 3), my = (int16_t)((int16_t)(mv >> 16) << 3);
                        if ((int)x + (mx >> 3) < -TPL_PAD)
                            mx = (int16_t)((-TPL_PAD - (int)x) << 3);
                        if ((int)x + BS + (mx >> 3) > TPL_PAD + (int)rf.max_width - 1)
                            mx = (int16_t)(((TPL_PAD + (int)rf.max_width - 1) - ((int)x + BS)) << 3);
                        if ((int)y + (my >> 3) < -TPL_PAD)
                            my = (int16_t)((-TPL_PAD - (int)y) << 3);
                        if ((int)y + BS + (my >> 3) > TPL_PAD + (int)rf.max_height - 1)
                            my = (int16_t)(((TPL_PAD + (int)rf.max_height - 1) - ((int)y + BS)) << 3);
                        ok = true, rfi = dir * 4 + ri;
                    }
                }
            }
            for (uint64_t todo = __ballot(ok); todo; todo &= todo - 1) {
                const int           src_lane = __builtin_ctzll(todo);
                int                 cmx = __shfl(mx, src_lane, 64), cmy = __shfl(my, src_lane, 64);
                const uint32_t      crf = (uint32_t)__shfl((int)rfi, src_lane, 64);
                const SvtHipTplRef &rf  = j.ref[crf >> 2][crf & 3];
                uint32_t            sad = 0;
                if constexpr (QPEL) {  // tpl_subpel_search, then the SAD against the compensated block when the vector is fractional
                    tpl_subpel(rf.src + ((ptrdiff_t)y + lr) * (ptrdiff_t)rf.src_stride + (ptrdiff_t)x + lc, (ptrdiff_t)rf.src_stride, (int)x, (int)y, mi_rows, mi_cols,
                               spx[0], cmx, cmy);
                    const uint32_t pv = ((cmx | cmy) & 7) ? tpl_compensate(rf.src, (ptrdiff_t)rf.src_stride, (int)x, (int)y, cmx, cmy, mi_rows, mi_cols)
                                                          : ld4(rf.src + ((ptrdiff_t)y + cmy / 8 + lr) * (ptrdiff_t)rf.src_stride + (ptrdiff_t)x + cmx / 8 + lc);
                    sad = __builtin_amdgcn_sad_u8(spx[0], pv, 0u);
                } else {
                const uint8_t *rp = rf.src + ((ptrdiff_t)y + cmy / 8 + lr) * (ptrdiff_t)rf.src_stride + (ptrdiff_t)x + cmx / 8 + lc;
#pragma unroll
                for (int k = 0; k < NDW; k++) sad = __builtin_amdgcn_sad_u8(spx[k], ld4(rp + 4 * k), sad);
                }
                const int64_t cost = wave_sum(sad);
                if (cost < best_inter)
                    best_inter = cost, best_ref_poc = rf.picture_number, best_rf_idx = (int32_t)crf, mv_row = cmy, mv_col = cmx;
            }
        }
        if (best_inter < best_intra)
            best_mode = NEWMV_MODE;
        if (best_mode == NEWMV_MODE) {
            const SvtHipTplRef &rf = j.ref[best_rf_idx < 4 ? 0 : 1][best_rf_idx & 3];
            const uint8_t      *rp = rf.src + ((ptrdiff_t)y + (mv_row >> 3)) * (ptrdiff_t)rf.src_stride + (ptrdiff_t)x + (mv_col >> 3);
            uint32_t            rps = rf.src_stride;
            if (QPEL && ((mv_col | mv_row) & 7)) {
                // the source-based residual of a fractional vector is taken against the compensated block: it is parked in the block's
                // own (not yet written) reconstruction area, which the transform block reads like any other prediction
                const uint32_t pv = tpl_compensate(rf.src, (ptrdiff_t)rf.src_stride, (int)x, (int)y, mv_col, mv_row, mi_rows, mi_cols);
                uint8_t       *o2 = dst + (size_t)lr * rs + lc;
                o2[0] = (uint8_t)pv, o2[1] = (uint8_t)(pv >> 8), o2[2] = (uint8_t)(pv >> 16), o2[3] = (uint8_t)(pv >> 24);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __syncthreads();
                rp = dst, rps = rs;
            }
            srcrf_dist = (quantize_error<BS, SUB>(a, src, ss, rp, rps, nullptr, 0, false, tile, &nonzero) << 4) << SUB;
        }
        if (j.store_src_stats && lane == 0) {
            SvtHipTplSrcStats s;
            memset(&s, 0, sizeof(s));
            s.srcrf_dist = srcrf_dist, s.srcrf_rate = 0, s.mv_row = (int16_t)mv_row, s.mv_col = (int16_t)mv_col, s.best_rf_idx = best_rf_idx;
            s.ref_frame_poc = best_ref_poc, s.best_mode = (uint8_t)best_mode, s.best_intra_mode = 0;
            *sst = s;
        }
    } else {
        srcrf_dist = sst->srcrf_dist, mv_row = sst->mv_row, mv_col = sst->mv_col, best_rf_idx = sst->best_rf_idx;
        best_ref_poc = sst->ref_frame_poc, best_mode = sst->best_mode;
    }
    // ---- reconstruction path
    uint8_t *o = dst + (size_t)lr * rs + lc;
    if (best_mode == NEWMV_MODE) {
        const SvtHipTplRef &rf = j.ref[best_rf_idx < 4 ? 0 : 1][best_rf_idx & 3];
        const uint8_t      *rp = rf.recon + ((ptrdiff_t)y + (mv_row >> 3) + lr) * (ptrdiff_t)rf.recon_stride + (ptrdiff_t)x + (mv_col >> 3) + lc;
        uint32_t            qv = 0;
        const bool          frac = QPEL && ((mv_col | mv_row) & 7);
        if (frac) {
            qv = tpl_compensate(rf.recon, (ptrdiff_t)rf.recon_stride, (int)x, (int)y, mv_col, mv_row, mi_rows, mi_cols);
        }
#pragma unroll
        for (int k = 0; k < NDW; k++) {
            const uint32_t v = frac ? qv : ld4(rp + 4 * k);
            o[4 * k] = (uint8_t)v, o[4 * k + 1] = (uint8_t)(v >> 8), o[4 * k + 2] = (uint8_t)(v >> 16), o[4 * k + 3] = (uint8_t)(v >> 24);
        }
    } else {
        // the DC prediction reads reconstructed samples of the left / top / top-left blocks: wait for them
        if (lane == 0) {
            constexpr int NC = BS / 16;  // cells per block side
            uint32_t deps[2 * NC + 1];
            int      nd = 0;
            for (int k = 0; k < NC; k++) {
                if (cx)
                    deps[nd++] = cell0 + k * a.a16 - 1;   // left column
                if (cy)
                    deps[nd++] = cell0 - a.a16 + k;        // top row
            }
            if (cx && cy)
                deps[nd++] = cell0 - a.a16 - 1;
            for (int k = 0; k < nd; k++) {
                uint32_t spins = 0;
                while (__hip_atomic_load(&a.flags[deps[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > SPIN_LIMIT) {
                        atomicExch(a.error, 1u);
                        break;
                    }
                }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t dc = dc_value<BS>(a.rec0, rs, x, y, a.W, a.H, nb);
#pragma unroll
        for (int k = 0; k < 4 * NDW; k++) o[k] = (uint8_t)dc;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the prediction (this wave's stores) is in memory before the transform block reads it back
    __syncthreads();
    const bool inv = !j.disable_intra_pred || j.is_ref;
    recon_error = quantize_error<BS, SUB>(a, src, ss, dst, rs, dst, rs, inv, tile, &nonzero);  // (its fence also publishes the reconstruction)
    if (SUB && inv && nonzero) {
        // the rows the sub-sampled transform left out repeat the reconstructed row above them (:1162-1180; only when the block has
        // coefficients: otherwise every row keeps its prediction).  Row (lr & ~mask) was stored by other lanes of this wave: the
        // fence + barrier inside quantize_error completed those stores.
        constexpr int MASK = (1 << SUB) - 1;
        if (lr & MASK) {
            const uint8_t *from = dst + (size_t)(lr & ~MASK) * rs + lc;
#pragma unroll
            for (int k = 0; k < NDW; k++) {
                const uint32_t v = ld4(from + 4 * k);
                o[4 * k] = (uint8_t)v, o[4 * k + 1] = (uint8_t)(v >> 8), o[4 * k + 2] = (uint8_t)(v >> 16), o[4 * k + 3] = (uint8_t)(v >> 24);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
    }
    if (lane == 0) {
        SvtHipTplStats st;
        memset(&st, 0, sizeof(st));
        st.srcrf_dist = srcrf_dist, st.recrf_dist = (recon_error << 4) << SUB;
        if (best_mode != NEWMV_MODE)
            st.srcrf_dist = (recon_error << 4) << SUB;
        st.recrf_dist = max64(st.srcrf_dist, st.recrf_dist);
        if (!j.tpl_i_slice && best_rf_idx != -1)
            st.mv_row = (int16_t)mv_row, st.mv_col = (int16_t)mv_col, st.ref_frame_poc = best_ref_poc;
        st.srcrf_dist = max64(1, st.srcrf_dist), st.recrf_dist = max64(1, st.recrf_dist), st.srcrf_rate = 1, st.recrf_rate = 1;
        // result_model_store: the synthesizer's grid, statistics normalised to its cell size
        const uint32_t cell = j.synth_blk_size;
        const uint32_t aw = (a.W + 7) & ~7u, ah = (a.H + 7) & ~7u;
        const uint32_t gstride = cell == 32 ? (aw + 31) >> 5 : (cell == 16 ? a.a16 : a.a16 << 1);
        const uint32_t grows = cell == 32 ? (ah + 31) >> 5 : (cell == 16 ? a.rows16 : a.rows16 << 1);
        if (BS < 32 && cell == 32) {
            // four 16x16 blocks share the cell and the reference's last one (z-order: right, then below) stays: this block
            // writes only if no later one is dispensed (at least half inside the picture)
            const uint32_t x0 = x & ~31u, y0 = y & ~31u, me = ((y >> 4) & 1) * 2 + ((x >> 4) & 1);
            bool           last = true;
            for (uint32_t k = me + 1; k < 4; k++) last = last && !(x0 + (k & 1) * 16 + 8 <= a.W && y0 + (k >> 1) * 16 + 8 <= a.H);
            if (last)
                j.stats[(size_t)(y >> 5) * gstride + (x >> 5)] = st;
        } else {
            const uint32_t per = BS / cell ? BS / cell : 1;
            if (per > 1) {
                const int64_t div = (int64_t)per * per;
                st.srcrf_dist = max64(1, st.srcrf_dist / div), st.recrf_dist = max64(1, st.recrf_dist / div);
            }
            for (uint32_t gy = 0; gy < per; gy++)
                for (uint32_t gx = 0; gx < per; gx++)
                    if (x / cell + gx < gstride && y / cell + gy < grows)
                        j.stats[(size_t)(y / cell + gy) * gstride + x / cell + gx] = st;
        }
    }
    // publish the reconstruction to the other CUs / XCDs, then the flags of the cells this block covers.  An agent-scope release
    // fence writes back EVERY dirty line of this XCD's L2 (buffer_wbl2): one per block made 68 % of the kernel's time.  Instead
    // every lane stores its samples of the block once more with agent scope (sc1: written through to the device's coherence
    // point; the values are the ones already there), waits for those stores, and lane 0 sets the flags.  Needs 4-byte aligned
    // rows (a.coherent_rows); otherwise the fence.
    if (a.coherent_rows) {
#pragma unroll
        for (int k = 0; k < NDW; k++) {
            uint32_t *q = (uint32_t *)(o + 4 * k);
            __hip_atomic_store(q, __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // vmcnt(0): the write-through stores are acknowledged
        __syncthreads();
    }
    if (lane == 0) {
        if (!a.coherent_rows)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        for (int k = 0; k < (BS / 16) * (BS / 16); k++) {
            const uint32_t fx = cx + (k & (BS / 16 - 1)), fy = cy + k / (BS / 16);
            if (fx < a.a16 && fy < a.rows16)
                __hip_atomic_store(&a.flags[fy * a.a16 + fx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// MODE 0: every block 16x16, transform 16x16 (tpl level 4); MODE 2: the same with the quarter-pel refinement of level 3.  MODE 1 (level 5): complete 64x64 blocks are dispensed as 32x32
// blocks, incomplete ones (right / bottom picture edge) as 16x16 blocks (svt_aom_tpl_disp_kernel, :2043-2051), transform on every 4th
// row in both.  A workgroup takes the next block in raster order of the block origins from the ticket counter: every block a block
// waits for (left, top, top-left) has a lower ticket, so it is running or done whatever the dispatch order.
template <int MODE>
__global__ __launch_bounds__(64) void tpl_kernel(TplArgs a) {
    __shared__ int32_t  tile[4 * 16 * 17];
    __shared__ uint8_t  nb[8 + 88 + 88];
    __shared__ uint32_t s_ticket;
    if (threadIdx.x == 0)
        s_ticket = atomicAdd(a.error + 1, 1u);
    __syncthreads();
    const uint32_t t = s_ticket;
    if (MODE == 0) {
        tpl_block<16, 0>(a, (t % a.a16) * 16, (t / a.a16) * 16, tile, nb);
        return;
    }
    if (MODE == 2) {  // level 3: level 4 + quarter-pel refinement
        tpl_block<16, 0, true>(a, (t % a.a16) * 16, (t / a.a16) * 16, tile, nb);
        return;
    }
    const uint32_t aw = (a.W + 7) & ~7u, ah = (a.H + 7) & ~7u;
    const uint32_t nfc = aw >> 6, n16c = ((aw & 63) + 15) >> 4, nfr = ah >> 6;  // complete columns, 16-wide columns behind them, complete rows
    const uint32_t even = 2 * nfc + n16c, per_row = 2 * even + 2 * n16c;        // blocks that start in a 16-row with / of a complete 64-row
    if (t < nfr * per_row) {
        uint32_t u = t % per_row, q = 0;
        if (u >= even) {
            u -= even, q = 1;
            if (u >= n16c) {
                u -= n16c, q = 2;
                if (u >= even)
                    u -= even, q = 3;
            }
        }
        const uint32_t y = (t / per_row) * 64 + q * 16;
        if (!(q & 1) && u < 2 * nfc)
            tpl_block<32, 2>(a, u * 32, y, tile, nb);
        else
            tpl_block<16, 2>(a, nfc * 64 + (u - ((q & 1) ? 0 : 2 * nfc)) * 16, y, tile, nb);
    } else {
        const uint32_t v = t - nfr * per_row;
        tpl_block<16, 2>(a, (v % a.a16) * 16, nfr * 64 + (v / a.a16) * 16, tile, nb);
    }
}

size_t flag_bytes(uint32_t width, uint32_t height) {
    const uint32_t aw = (width + 7) & ~7u, ah = (height + 7) & ~7u;
    return ((size_t)((aw + 15) >> 4) * ((ah + 15) >> 4) * sizeof(uint32_t) + 255) & ~(size_t)255;
}

}  // namespace

extern "C" int32_t svt_hip_tpl_dispenser_frame(const SvtHipTplFrameJob *job, void *stream) {
    auto bad = [](const char *m) {
        set_error("svt_hip_tpl_dispenser_frame: %s", m);
        return (int32_t)SVT_HIP_ERR_BAD_PARAMETER;
    };
    if (!job)
        return bad("NULL job");
    const SvtHipPlane8 &s = job->src, &r = job->recon;
    if (!s.buf || !r.buf || s.width < 16 || s.height < 16 || r.width != s.width || r.height != s.height)
        return bad("source / reconstruction planes missing or of different size");
    if (s.org_x < TPL_PAD || s.org_y < TPL_PAD || r.org_x < TPL_PAD || r.org_y < TPL_PAD)
        return bad("planes need >= 32 samples of padding (TPL_PADX / TPL_PADY: the clipped vectors reach that far, reference_object.c:439-442)");
    if (s.stride < s.width + 2u * s.org_x || r.stride < r.width + 2u * r.org_x)
        return bad("stride smaller than the padded width");
    const bool b32 = job->blk_size == 32;
    if (job->blk_size != 0 && job->blk_size != 16 && !b32)
        return bad("blk_size must be 16 (or 0) or 32");
    if (job->quarter_pel > 1 || (job->quarter_pel && b32))
        return bad("quarter_pel is 0 or 1, and comes with 16x16 blocks only (set_tpl_params)");
    if (b32 ? job->subsample_tx != 2 : job->subsample_tx != 0)
        return bad("subsample_tx must be 0 with 16x16 blocks and 2 with 32x32 blocks");
    if (job->synth_blk_size != 16 && job->synth_blk_size != 8 && !(b32 && job->synth_blk_size == 32))
        return bad("synth_blk_size must be 16 or 8 (or 32 with 32x32 blocks)");
    if (job->pf_shape > 2)
        return bad("pf_shape must be 0, 1 or 2");
    if (!job->stats || !job->src_stats)
        return bad("NULL output array");
    if (!job->workspace || job->workspace_bytes < svt_hip_tpl_workspace_bytes(s.width, s.height))
        return bad("workspace too small (svt_hip_tpl_workspace_bytes)");
    const bool need_me = !job->src_data_ready && !job->i_slice;
    if (need_me && (!job->me_mv_array || !job->me_candidate_array || !job->total_me_candidate_index || !job->max_cand || !job->max_refs || !job->stored_pus))
        return bad("ME results missing");
    for (int l = 0; l < SVT_HIP_ME_MAX_LIST; l++)
        for (int q = 0; q < SVT_HIP_ME_MAX_REF; q++) {
            const SvtHipTplRef &f = job->ref[l][q];
            if (!f.src) {
                // the kernel gates on `usable` alone and reads src / recon of every usable reference
                if (f.usable)
                    return bad("reference picture marked usable but its src plane is NULL");
                continue;
            }
            if (!f.recon || f.src_stride < s.width + 2u * TPL_PAD || f.recon_stride < s.width + 2u * TPL_PAD)
                return bad("reference picture: reconstruction missing or stride too small");
            // the clipped vector keeps a block inside max_width + TPL_PAD: that must be inside the padded plane
            if (f.max_width > s.width || f.max_height > s.height)
                return bad("reference picture: max_width / max_height beyond the picture (its planes must be padded by >= 32 samples around them)");
        }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t    st = resolve_stream(stream);
    const uint32_t aw = (s.width + 7) & ~7u, ah = (s.height + 7) & ~7u, a16 = (aw + 15) >> 4, rows16 = (ah + 15) >> 4;
    // level 5: 32x32 blocks in the complete 64x64 blocks, 16x16 blocks in the incomplete ones (see tpl_kernel)
    const uint32_t nfc = aw >> 6, n16c = ((aw & 63) + 15) >> 4, nfr = ah >> 6;
    const size_t   blocks = b32 ? (size_t)nfr * (4 * nfc + 4 * n16c) + (size_t)(rows16 - 4 * nfr) * a16 : (size_t)a16 * rows16;
    uint8_t       *ws = (uint8_t *)job->workspace;
    const size_t   fb = flag_bytes(s.width, s.height);
    SVT_HIP_CHECK(hipMemsetAsync(ws, 0, fb + 256, st));
    TplArgs a;
    a.j = *job;
    a.src0 = s.buf + (size_t)s.org_y * s.stride + s.org_x, a.rec0 = r.buf + (size_t)r.org_y * r.stride + r.org_x;
    a.W = s.width, a.H = s.height, a.a16 = a16, a.rows16 = rows16, a.coherent_rows = (r.stride % 4 == 0 && ((uintptr_t)(r.buf + (size_t)r.org_y * r.stride + r.org_x) % 4) == 0 && !job->publish_fence) ? 1u : 0u, a.flags = (uint32_t *)ws, a.error = (uint32_t *)(ws + fb);
    if (b32)
        hipLaunchKernelGGL((tpl_kernel<1>), dim3((uint32_t)blocks), dim3(64), 0, st, a);
    else if (job->quarter_pel)
        hipLaunchKernelGGL((tpl_kernel<2>), dim3((uint32_t)blocks), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((tpl_kernel<0>), dim3((uint32_t)blocks), dim3(64), 0, st, a);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" uint64_t svt_hip_tpl_workspace_bytes(uint32_t width, uint32_t height) {
    return flag_bytes(width, height) + 256;
}
extern "C" uint64_t svt_hip_tpl_status_offset(uint32_t width, uint32_t height) { return flag_bytes(width, height); }

SVT_HIP_MODULE_WARMUP(tpl)
