// txfm_block.hpp — the fused transform block of txfm.hip as a device function: [source - prediction ->] forward 2-D ->
// 64-pt energy / repack -> [SATD ->] quantise -> eob -> inverse 2-D -> add + clip for ONE transform block held by L = max(W, H)
// adjacent lanes of a wavefront.  Used by txfm_kernel (txfm.hip: batches of blocks) and by the TPL dispenser (tpl.hip: one
// block per wave inside its dependency-ordered kernel).  See txfm.hip for the mapping.
#pragma once
#include <cstdint>

#include "../../include/svt_hip_txfm.h"
#include "common.hpp"
#include "txfm_device.hpp"

namespace svthip {
namespace txb {
using namespace svthip::txd;


// ---- per-size constants (transforms.h:26-49, inv_transforms.c:17-35; see oracle/src/orc_txfm.c) ----
constexpr int8_t FWD_SHIFT[5][5][3] = {
    {{2, 0, 0}, {2, -1, 0}, {2, -1, 0}, {0, 0, 0}, {0, 0, 0}},   {{2, -1, 0}, {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, 0, 0}},
    {{2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, 0}}, {{0, 0, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0}, {0, -2, -2}},
    {{0, 0, 0}, {0, 0, 0}, {2, -4, 0}, {2, -4, -2}, {0, -2, -2}}};
constexpr int8_t FWD_COS_COL[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
constexpr int8_t FWD_COS_ROW[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
constexpr int8_t INV_SHIFT0[5][5]  = {{0, 0, -1, 0, 0}, {0, -1, -1, -2, 0}, {-1, -1, -2, -1, -2}, {0, -2, -1, -2, -1}, {0, 0, -2, -1, -2}};
__device__ const uint8_t VTX_D[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
__device__ const uint8_t HTX_D[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int pow2floor(int v) {
    int p = 1;
    while (p * 2 <= v) p *= 2;
    return p;
}
#ifndef SVT_HIP_TX64_WAVES
#define SVT_HIP_TX64_WAVES 1  // waves per SIMD the 64-point kernels are compiled for (1: no cap, 211 VGPRs = 2 waves)
#endif
template <int W, int H>
struct Geo {
    static constexpr int L  = cmax(W, H);
    static constexpr int PW = W + 1;
    static constexpr int NT = cmax(1, 64 / L);  // one wave per workgroup: its barriers cost nothing and waves never wait for each other
    static constexpr int IW = cmin(W, 32), IH = cmin(H, 32);
    static constexpr int MINW = L == 32 ? 3 : (L == 8 ? 8 : (L == 64 ? SVT_HIP_TX64_WAVES : 1));  // waves per SIMD the register allocation aims for: 3 for the 32-wide kernels (150 VGPRs, no scratch: 2.86 TB/s; at 4 waves / 128 VGPRs they spill 104 B per lane: 2.31 TB/s); 8 for the 8-wide ones (64 VGPRs, 12 B of scratch: +4 %); the 16-wide take what they need (96 VGPRs, 5 waves: capped at 80 they spill 64 B and lose 15 %)
    static constexpr int WI = clog2(W) - 2, HI = clog2(H) - 2;
    static constexpr bool RECT = (W == 2 * H) || (H == 2 * W);
};

// ---- quantiser (per coefficient; semantics of full_loop.c:25-75, 145-194, 278-338, 383-449) ----
struct QP {
    int32_t        zbin[2], round[2], quant[2], qshift[2], dequant[2];
    int32_t        log_scale, mode;
    const uint8_t *qm, *iqm;
    bool           simple;  // no quantisation matrices and 16-bit table entries: quant_small() applies to small coefficients
};
__device__ __forceinline__ int32_t rpot(int32_t v, int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ void load_qp(QP &q, const SvtHipTxfmDesc &d, const uint8_t *base) {
    q.log_scale = d.log_scale, q.mode = d.quant_mode;
    bool small = true;
    for (int i = 0; i < 2; i++) {
        q.zbin[i]    = rpot(d.zbin[i], d.log_scale);
        q.round[i]   = rpot(d.round[i], d.log_scale);
        q.quant[i]   = d.quant[i];
        q.qshift[i]  = d.quant_shift[i];
        q.dequant[i] = d.dequant[i];
        // the ranges the reference's int16 tables can hold
        small = small && q.zbin[i] >= 0 && q.zbin[i] < 65536 && q.round[i] >= 0 && q.round[i] < 32768 && q.quant[i] >= -32768 && q.quant[i] < 32768 &&
            q.qshift[i] >= 0 && q.qshift[i] < 65536 && q.dequant[i] >= 0 && q.dequant[i] < 32768;
    }
    q.qm  = d.qm_off == SVT_HIP_NO_OFFSET ? nullptr : base + d.qm_off;
    q.iqm = d.iqm_off == SVT_HIP_NO_OFFSET ? nullptr : base + d.iqm_off;
    q.simple = small && !q.qm && !q.iqm && d.log_scale >= 0 && d.log_scale <= 2;
}
__device__ __forceinline__ int64_t clamp_i16(int64_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
// a * b with the exact 64-bit result; when a fits 32 bits (always, for coefficients in the range a bit depth can produce)
// this is one v_mad_i64_i32 instead of the four quarter-rate multiplies of a generic 64 x 64 product
__device__ __forceinline__ int64_t mul_64x32(int64_t a, int32_t b) {
    return a == (int64_t)(int32_t)a ? (int64_t)(int32_t)a * (int64_t)b : a * (int64_t)b;
}
__device__ __forceinline__ void quant_one(const QP &q, int32_t c, uint32_t rc, int32_t &qc, int32_t &dqc) {
    const int     ac   = rc != 0;
    const int32_t sign = c < 0 ? -1 : 0;
    const int32_t absc = (c ^ sign) - sign;
    const int32_t wt = q.qm ? q.qm[rc] : 32, iwt = q.iqm ? q.iqm[rc] : 32;
    const int     ls = q.log_scale;
    int32_t       aq = 0, dq = q.dequant[ac];
    if (q.mode == SVT_HIP_QUANT_B) {
        if (mul32(absc, wt) >= (q.zbin[ac] << 5)) {
            int64_t tmp = clamp_i16((int64_t)add32(absc, q.round[ac]));
            tmp *= wt;
            aq = (int32_t)(mul_64x32((mul_64x32(tmp, q.quant[ac]) >> 16) + tmp, q.qshift[ac]) >> (16 - ls + 5));
            dq = (q.dequant[ac] * iwt + 16) >> 5;
        }
    } else if (q.mode == SVT_HIP_QUANT_B_HBD) {
        const int32_t cw = mul32(c, wt);
        if (cw >= q.zbin[ac] * 32 || cw <= -q.zbin[ac] * 32) {
            const int64_t tmpw = mul_64x32((int64_t)absc + q.round[ac], wt);
            const int64_t tmp2 = (mul_64x32(tmpw, q.quant[ac]) >> 16) + tmpw;
            aq = (int32_t)(mul_64x32(tmp2, q.qshift[ac]) >> (16 - ls + 5));
            dq = (q.dequant[ac] * iwt + 16) >> 5;
        }
    } else if (q.mode == SVT_HIP_QUANT_FP) {
        if (!q.qm && !q.iqm) {
            if (((int64_t)absc << (1 + ls)) >= (int64_t)q.dequant[ac]) {
                const int64_t a2 = clamp_i16((int64_t)absc + q.round[ac]);
                aq = (int32_t)(mul_64x32(a2, q.quant[ac]) >> (16 - ls));
            }
        } else {
            dq = (q.dequant[ac] * iwt + 16) >> 5;
            if ((int64_t)absc * wt >= (int64_t)(q.dequant[ac] << (5 - (1 + ls)))) {
                const int64_t a2 = clamp_i16((int64_t)absc + q.round[ac]);
                aq = (int32_t)(mul_64x32(a2 * wt, q.quant[ac]) >> (16 - ls + 5));
            }
        }
    } else {  // SVT_HIP_QUANT_FP_HBD
        if (q.qm || q.iqm) {
            dq = (q.dequant[ac] * iwt + 16) >> 5;
            if ((int64_t)absc * wt >= (int64_t)(q.dequant[ac] << (5 - (1 + ls)))) {
                const int64_t tmp = (int64_t)absc + q.round[ac];
                aq = (int32_t)(mul_64x32(mul_64x32(tmp, q.quant[ac]), wt) >> (16 - ls + 5));
            }
        } else {
            if ((int32_t)((uint32_t)absc << (1 + ls)) >= q.dequant[ac]) {
                const int64_t tmp = (int64_t)absc + q.round[ac];
                aq = (int32_t)(mul_64x32(tmp, q.quant[ac]) >> (16 - ls));
            }
        }
    }
    qc                = (aq ^ sign) - sign;
    const int32_t adq = mul32(aq, dq) >> ls;
    dqc               = (adq ^ sign) - sign;
}

// quant_one for the common case: no quantisation matrix (weights 32, so the << 5 / >> 5 pairs cancel), 16-bit table entries
// (q.simple) and |c| < 2^15.  Every product then fits 32 bits except x * quant_shift (38 bits: one v_mad_i64_i32), where
//   (((t << 5) * quant) >> 16) + (t << 5)  ==  ((t * quant) >> 11) + (t << 5)      exactly (t * quant fits 32 bits),
// so the results are bit-identical to the general path at about a third of its instructions.
template <bool BTYPE>  // BTYPE: quantize_b family (zbin, quant + quant_shift); otherwise the fp family
__device__ __forceinline__ void quant_small(const QP &q, int32_t c, int ac, int32_t &qc, int32_t &dqc) {
    const int32_t sign = c >> 31;
    const int32_t absc = (c ^ sign) - sign;  // < 2^15
    const int     ls   = q.log_scale;
    int32_t       aq;
    if (BTYPE) {
        int32_t t = absc + q.round[ac];  // < 2^16
        t         = q.mode == SVT_HIP_QUANT_B && t > 32767 ? 32767 : t;
        const int32_t x = (mul_i24(t, q.quant[ac]) >> 11) + (t << 5);  // 0 <= x < 2^22
        // x * quant_shift is a 38-bit product of two 24-bit values: low word + high 16 bits by the full-rate 24-bit multipliers and
        // one v_alignbit for the shift, instead of a quarter-rate 64-bit multiply-add and a 64-bit shift
        const uint32_t lo = mul_u24((uint32_t)x, (uint32_t)q.qshift[ac]), hi = mulhi_u24((uint32_t)x, (uint32_t)q.qshift[ac]);
        aq                = (int32_t)__builtin_amdgcn_alignbit(hi, lo, (uint32_t)(21 - ls));
        aq              = absc >= q.zbin[ac] ? aq : 0;
    } else {
        int32_t a2 = absc + q.round[ac];
        a2         = q.mode == SVT_HIP_QUANT_FP && a2 > 32767 ? 32767 : a2;
        aq         = mul_i24(a2, q.quant[ac]) >> (16 - ls);
        aq         = (absc << (1 + ls)) >= q.dequant[ac] ? aq : 0;
    }
    qc                = (aq ^ sign) - sign;
    const int32_t adq = mul_i24(aq, q.dequant[ac]) >> ls;  // aq < 2^19, dequant < 2^15
    dqc               = (adq ^ sign) - sign;
}

// largest magnitude of N values: running signed maximum and minimum, two values per v_max3 / v_min3 (N instructions instead of the
// 3 N of |v| then max); hi >= 0 >= lo, so max(hi, -lo) is it (as unsigned: -INT_MIN is 2^31)
template <int N>
__device__ __forceinline__ uint32_t max_abs(const int32_t *v) {
    int32_t hi = 0, lo = 0;
#pragma unroll
    for (int i = 0; i + 1 < N; i += 2) {
        hi = max(max(v[i], v[i + 1]), hi);
        lo = min(min(v[i], v[i + 1]), lo);
    }
    if (N & 1)
        hi = max(v[N - 1], hi), lo = min(v[N - 1], lo);
    const uint32_t a = (uint32_t)hi, b = 0u - (uint32_t)lo;
    return a > b ? a : b;
}
template <int L>
__device__ __forceinline__ uint32_t group_max(uint32_t v) {
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v                = o > v ? o : v;
    }
    return v;
}
template <int L>
__device__ __forceinline__ uint32_t group_sum32(uint32_t v) {
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int L>
__device__ __forceinline__ uint64_t group_sum64(uint64_t v) {
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// highbd_clip_pixel_add with the check_range clamp of the residual (inv_transforms.c:2490-2520); everything fits 32 bits
// for bit depths up to 12.  Both clamps are one v_med3_i32 (the compiler only forms it for constant bounds).
struct PixelClip {
    int32_t mn, mx, hi;
    __device__ __forceinline__ explicit PixelClip(int bd) : mx((1 << (7 + bd)) - 1 + (914 << (bd - 7))), hi((1 << bd) - 1) { mn = -mx - 1; }
    __device__ __forceinline__ static int32_t med3(int32_t v, int32_t lo, int32_t up) {
        int32_t r;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(up));
        return r;
    }
    __device__ __forceinline__ uint16_t add(uint32_t dest, int32_t trans) const { return (uint16_t)med3((int32_t)dest + med3(trans, mn, mx), 0, hi); }
};

// Store the IW x IH coefficient tile held in LDS (row pitch PW) to a dense global array with all L lanes of the
// transform block: consecutive lanes write consecutive 16-byte chunks, so every wave-level store covers whole cache
// lines (a lane-per-row store touches 16 bytes of L different lines and amplifies the HBM write traffic ~4x).
template <int IW, int IH, int L, int PW>
__device__ __forceinline__ void coop_store_tile(const int32_t *__restrict__ lds, int32_t *__restrict__ g, int t) {
    constexpr int NCH = IW * IH / 4;
    if ((((uintptr_t)g) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < (NCH + L - 1) / L; k++) {
            const int q = k * L + t;
            if (q < NCH) {
                const int e = 4 * q, r = e / IW, c = e % IW;
                int4      v;
                v.x = lds[r * PW + c], v.y = lds[r * PW + c + 1], v.z = lds[r * PW + c + 2], v.w = lds[r * PW + c + 3];
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i nv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(nv, (v4i *)g + q);  // written once, never read by this kernel: keep it out of the caches
            }
        }
    } else {
        for (int e = t; e < IW * IH; e += L) g[e] = lds[(e / IW) * PW + e % IW];
    }
}

// One transform block: `d` its descriptor, `result` where its SvtHipTxfmResult goes, `live` false for a lane group without a
// block (it still takes part in the barriers), t = lane index inside the group (0 .. L-1), lds = the group's H x (W+1) int32 tile.
// Every lane of the workgroup must call this (it contains __syncthreads).
// BLKERR (the TPL dispenser's get_quantize_error, src_ops_process.c:225-249; sizes without a 64-point side): result->three_quad_energy
// = svt_av1_block_error = sum of (coeff - dqcoeff)^2 over the block, taken in the quantiser loop where both values are in registers, so
// that neither array has to be stored and read back (coeff_off / dqcoeff_off may be SVT_HIP_NO_OFFSET).
template <int W, int H, bool BLKERR = false>
__device__ __forceinline__ void txfm_block(uint8_t *__restrict__ base, const SvtHipTxfmDesc &d, SvtHipTxfmResult *__restrict__ result,
                                           const bool live, const int t, int32_t *__restrict__ lds) {
    static_assert(!BLKERR || (W != 64 && H != 64), "three_quad_energy is the 64-point energy for those sizes");
    using G = Geo<W, H>;
    constexpr int L = G::L, PW = G::PW, IW = G::IW, IH = G::IH;
    const int vk = VTX_D[d.tx_type & 15], hk = HTX_D[d.tx_type & 15];
    const bool ud = vk == 2, lr = hk == 2;
    const int  bd = d.bit_depth;
    const int  bdi = bd == 8 ? 0 : (bd == 10 ? 1 : (bd == 12 ? 2 : -1));  // row of INV_FAST_OK
    const bool do_fwd = d.flags & SVT_HIP_TX_FWD, do_inv = d.flags & SVT_HIP_TX_INV;
    constexpr int sh0 = FWD_SHIFT[G::WI][G::HI][0], sh1 = FWD_SHIFT[G::WI][G::HI][1], sh2 = FWD_SHIFT[G::WI][G::HI][2];
    int32_t  row[W];
#pragma unroll
    for (int c = 0; c < W; c++) row[c] = 0;

    // ------------------------------------------------------------------ forward: columns
    if (do_fwd) {
        if (live && t < W) {
            int32_t        v[H];
            {   // FLIPADST columns read the rows bottom-up: walk the pointers, the register index stays constant
                const ptrdiff_t step = ud ? -(ptrdiff_t)d.residual_stride : (ptrdiff_t)d.residual_stride;
                const ptrdiff_t row0 = (ud ? (ptrdiff_t)(H - 1) * d.residual_stride : 0) + t;
                if (!(d.flags & SVT_HIP_TX_SRC_PRED)) {
                    const int16_t *rp = (const int16_t *)(base + d.residual_off) + row0;
#pragma unroll
                    for (int r = 0; r < H; r++, rp += step) v[r] = (int32_t)((uint32_t)(int32_t)__builtin_nontemporal_load(rp) << sh0);  // read once
                } else {  // residual = source - prediction (svt_aom_[highbd_]subtract_block), never materialised
                    const ptrdiff_t pstep = ud ? -(ptrdiff_t)d.pred_stride : (ptrdiff_t)d.pred_stride;
                    const ptrdiff_t prow0 = (ud ? (ptrdiff_t)(H - 1) * d.pred_stride : 0) + t;
                    if (d.flags & SVT_HIP_TX_PIXEL16) {
                        const uint16_t *sp = (const uint16_t *)(base + d.residual_off) + row0;
                        const uint16_t *pp = (const uint16_t *)(base + d.pred_off) + prow0;
#pragma unroll
                        for (int r = 0; r < H; r++, sp += step, pp += pstep)
                            v[r] = (int32_t)((uint32_t)(int32_t)(int16_t)(*sp - *pp) << sh0);
                    } else {
                        const uint8_t *sp = base + d.residual_off + row0, *pp = base + d.pred_off + prow0;
#pragma unroll
                        for (int r = 0; r < H; r++, sp += step, pp += pstep) v[r] = (int32_t)((uint32_t)((int32_t)*sp - (int32_t)*pp) << sh0);
                    }
                }
            }
            // 24-bit multiplies and 32-bit sums whenever the whole wave's inputs are small enough for them to be exact
            // (the limits leave room for the rounding term of the shift that follows)
            if (__all(max_abs<H>(v) <= (uint32_t)FWD_FAST_LIMIT[G::HI][kind_index(vk)][FWD_COS_COL[G::WI][G::HI] - 10])) {
                fwd1d<Fast, H>(v, vk, FWD_COS_COL[G::WI][G::HI]);
                if constexpr (sh1 < 0) {
#pragma unroll
                    for (int r = 0; r < H; r++) v[r] = Fast::rs(v[r], -sh1);
                }
            } else {
                fwd1d<Exact, H>(v, vk, FWD_COS_COL[G::WI][G::HI]);
                if constexpr (sh1 < 0) {
#pragma unroll
                    for (int r = 0; r < H; r++) v[r] = rshift64(v[r], -sh1);
                }
            }
            const int cc = lr ? W - 1 - t : t;
#pragma unroll
            for (int r = 0; r < H; r++) lds[r * PW + cc] = v[r];
        }
        __syncthreads();
        // -------------------------------------------------------------- forward: rows
        if (live && t < H) {
#pragma unroll
            for (int c = 0; c < W; c++) row[c] = lds[t * PW + c];
            if (__all(max_abs<W>(row) <= (uint32_t)FWD_FAST_LIMIT[G::WI][kind_index(hk)][FWD_COS_ROW[G::WI][G::HI] - 10])) {
                fwd1d<Fast, W>(row, hk, FWD_COS_ROW[G::WI][G::HI]);
                if constexpr (sh2 < 0) {
#pragma unroll
                    for (int c = 0; c < W; c++) row[c] = Fast::rs(row[c], -sh2);
                }
            } else {
                fwd1d<Exact, W>(row, hk, FWD_COS_ROW[G::WI][G::HI]);
                if constexpr (sh2 < 0) {
#pragma unroll
                    for (int c = 0; c < W; c++) row[c] = rshift64(row[c], -sh2);
                }
            }
            const int kw = W >> d.shape, kh = H >> d.shape;
#pragma unroll
            for (int c = 0; c < W; c++) {
                int32_t x = row[c];
                if (G::RECT)
                    x = rshift64((int64_t)x * 5793, 12);
                row[c] = (t < kh && c < kw) ? x : 0;
            }
        }
    }
    // ---------------------------------------------------------------------- 64-point energy (svt_handle_transformWxH)
    uint64_t energy = 0;
    if constexpr (W == 64 || H == 64) {
        if (do_fwd && live && t < H) {
#pragma unroll
            for (int c = 0; c < W; c++)
                if (t >= IH || c >= IW)
                    energy += (uint64_t)((int64_t)row[c] * (int64_t)row[c]);
        }
        energy = group_sum64<L>(energy);
    }
    // ---------------------------------------------------------------------- transform-domain cost (svt_aom_satd)
    uint32_t satd = 0;
    if (do_fwd && (d.flags & SVT_HIP_TX_SATD)) {
        if (live && t < IH) {
#pragma unroll
            for (int c = 0; c < IW; c++) satd += (uint32_t)(row[c] < 0 ? -row[c] : row[c]);
        }
        satd = group_sum32<L>(satd);
    }
    // ---------------------------------------------------------------------- coefficients out / quantise
    uint32_t eob = 0;
    // this lane's scan positions are requested before the coefficient stores below (same arena: see the prediction loads)
    uint32_t iscv[(IW + 1) / 2];
    {
        const bool      qon = live && t < IH && d.quant_mode != SVT_HIP_QUANT_NONE;
        const uint16_t *isp = (const uint16_t *)(base + (qon ? d.iscan_off : 0)) + (qon ? t * IW : 0);
#pragma unroll
        for (int c = 0; c < IW; c += 2) iscv[c / 2] = qon ? ((uint32_t)isp[c] | ((uint32_t)isp[c + 1] << 16)) : 0u;
    }
    if (live && do_fwd && d.coeff_off != SVT_HIP_NO_OFFSET && (d.flags & SVT_HIP_TX_FULLCOEFF) && t < H) {
        int32_t *co = (int32_t *)(base + d.coeff_off) + t * W;
#pragma unroll
        for (int c = 0; c < W; c++) co[c] = row[c];
    }
    if (live && t < IH) {
        if (do_fwd && d.coeff_off != SVT_HIP_NO_OFFSET && !(d.flags & SVT_HIP_TX_FULLCOEFF)) {
            int32_t *co = (int32_t *)(base + d.coeff_off) + t * IW;
#pragma unroll
            for (int c = 0; c < IW; c++) co[c] = row[c];
        }
        if (d.quant_mode != SVT_HIP_QUANT_NONE) {
            QP q;
            load_qp(q, d, base);
            if (!do_fwd) {  // quantise coefficients that already live in memory
                const int32_t *ci = (const int32_t *)(base + d.coeff_off) + t * IW;
#pragma unroll
                for (int c = 0; c < IW; c++) row[c] = ci[c];
            }
            // one decision per wave: every coefficient small and plain tables -> the branch-free short form
            const bool btype = q.mode == SVT_HIP_QUANT_B || q.mode == SVT_HIP_QUANT_B_HBD;
            const int  path  = (q.simple && max_abs<IW>(row) <= 32767u) ? (btype ? 1 : 2) : 0;
            const bool all1 = __all(path == 1), all2 = __all(path == 2);
#pragma unroll
            for (int c = 0; c < IW; c++) {
                const uint32_t rc = (uint32_t)(t * IW + c);
                const int      ac = (c > 0) | (t > 0);
                int32_t        qc, dqc;
                if (all1)
                    quant_small<true>(q, row[c], ac, qc, dqc);
                else if (all2)
                    quant_small<false>(q, row[c], ac, qc, dqc);
                else
                    quant_one(q, row[c], rc, qc, dqc);
                const uint32_t pos = qc ? ((iscv[c / 2] >> (16 * (c & 1))) & 0xffffu) + 1u : 0u;
                eob                = pos > eob ? pos : eob;
                lds[t * PW + c] = qc;  // staged for the coalesced store below
                if constexpr (BLKERR) {
                    const int64_t e = (int64_t)row[c] - (int64_t)dqc;
                    energy += (uint64_t)(e * e);
                }
                row[c]          = dqc;
            }
        } else if (do_inv && !do_fwd) {
            const int32_t *dqi = (const int32_t *)(base + d.dqcoeff_off) + t * IW;
#pragma unroll
            for (int c = 0; c < IW; c++) row[c] = dqi[c];
        }
    }
    eob = group_max<L>(eob);
    if constexpr (BLKERR)
        energy = group_sum64<L>(energy);
    {   // qcoeff / dqcoeff leave through LDS so that the stores are line-coalesced (all L lanes of the block take part)
        const bool quant = live && d.quant_mode != SVT_HIP_QUANT_NONE;
        __syncthreads();
        if (quant && d.qcoeff_off != SVT_HIP_NO_OFFSET)
            coop_store_tile<IW, IH, L, PW>(lds, (int32_t *)(base + d.qcoeff_off), t);
        __syncthreads();
        if (quant && t < IH) {
#pragma unroll
            for (int c = 0; c < IW; c++) lds[t * PW + c] = row[c];
        }
        __syncthreads();
        if (quant && d.dqcoeff_off != SVT_HIP_NO_OFFSET)
            coop_store_tile<IW, IH, L, PW>(lds, (int32_t *)(base + d.dqcoeff_off), t);
    }
    if (live && t == 0) {
        SvtHipTxfmResult r;
        r.three_quad_energy = energy;
        r.eob               = (uint16_t)eob;
        r.pad_ = 0, r.satd = satd;
        *result = r;
    }
    // ---------------------------------------------------------------------- inverse: rows
    __syncthreads();
    if (live && do_inv && t < H) {
        const int range_row = bd == 8 ? 16 : (bd == 10 ? 18 : 20);
        const int clamp_in  = bd + 8;
        bool      any       = false;
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = (t < IH && c < IW) ? row[c] : 0;
            if (G::RECT)
                x = rshift64((int64_t)x * 2896, 12);
            row[c] = clampv<true>(x, clamp_in);
            any |= row[c] != 0;
        }
        if (any) {  // an all-zero row stays all-zero through every 1-D kernel
            constexpr int ish0 = INV_SHIFT0[G::WI][G::HI];
            if (__all(bdi >= 0 && INV_FAST_OK[bdi < 0 ? 0 : bdi][0][G::WI][kind_index(hk)] != 0)) {
                inv1d<Fast, W>(row, hk, range_row);
                if constexpr (ish0 < 0) {
#pragma unroll
                    for (int c = 0; c < W; c++) row[c] = Fast::rs(row[c], -ish0);
                }
            } else {
                inv1d<Exact, W>(row, hk, range_row);
                if constexpr (ish0 < 0) {
#pragma unroll
                    for (int c = 0; c < W; c++) row[c] = rshift64(row[c], -ish0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < W; c++) lds[t * PW + c] = row[c];
    }
    __syncthreads();
    // ---------------------------------------------------------------------- inverse: columns + reconstruction
    if (live && do_inv && t < W) {
        int32_t   v[H];
        const int range_col = bd == 12 ? 18 : 16, col_clamp = bd + 6 > 16 ? bd + 6 : 16;
        const int cc = lr ? W - 1 - t : t;
#pragma unroll
        for (int r = 0; r < H; r++) v[r] = r < IH ? clampv<true>(lds[r * PW + cc], col_clamp) : 0;  // rows >= IH of a 64-point column are zero: the network below folds
        if (__all(bdi >= 0 && INV_FAST_OK[bdi < 0 ? 0 : bdi][1][G::HI][kind_index(vk)] != 0)) {
            inv1d<Fast, H>(v, vk, range_col);
#pragma unroll
            for (int r = 0; r < H; r++) v[r] = Fast::rs(v[r], 4);
        } else {
            inv1d<Exact, H>(v, vk, range_col);
#pragma unroll
            for (int r = 0; r < H; r++) v[r] = rshift64(v[r], 4);
        }
        // FLIPADST columns: output row r of the network is picture row H-1-r — flip the addresses, not the register index
        // (a run-time register index costs a 16-way select per element)
        const ptrdiff_t r0 = ud ? H - 1 : 0, rs = ud ? -(ptrdiff_t)d.recon_stride : (ptrdiff_t)d.recon_stride;
        // all prediction samples are requested before the first reconstruction sample is stored: a load behind a store to
        // the same arena could not be moved ahead of it by the compiler, and waiting for it also waits for that store
        uint32_t predv[(H + 1) / 2];
        {
            const ptrdiff_t ps = ud ? -(ptrdiff_t)d.pred_stride : (ptrdiff_t)d.pred_stride;
            if (d.flags & SVT_HIP_TX_PIXEL16) {
                const uint16_t *pr = (const uint16_t *)(base + d.pred_off) + r0 * (ptrdiff_t)d.pred_stride + t;
#pragma unroll
                for (int r = 0; r < H; r += 2) predv[r / 2] = (uint32_t)__builtin_nontemporal_load(pr + (ptrdiff_t)r * ps) | ((uint32_t)__builtin_nontemporal_load(pr + (ptrdiff_t)(r + 1) * ps) << 16);
            } else {
                const uint8_t *pr = base + d.pred_off + r0 * (ptrdiff_t)d.pred_stride + t;
#pragma unroll
                for (int r = 0; r < H; r += 2) predv[r / 2] = (uint32_t)pr[(ptrdiff_t)r * ps] | ((uint32_t)pr[(ptrdiff_t)(r + 1) * ps] << 16);
            }
        }
        const PixelClip clip((d.flags & SVT_HIP_TX_PIXEL16) ? bd : 8);
        if (d.flags & SVT_HIP_TX_PIXEL16) {
            uint16_t *rc = (uint16_t *)(base + d.recon_off) + r0 * (ptrdiff_t)d.recon_stride + t;
#pragma unroll
            for (int r = 0; r < H; r++, rc += rs) __builtin_nontemporal_store(clip.add((uint16_t)(predv[r / 2] >> (16 * (r & 1))), v[r]), rc);
        } else {
            uint8_t *rc = base + d.recon_off + r0 * (ptrdiff_t)d.recon_stride + t;
#pragma unroll
            for (int r = 0; r < H; r++, rc += rs) *rc = (uint8_t)clip.add((uint16_t)(predv[r / 2] >> (16 * (r & 1))), v[r]);
        }
    }
}

}  // namespace txb
}  // namespace svthip
