// me_frame.hip — whole-picture open-loop motion estimation on gfx950.
//
// Replaces the per-64x64 loop of the reference's ME kernel thread (Source/Lib/Codec/me_process.c:174-290)
// and everything it calls (svt_aom_motion_estimation_b64, motion_estimation.c:3146-3223).
//
// Mapping: ONE workgroup (4 wave64) owns ONE 64x64 block (b64) of ONE picture and walks the whole
// per-block pipeline — zero-MV SADs, pre-HME, HME level 0/1/2, search-centre selection, reference
// pruning, full-pel 85-PU search, candidate list — without leaving the chip: the three source blocks
// (64x64, 32x32, 16x16) and every search window live in LDS, intermediate state never touches HBM.
// Data-dependent control flow (early exits, pruning, search-area adaptation) is evaluated by lane 0 on
// LDS state between barriers; the pixel work (SAD searches, 8x8 SAD pyramids, arg-min reductions) is
// spread over all 256 lanes.  A launch covers all b64 of all pictures of a batch (grid = b64 x jobs), so
// thousands of independent workgroups hide each other's barriers and global-load latency.
//
// Exactness: integer types, wrap-arounds and tie-breaking follow the reference expression by expression
// (see oracle/src/orc_me.c for the line-by-line citations; this file mirrors its structure).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "sad_device.hpp"

using namespace svthip;
using namespace svthip::dev;

namespace {

#define NL 2
#define NR 4
#define MAX_SAD_VALUE_ (128 * 128 * 255) /* motion_estimation.h:85 */
#define MAX_U32_ 0xFFFFFFFFu

// LDS window buffers.  The pre-HME / HME level 0 / level 1 stages search down-scaled pictures and do not need the full
// resolution source block, so their window buffer also covers it (and the full-pel bookkeeping); the block is staged again
// before the first full resolution stage.  Bigger windows than the buffer: several passes.
constexpr uint32_t ME_HME_WIN_DW = 3200;                          // 12.5 KiB: pre-HME, HME level 0 / 1 (8 workgroups per CU)
constexpr uint32_t ME_WIN_DW     = ME_HME_WIN_DW - 64 * 16 - 256;  // HME level 2 and full-pel
constexpr uint32_t fp_pitch_c(uint32_t tw) { return (((tw + 3) >> 2) + 17) | 1u; }  // = fp_pitch()
// widest full-pel tile staged at once (positions): the tile's window — (rows of positions + 63) x fp_pitch(width) dwords —
// must hold at least a few rows of positions
constexpr uint32_t FP_TILE_W = ME_WIN_DW / fp_pitch_c(64) >= 63 + 8 ? 64 : 32;
static_assert(ME_WIN_DW / fp_pitch_c(FP_TILE_W) >= 63 + 8, "full-pel window buffer too small for one tile");

struct PreHme {
    uint64_t sad;
    uint16_t sa_w, sa_h;
    int16_t  col, row;
    uint8_t  valid, pad_[7];
};

struct B64State {
    uint32_t org_x, org_y, b64_w, b64_h;
    SvtHipMeSearchResult sr[NL][NR];
    uint32_t reduce_div[NL][NR];
    uint32_t zz_sad[NL][NR];
    PreHme   ph[NL][NR][2];
    uint8_t  performed_phme[NL][NR][2];
    int16_t  l0x[NL][NR][2][2], l0y[NL][NR][2][2], l1x[NL][NR][2][2], l1y[NL][NR][2][2], l2x[NL][NR][2][2],
        l2y[NL][NR][2][2];
    uint64_t l0s[NL][NR][2][2], l1s[NL][NR][2][2], l2s[NL][NR][2][2];
    SvtHipSearchArea l0_min, l0_max;
    // scratch shared between lane 0 and the workgroup
    uint32_t wg_sum;
    // integer search of the current reference
    int16_t  xc, yc, sw, sh, ox, oy;
    int32_t  do_centre, need_zero_sad, need_hme_sad;
    int16_t  stx, sty, stw, sth;  // block of positions whose window is currently staged (full-pel)
    int32_t  staged, restage;
    uint32_t zero_sad, hme_mv_sad;
    // integer search set-up of every reference, prepared in one pass (lane = reference) when it does not depend on the search results of
    // another reference (fullpel_prepare_all): centre, whether the centre is probed, the final window (size and origin)
    int16_t  fp_xc[NL * NR], fp_yc[NL * NR], fp_sw[NL * NR], fp_sh[NL * NR], fp_ox[NL * NR], fp_oy[NL * NR];
    uint8_t  fp_centre[NL * NR];
    uint32_t fp_inv[NL * NR];        // ceil(2^32 / final window width): position index -> (row, column) without a division per lane
    uint32_t first_ref_sad64;        // p_sb_best_sad[0][0][0] (read by later references, :1359)
    uint64_t me_sad_sum[NL][NR];     // sum of the 64 best 8x8 SADs per reference (me_prune_ref, :1605-1611)
};

struct MeLds {
    static constexpr uint32_t WIN_DW = ME_WIN_DW, HME_WIN_DW = ME_HME_WIN_DW;
    SearchShared sh;
    B64State     st;
    union {
        alignas(16) uint32_t win[ME_HME_WIN_DW];
        struct {
            alignas(16) uint32_t fp_win_[ME_WIN_DW];
            alignas(16) uint32_t src_full[64 * 16];
            uint64_t bestkey[85];
            uint32_t me_dist[85];
        };
    };
    alignas(16) uint32_t src_q[32 * 8];
    alignas(16) uint32_t src_s[16 * 4];
};
static_assert(sizeof(MeLds::win) >= (ME_WIN_DW + 64 * 16) * 4 + 85 * 12, "full-pel view must fit the HME window buffer");

typedef __attribute__((address_space(1))) uint32_t g_u32;  // the result arrays are global memory (the pointers come out of LDS)
#define MINV(a, b) ((a) < (b) ? (a) : (b))
#define MAXV(a, b) ((a) > (b) ? (a) : (b))
#define ABSV(a) ((a) < 0 ? -(a) : (a))

// x / d for the divisors of the set-up code, which are powers of two in every preset (numbers of search regions, search-range divisors,
// 64 x 64 samples): a shift when d is one, the division otherwise (both branches only run when a lane of the wave needs the division —
// it is ~40 instructions, a third of them quarter-rate, in sections that one wave runs alone)
__device__ __forceinline__ uint32_t div_small(uint32_t x, uint32_t d) {
    return (d & (d - 1u)) == 0u && d ? x >> (31u - (uint32_t)__builtin_clz(d)) : x / d;
}

__device__ __forceinline__ uint16_t scaled_dist(uint16_t dist) {
    return (uint16_t)(((dist * 5) / 8) + ((dist % 8) == 0 ? 0 : 1));
}
__device__ __forceinline__ uint16_t pic_dist(const SvtHipMeParams &p, int li, int ri) {
    const int64_t d = (int64_t)p.picture_number - (int64_t)p.ref_picture_number[li][ri];
    return (uint16_t)(int16_t)ABSV(d);
}
__device__ __forceinline__ const uint8_t *plane_at(const SvtHipPlane8 &pl, int x, int y) {
    return pl.buf + (ptrdiff_t)((int)pl.org_y + y) * (ptrdiff_t)pl.stride + (int)pl.org_x + x;
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *g) { return load_u32_any(g); }

// This lane's share of the SAD between the staged 64x64 source (LDS rows of 16 dwords, every `row_step`-th row) and a block
// in global memory at any byte address: 16 bytes per load, `ndw` = block width in dwords (b64_w / 4, a multiple of 2).
__device__ __forceinline__ uint32_t block_sad16(const uint32_t *src, const uint8_t *ref, uint32_t ref_stride, uint32_t ndw, uint32_t rows,
                                                uint32_t row_step, uint32_t lane, uint32_t nlanes) {
    const uint32_t nch = (ndw + 3) >> 2, inv = make_inv_small(nch);
    uint32_t       acc = 0;
    for (uint32_t idx = lane; idx < nch * rows; idx += nlanes) {
        const uint32_t r = fast_div(idx, inv), ch = idx - r * nch;
        const uint4    sv = *(const uint4 *)&src[(r * row_step) * 16 + 4 * ch];
        const u128v    v  = load_u128_any(ref + (size_t)sad_mul_u24(r * row_step, ref_stride) + 16 * ch);
        const uint32_t nd = ndw - 4 * ch;
        acc = __builtin_amdgcn_sad_u8(sv.x, v.x, acc);
        acc = __builtin_amdgcn_sad_u8(sv.y, v.y, acc);      // ndw is even: dword 1 of a chunk is always inside the block
        if (nd > 2) {
            acc = __builtin_amdgcn_sad_u8(sv.z, v.z, acc);
            acc = __builtin_amdgcn_sad_u8(sv.w, v.w, acc);
        }
    }
    return acc;
}

// Workgroup-wide SAD between the staged 64x64 source (rows 0,2,4.. when row_step == 2) and a global block.
// Result in L.st.wg_sum (valid after the trailing barrier).  width must be a multiple of 8.
template <class LDS>
__device__ void wg_block_sad(LDS &L, const uint8_t *ref, uint32_t ref_stride, uint32_t width, uint32_t rows,
                             uint32_t row_step) {
    if (threadIdx.x == 0)
        L.st.wg_sum = 0;
    __syncthreads();
    uint32_t acc = block_sad16(L.src_full, ref, ref_stride, width >> 2, rows, row_step, threadIdx.x, blockDim.x);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&L.st.wg_sum, acc);
    __syncthreads();
}

// The window clamp shared by hme_level_0/1/2 (motion_estimation.c:837-888, 940-990, 1041-1084).
__device__ void hme_clamp(int16_t org_x, int16_t org_y, int16_t pad_w, int16_t pad_h, int16_t W, int16_t H, int16_t *pox,
                          int16_t *poy, int16_t *pw, int16_t *ph) {
    int16_t ox = *pox, oy = *poy, sa_w = *pw, sa_h = *ph;
    if ((org_x + ox) < -pad_w) {
        ox   = (int16_t)(-pad_w - org_x);
        sa_w = (int16_t)(sa_w - (-pad_w - (org_x + ox)));
    }
    if ((org_x + ox) > W - 1)
        ox = (int16_t)(ox - ((org_x + ox) - (W - 1)));
    if ((org_x + ox + sa_w) > W)
        sa_w = (int16_t)MAXV(1, sa_w - ((org_x + ox + sa_w) - W));
    sa_w = (sa_w < 8) ? sa_w : (int16_t)(sa_w & ~0x07);
    if ((org_y + oy) < -pad_h) {
        oy   = (int16_t)(-pad_h - org_y);
        sa_h = (int16_t)(sa_h - (-pad_h - (org_y + oy)));
    }
    if ((org_y + oy) > H - 1)
        oy = (int16_t)(oy - ((org_y + oy) - (H - 1)));
    if ((org_y + oy + sa_h) > H)
        sa_h = (int16_t)MAXV(1, sa_h - ((org_y + oy + sa_h) - H));
    *pox = ox, *poy = oy, *pw = sa_w, *ph = sa_h;
}

// Fill one search descriptor for an HME-style call (SUB_SAD/FULL_SAD argument rewrite of
// motion_estimation.c:891-909).
__device__ void set_desc(SearchDesc &d, const SvtHipMeParams &p, const SvtHipPlane8 &rp, int x_tl, int y_tl, int16_t sa_w,
                         int16_t sa_h, uint32_t skip) {
    const bool full = p.hme_search_method == 1;
    d.ref           = rp.buf + (ptrdiff_t)y_tl * (ptrdiff_t)rp.stride + x_tl;
    d.ref_stride    = full ? rp.stride : rp.stride * 2;
    d.raw_stride    = rp.stride;
    d.sa_w          = sa_w;
    d.sa_h          = sa_h;
    d.skip          = skip;
}

template <class SH>
__device__ void decode_result(const SH &sh, uint32_t i, bool sub, uint64_t *sad, int16_t *x, int16_t *y) {
    const uint64_t key = sh.best[i];
    uint64_t       s   = key >> 32;
    if (key != KEY_NONE) {
        const uint32_t idx = (uint32_t)key;  // row << 16 | column
        *x = (int16_t)(idx & 0xffffu);
        *y = (int16_t)(idx >> 16);
    }
    if (sub)
        s *= 2;
    *sad = s;
}

__device__ void set_quadrants(int16_t x[2][2], int16_t y[2][2], uint64_t s[2][2], int16_t vx, int16_t vy, uint64_t vs) {
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) x[a][b] = vx, y[a][b] = vy, s[a][b] = vs;
}

__device__ void best_quadrant(int16_t x[2][2], int16_t y[2][2], uint64_t s[2][2], int16_t *bx, int16_t *by, uint64_t *bs) {
    *bx = x[0][0], *by = y[0][0], *bs = s[0][0];
    if (s[1][0] < *bs) *bx = x[1][0], *by = y[1][0], *bs = s[1][0];
    if (s[0][1] < *bs) *bx = x[0][1], *by = y[0][1], *bs = s[0][1];
    if (s[1][1] < *bs) *bx = x[1][1], *by = y[1][1], *bs = s[1][1];
}

// Final search-window clamp of integer_search_b64 (motion_estimation.c:1442-1561): window of sw x sh positions centred
// on (xc, yc), clipped to the padded reference; returns its origin relative to the b64 and the clipped size.
__device__ void clamp_me_window(int16_t xc, int16_t yc, int16_t ox_b, int16_t oy_b, int16_t W, int16_t H, int16_t pad, int16_t *psw,
                                int16_t *psh, int16_t *pox, int16_t *poy) {
    int16_t sw = *psw, sh_ = *psh;
    int16_t ox = (int16_t)(xc - (sw >> 1)), oy = (int16_t)(yc - (sh_ >> 1));
    ox  = ((ox_b + ox) < -pad) ? (int16_t)(-pad - ox_b) : ox;
    sw  = ((ox_b + ox) < -pad) ? (int16_t)(sw - (-pad - (ox_b + ox))) : sw;
    ox  = ((ox_b + ox) > W - 1) ? (int16_t)(ox - ((ox_b + ox) - (W - 1))) : ox;
    sw  = ((ox_b + ox + sw) > W) ? (int16_t)MAXV(1, sw - ((ox_b + ox + sw) - W)) : sw;
    sw  = (sw < 8) ? sw : (int16_t)(sw & ~0x07);
    oy  = ((oy_b + oy) < -pad) ? (int16_t)(-pad - oy_b) : oy;
    sh_ = ((oy_b + oy) < -pad) ? (int16_t)(sh_ - (-pad - (oy_b + oy))) : sh_;
    oy  = ((oy_b + oy) > H - 1) ? (int16_t)(oy - ((oy_b + oy) - (H - 1))) : oy;
    sh_ = ((oy_b + oy + sh_) > H) ? (int16_t)MAXV(1, sh_ - ((oy_b + oy + sh_) - H)) : sh_;
    *psw = sw, *psh = sh_, *pox = ox, *poy = oy;
}

// ------------------------------------------------------------------------------------------------
// full-pel 85-PU search (open_loop_me_fullpel_search_sblock semantics, motion_estimation.c:428-560 with the ext_*
// SAD pyramids of :210-425).
//
// fp_stage puts the window of a tw x th block of search positions into L.win: (th + 63) rows of `fp_pitch(tw)` dwords,
// row 0 / byte 0 = the sample that position (0,0) puts under source sample (0,0).
// fp_search evaluates the cw x ch positions whose top-left one sits at (x0, y0) of the staged block (x0 % 4 == 0).
// Work item = (quad of 4 horizontally adjacent positions, one 16x16 block of the source): the four 8x8 SADs of the
// 16x16 are accumulated in registers with v_qsad_pk_u16_u8, the 16x16 / 32x32 / 64x64 sums come from shuffles over the
// 16 lanes that share the quad, each lane keeps the minimum over its four positions and only then touches the 85
// (sad << 32 | raster order) keys in LDS => first minimum in raster order wins, as in the reference.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fp_pitch(uint32_t tw) { return (((tw + 3) >> 2) + 17) | 1u; }

template <class LDS>
__device__ void fp_stage(LDS &L, const uint8_t *win_org, uint32_t stride, uint32_t tw, uint32_t th) {
    stage_rows16(L.win, fp_pitch(tw), win_org, stride, th + 63, blockDim.x, threadIdx.x);
    __syncthreads();
}

// v + (v of the lane selected by a DPP control); all lanes of the 16-lane row take part
template <int CTRL> __device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <class LDS>
__device__ void fp_search(LDS &L, uint32_t pitch, uint32_t x0, uint32_t y0, uint32_t cw, uint32_t ch, uint32_t order0,
                          uint32_t order_pitch, bool sub) {
    const uint32_t tid = threadIdx.x;
    const uint32_t nq = (cw + 3) >> 2, nitems = nq * ch * 16, inv_nq = make_inv_small(nq);
    // lane (tid & 15) = z-order index of this lane's 16x16 block (Appendix A.1): the four 16x16 of a 32x32 are the four
    // lanes of a DPP quad, the four 32x32 are the four quads of a DPP row
    const uint32_t zo = tid & 15;
    const uint32_t zy = 2 * (zo >> 3) + ((zo >> 1) & 1), zx = 2 * ((zo >> 2) & 1) + (zo & 1);
    const uint32_t *s = &L.src_full[(16 * zy) * 16 + 4 * zx];
    for (uint32_t base = 0; base < nitems; base += blockDim.x) {
        if (base + (tid & ~63u) >= nitems)
            break;  // nothing left for this wave (the centre probe is ONE item group: three of the four waves skip it)
        const uint32_t item = base + tid;
        const bool     on   = item < nitems;  // uniform over each group of 16 lanes
        const uint32_t qi = on ? item >> 4 : 0, y = fast_div(qi, inv_nq), q = qi - sad_mul_u24(y, nq);  // (24-bit products: full rate)
        const uint32_t *w = &L.win[sad_mul_u24(y0 + y + 16 * zy, pitch) + (x0 >> 2) + q + 4 * zx];
        uint64_t        a00 = 0, a01 = 0, a10 = 0, a11 = 0;
#pragma unroll
        for (uint32_t r = 0; r < 16; r++) {
            if (sub && (r & 1))
                continue;
            const uint32_t d0 = w[r * pitch], d1 = w[r * pitch + 1], d2 = w[r * pitch + 2], d3 = w[r * pitch + 3], d4 = w[r * pitch + 4];
            const uint4    sv = *(const uint4 *)&s[r * 16];
            if (r < 8) {
                a00 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d0, d1), sv.x, a00);
                a00 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d1, d2), sv.y, a00);
                a01 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d2, d3), sv.z, a01);
                a01 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d3, d4), sv.w, a01);
            } else {
                a10 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d0, d1), sv.x, a10);
                a10 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d1, d2), sv.y, a10);
                a11 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d2, d3), sv.z, a11);
                a11 = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d3, d4), sv.w, a11);
            }
        }
        // running minima over this lane's four positions as packed (sad << 2 | position) keys: a SAD is below 2^20, the four
        // positions are in raster order, so the smallest key is the first minimum; the 64-bit (sad, order) keys are only built
        // for the LDS update
        uint32_t k8[4] = {~0u, ~0u, ~0u, ~0u}, k16 = ~0u, k32 = ~0u, k64 = ~0u;
#pragma unroll
        for (uint32_t pp = 0; pp < 4; pp++) {
            const uint32_t lo_hi_shift = 16 * (pp & 1);
            const uint32_t c0 = ((pp < 2 ? (uint32_t)a00 : (uint32_t)(a00 >> 32)) >> lo_hi_shift) & 0xffff;
            const uint32_t c1 = ((pp < 2 ? (uint32_t)a01 : (uint32_t)(a01 >> 32)) >> lo_hi_shift) & 0xffff;
            const uint32_t c2 = ((pp < 2 ? (uint32_t)a10 : (uint32_t)(a10 >> 32)) >> lo_hi_shift) & 0xffff;
            const uint32_t c3 = ((pp < 2 ? (uint32_t)a11 : (uint32_t)(a11 >> 32)) >> lo_hi_shift) & 0xffff;
            const uint32_t s16 = c0 + c1 + c2 + c3;
            uint32_t       s32 = dpp_add<0xB1>(s16);   // quad_perm [1,0,3,2]
            s32                = dpp_add<0x4E>(s32);   // quad_perm [2,3,0,1]
            uint32_t s64       = dpp_add<0x124>(s32);  // row_ror:4
            s64                = dpp_add<0x128>(s64);  // row_ror:8
            if (4 * q + pp < cw) {
                k8[0] = min(k8[0], (c0 << 2) | pp), k8[1] = min(k8[1], (c1 << 2) | pp);
                k8[2] = min(k8[2], (c2 << 2) | pp), k8[3] = min(k8[3], (c3 << 2) | pp);
                k16 = min(k16, (s16 << 2) | pp), k32 = min(k32, (s32 << 2) | pp), k64 = min(k64, (s64 << 2) | pp);
            }
        }
        if (on && k16 != ~0u) {
            const uint32_t sh1 = sub ? 1 : 0;  // sub-sampled rows: SAD x 2 (motion_estimation.c:520-530)
            const uint32_t ord_row = order0 + sad_mul_u24(y, order_pitch) + 4 * q;
#define FP_KEY(k) (((unsigned long long)(((k) >> 2) << sh1) << 32) | (ord_row + ((k) & 3u)))
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) atomicMin((unsigned long long *)&L.bestkey[21 + 4 * zo + c], FP_KEY(k8[c]));
            atomicMin((unsigned long long *)&L.bestkey[5 + zo], FP_KEY(k16));
            if ((zo & 3) == 0)
                atomicMin((unsigned long long *)&L.bestkey[1 + (zo >> 2)], FP_KEY(k32));
            if (zo == 0)
                atomicMin((unsigned long long *)&L.bestkey[0], FP_KEY(k64));
#undef FP_KEY
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// candidate construction (motion_estimation.c:2602-2905), one lane per PU n
// ------------------------------------------------------------------------------------------------
__device__ const uint8_t z_to_raster_d[85] = {
    0,  1,  2,  3,  4,  5,  6,  9,  10, 7,  8,  11, 12, 13, 14, 17, 18, 15, 16, 19, 20, 21, 22, 29, 30, 23, 24, 31, 32,
    37, 38, 45, 46, 39, 40, 47, 48, 25, 26, 33, 34, 27, 28, 35, 36, 41, 42, 49, 50, 43, 44, 51, 52, 53, 54, 61, 62, 55,
    56, 63, 64, 69, 70, 77, 78, 71, 72, 79, 80, 57, 58, 65, 66, 59, 60, 67, 68, 73, 74, 81, 82, 75, 76, 83, 84};
__device__ __forceinline__ uint8_t pack_cand(uint32_t direction, uint32_t l0, uint32_t l1, uint32_t r0, uint32_t r1) {
    return (uint8_t)((direction & 3) | ((l0 & 3) << 2) | ((l1 & 3) << 4) | ((r0 & 1) << 6) | ((r1 & 1) << 7));
}
__device__ __forceinline__ int use_me_pu(const SvtHipMeParams &p, uint32_t n) {
    return p.enable_me_16x16 ? (p.enable_me_8x8 || n < 21) : (n < 5);
}

#define GS(li, ri, n) gs[((li) * NR + (ri)) * 85 + (n)]
#define GM(li, ri, n) gm[((li) * NR + (ri)) * 85 + (n)]

template <class LDS>
__device__ void cand_single_ref(LDS &L, const SvtHipMeParams &p, const g_u32 *gs, const g_u32 *gm, uint32_t n,
                                uint32_t *mv, uint8_t *cand) {
    const uint8_t pu = z_to_raster_d[n];
    L.me_dist[pu]    = GS(0, 0, n);
    if (!L.st.sr[0][0].do_ref)
        return;
    if (use_me_pu(p, n)) {
        cand[pu * p.max_cand] = pack_cand(0, 0, 0, 0, 0);
        mv[pu * p.max_refs]   = GM(0, 0, n);
    }
}

template <class LDS>
__device__ void cand_mrp_off(LDS &L, const SvtHipMeParams &p, const g_u32 *gs, const g_u32 *gm, uint32_t n,
                             uint32_t nlist, uint32_t *mv, uint8_t *cand, uint8_t *total) {
    const uint8_t org0 = L.st.sr[0][0].do_ref, org1 = (uint8_t)((nlist == 1) ? 0 : L.st.sr[1][0].do_ref);
    if (nlist < 2 || !L.st.sr[1][0].do_ref)
        nlist = 1;
    const uint32_t prune_th = (org0 && org1) ? (uint32_t)p.prune_me_candidates_th : 0;
    const uint8_t  pu  = z_to_raster_d[n];
    uint8_t        off = 0;
    const int      use = use_me_pu(p, n);
    uint8_t       *ca  = cand + pu * p.max_cand;
    uint8_t        dr[2] = {org0, org1};
    const uint32_t s0 = GS(0, 0, n), s1 = GS(1, 0, n);
    const uint32_t best = (org0 && org1) ? MINV(s0, s1) : org0 ? s0 : s1;
    L.me_dist[pu]       = best;
    int min_list        = -1;
    if (p.use_best_unipred_cand_only && dr[0] && dr[1])
        min_list = s0 < s1 ? 0 : 1;
    for (uint32_t li = 0; li < nlist && (use || off == 0); ++li) {
        if (dr[li] == 0)
            continue;
        const uint32_t sl = li ? s1 : s0;
        if (prune_th > 0) {
            const uint32_t d = (sl - best) * 100u;
            if (d > (uint32_t)(best * prune_th)) {
                dr[li] = 0;
                continue;
            }
        }
        if (min_list != -1 && min_list != (int)li) {
            if (use)
                mv[pu * p.max_refs + (li ? p.max_l0 : 0)] = GM(li, 0, n);
            continue;
        }
        if (use) {
            ca[off] = pack_cand(li, 0, 0, li == 0 ? li : 24, li == 1 ? li : 24);
            mv[pu * p.max_refs + (li ? p.max_l0 : 0)] = GM(li, 0, n);
        }
        off++;
    }
    if (dr[0] && dr[1] && use) {
        ca[off]   = pack_cand(2, 0, 0, 0, 1);
        total[pu] = (uint8_t)(off + 1);
    }
}

template <class LDS>
__device__ void cand_general(LDS &L, const SvtHipMeParams &p, const g_u32 *gs, const g_u32 *gm, uint32_t n,
                             uint32_t nlist, uint32_t *mv, uint8_t *cand, uint8_t *total) {
    const uint8_t pu  = (n > 4) ? z_to_raster_d[n] : (uint8_t)n;
    uint8_t       off = 0;
    const int     use = use_me_pu(p, n);
    uint8_t      *ca  = cand + pu * p.max_cand;
    uint32_t      drm = 0;  // bit (li*4+ri)
    const uint32_t prune_th = (uint32_t)p.prune_me_candidates_th;
    uint32_t       best     = MAX_U32_;
    for (uint32_t li = 0; li < nlist; li++)
        for (uint32_t ri = 0; ri < p.num_of_ref_pic_to_search[li]; ri++) {
            if (!L.st.sr[li][ri].do_ref)
                continue;
            drm |= 1u << (li * 4 + ri);
            best = GS(li, ri, n) < best ? GS(li, ri, n) : best;
        }
    L.me_dist[pu] = best;
    for (uint32_t li = 0; li < nlist && (use || off == 0); ++li)
        for (uint32_t ri = 0; ri < p.num_of_ref_pic_to_search[li] && (use || off == 0); ++ri) {
            if (!(drm & (1u << (li * 4 + ri))))
                continue;
            if (prune_th > 0) {
                const uint32_t d = (GS(li, ri, n) - best) * 100u;
                if (d > (uint32_t)(best * prune_th)) {
                    drm &= ~(1u << (li * 4 + ri));
                    continue;
                }
            }
            if (use) {
                ca[off] = pack_cand(li, ri, ri, li == 0 ? li : 24, li == 1 ? li : 24);
                mv[pu * p.max_refs + (li ? p.max_l0 : 0) + ri] = GM(li, ri, n);
            }
            off++;
        }
    if (nlist == 2 && use) {
        for (uint32_t a = 0; a < p.num_of_ref_pic_to_search[0]; a++)
            for (uint32_t b = 0; b < p.num_of_ref_pic_to_search[1]; b++) {
                if (p.only_l_bwd && (a > 0 || b > 0))
                    continue;
                if ((drm & (1u << a)) && (drm & (1u << (4 + b))))
                    ca[off++] = pack_cand(2, a, b, 0, 1);
            }
        if (!p.only_l_bwd) {
            for (uint32_t a = 1; a < p.num_of_ref_pic_to_search[0]; a++)
                if ((drm & 1u) && (drm & (1u << a)))
                    ca[off++] = pack_cand(2, 0, a, 0, 0);
            if (p.num_of_ref_pic_to_search[1] == 3 && (drm & (1u << 4)) && (drm & (1u << 6)))
                ca[off++] = pack_cand(2, 0, 2, 1, 1);
        }
    }
    if (use)
        total[pu] = off;
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// Phase timing for kernel tuning (make PROF=1): per-phase wall-clock ticks (100 MHz) summed over all workgroups.
#ifdef SVT_HIP_ME_PROFILE
__device__ unsigned long long g_me_prof[16];
#define ME_SITE(k)             \
    do {                       \
        if (threadIdx.x == 0)  \
            L.sh.prof_site = k; \
    } while (0)
#define ME_PHASE(i)                                                        \
    do {                                                                   \
        if (threadIdx.x == 0) {                                            \
            const unsigned long long t_ = wall_clock64();                  \
            atomicAdd(&g_me_prof[i], t_ - prof_last);                      \
            prof_last = t_;                                                \
        }                                                                  \
    } while (0)
#else
#define ME_PHASE(i) \
    do {            \
    } while (0)
#define ME_SITE(k) \
    do {           \
    } while (0)
#endif


// ------------------------------------------------------------------------------------------------
// Stage functions.  Every stage of svt_aom_motion_estimation_b64 is a device function over the per-b64 state in LDS
// (B64State); `Ctx` carries the uniform geometry.  (A staged variant — one workgroup per b64 x reference and stage, handing
// results over through global records — was measured at 1.46 ms against 0.98 ms for the one-launch kernel on the bench
// workload: the kernel is instruction-issue bound, and the staged form repeats the scalar set-up per reference.)
// ------------------------------------------------------------------------------------------------
struct Ctx {
    const SvtHipMeFrameJob *job;
    uint32_t  b64, org_x, org_y, b64_w, b64_h, aw, ah;
    bool      hme_sub, me_sub;
    int       tl, nlists, R0, R1, nref;
};
// Values every stage derives its indices from are made opaque at the top of each stage: otherwise the optimiser hoists the
// index arithmetic of ALL stages out of the kernel's step loop and keeps dozens of results alive across the searches — which
// under the 64-register budget means spilling values as cheap as tid >> 2.
__device__ __forceinline__ uint32_t opaque_v(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t opaque_s(uint32_t v) {
    asm volatile("" : "+s"(v));
    return v;
}
#define ME_CTX_LOCALS(c)                                                                                                      \
    const SvtHipMeFrameJob &job = *(c).job;                                                                                   \
    const SvtHipMeParams   &p   = job.prm;                                                                                    \
    B64State               &S   = L.st;                                                                                       \
    const uint32_t tid = opaque_v(threadIdx.x), org_x = opaque_s((c).org_x), org_y = opaque_s((c).org_y);                       \
    const uint32_t b64_w = opaque_s((c).b64_w), b64_h = opaque_s((c).b64_h);                                                    \
    const uint32_t aw = (c).aw, ah = (c).ah;                                                                                  \
    const bool     hme_sub = (c).hme_sub, me_sub = (c).me_sub;                                                                \
    const int      tl = (c).tl, nlists = (c).nlists, R0 = (c).R0, R1 = (c).R1, nref = (c).nref;                               \
    /* p_sb_best_sad / p_sb_best_mv of this b64, recomputed where needed: two pointers held across every stage were spilled */ \
    g_u32 *const gs = (g_u32 *)(job.out.best_sad + (size_t)(c).b64 * NL * NR * 85), *const gm = (g_u32 *)(job.out.best_mv + (size_t)(c).b64 * NL * NR * 85); \
    (void)job, (void)p, (void)S, (void)tid, (void)org_x, (void)org_y, (void)b64_w, (void)b64_h, (void)aw, (void)ah;           \
    (void)hme_sub, (void)me_sub, (void)tl, (void)nlists, (void)R0, (void)R1, (void)nref, (void)gs, (void)gm

__device__ bool make_ctx(Ctx &c, const SvtHipMeFrameJob &job, uint32_t bx) {
    const SvtHipMeParams &p = job.prm;
    c.job = &job;
    c.aw = (job.src.full.width + 7u) & ~7u, c.ah = (job.src.full.height + 7u) & ~7u;
    const uint32_t bw64 = (c.aw + 63) / 64, bh64 = (c.ah + 63) / 64;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup x goes to XCD
    // x % 8 (the grid's x extent is a multiple of 8).  Give every XCD one contiguous band of b64 rows of the picture:
    // neighbouring b64 share most of their search windows, which then hit in that XCD's L2.
    const uint32_t nb64 = bw64 * bh64, band = (nb64 + 7) / 8;
    c.b64 = (bx & 7) * band + (bx >> 3);
    if ((bx >> 3) >= band || c.b64 >= nb64)
        return false;
    c.org_x = (c.b64 % bw64) * 64, c.org_y = (c.b64 / bw64) * 64;
    c.b64_w = (c.aw - c.org_x) < 64 ? c.aw - c.org_x : 64, c.b64_h = (c.ah - c.org_y) < 64 ? c.ah - c.org_y : 64;
    c.hme_sub = p.hme_search_method == 0, c.me_sub = p.me_search_method == 0;
    c.tl = p.temporal_layer_index, c.nlists = p.num_of_list_to_search;
    c.R0 = p.num_of_ref_pic_to_search[0], c.R1 = c.nlists > 1 ? p.num_of_ref_pic_to_search[1] : 0, c.nref = c.R0 + c.R1;
    return true;
}


// source blocks of this b64 (me_process.c:183-214): which = 1 full | 2 quarter | 4 sixteenth
template <class LDS>
__device__ void stage_sources(LDS &L, const Ctx &c, int which) {
    ME_CTX_LOCALS(c);
    // whole 64 x 64 / 32 x 32 / 16 x 16 tiles (a partial b64 at the picture edge reads into the padding, as the reference's
    // SIMD kernels do; only the b64_w x b64_h part is ever used)
    if (which & 1)
        stage_rows16(L.src_full, 16, plane_at(job.src.full, (int)org_x, (int)org_y), job.src.full.stride, 64, blockDim.x, tid);
    if (which & 2)
        stage_rows16(L.src_q, 8, plane_at(job.src.quarter, (int)(org_x >> 1), (int)(org_y >> 1)), job.src.quarter.stride, 32, blockDim.x, tid);
    if (which & 4)
        stage_rows16(L.src_s, 4, plane_at(job.src.sixteenth, (int)(org_x >> 2), (int)(org_y >> 2)), job.src.sixteenth.stride, 16, blockDim.x, tid);
}

// The per-reference rules between the stages (initialisation, the three pruning rules, the centre selection) run with ONE LANE PER
// REFERENCE SLOT (list, index) of the first wave instead of a loop in lane 0: a loop over the slots is a chain of dependent LDS round
// trips (the centre selection alone was 8 % of a workgroup's life), eight lanes read their slots at once and meet in three shuffles.
struct Slot {
    bool slot, searched;
    int  li, ri;
};
__device__ __forceinline__ Slot slot_of(uint32_t tid, const SvtHipMeParams &p, int nlists) {
    Slot s;
    s.slot = tid < (uint32_t)(NL * NR);
    s.li = s.slot ? (int)tid / NR : 0, s.ri = s.slot ? (int)tid % NR : 0;
    s.searched = s.slot && s.li < nlists && s.ri < (int)p.num_of_ref_pic_to_search[s.li];
    return s;
}
static_assert(NL * NR == 8, "slots_min covers eight lanes");
__device__ __forceinline__ uint64_t slots_min(uint64_t v) {  // minimum over lanes 0..7 (every lane of the wave takes part)
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const uint64_t o = __shfl_xor((unsigned long long)v, off, 64);
        v                = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t slots_min(uint32_t v) {
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v                = o < v ? o : v;
    }
    return v;
}

// init_me_hme_data (motion_estimation.c:3080-3140); lane = reference slot
template <class LDS>
__device__ void init_state_lane0(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    if (tid == 0) {
        S.first_ref_sad64 = 0;
        S.org_x = org_x, S.org_y = org_y, S.b64_w = b64_w, S.b64_h = b64_h;
        S.l0_min = p.hme_l0_sa_min, S.l0_max = p.hme_l0_sa_max;
    }
    if (tid < (uint32_t)(NL * NR)) {
        const int i = (int)tid / NR, j = (int)tid % NR;
        S.me_sad_sum[i][j]  = 0;
        S.sr[i][j].hme_sad  = MAX_U32_;
        S.sr[i][j].hme_sc_x = S.sr[i][j].hme_sc_y = 0;
        S.sr[i][j].do_ref                         = 1;
        S.sr[i][j].pad_[0] = S.sr[i][j].pad_[1] = S.sr[i][j].pad_[2] = 0;
        S.reduce_div[i][j] = 1;
        S.zz_sad[i][j]     = ~0u;
        for (int k = 0; k < 2; k++) {
            S.ph[i][j][k].valid = 0, S.ph[i][j][k].sad = 0, S.ph[i][j][k].col = S.ph[i][j][k].row = 0;
            S.performed_phme[i][j][k] = 0;
        }
        set_quadrants(S.l0x[i][j], S.l0y[i][j], S.l0s[i][j], 0, 0, 0);
        set_quadrants(S.l1x[i][j], S.l1y[i][j], S.l1s[i][j], 0, 0, 0);
        set_quadrants(S.l2x[i][j], S.l2y[i][j], S.l2s[i][j], 0, 0, 0);
    }
}

// zero-MV SADs of all references (init_zz_sad, motion_estimation.c:2452-2470) in one pass: wave w takes references
// w, w + 4, ...; 64 x h/2 samples on every other row against the staged source; ends with a barrier
template <class LDS>
__device__ void zz_sad_all(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    const uint32_t wv = tid >> 6, lane = tid & 63, nwv = blockDim.x >> 6;
    const uint32_t ndw = b64_w >> 2, rows = b64_h >> 1;
    for (int f = (int)wv; f < nref; f += (int)nwv) {
        const int li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
        if (!(tl > 0 || li == 0))
            continue;
        const SvtHipPlane8 &rp  = job.ref[li][ri].full;
        const uint8_t      *ref = plane_at(rp, (int16_t)org_x, (int16_t)org_y);
        uint32_t            acc = block_sad16(L.src_full, ref, rp.stride, ndw, rows, 2, lane, 64);
        acc = wave_sum(acc);
        if (lane == 0) {
            uint32_t z = acc << 1;
            z          = div_small(z * 64 * 64, b64_w * b64_h);
            S.zz_sad[li][ri] = z;
        }
    }
    __syncthreads();
}

// zz-SAD based reference pruning (init_zz_sad, motion_estimation.c:2471-2504); first wave, lane = reference slot
template <class LDS>
__device__ void zz_prune_lane0(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    if (tid < 64) {
        const Slot     q    = slot_of(tid, p, nlists);
        const uint32_t z    = S.zz_sad[q.li][q.ri];
        const uint32_t best = slots_min((q.searched && (tl > 0 || q.li == 0)) ? z : MAX_U32_);
        if (q.searched && q.ri >= 1) {
            bool drop = tl > 0 && best < p.zz_sad_th && (uint32_t)((z - best) * 100u) > (uint32_t)(p.zz_sad_pct * best);
            if (p.me_safe_limit_zz_th)
                drop = drop || (p.hierarchical_levels > 0 && nlists == 2 && tl >= p.hierarchical_levels && p.similar_brightness_refs &&
                                S.zz_sad[0][0] < p.me_safe_limit_zz_th && S.zz_sad[1][0] < p.me_safe_limit_zz_th);
            if (drop)
                S.sr[q.li][q.ri].do_ref = 0;
        }
    }
}

// The four searching stages (pre-HME, HME level 0 / 1 / 2) are written as a set-up part and a finish part around ONE shared
// call of wg_multi_search in the kernel's step loop: a single inlined copy of the search (four copies do not fit the register
// budget of 8 workgroups per CU, and an out-of-line copy saves and restores 16 registers per call through scratch memory —
// measured 0.5 GB of extra traffic per launch).  `part` 0 = set-up (fills the descriptors and `a`), 1 = finish.
struct SearchArgs {
    bool            search;
    uint32_t        nd;
    const uint32_t *src;
    uint32_t        row_dw, bw, bh, cap;
};

// prehme_b64 (motion_estimation.c:1792-1866) for references [f0, f1): one lane per (reference, region) sets its
// descriptor up, ONE wg_multi_search call searches them all, the same lane decodes.  Ends with a barrier.
template <class LDS>
__device__ void prehme_round(LDS &L, const Ctx &c, int f0, int f1, bool searching, int part, SearchArgs &a) {
    ME_CTX_LOCALS(c);
    const int      f  = f0 + (int)(tid >> 1), si = (int)(tid & 1);
    const int      li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
    const uint32_t nd = 2u * (uint32_t)(f1 - f0);
    const bool     mine = tid < nd;
    if (part == 0 && mine) {
        PreHme     &d  = S.ph[li][ri][si];
        SearchDesc &sd = L.sh.desc[tid];
        sd.aux_go      = 0;
        const SvtHipPlane8 &rp = job.ref[li][ri].sixteenth;
        sd.sa_w = sd.sa_h = 0, sd.skip = 0, sd.ref = rp.buf, sd.ref_stride = sd.raw_stride = rp.stride;
        if (tl > 0 || li == 0) {
            const uint32_t factor = scaled_dist(pic_dist(p, li, ri));
            const PreHme   o      = S.ph[0][ri][si];
            // check_prehme_early_exit (:1763-1789)
            if (p.me_early_exit_th && S.zz_sad[li][ri] < p.me_early_exit_th) {
                d.col = d.row = 0, d.sad = 0, d.valid = 1;
            } else if (p.prehme_l1_early_exit && li == 1 && o.valid &&
                       ((o.sad < (32 * 32)) || ((ABSV(o.col) < 16) && (ABSV(o.row) < 16)))) {
                d.col = (int16_t)-o.col, d.row = (int16_t)-o.row, d.sad = o.sad, d.valid = 1;
            } else if (!S.sr[li][ri].do_ref) {
                d.col = d.row = 0, d.sad = MAX_U32_;
            } else {
                d.sa_w = (uint16_t)MINV((uint32_t)(p.prehme_sa_min[si].width * factor), (uint32_t)p.prehme_sa_max[si].width);
                d.sa_h = (uint16_t)MINV((uint32_t)(p.prehme_sa_min[si].height * factor), (uint32_t)p.prehme_sa_max[si].height);
                // prehme_core (:1638-1736)
                const int16_t ox16 = (int16_t)(((int16_t)org_x) >> 2), oy16 = (int16_t)(((int16_t)org_y) >> 2);
                int16_t pad_w = (int16_t)rp.org_x - 1, pad_h = (int16_t)rp.org_y - 1;
                int16_t sa_w = (int16_t)d.sa_w, sa_h = (int16_t)d.sa_h;
                int16_t ox = -(int16_t)(sa_w >> 1), oy = -(int16_t)(sa_h >> 1);
                const int16_t W = (int16_t)rp.width, H = (int16_t)rp.height;
                ox   = ((ox16 + ox) < -pad_w) ? (int16_t)(-pad_w - ox16) : ox;
                sa_w = ((ox16 + ox) < -pad_w) ? (int16_t)(sa_w - (-pad_w - (ox16 + ox))) : sa_w;
                ox   = ((ox16 + ox) > W - 1) ? (int16_t)(ox - ((ox16 + ox) - (W - 1))) : ox;
                sa_w = ((ox16 + ox + sa_w) > W) ? (int16_t)MAXV(1, sa_w - ((ox16 + ox + sa_w) - W)) : sa_w;
                oy   = ((oy16 + oy) < -pad_h) ? (int16_t)(-pad_h - oy16) : oy;
                sa_h = ((oy16 + oy) < -pad_h) ? (int16_t)(sa_h - (-pad_h - (oy16 + oy))) : sa_h;
                oy   = ((oy16 + oy) > H - 1) ? (int16_t)(oy - ((oy16 + oy) - (H - 1))) : oy;
                sa_h = ((oy16 + oy + sa_h) > H) ? (int16_t)MAXV(1, sa_h - ((oy16 + oy + sa_h) - H)) : sa_h;
                const int16_t x_tl = (int16_t)(((int16_t)rp.org_x + ox16) + ox);
                const int16_t y_tl = (int16_t)(((int16_t)rp.org_y + oy16) + oy);
                const uint32_t bw = b64_w >> 2, bh = hme_sub ? (b64_h >> 2) >> 1 : (b64_h >> 2);
                set_desc(sd, p, rp, x_tl, y_tl, sa_w, sa_h, p.prehme_skip_search_line && bw == 16 && bh <= 16);
                sd.aux_x = ox, sd.aux_y = oy, sd.aux_go = 1;
                S.performed_phme[li][ri][si] = 1;
            }
        } else {
            d.col = (int16_t)-S.ph[0][ri][si].col;
            d.row = (int16_t)-S.ph[0][ri][si].row;
            d.sad = S.ph[0][ri][si].sad;
        }
    }
    if (part == 0) {
        a = SearchArgs{searching, nd, L.src_s, hme_sub ? 8u : 4u, b64_w >> 2, hme_sub ? (b64_h >> 2) >> 1 : (b64_h >> 2), LDS::HME_WIN_DW};
        return;
    }
    if (searching) {
        if (mine && L.sh.desc[tid].aux_go) {
            PreHme       &d    = S.ph[li][ri][si];
            const int16_t q_ox = L.sh.desc[tid].aux_x, q_oy = L.sh.desc[tid].aux_y;
            decode_result(L.sh, tid, hme_sub, &d.sad, &d.col, &d.row);
            d.col = (int16_t)(d.col + q_ox);
            d.col = (int16_t)(d.col * 4);
            d.row = (int16_t)(d.row + q_oy);
            d.row = (int16_t)(d.row * 4);
            d.valid = 1;
        }
    }
    __syncthreads();
}

// pre-HME based reference pruning (:1851-1865); first wave, lane = reference slot
template <class LDS>
__device__ void phme_prune_lane0(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    if (tid < 64) {
        const Slot     q    = slot_of(tid, p, nlists);
        const uint32_t sad  = (uint32_t)MINV(S.ph[q.li][q.ri][0].sad, S.ph[q.li][q.ri][1].sad);
        const uint32_t best = slots_min((q.searched && (tl > 0 || q.li == 0)) ? sad : MAX_U32_);
        if (tl > 0 && best < p.phme_sad_th && q.searched && q.ri != 0 && S.sr[q.li][q.ri].do_ref &&
            (uint32_t)((sad - best) * 100u) > (uint32_t)(p.phme_sad_pct * best))
            S.sr[q.li][q.ri].do_ref = 0;
    }
}

// HME level 0 (motion_estimation.c:1976-2106) for references [f0, f1); one lane per (reference, quadrant).  Ends with a barrier.
template <class LDS>
__device__ void hme_l0_round(LDS &L, const Ctx &c, int f0, int f1, int part, SearchArgs &a) {
    ME_CTX_LOCALS(c);
    const uint32_t nd   = (uint32_t)(f1 - f0) * 4u;
    const bool     mine = tid < nd;
    const int      f    = f0 + (int)(tid >> 2);
    const uint32_t qi = tid & 3, sw_ = qi & 1, sh_ = qi >> 1;
    const int      li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
    if (part == 0 && mine) {
        SearchDesc         &sd = L.sh.desc[tid];
        sd.aux_go              = 0;
        const SvtHipPlane8 &rp = job.ref[li][ri].sixteenth;
        sd.sa_w = sd.sa_h = 0, sd.skip = 0, sd.ref = rp.buf, sd.ref_stride = sd.raw_stride = rp.stride;
        const int psi = S.ph[li][ri][0].sad <= S.ph[li][ri][1].sad ? 0 : 1;
        if (p.me_early_exit_th && S.zz_sad[li][ri] < (p.me_early_exit_th >> 2)) {
            S.l0x[li][ri][sw_][sh_] = 0, S.l0y[li][ri][sw_][sh_] = 0, S.l0s[li][ri][sw_][sh_] = 0;
        } else if (p.prev_me_stage_based_exit_th && S.performed_phme[li][ri][psi] &&
                   S.ph[li][ri][psi].sad < (p.prev_me_stage_based_exit_th >> 4)) {
            S.l0x[li][ri][sw_][sh_] = S.ph[li][ri][psi].col, S.l0y[li][ri][sw_][sh_] = S.ph[li][ri][psi].row;
            S.l0s[li][ri][sw_][sh_] = S.ph[li][ri][psi].sad;
        } else if (!S.sr[li][ri].do_ref) {
            S.l0x[li][ri][sw_][sh_] = 0, S.l0y[li][ri][sw_][sh_] = 0, S.l0s[li][ri][sw_][sh_] = MAX_U32_;
        } else if (tl > 0 || li == 0) {
            // get_hme_l0_search_area (:1870-1937); hme_l0_sa is restored right after use (:2069-2073)
            SvtHipSearchArea mn = S.l0_min, mx = S.l0_max;
            if (p.enable_me_sr_adjustment && p.distance_based_hme_resizing) {
                uint8_t is_hor = 1, is_ver = 1, is_still = 0;
                if (p.reduce_hme_l0_sr_th_min && p.reduce_hme_l0_sr_th_max && (li || ri)) {
                    const int16_t mvx = S.l0x[0][0][0][0], mvy = S.l0y[0][0][0][0];
                    is_ver   = (ABSV(mvx) < p.reduce_hme_l0_sr_th_min) && (ABSV(mvy) > p.reduce_hme_l0_sr_th_max);
                    is_hor   = (ABSV(mvx) > p.reduce_hme_l0_sr_th_max) && (ABSV(mvy) < p.reduce_hme_l0_sr_th_min);
                    is_still = (ABSV(mvx) < (p.reduce_hme_l0_sr_th_min * 3)) && (ABSV(mvy) < (p.reduce_hme_l0_sr_th_min * 3));
                }
                uint8_t xo = 1, yo = 1;
                if (!is_ver) yo = 2;
                if (!is_hor) xo = 2;
                if (p.enable_me_sr_adjustment == 2 && is_still) xo = yo = 4;
                mn.width  = (uint16_t)(mn.width / (xo + ri));
                mn.height = (uint16_t)(mn.height / (yo + ri));
                mx.width  = (uint16_t)(mx.width / (xo + ri));
                mx.height = (uint16_t)(mx.height / (yo + ri));
            }
            const int32_t factor = scaled_dist(pic_dist(p, li, ri));
            int16_t       w      = (int16_t)div_small(mn.width, p.num_hme_sa_w);
            w = (int16_t)MINV((((w * factor) + 15) & ~0x0F), (int32_t)((div_small(mx.width, p.num_hme_sa_w) + 15) & ~0x0Fu));
            int16_t h = (int16_t)div_small(mn.height, p.num_hme_sa_h);
            h         = (int16_t)MINV((h * factor), (int32_t)div_small(mx.height, p.num_hme_sa_h));
            const int16_t ox16 = (int16_t)(((int16_t)org_x) >> 2), oy16 = (int16_t)(((int16_t)org_y) >> 2);
            // hme_level_0 (:820-920)
            int16_t       sa_w = (int16_t)((w + 7) & ~0x07), sa_h = h;
            const int16_t xd = (int16_t)(sa_w * sw_), yd = (int16_t)(sa_h * sh_);
            int16_t       ox = (int16_t)(-(int16_t)((sa_w * p.num_hme_sa_w) >> 1) + xd);
            int16_t       oy = (int16_t)(-(int16_t)((sa_h * p.num_hme_sa_h) >> 1) + yd);
            hme_clamp(ox16, oy16, (int16_t)(rp.org_x - 1), (int16_t)(rp.org_y - 1), (int16_t)rp.width, (int16_t)rp.height, &ox,
                      &oy, &sa_w, &sa_h);
            const int16_t x_tl = (int16_t)(((int16_t)rp.org_x + ox16) + ox);
            const int16_t y_tl = (int16_t)(((int16_t)rp.org_y + oy16) + oy);
            set_desc(sd, p, rp, x_tl, y_tl, sa_w, sa_h, 0);
            sd.aux_x = ox, sd.aux_y = oy, sd.aux_go = 1;
        }
    }
    if (part == 0) {
        a = SearchArgs{true, nd, L.src_s, hme_sub ? 8u : 4u, b64_w >> 2, hme_sub ? (b64_h >> 2) >> 1 : (b64_h >> 2), LDS::HME_WIN_DW};
        return;
    }
    const bool go = mine && L.sh.desc[tid].aux_go;
    const int16_t q_ox = L.sh.desc[tid & 31].aux_x, q_oy = L.sh.desc[tid & 31].aux_y;
    if (go) {
        int16_t mx = S.l0x[li][ri][sw_][sh_], my = S.l0y[li][ri][sw_][sh_];
        decode_result(L.sh, tid, hme_sub, &S.l0s[li][ri][sw_][sh_], &mx, &my);
        mx = (int16_t)(mx + q_ox), mx = (int16_t)(mx * 4);
        my = (int16_t)(my + q_oy), my = (int16_t)(my * 4);
        S.l0x[li][ri][sw_][sh_] = mx, S.l0y[li][ri][sw_][sh_] = my;
    }
    __syncthreads();
    // the pre-HME vector replaces the worst quadrant if it is better (:2086-2104): one lane per reference
    if (go && qi == 0 && p.prehme_enable) {
        uint8_t  bw_ = 0, bh_ = 0;
        uint64_t mx = 0;
        if (S.l0s[li][ri][0][0] > mx) mx = S.l0s[li][ri][0][0], bw_ = 0, bh_ = 0;
        if (S.l0s[li][ri][1][0] > mx) mx = S.l0s[li][ri][1][0], bw_ = 1, bh_ = 0;
        if (S.l0s[li][ri][0][1] > mx) mx = S.l0s[li][ri][0][1], bw_ = 0, bh_ = 1;
        if (S.l0s[li][ri][1][1] > mx) bw_ = 1, bh_ = 1;
        const int si = S.ph[li][ri][0].sad <= S.ph[li][ri][1].sad ? 0 : 1;
        if (S.ph[li][ri][si].sad < S.l0s[li][ri][bw_][bh_]) {
            S.l0s[li][ri][bw_][bh_] = S.ph[li][ri][si].sad;
            S.l0x[li][ri][bw_][bh_] = S.ph[li][ri][si].col;
            S.l0y[li][ri][bw_][bh_] = S.ph[li][ri][si].row;
        }
    }
    __syncthreads();
}

// HME level 1 (motion_estimation.c:2111-2192) for references [f0, f1).  Ends with a barrier.
template <class LDS>
__device__ void hme_l1_round(LDS &L, const Ctx &c, int f0, int f1, int part, SearchArgs &a) {
    ME_CTX_LOCALS(c);
    const uint32_t nd   = (uint32_t)(f1 - f0) * 4u;
    const bool     mine = tid < nd;
    const int      f    = f0 + (int)(tid >> 2);
    const uint32_t qi = tid & 3, sw_ = qi & 1, sh_ = qi >> 1;
    const int      li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
    if (part == 0 && mine) {
        SearchDesc         &sd = L.sh.desc[tid];
        sd.aux_go              = 0;
        const SvtHipPlane8 &rp = job.ref[li][ri].quarter;
        sd.sa_w = sd.sa_h = 0, sd.skip = 0, sd.ref = rp.buf, sd.ref_stride = sd.raw_stride = rp.stride;
        if (tl > 0 || li == 0) {
            if (p.me_early_exit_th && S.zz_sad[li][ri] < (p.me_early_exit_th >> 2)) {
                S.l1x[li][ri][sw_][sh_] = 0, S.l1y[li][ri][sw_][sh_] = 0, S.l1s[li][ri][sw_][sh_] = 0;
            } else if (!S.sr[li][ri].do_ref) {
                S.l1x[li][ri][sw_][sh_] = 0, S.l1y[li][ri][sw_][sh_] = 0, S.l1s[li][ri][sw_][sh_] = MAX_U32_;
            } else if (p.prev_me_stage_based_exit_th && S.l0s[li][ri][sw_][sh_] < (p.prev_me_stage_based_exit_th >> 5)) {
                S.l1x[li][ri][sw_][sh_] = S.l0x[li][ri][sw_][sh_];
                S.l1y[li][ri][sw_][sh_] = S.l0y[li][ri][sw_][sh_];
                S.l1s[li][ri][sw_][sh_] = S.l0s[li][ri][sw_][sh_];
            } else {
                // hme_level_1 (:923-1022)
                const int16_t ox4 = (int16_t)(((int16_t)org_x) >> 1), oy4 = (int16_t)(((int16_t)org_y) >> 1);
                int16_t sa_w = (int16_t)(((int16_t)p.hme_l1_sa.width + 7) & ~0x07), sa_h = (int16_t)p.hme_l1_sa.height;
                int16_t ox = (int16_t)(-(sa_w >> 1) + (int16_t)(S.l0x[li][ri][sw_][sh_] >> 1));
                int16_t oy = (int16_t)(-(sa_h >> 1) + (int16_t)(S.l0y[li][ri][sw_][sh_] >> 1));
                hme_clamp(ox4, oy4, (int16_t)(rp.org_x - 1), (int16_t)(rp.org_y - 1), (int16_t)rp.width, (int16_t)rp.height, &ox,
                          &oy, &sa_w, &sa_h);
                const int16_t x_tl = (int16_t)(((int16_t)rp.org_x + ox4) + ox);
                const int16_t y_tl = (int16_t)(((int16_t)rp.org_y + oy4) + oy);
                set_desc(sd, p, rp, x_tl, y_tl, sa_w, sa_h, 0);
                sd.aux_x = ox, sd.aux_y = oy, sd.aux_go = 1;
            }
        }
    }
    if (part == 0) {
        a = SearchArgs{true, nd, L.src_q, hme_sub ? 16u : 8u, b64_w >> 1, hme_sub ? (b64_h >> 1) >> 1 : (b64_h >> 1), LDS::HME_WIN_DW};
        return;
    }
    const bool go = mine && L.sh.desc[tid].aux_go;
    const int16_t q_ox = L.sh.desc[tid & 31].aux_x, q_oy = L.sh.desc[tid & 31].aux_y;
    if (go) {
        int16_t mx = S.l1x[li][ri][sw_][sh_], my = S.l1y[li][ri][sw_][sh_];
        decode_result(L.sh, tid, hme_sub, &S.l1s[li][ri][sw_][sh_], &mx, &my);
        mx = (int16_t)(mx + q_ox), mx = (int16_t)(mx * 2);
        my = (int16_t)(my + q_oy), my = (int16_t)(my * 2);
        S.l1x[li][ri][sw_][sh_] = mx, S.l1y[li][ri][sw_][sh_] = my;
    }
    __syncthreads();
}

// HME level 2 (motion_estimation.c:2197-2247) for references [f0, f1).  Ends with a barrier.
template <class LDS>
__device__ void hme_l2_round(LDS &L, const Ctx &c, int f0, int f1, int part, SearchArgs &a) {
    ME_CTX_LOCALS(c);
    const uint32_t nd   = (uint32_t)(f1 - f0) * 4u;
    const bool     mine = tid < nd;
    const int      f    = f0 + (int)(tid >> 2);
    const uint32_t qi = tid & 3, sw_ = qi & 1, sh_ = qi >> 1;
    const int      li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
    if (part == 0 && mine) {
        SearchDesc         &sd = L.sh.desc[tid];
        sd.aux_go              = 0;
        const SvtHipPlane8 &rp = job.ref[li][ri].full;
        sd.sa_w = sd.sa_h = 0, sd.skip = 0, sd.ref = rp.buf, sd.ref_stride = sd.raw_stride = rp.stride;
        if (tl > 0 || li == 0) {
            if (p.prev_me_stage_based_exit_th && S.l1s[li][ri][sw_][sh_] < (p.prev_me_stage_based_exit_th >> 2)) {
                S.l2x[li][ri][sw_][sh_] = S.l1x[li][ri][sw_][sh_];
                S.l2y[li][ri][sw_][sh_] = S.l1y[li][ri][sw_][sh_];
                S.l2s[li][ri][sw_][sh_] = S.l1s[li][ri][sw_][sh_];
            } else {
                // hme_level_2 (:1025-1113)
                int16_t sa_w = (int16_t)(((int16_t)p.hme_l2_sa.width + 7) & ~0x07), sa_h = (int16_t)p.hme_l2_sa.height;
                int16_t ox = (int16_t)(-(sa_w >> 1) + S.l1x[li][ri][sw_][sh_]);
                int16_t oy = (int16_t)(-(sa_h >> 1) + S.l1y[li][ri][sw_][sh_]);
                hme_clamp((int16_t)org_x, (int16_t)org_y, 63, 63, (int16_t)rp.width, (int16_t)rp.height, &ox, &oy, &sa_w, &sa_h);
                const int16_t x_tl = (int16_t)(((int16_t)rp.org_x + (int16_t)org_x) + ox);
                const int16_t y_tl = (int16_t)(((int16_t)rp.org_y + (int16_t)org_y) + oy);
                set_desc(sd, p, rp, x_tl, y_tl, sa_w, sa_h, 0);
                sd.aux_x = ox, sd.aux_y = oy, sd.aux_go = 1;
            }
        }
    }
    if (part == 0) {
        a = SearchArgs{true, nd, L.src_full, hme_sub ? 32u : 16u, b64_w, hme_sub ? b64_h >> 1 : b64_h, LDS::WIN_DW};
        return;
    }
    const bool go = mine && L.sh.desc[tid].aux_go;
    const int16_t q_ox = L.sh.desc[tid & 31].aux_x, q_oy = L.sh.desc[tid & 31].aux_y;
    if (go) {
        int16_t mx = S.l2x[li][ri][sw_][sh_], my = S.l2y[li][ri][sw_][sh_];
        decode_result(L.sh, tid, hme_sub, &S.l2s[li][ri][sw_][sh_], &mx, &my);
        S.l2x[li][ri][sw_][sh_] = (int16_t)(mx + q_ox);
        S.l2y[li][ri][sw_][sh_] = (int16_t)(my + q_oy);
    }
    __syncthreads();
}

// set_final_seach_centre_sb (:2252-2450) + hme_prune_ref_and_adjust_sr (:2547-2588); first wave, lane = reference slot
template <class LDS>
__device__ void centre_prune_lane0(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    if (tid < 64) {
        const Slot q = slot_of(tid, p, nlists);
        // the reference's loop carries (x, y, sad) from one reference to the next: a reference whose centre is not taken from the HME
        // (list 1 at temporal layer 0) gets the zero vector and the SAD of the reference before it in loop order, which then is the last
        // reference of list 0; without a level to take the centre from everything stays zero
        const bool from_l2 = p.enable_hme_level2_flag, from_l1 = !from_l2 && p.enable_hme_level1_flag;
        const bool from_l0 = !from_l2 && !from_l1 && p.enable_hme_level0_flag;
        const bool taken   = p.enable_hme_flag && (from_l0 || from_l1 || from_l2);
        int16_t    hx = 0, hy = 0;
        uint64_t   hsad = 0;
        if (taken && q.searched && (tl > 0 || q.li == 0)) {
            if (from_l2)
                best_quadrant(S.l2x[q.li][q.ri], S.l2y[q.li][q.ri], S.l2s[q.li][q.ri], &hx, &hy, &hsad);
            else if (from_l1)
                best_quadrant(S.l1x[q.li][q.ri], S.l1y[q.li][q.ri], S.l1s[q.li][q.ri], &hx, &hy, &hsad);
            else
                best_quadrant(S.l0x[q.li][q.ri], S.l0y[q.li][q.ri], S.l0s[q.li][q.ri], &hx, &hy, &hsad);
        }
        const uint64_t last_l0 = __shfl((unsigned long long)hsad, R0 > 0 ? R0 - 1 : 0, 64);  // slot (0, R0 - 1)
        if (q.searched && !(tl > 0 || q.li == 0))
            hx = hy = 0, hsad = (taken && R0 > 0) ? last_l0 : 0;
        // what the pruning rules below read: the slot's record as the loop above leaves it (untouched slots keep their initial values)
        SvtHipMeSearchResult &r = S.sr[q.li][q.ri];
        int16_t               sx = r.hme_sc_x, sy = r.hme_sc_y;
        uint64_t              sad = q.slot ? r.hme_sad : ~(uint64_t)0;
        if (q.searched) {
            sx = hx, sy = hy, sad = hsad;
            r.hme_sc_x = hx, r.hme_sc_y = hy, r.hme_sad = hsad;
        }
        if (p.enable_hme_flag && !p.me_mctf) {  // prune_ref = enable_hme_flag && me_type != ME_MCTF (:3173)
            const uint16_t th = p.prune_ref_if_hme_sad_dev_bigger_than_th;
            if (p.enable_me_hme_ref_pruning && th != (uint16_t)~0) {
                const uint64_t best = slots_min(sad);
                if (q.slot && q.ri >= 1 && (sad - best) * 100 > (th * best))
                    r.do_ref = 0;
            }
            if (p.enable_me_sr_adjustment && q.slot) {
                if (ABSV(sx) <= p.reduce_me_sr_based_on_mv_length_th && ABSV(sy) <= p.reduce_me_sr_based_on_mv_length_th &&
                    sad < p.stationary_hme_sad_abs_th)
                    S.reduce_div[q.li][q.ri] = p.stationary_me_sr_divisor;
                else if (sad < p.reduce_me_sr_based_on_hme_sad_abs_th)
                    S.reduce_div[q.li][q.ri] = p.me_sr_divisor_for_low_hme_sad;
            }
        }
    }
}

// Parts 1 and 2 of integer_search_b64 (below) for ALL references at once, lane = reference: with me_early_exit_th != 0 they read only
// the parameters, the HME results and the zero-vector SAD of their own reference (no check_00_center SADs, no first-reference rule),
// so five single-lane passes between barriers become one pass of five lanes.  Caller: uniform `p.me_early_exit_th != 0`.
template <class LDS>
__device__ void fullpel_prepare_all(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    const int16_t W = (int16_t)aw, H = (int16_t)ah, pad = 63;
    const int16_t ox_b = (int16_t)org_x, oy_b = (int16_t)org_y;
    if ((int)tid < nref) {
        const int f = (int)tid, li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0;
        uint16_t  dist = pic_dist(p, li, ri);
        int16_t   xc = S.sr[li][ri].hme_sc_x, yc = S.sr[li][ri].hme_sc_y;
        int16_t   sw = (int16_t)p.me_sa_min.width, sh_ = (int16_t)p.me_sa_min.height;
        if (!p.me_mctf)  // motion_estimation.c:1302
            dist = scaled_dist(dist);
        sw  = (int16_t)MINV((sw * dist), p.me_sa_max.width);
        sh_ = (int16_t)MINV((sh_ * dist), p.me_sa_max.height);
        if (p.mv_sa_adj_enabled && (!p.mv_sa_adj_nearest_ref_only || ri == 0)) {
            if (ABSV(xc) > p.mv_sa_adj_mv_size_th) sw = (int16_t)(sw * p.mv_sa_adj_sa_multiplier);
            if (ABSV(yc) > p.mv_sa_adj_mv_size_th) sh_ = (int16_t)(sh_ * p.mv_sa_adj_sa_multiplier);
        }
        sw  = (int16_t)((MAXV(1u, div_small((uint32_t)sw, S.reduce_div[li][ri])) + 7) & ~0x07u);
        sh_ = (int16_t)MAXV(3u, div_small((uint32_t)sh_, S.reduce_div[li][ri]));
        if (S.zz_sad[li][ri] < (p.me_early_exit_th / 6))
            sw = sh_ = 1;
        const int g = li * NR + ri;
        S.fp_xc[g] = xc, S.fp_yc[g] = yc, S.fp_sw[g] = sw, S.fp_sh[g] = sh_;
        S.fp_centre[g] = (uint8_t)(p.me_8x8_var_enabled && (sw * sh_ > 24));
    }
    __syncthreads();
    // The centre probes (:1393-1441) of all references, one wave per reference, straight from the reference plane: lane = one 8x8 block
    // of the b64 in the order of its p_best_sad_8x8 entry (4 * z-order index of its 16x16 + quadrant), 8 rows of two v_sad_u8 — instead of
    // a staged window and a 4-position quad search of which one position is wanted.  The 64 SADs stay in LDS for the reference's turn
    // (they are the position-0 entries of its 85 minima: fullpel_ref), the variance of them decides the window size below.
    uint16_t *const probe8  = (uint16_t *)L.src_q;  // [reference][64]; the down-scaled source blocks are dead by now
    uint32_t *const probe_v = L.src_s;              // [reference]
    {
        const uint32_t wv = tid >> 6, lane = tid & 63, sh1 = me_sub ? 1u : 0u;
        const uint32_t zo = lane >> 2, cq = lane & 3;
        const uint32_t zy = 2 * (zo >> 3) + ((zo >> 1) & 1), zx = 2 * ((zo >> 2) & 1) + (zo & 1);
        const uint32_t by = 2 * zy + (cq >> 1), bx = 2 * zx + (cq & 1);
        for (int f = (int)wv; f < nref; f += (int)(blockDim.x >> 6)) {
            const int li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0, g = li * NR + ri;
            if (!S.sr[li][ri].do_ref || !S.fp_centre[g])  // uniform over the wave
                continue;
            const SvtHipPlane8 &rp  = job.ref[li][ri].full;
            const uint8_t      *ref = plane_at(rp, (int)ox_b + S.fp_xc[g] + (int)(8 * bx), (int)oy_b + S.fp_yc[g] + (int)(8 * by));
            const uint32_t     *sb  = &L.src_full[(8 * by) * 16 + 2 * bx];
            uint32_t            acc = 0;
#pragma unroll
            for (uint32_t r = 0; r < 8; r++) {
                if (me_sub && (r & 1))
                    continue;
                const uint8_t *rr = ref + (size_t)(r * rp.stride);
                acc = __builtin_amdgcn_sad_u8(sb[r * 16], load_u32_any(rr), acc);
                acc = __builtin_amdgcn_sad_u8(sb[r * 16 + 1], load_u32_any(rr + 4), acc);
            }
            probe8[g * 64 + (int)lane] = (uint16_t)acc;  // <= 64 * 255
            const uint32_t mean = (wave_sum_all(acc) << sh1) / 64;
            const int32_t  dv   = (int32_t)(acc << sh1) - (int32_t)mean;
            const uint32_t ssq  = wave_sum((uint32_t)(dv * dv));
            if (lane == 0)
                probe_v[g] = ssq / 64;
        }
    }
    __syncthreads();
    // window size from the variance, final window (:1393-1561): lane = reference
    if ((int)tid < nref) {
        const int f = (int)tid, li = f < R0 ? 0 : 1, ri = f < R0 ? f : f - R0, g = li * NR + ri;
        int16_t   sw = S.fp_sw[g], sh_ = S.fp_sh[g], ox, oy;
        if (S.sr[li][ri].do_ref && S.fp_centre[g]) {
            const uint32_t var = probe_v[g];
            if (var > p.me_sr_mult2_th) {
                sw  = (int16_t)((MAXV(1, sw * 3 / 2) + 7) & ~0x7);
                sh_ = (int16_t)MAXV(1, sh_ * 3 / 2);
            }
            if (var < p.me_sr_div4_th) {
                sw  = (int16_t)((MAXV(1, sw >> 2) + 7) & ~0x7);
                sh_ = (int16_t)MAXV(1, sh_ >> 2);
                sh_ = (int16_t)MAXV(3, sh_);
            } else if (var < p.me_sr_div2_th) {
                sw  = (int16_t)((MINV(sw, sw >> 1) + 7) & ~0x7);
                sh_ = (int16_t)MINV(sh_, sh_ >> 1);
                sh_ = (int16_t)MAXV(3, sh_);
            }
        }
        clamp_me_window(S.fp_xc[g], S.fp_yc[g], ox_b, oy_b, W, H, pad, &sw, &sh_, &ox, &oy);
        S.fp_sw[g] = sw, S.fp_sh[g] = sh_, S.fp_ox[g] = ox, S.fp_oy[g] = oy;
        S.fp_inv[g] = make_inv_small((uint32_t)(sw > 0 ? sw : 0));
    }
    __syncthreads();
}

// integer_search_b64 (motion_estimation.c:1249-1586) for one reference; all threads.  `store`: write the winners to the
// output arrays and accumulate the 8x8 SAD sum (false when the reference is only evaluated for first_ref_sad64).  `prepared`:
// parts 1 and 2 come from fullpel_prepare_all.
template <class LDS>
__device__ void fullpel_ref(LDS &L, const Ctx &c, int li, int ri, bool store, bool prepared) {
    ME_CTX_LOCALS(c);
    const int16_t W = (int16_t)aw, H = (int16_t)ah, pad = 63;
    const int16_t ox_b = (int16_t)org_x, oy_b = (int16_t)org_y;
    const SvtHipPlane8 &rp = job.ref[li][ri].full;
    if (prepared) {
        // centre, final window and the centre probe's SADs come from fullpel_prepare_all: the 85 minima start from the probe's values
        // (position 0 of the scan order, :1393-1441) or empty
        const int g = li * NR + ri;
        if (tid == 0) {
            S.xc = S.fp_xc[g], S.yc = S.fp_yc[g], S.sw = S.fp_sw[g], S.sh = S.fp_sh[g], S.ox = S.fp_ox[g], S.oy = S.fp_oy[g];
            S.restage = 1;
        }
        if (S.fp_centre[g]) {  // uniform
            if (tid < 64) {
                const uint32_t sh1 = me_sub ? 1u : 0u;
                const uint32_t v   = ((const uint16_t *)L.src_q)[g * 64 + (int)tid];
                uint32_t       s16 = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
                s16                = dpp_add<0x4E>(s16);  // quad_perm [2,3,0,1]: the 16x16 of this lane's quad
                uint32_t s32       = dpp_add<0x124>(s16); // row_ror:4
                s32                = dpp_add<0x128>(s32); // row_ror:8: the 32x32 of this lane's row of 16
                const uint32_t s64 = wave_sum_all(v);
                L.bestkey[21 + tid] = (uint64_t)(v << sh1) << 32;
                if ((tid & 3) == 0)
                    L.bestkey[5 + (tid >> 2)] = (uint64_t)(s16 << sh1) << 32;
                if ((tid & 15) == 0)
                    L.bestkey[1 + (tid >> 4)] = (uint64_t)(s32 << sh1) << 32;
                if (tid == 0)
                    L.bestkey[0] = (uint64_t)(s64 << sh1) << 32;
            }
        } else {
            for (uint32_t pu = tid; pu < 85; pu += blockDim.x) L.bestkey[pu] = ((uint64_t)MAX_SAD_VALUE_ << 32) | 0xffffffffu;
        }
        __syncthreads();
    } else {
    // part 1: search area from settings + HME results
    if (tid == 0) {
        uint16_t dist = pic_dist(p, li, ri);
        int16_t  xc = S.sr[li][ri].hme_sc_x, yc = S.sr[li][ri].hme_sc_y;
        int16_t  sw = (int16_t)p.me_sa_min.width, sh_ = (int16_t)p.me_sa_min.height;
        if (!p.me_mctf)  // motion_estimation.c:1302
            dist = scaled_dist(dist);
        sw          = (int16_t)MINV((sw * dist), p.me_sa_max.width);
        sh_         = (int16_t)MINV((sh_ * dist), p.me_sa_max.height);
        if (p.mv_sa_adj_enabled && (!p.mv_sa_adj_nearest_ref_only || ri == 0)) {
            if (ABSV(xc) > p.mv_sa_adj_mv_size_th) sw = (int16_t)(sw * p.mv_sa_adj_sa_multiplier);
            if (ABSV(yc) > p.mv_sa_adj_mv_size_th) sh_ = (int16_t)(sh_ * p.mv_sa_adj_sa_multiplier);
        }
        sw  = (int16_t)((MAXV(1u, div_small((uint32_t)sw, S.reduce_div[li][ri])) + 7) & ~0x07u);
        sh_ = (int16_t)MAXV(3u, div_small((uint32_t)sh_, S.reduce_div[li][ri]));
        S.xc = xc, S.yc = yc, S.sw = sw, S.sh = sh_;
        S.need_zero_sad = 0, S.need_hme_sad = 0;
        if (p.me_early_exit_th) {
            if (S.zz_sad[li][ri] < (p.me_early_exit_th / 6))
                S.sw = S.sh = 1;
        } else if ((xc != 0 || yc != 0) && p.is_ref) {
            // check_00_center needs two 64x(h/2) SADs (:1139-1206)
            S.need_zero_sad = 1;  // me_early_exit_th == 0 here, so zz_sad is not available
            S.need_hme_sad  = 1;
            int16_t cx = xc, cy = yc;
            const int16_t RW = (int16_t)rp.width, RH = (int16_t)rp.height;
            cx = ((ox_b + cx) < -pad) ? (int16_t)(-pad - ox_b) : cx;
            cx = ((ox_b + cx) > RW - 1) ? (int16_t)(cx - ((ox_b + cx) - (RW - 1))) : cx;
            cy = ((oy_b + cy) < -pad) ? (int16_t)(-pad - oy_b) : cy;
            cy = ((oy_b + cy) > RH - 1) ? (int16_t)(cy - ((oy_b + cy) - (RH - 1))) : cy;
            S.xc = cx, S.yc = cy;
        }
    }
    __syncthreads();
    if (S.need_zero_sad) {
        wg_block_sad(L, plane_at(rp, ox_b, oy_b), rp.stride, b64_w, b64_h >> 1, 2);
        if (tid == 0)
            S.zero_sad = S.wg_sum;
        __syncthreads();
        wg_block_sad(L, plane_at(rp, ox_b + S.xc, oy_b + S.yc), rp.stride, b64_w, b64_h >> 1, 2);
        if (tid == 0)
            S.hme_mv_sad = S.wg_sum;
        __syncthreads();
    }
    // part 2: finish check_00_center, SR adjustment, decide on the centre probe
    if (tid == 0) {
        int16_t       sw = S.sw, sh_ = S.sh;
        const int16_t sw0 = sw, sh0 = sh_;
        if (!p.me_early_exit_th) {
            uint8_t  accurate     = 1;
            uint64_t best_hme_sad = ~(uint64_t)0;
            if (S.need_zero_sad) {
                const uint32_t zero_sad = S.zero_sad << 1;
                const uint32_t hme_sad  = S.hme_mv_sad << 1;
                const uint64_t zc = (uint64_t)(zero_sad << 8), hc = (uint64_t)(hme_sad << 8);
                const uint64_t cost = MINV(zc, hc);
                if (cost == zc)
                    S.xc = 0, S.yc = 0;
                best_hme_sad = hme_sad;
                if (S.xc == 0 && S.yc == 0)
                    accurate = 0;
            }
            if (p.enable_me_sr_adjustment == 2) {
                if ((accurate && (best_hme_sad < (24 * 24))) || (p.is_ref && S.sr[li][ri].hme_sad < (24 * 24)))
                    sh_ = (int16_t)(sh_ / 2);
                if ((li || ri) && S.first_ref_sad64 < 5000 && sh_ == sh0 && sw == sw0) {
                    sh_ = (int16_t)(sh_ >> 1);
                    sw  = (int16_t)(sw >> 1);
                }
            }
        }
        S.sw = sw, S.sh = sh_;
        S.do_centre = p.me_8x8_var_enabled && (sw * sh_ > 24);
        S.staged    = 0;
        if (S.do_centre) {
            // Stage once for the centre probe AND the search: the block of positions spanned by the centre and by
            // the window these settings give (the 8x8-variance rule below can only shrink it around the same
            // centre).  Origin congruent to the centre mod 4 keeps both dword-aligned in the staged tile.
            int16_t cw_ = sw, chh = sh_, cx0, cy0;
            clamp_me_window(S.xc, S.yc, ox_b, oy_b, W, H, pad, &cw_, &chh, &cx0, &cy0);
            int x0 = MINV((int)S.xc, (int)cx0), x1 = MAXV((int)S.xc + 1, (int)cx0 + (cw_ > 0 ? cw_ : 0));
            int y0 = MINV((int)S.yc, (int)cy0), y1 = MAXV((int)S.yc + 1, (int)cy0 + (chh > 0 ? chh : 0));
            x0 -= (x0 - (int)S.xc) & 3;
            if (x1 - x0 > (int)FP_TILE_W || (uint32_t)(y1 - y0 + 63) * fp_pitch((uint32_t)(x1 - x0)) > LDS::WIN_DW)
                x0 = S.xc, x1 = S.xc + 1, y0 = S.yc, y1 = S.yc + 1;  // does not fit: the centre alone
            S.stx = (int16_t)x0, S.sty = (int16_t)y0, S.stw = (int16_t)(x1 - x0), S.sth = (int16_t)(y1 - y0);
            S.staged = 1;
        }
    }
    for (uint32_t pu = tid; pu < 85; pu += blockDim.x) L.bestkey[pu] = ((uint64_t)MAX_SAD_VALUE_ << 32) | 0xffffffffu;
    __syncthreads();
    if (S.do_centre) {
        fp_stage(L, plane_at(rp, ox_b + S.stx, oy_b + S.sty), rp.stride, (uint32_t)S.stw, (uint32_t)S.sth);
        fp_search(L, fp_pitch((uint32_t)S.stw), (uint32_t)(S.xc - S.stx), (uint32_t)(S.yc - S.sty), 1, 1, 0, 1, me_sub);
        if (tid < 64) {  // variance of the 64 8x8 SADs at the centre (:1393-1441)
            const uint32_t v    = (uint32_t)(L.bestkey[21 + tid] >> 32);
            const uint32_t mean = (uint32_t)(L.bestkey[0] >> 32) / 64;
            const int32_t  dv   = (int32_t)v - (int32_t)mean;
            const uint32_t ssq  = wave_sum((uint32_t)(dv * dv));
            if (tid == 0) {
                int16_t        sw = S.sw, sh_ = S.sh;
                const uint32_t var = ssq / 64;
                if (var > p.me_sr_mult2_th) {
                    sw  = (int16_t)((MAXV(1, sw * 3 / 2) + 7) & ~0x7);
                    sh_ = (int16_t)MAXV(1, sh_ * 3 / 2);
                }
                if (var < p.me_sr_div4_th) {
                    sw  = (int16_t)((MAXV(1, sw >> 2) + 7) & ~0x7);
                    sh_ = (int16_t)MAXV(1, sh_ >> 2);
                    sh_ = (int16_t)MAXV(3, sh_);
                } else if (var < p.me_sr_div2_th) {
                    sw  = (int16_t)((MINV(sw, sw >> 1) + 7) & ~0x7);
                    sh_ = (int16_t)MINV(sh_, sh_ >> 1);
                    sh_ = (int16_t)MAXV(3, sh_);
                }
                S.sw = sw, S.sh = sh_;
            }
        }
        __syncthreads();
    }
    // part 3: final window (:1442-1561)
    if (tid == 0) {
        int16_t sw = S.sw, sh_ = S.sh, ox, oy;
        clamp_me_window(S.xc, S.yc, ox_b, oy_b, W, H, pad, &sw, &sh_, &ox, &oy);
        S.sw = sw, S.sh = sh_, S.ox = ox, S.oy = oy;
        // reuse the staged block if the final window lies inside it, dword-aligned
        S.restage = !(S.staged && ox >= S.stx && oy >= S.sty && ox + (sw > 0 ? sw : 0) <= S.stx + S.stw &&
                      oy + (sh_ > 0 ? sh_ : 0) <= S.sty + S.sth && ((ox - S.stx) & 3) == 0);
    }
    __syncthreads();
    }
    {
        const int      ox = S.ox, oy = S.oy;
        const uint32_t sw = (uint32_t)(S.sw > 0 ? S.sw : 0), sh_ = (uint32_t)(S.sh > 0 ? S.sh : 0);
        if (!S.restage) {
            fp_search(L, fp_pitch((uint32_t)S.stw), (uint32_t)(ox - S.stx), (uint32_t)(oy - S.sty), sw, sh_, 1, sw, me_sub);
        } else {
            // tiles that fit the window buffer
            const uint32_t tw = sw < FP_TILE_W ? sw : FP_TILE_W;
            const uint32_t pitch_t = fp_pitch(tw);
            uint32_t       th      = tw ? MINV(sh_, LDS::WIN_DW / pitch_t - 63) : 0;
            for (uint32_t ty = 0; th && ty < sh_; ty += th)
                for (uint32_t tx = 0; tx < sw; tx += tw) {
                    const uint32_t cw = MINV(tw, sw - tx), ch = MINV(th, sh_ - ty);
                    fp_stage(L, plane_at(rp, ox_b + ox + (int)tx, oy_b + oy + (int)ty), rp.stride, cw, ch);
                    fp_search(L, fp_pitch(cw), 0, 0, cw, ch, 1 + ty * sw + tx, sw, me_sub);
                }
        }
        // keys -> p_sb_best_sad / p_sb_best_mv of this reference
        uint32_t part = 0;  // this lane's share of the sum of the 64 8x8 SADs (tab8x8 is a permutation, :1608-1611)
        for (uint32_t pu = tid; pu < 85; pu += blockDim.x) {
            const uint64_t key    = L.bestkey[pu];
            const uint32_t ord    = (uint32_t)key;
            const uint32_t my_sad = (uint32_t)(key >> 32);
            if (li == 0 && ri == 0 && pu == 0)
                S.first_ref_sad64 = my_sad;
            if (pu >= 21)
                part += my_sad;
            if (!store)
                continue;
            gs[(li * NR + ri) * 85 + pu] = my_sad;
            if (ord != 0xffffffffu) {
                int16_t mx, my;
                if (ord == 0) {
                    mx = S.xc, my = S.yc;
                } else {
                    const uint32_t pos = ord - 1, row = prepared ? fast_div(pos, S.fp_inv[li * NR + ri]) : pos / sw;
                    mx = (int16_t)((int)(pos - row * sw) + ox), my = (int16_t)((int)row + oy);
                }
                gm[(li * NR + ri) * 85 + pu] = ((uint32_t)(uint16_t)my << 16) | (uint16_t)mx;
            }
        }
        part = wave_sum(part);
        if (store && (tid & 63) == 0 && part)
            atomicAdd((unsigned long long *)&S.me_sad_sum[li][ri], (unsigned long long)part);
    }
    __syncthreads();
}

// me_prune_ref (motion_estimation.c:1592-1635); first wave, lane = reference slot
template <class LDS>
__device__ void me_prune_lane0(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    if (tid < 64) {
        const Slot            q   = slot_of(tid, p, nlists);
        SvtHipMeSearchResult &r   = S.sr[q.li][q.ri];
        uint64_t              sad = q.slot ? r.hme_sad : ~(uint64_t)0;
        if (q.searched) {
            sad       = r.do_ref ? S.me_sad_sum[q.li][q.ri] : (uint64_t)(MAX_SAD_VALUE_ * 64);
            r.hme_sad = sad;
        }
        const uint16_t th = p.prune_ref_if_me_sad_dev_bigger_than_th;
        if (th != (uint16_t)~0) {
            const uint64_t best = slots_min(sad);
            if (q.slot && q.ri >= 1 && (sad - best) * 100 > (th * best))
                r.do_ref = 0;
        }
    }
}

// candidates (motion_estimation.c:3196-3211), compute_distortion (:3034-3077) and the per-reference search results
template <class LDS>
__device__ void finalize_b64(LDS &L, const Ctx &c) {
    ME_CTX_LOCALS(c);
    const uint32_t b64 = c.b64;
    // ---- candidates (motion_estimation.c:3196-3211) ----
    const SvtHipMeFrameOut &out    = job.out;
    const uint32_t          stored = p.enable_me_16x16 ? (p.enable_me_8x8 ? 85u : 21u) : 5u;
    uint32_t               *o_mv   = out.me_mv_array + (size_t)b64 * stored * p.max_refs;
    uint8_t                *o_cand = out.me_candidate_array + (size_t)b64 * stored * p.max_cand;
    uint8_t                *o_tot  = out.total_me_candidate_index + (size_t)b64 * stored;
    const bool single = p.num_of_ref_pic_to_search[0] == 1 && p.num_of_ref_pic_to_search[1] == 0;
    const bool mrpoff = p.num_of_ref_pic_to_search[0] == 1 && p.num_of_ref_pic_to_search[1] == 1;
    if ((single || mrpoff) && tid < stored)
        o_tot[tid] = 1;
    __syncthreads();
    if (tid < p.max_number_of_pus_per_sb && tid < 85) {
        if (single)
            cand_single_ref(L, p, gs, gm, tid, o_mv, o_cand);
        else if (mrpoff)
            cand_mrp_off(L, p, gs, gm, tid, (uint32_t)nlists, o_mv, o_cand, o_tot);
        else
            cand_general(L, p, gs, gm, tid, (uint32_t)nlists, o_mv, o_cand, o_tot);
    }
    __syncthreads();

    // ---- compute_distortion (motion_estimation.c:3034-3077) + result write-back ----
    // one wave: lane i holds the 8x8 entry i (and the 16x16 / 32x32 / 64x64 entries in its first lanes); sums by wave reductions
    // (the sum of squared deviations in 64 bits, as the reference's uint64 accumulation)
    if (tid < 64) {
        const uint32_t e8 = L.me_dist[21 + tid];
        const uint32_t d8 = wave_sum_all(e8);
        const uint32_t d16 = wave_sum_all(tid < 16 ? L.me_dist[5 + tid] : 0u), d32 = wave_sum_all(tid < 4 ? L.me_dist[1 + tid] : 0u);
        const uint64_t mean = d8 / 64;
        const int64_t  dv   = (int64_t)e8 - (int64_t)mean;
        const uint64_t sq   = (uint64_t)(dv * dv);
        const uint32_t lo = wave_sum_all((uint32_t)sq & 0xffffu), mid = wave_sum_all((uint32_t)(sq >> 16) & 0xffffu),
                       hi = wave_sum_all((uint32_t)(sq >> 32) & 0xffffu), top = wave_sum_all((uint32_t)(sq >> 48));
        if (tid == 0) {
            const uint64_t ssq = (uint64_t)lo + ((uint64_t)mid << 16) + ((uint64_t)hi << 32) + ((uint64_t)top << 48);
            const uint32_t d64 = L.me_dist[0];
            const uint32_t pix = b64_w * b64_h;
            out.me_8x8_cost_variance[b64] = (uint32_t)(ssq / 64);
            out.rc_me_distortion[b64]     = p.input_resolution_le_480p ? d8 : d16;
            out.me_64x64_distortion[b64]  = div_small(d64 * 4096u, pix);
            out.me_32x32_distortion[b64]  = div_small(d32 * 4096u, pix);
            out.me_16x16_distortion[b64]  = div_small(d16 * 4096u, pix);
            out.me_8x8_distortion[b64]    = div_small(d8 * 4096u, pix);
        }
    }
    if (tid < NL * NR)
        out.search_results[(size_t)b64 * NL * NR + tid] = (&S.sr[0][0])[tid];
}


// Tuning aid (make ABLATE=1): svt_hip_debug_me_stop(k) makes every workgroup return behind stage k, so that the time of the
// kernel truncated there can be measured with the real overlap between workgroups (results are garbage).
#ifdef SVT_HIP_ME_ABLATE
__device__ int g_me_stop = 99;
#define ME_STOP(k)          \
    do {                    \
        if (me_stop == (k)) \
            return;         \
    } while (0)
#else
#define ME_STOP(k) \
    do {           \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// One launch, one workgroup per b64: all stages back to back.
// ------------------------------------------------------------------------------------------------
#ifndef SVT_HIP_ME_WGS
#define SVT_HIP_ME_WGS 8  // workgroups per CU the register allocation aims for (64 VGPRs)
#endif
// MCTF: a name tag only (the code is the same): launches whose pictures all run in ME_MCTF mode (the temporal filter's use of
// this function) show up as me_b64_kernel<true> in profiles, apart from the open-loop launches
template <bool MCTF>
__global__ __launch_bounds__(WG_THREADS, SVT_HIP_ME_WGS) void me_b64_kernel(const SvtHipMeFrameJob *__restrict__ jobs) {
    __shared__ MeLds L;
#ifdef SVT_HIP_ME_PROFILE
    unsigned long long prof_last = wall_clock64();
#endif
    // the job descriptor (parameters, plane descriptors, output pointers) is read hundreds of times by the scalar stage
    // logic: keep a copy in LDS instead of going back to memory for every field
    __shared__ SvtHipMeFrameJob sjob;
    for (uint32_t i = threadIdx.x; i < sizeof(SvtHipMeFrameJob) / 4; i += blockDim.x)
        ((uint32_t *)&sjob)[i] = ((const uint32_t *)&jobs[blockIdx.y])[i];
    __syncthreads();
    Ctx c;
    if (!make_ctx(c, sjob, blockIdx.x))
        return;
#ifdef SVT_HIP_ME_ABLATE
    const int me_stop = g_me_stop;
#endif
    ME_CTX_LOCALS(c);
    stage_sources(L, c, 7);
    for (uint32_t i = tid; i < NL * NR * 85; i += blockDim.x) gs[i] = 0, gm[i] = 0;
    init_state_lane0(L, c);
    __syncthreads();
    ME_PHASE(0);
    ME_STOP(0);
    if (p.me_early_exit_th || p.me_safe_limit_zz_th) {
        zz_sad_all(L, c);
        zz_prune_lane0(L, c);
        __syncthreads();
    }
    ME_PHASE(1);
    ME_STOP(1);
    {
        // List 1 reads list 0's pre-HME results for the l1 early exit (:1771-1781) and for the tl == 0 mirror (:1853-1860):
        // only then does it need a round of its own.  With distance-based resizing, references other than the first read
        // the first one's quadrant-(0,0) level-0 vector (get_hme_l0_search_area, :1881-1890): the first reference then
        // gets a round of its own.
        const bool pre = p.prehme_enable, split = nlists == 2 && (p.prehme_l1_early_exit || tl == 0);
        const bool l0 = p.enable_hme_flag && p.enable_hme_level0_flag, l1 = p.enable_hme_flag && p.enable_hme_level1_flag;
        const bool l2 = p.enable_hme_flag && p.enable_hme_level2_flag;
        const bool dep = p.enable_me_sr_adjustment && p.distance_based_hme_resizing && p.reduce_hme_l0_sr_th_min &&
            p.reduce_hme_l0_sr_th_max && nref > 1;
        for (int step = 0; step < 6; step++) {
            if (step == 2) {  // after pre-HME
                if (pre) {
                    phme_prune_lane0(L, c);
                    __syncthreads();
                }
                ME_PHASE(2);
                ME_STOP(2);
            } else if (step == 4) {
                ME_PHASE(3);
                ME_STOP(3);
            } else if (step == 5) {
                ME_PHASE(4);
                ME_STOP(4);
                // the down-scaled stages' windows may have overwritten the full resolution source block: stage it again
                // (the barrier that ends the last search orders the window reads before these writes)
                if (pre || l0 || l1) {
                    stage_sources(L, c, 1);
                    __syncthreads();
                }
            }
            const bool active = step == 0 ? pre : step == 1 ? (pre && split) : step == 2 ? l0 : step == 3 ? (l0 && dep) : step == 4 ? l1 : l2;
            if (!active)
                continue;
            const int f0 = step == 1 ? R0 : (step == 3 ? 1 : 0);
            const int f1 = step == 0 ? (split ? R0 : nref) : (step == 2 ? (dep ? 1 : nref) : nref);
            SearchArgs a;
            for (int part = 0; part < 2; part++) {
                if (step < 2)
                    prehme_round(L, c, f0, f1, step == 0 || tl > 0, part, a);
                else if (step < 4)
                    hme_l0_round(L, c, f0, f1, part, a);
                else if (step == 4)
                    hme_l1_round(L, c, f0, f1, part, a);
                else
                    hme_l2_round(L, c, f0, f1, part, a);
                if (part == 0 && a.search) {
                    ME_SITE(step < 2 ? 0 : (step < 4 ? 1 : step - 2));
                    wg_multi_search(L.sh, a.nd, a.src, a.row_dw, a.bw, a.bh, L.win, a.cap);
                }
            }
        }
    }
    ME_PHASE(5);
    ME_STOP(5);
    centre_prune_lane0(L, c);
    __syncthreads();
    ME_PHASE(6);
    ME_STOP(6);
    // ME_MCTF (the temporal filter's use of this function): a block whose first reference already matches well keeps its HME
    // vector and skips the full-pel search (:3179-3183); no pruning, candidates or distortion statistics (:3173, 3196)
    const bool tf_exit = p.me_mctf && S.sr[0][0].hme_sad < p.tf_me_exit_th;  // uniform: LDS value written before the last barrier
    if (!tf_exit) {
        const bool prepared = p.me_early_exit_th != 0;  // uniform
        if (prepared)
            fullpel_prepare_all(L, c);
        for (int li = 0; li < nlists; ++li)
            for (int ri = 0; ri < p.num_of_ref_pic_to_search[li]; ++ri)
                if (S.sr[li][ri].do_ref)  // uniform
                    fullpel_ref(L, c, li, ri, true, prepared);
    }
    ME_PHASE(7);
    ME_STOP(7);
    if (p.me_mctf) {
        if (tid < NL * NR)
            job.out.search_results[(size_t)c.b64 * NL * NR + tid] = (&S.sr[0][0])[tid];
        return;
    }
    if (p.enable_hme_flag && p.enable_me_hme_ref_pruning) {
        me_prune_lane0(L, c);
        __syncthreads();
    }
    ME_PHASE(8);
    finalize_b64(L, c);
    ME_PHASE(10);
}

}  // namespace

#ifdef SVT_HIP_ME_ABLATE
extern "C" __attribute__((visibility("default"))) int32_t svt_hip_debug_me_stop(int32_t k) {
    SVT_HIP_CHECK(hipDeviceSynchronize());
    const int32_t stop = k & 0xff, skip = k >> 8;  // bits 8.. = what wg_multi_search leaves out (see g_ms_skip)
    SVT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_me_stop), &stop, sizeof(stop)));
    SVT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ms_skip), &skip, sizeof(skip)));
    return SVT_HIP_OK;
}
#endif

#ifdef SVT_HIP_ME_PROFILE
// tuning aid, only in PROF=1 builds: copies (and optionally clears) the phase counters
extern "C" __attribute__((visibility("default"))) int32_t svt_hip_debug_me_profile(unsigned long long out[32], int32_t reset) {
    SVT_HIP_CHECK(hipDeviceSynchronize());
    SVT_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_me_prof), sizeof(unsigned long long) * 11));
    SVT_HIP_CHECK(hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_ms_prof), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {};
        SVT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_me_prof), z, sizeof(z)));
        SVT_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ms_prof), z, sizeof(unsigned long long) * 16));
    }
    return SVT_HIP_OK;
}
#endif

extern "C" uint32_t svt_hip_me_b64_count(uint32_t width, uint32_t height) {
    const uint32_t aw = (width + 7u) & ~7u, ah = (height + 7u) & ~7u;
    return ((aw + 63) / 64) * ((ah + 63) / 64);
}

static int32_t validate_job(const SvtHipMeFrameJob &j, uint32_t idx) {
    const SvtHipMeParams &p = j.prm;
    auto bad = [&](const char *what) {
        set_error("svt_hip_me_frames: job %u: %s", idx, what);
        return (int32_t)SVT_HIP_ERR_BAD_PARAMETER;
    };
    if (p.num_hme_sa_w != 2 || p.num_hme_sa_h != 2)
        return bad("num_hme_sa_w/h must be 2 (as in the reference, motion_estimation.c:1945)");
    if (p.num_of_list_to_search < 1 || p.num_of_list_to_search > 2)
        return bad("num_of_list_to_search must be 1 or 2");
    for (int l = 0; l < 2; l++)
        if (p.num_of_ref_pic_to_search[l] > SVT_HIP_ME_MAX_REF)
            return bad("too many reference pictures");
    if (p.num_of_ref_pic_to_search[0] < 1)
        return bad("list 0 needs at least one reference");
    if (p.max_number_of_pus_per_sb != 85)
        return bad("max_number_of_pus_per_sb must be 85 (SQUARE_PU_COUNT)");
    if (p.max_refs == 0 || p.max_cand == 0)
        return bad("max_refs / max_cand not set");
    if (!j.src.full.buf || !j.src.quarter.buf || !j.src.sixteenth.buf)
        return bad("source pyramid incomplete");
    if (j.src.full.width < 64 || j.src.full.height < 64 || j.src.full.width > 16384 || j.src.full.height > 8704)
        return bad("unsupported picture size");
    if (j.src.full.org_x < 64 || j.src.full.org_y < 64)
        return bad("full-resolution planes need >= 64 samples of padding");
    // geometry the kernels rely on (include/svt_hip_me.h "Memory contract"): decimated planes are exactly 1/2 and 1/4 of the
    // full one, padded by at least 32 / 16 samples, rows at least as long as picture + both paddings
    auto pyramid_ok = [](const SvtHipPyramid8 &y) {
        const SvtHipPlane8 *pl[3] = {&y.full, &y.quarter, &y.sixteenth};
        const uint32_t      pad[3] = {64, 32, 16};
        for (int i = 0; i < 3; i++) {
            if (pl[i]->org_x < pad[i] || pl[i]->org_y < pad[i])
                return false;
            if (pl[i]->width != (y.full.width >> i) || pl[i]->height != (y.full.height >> i))
                return false;
            if (pl[i]->stride < (uint32_t)pl[i]->width + 2u * pl[i]->org_x)
                return false;
        }
        return true;
    };
    if (!pyramid_ok(j.src))
        return bad("source pyramid: decimated planes must be width>>1 / width>>2 with >= 32 / 16 samples of padding and stride >= width + 2 * org_x");
    for (int l = 0; l < p.num_of_list_to_search; l++)
        for (int r = 0; r < p.num_of_ref_pic_to_search[l]; r++) {
            const SvtHipPyramid8 &y = j.ref[l][r];
            if (!y.full.buf || !y.quarter.buf || !y.sixteenth.buf)
                return bad("reference pyramid incomplete");
            if (y.full.width != j.src.full.width || y.full.height != j.src.full.height)
                return bad("reference size differs from the source (scaled references are not supported)");
            if (y.full.org_x < 64 || y.full.org_y < 64)
                return bad("reference planes need >= 64 samples of padding");
            if (!pyramid_ok(y))
                return bad("reference pyramid: decimated planes must be width>>1 / width>>2 with >= 32 / 16 samples of padding and stride >= width + 2 * org_x");
        }
    const void *outs[] = {j.out.best_sad, j.out.best_mv, j.out.search_results, j.out.me_mv_array, j.out.me_candidate_array,
                          j.out.total_me_candidate_index, j.out.me_64x64_distortion, j.out.me_32x32_distortion,
                          j.out.me_16x16_distortion, j.out.me_8x8_distortion, j.out.me_8x8_cost_variance,
                          j.out.rc_me_distortion};
    for (const void *o : outs)
        if (!o)
            return bad("NULL output array");
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_me_validate_jobs(const SvtHipMeFrameJob *jobs, uint32_t n_jobs, uint32_t *max_b64_out) {
    if (!jobs || n_jobs == 0 || n_jobs > 65535) {
        set_error("svt_hip_me_frames: bad job count");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    uint32_t max_b64 = 0;
    for (uint32_t i = 0; i < n_jobs; i++) {
        const int32_t rc = validate_job(jobs[i], i);
        if (rc != SVT_HIP_OK)
            return rc;
        const uint32_t nb = svt_hip_me_b64_count(jobs[i].src.full.width, jobs[i].src.full.height);
        max_b64           = nb > max_b64 ? nb : max_b64;
    }
    if (max_b64_out)
        *max_b64_out = max_b64;
    return SVT_HIP_OK;
}

static int32_t launch_me(const SvtHipMeFrameJob *d_jobs, uint32_t n_jobs, uint32_t max_b64, hipStream_t st, bool mctf) {
    if (mctf)
        hipLaunchKernelGGL(me_b64_kernel<true>, dim3((max_b64 + 7) / 8 * 8, n_jobs), dim3(WG_THREADS), 0, st, d_jobs);
    else
        hipLaunchKernelGGL(me_b64_kernel<false>, dim3((max_b64 + 7) / 8 * 8, n_jobs), dim3(WG_THREADS), 0, st, d_jobs);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_me_frames_dev(const SvtHipMeFrameJob *d_jobs, uint32_t n_jobs, uint32_t max_b64, void *stream) {
    if (!d_jobs || n_jobs == 0 || n_jobs > 65535 || max_b64 == 0) {
        set_error("svt_hip_me_frames_dev: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    return launch_me(d_jobs, n_jobs, max_b64, resolve_stream(stream), false);
}

extern "C" int32_t svt_hip_me_frames(const SvtHipMeFrameJob *jobs, uint32_t n_jobs, void *stream) {
    uint32_t      max_b64 = 0;
    const int32_t vrc     = svt_hip_me_validate_jobs(jobs, n_jobs, &max_b64);
    if (vrc != SVT_HIP_OK)
        return vrc;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t st = resolve_stream(stream);
    // Job descriptors travel through the per-thread staging ring (common.hpp): no allocation on the launch path.
    const SvtHipMeFrameJob *d_jobs = (const SvtHipMeFrameJob *)stage_descriptors(jobs, sizeof(SvtHipMeFrameJob) * n_jobs, st);
    if (!d_jobs)
        return SVT_HIP_ERR_RUNTIME;
    bool mctf = true;  // (host copy of the descriptors: the mode of every picture is known here)
    for (uint32_t i = 0; i < n_jobs; i++) mctf = mctf && jobs[i].prm.me_mctf;
    const int32_t rc = launch_me(d_jobs, n_jobs, max_b64, st, mctf);
    stage_commit(st);
    return rc;
}

extern "C" int32_t svt_hip_install_rtcd_me(void **table, uint32_t n_slots) {
    if (!table || n_slots < SVT_HIP_SLOT_ME_COUNT) {
        set_error("svt_hip_install_rtcd_me: table too small");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;  // caller keeps its CPU pointers
    void *fn[SVT_HIP_SLOT_ME_COUNT] = {
        (void *)svt_sad_loop_kernel_hip,
        (void *)svt_nxm_sad_kernel_hip,
        (void *)svt_ext_all_sad_calculation_8x8_16x16_hip,
        (void *)svt_ext_eight_sad_calculation_32x32_64x64_hip,
        (void *)svt_ext_sad_calculation_8x8_16x16_hip,
        (void *)svt_ext_sad_calculation_32x32_64x64_hip,
        (void *)svt_aom_downsample_2d_hip,
        (void *)svt_compute_interm_var_four8x8_hip,
        (void *)svt_compute_sub_mean_8x8_hip,
        (void *)svt_compute_mean_8x8_hip,
        (void *)svt_compute_mean_square_values_8x8_hip,
    };
    for (uint32_t i = 0; i < SVT_HIP_SLOT_ME_COUNT; i++)
        if (table[i])
            *(void **)table[i] = fn[i];
    return SVT_HIP_OK;
}

SVT_HIP_MODULE_WARMUP(me_frame)
