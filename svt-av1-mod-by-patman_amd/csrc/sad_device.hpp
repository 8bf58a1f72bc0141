// sad_device.hpp — workgroup-level exhaustive SAD search for gfx950 (wave64, LDS-staged windows).
//
// Replaces the triple loop of svt_sad_loop_kernel (Source/Lib/C_DEFAULT/compute_sad_c.c:58-101 of the
// reference) by:
//   * staging the search window(s) once into LDS with coalesced, dword-aligned global loads
//     (unaligned window origins are re-aligned with v_alignbyte_b32 while staging),
//   * one lane per QUAD of 4 horizontally adjacent search positions, using v_qsad_pk_u16_u8
//     (4 positions x 4 pixels per instruction) on dwords read from LDS,
//   * a 64-bit key (sad << 32 | raster index) reduced with ds_min_u64, which reproduces the
//     reference's "first minimum in raster order, strict <" tie-breaking exactly.
// Several independent searches that share one source block (the 4 HME quadrants, both pre-HME
// regions ...) are processed in one pass so that all 256 lanes have work.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svthip {
namespace dev {

constexpr int      WG_THREADS = 256;
constexpr int      MAX_SEARCH = 32;
constexpr uint32_t SEG_ALIGN  = 4;    // segments of the window buffer start on 16-byte boundaries
constexpr uint64_t KEY_NONE   = ((uint64_t)0xffffffu << 32) | 0xffffffffu;  // best starts at 0xffffff
constexpr int      MAX_PASS   = 9;     // passes the plan-once path can hold (more: the per-pass planner takes over)

struct SearchDesc {
    const uint8_t *ref;         // global: window origin (search position (0,0), block row 0)
    uint32_t       ref_stride;  // bytes between consecutive block rows
    uint32_t       raw_stride;  // bytes between consecutive search rows (src_stride_raw of the reference)
    int32_t        sa_w, sa_h;
    uint32_t       skip;  // search only odd rows (skip_search_line && bw==16 && bh<=16)
    // --- plan (filled by search_plan) ---
    uint32_t k;         // ref_stride / raw_stride
    uint32_t fast;      // LDS fast path usable
    uint32_t nq;        // quads per search row
    uint32_t no;        // work items per search row: pairs of quads (8 adjacent positions), the last one may be half
    uint32_t n_srows;   // searched rows
    uint32_t pitch_dw;  // LDS row pitch of the staged window (dwords)
    uint32_t rows_per_pos;  // raw rows spanned by one search position: (bh-1)*k + 1
    uint32_t inv_no;    // ceil(2^32 / no): exact quotients for the small dividends used here
    // --- caller's per-descriptor state that has to survive the search (kept here rather than in registers: live values
    // across the inlined search are what the register allocator spills) ---
    int16_t  aux_x, aux_y;
    uint32_t aux_go;
};

struct SearchSeg {
    uint32_t d;          // descriptor
    uint32_t j0, nj;     // searched-row range [j0, j0+nj)
    uint32_t lds_dw;     // window offset in the LDS window buffer
    uint32_t nstage;     // raw rows staged
    uint32_t item_base;  // first work item of this segment
};

template <int NS> struct SearchSharedT {
    static constexpr int MAXS = NS;  // descriptors (and segments per pass) this instance can hold
    SearchDesc desc[NS];
    SearchSeg  seg[NS];
    uint64_t   best[NS];
    uint32_t   nseg, nitems, nstage_dw;
    uint32_t   next_d, next_j;  // continuation point of the per-pass planner
    // plan of ALL passes (made once when no descriptor has to be split): pass p = segments [pass_seg0[p], pass_seg0[p+1]),
    // pass_item0[p+1] - pass_item0[p] work items
    uint32_t   npass;           // 0: no such plan, the per-pass planner runs
    uint32_t   pass_item0[MAX_PASS + 1];
    uint8_t    pass_seg0[MAX_PASS + 3];
#ifdef SVT_HIP_ME_PROFILE
    uint32_t   prof_site;  // PROF=1 builds: which caller the phase times are booked to
#endif
};
using SearchShared = SearchSharedT<MAX_SEARCH>;

#ifdef SVT_HIP_ME_ABLATE
static __device__ int g_ms_skip;  // ABLATE=1 builds: bit 0 = no window staging loads, bit 1 = no SAD arithmetic, bit 2 = no work items (instruction accounting)
#endif
#ifdef SVT_HIP_ME_PROFILE
static __device__ unsigned long long g_ms_prof[16];  // [call site][plan, stage, search, -] (per translation unit)
#define MS_PHASE(i)                                                 \
    do {                                                            \
        if (threadIdx.x == 0) {                                     \
            const unsigned long long t_ = wall_clock64();           \
            atomicAdd(&g_ms_prof[(sh.prof_site & 3u) * 4 + i], t_ - ms_last);                 \
            ms_last = t_;                                           \
        }                                                           \
    } while (0)
#else
#define MS_PHASE(i) \
    do {            \
    } while (0)
#endif

__device__ __forceinline__ uint64_t pair64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
// One dword from any byte address: amdhsa runs the memory pipeline in unaligned-access mode, so this is a single
// global_load_dword (window origins are arbitrary byte offsets into the padded planes).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// Picture pointers reach the kernels through descriptors kept in LDS, so the compiler sees generic pointers and would emit
// flat loads (which also count against the LDS counter: every wait on one then waits for outstanding LDS traffic too).
// They are always global memory: say so.
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *g) {
    return *(const __attribute__((address_space(1))) u32_unaligned *)g;
}
// 16 bytes from any byte address: one global_load_dwordx4 (unaligned-access mode, as above)
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u128_unaligned;
typedef uint32_t __attribute__((ext_vector_type(4))) u128v;
__device__ __forceinline__ u128v load_u128_any(const uint8_t *g) {
    return *(const __attribute__((address_space(1))) u128_unaligned *)g;
}

// products of two values below 2^24 with a result below 2^32: one full-rate v_mul_u32_u24 instead of the quarter-rate v_mul_lo_u32
// (declared as the LLVM intrinsic: `__umul24` is expanded to shifts and a generic multiply that the selector does not always narrow again)
extern "C" __device__ unsigned sad_mul_u24(unsigned, unsigned) __asm("llvm.amdgcn.mul.u24");

// Copy `rows` rows of `pitch` dwords from global memory (row r starts at byte g0 + r * rstride, any alignment) into LDS rows
// of `pitch` dwords at w0, with `lanes` lanes of which this one is number `lid`.  2 / 4 / 8 / 16 / 32 lanes share a row, one
// 16-byte load each (a fourth of the load / address instructions of a dword-per-lane copy), two rows in flight per lane.
// Reads up to 15 bytes past the last dword of a row: callers guarantee those bytes exist (they are never used) — for
// picture planes they are the next row's first bytes, see the contract in include/svt_hip_me.h.
__device__ __forceinline__ void stage_rows16(uint32_t *__restrict__ w0, uint32_t pitch, const uint8_t *g0, uint32_t rstride,
                                             uint32_t rows, uint32_t lanes, uint32_t lid) {
    const uint32_t nch = (pitch + 3) >> 2;
    const uint32_t lg  = nch <= 2 ? 1u : (nch <= 4 ? 2u : (nch <= 8 ? 3u : (nch <= 16 ? 4u : 5u)));
    const uint32_t lpr = 1u << lg, rpi = lanes >> lg;  // lanes per row, rows per iteration
    const uint32_t c0 = lid & (lpr - 1), r0 = lid >> lg;
    for (uint32_t ch = c0; ch < nch; ch += lpr) {
        const uint32_t  nd = pitch - 4 * ch;  // dwords of this chunk inside the row (>= 1)
        const uint8_t  *g  = g0 + 16 * ch;
        uint32_t       *w  = w0 + 4 * ch;
        for (uint32_t row = r0; row < rows; row += 2 * rpi) {
            const uint32_t row1 = row + rpi;
            const bool     two  = row1 < rows;
            const u128v    a    = load_u128_any(g + (size_t)sad_mul_u24(row, rstride));  // rows and strides are far below 2^24
            u128v          b    = {0u, 0u, 0u, 0u};
            if (two)
                b = load_u128_any(g + (size_t)sad_mul_u24(row1, rstride));
            uint32_t *wa = w + sad_mul_u24(row, pitch), *wb = w + sad_mul_u24(row1, pitch);
            wa[0] = a.x;
            if (nd > 1) wa[1] = a.y;
            if (nd > 2) wa[2] = a.z;
            if (nd > 3) wa[3] = a.w;
            if (two) {
                wb[0] = b.x;
                if (nd > 1) wb[1] = b.y;
                if (nd > 2) wb[2] = b.z;
                if (nd > 3) wb[3] = b.w;
            }
        }
    }
}

// n / d for n * d < 2^32 with inv = ceil(2^32 / d) (d >= 2; d == 1 is handled by the caller passing inv = 0)
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t inv) { return inv ? __umulhi(n, inv) : n; }
// ceil(2^32 / d) == floor((2^32 - 1) / d) + 1 for every d >= 1: a 32-bit division (a 64-bit one is ~100 instructions here)
__device__ __forceinline__ uint32_t make_inv(uint32_t d) { return d > 1 ? 0xffffffffu / d + 1u : 0u; }
// the same from a table for the divisors the search plans use most (items per search row): a 32-bit division is ~40 instructions,
// a third of them quarter-rate, in a section that one wave runs alone
__device__ const uint32_t INV_SMALL[65] = {
    0u, 0u, 0x80000000u, 0x55555556u, 0x40000000u, 0x33333334u, 0x2aaaaaabu, 0x24924925u, 0x20000000u, 0x1c71c71du, 0x1999999au, 0x1745d175u, 0x15555556u,
    0x13b13b14u, 0x12492493u, 0x11111112u, 0x10000000u, 0x0f0f0f10u, 0x0e38e38fu, 0x0d79435fu, 0x0ccccccdu, 0x0c30c30du, 0x0ba2e8bbu, 0x0b21642du, 0x0aaaaaabu,
    0x0a3d70a4u, 0x09d89d8au, 0x097b425fu, 0x0924924au, 0x08d3dcb1u, 0x08888889u, 0x08421085u, 0x08000000u, 0x07c1f07du, 0x07878788u, 0x07507508u, 0x071c71c8u,
    0x06eb3e46u, 0x06bca1b0u, 0x06906907u, 0x06666667u, 0x063e7064u, 0x06186187u, 0x05f417d1u, 0x05d1745eu, 0x05b05b06u, 0x0590b217u, 0x0572620bu, 0x05555556u,
    0x0539782au, 0x051eb852u, 0x05050506u, 0x04ec4ec5u, 0x04d4873fu, 0x04bda130u, 0x04a7904bu, 0x04924925u, 0x047dc120u, 0x0469ee59u, 0x0456c798u, 0x04444445u,
    0x04325c54u, 0x04210843u, 0x04104105u, 0x04000000u};
__device__ __forceinline__ uint32_t make_inv_small(uint32_t d) { return d <= 64 ? INV_SMALL[d] : make_inv(d); }

// SAD of 4 adjacent positions (window dword `w` onwards) against one block.  src rows are dwords in LDS.
__device__ __forceinline__ void quad_sad(const uint32_t *__restrict__ w, uint32_t w_row_dw,
                                         const uint32_t *__restrict__ s, uint32_t s_row_dw, uint32_t bw, uint32_t bh,
                                         uint32_t out[4]) {
    const uint32_t nfull = bw >> 2, tail = bw & 3;
    const uint32_t rows_per_flush = nfull ? (64u / nfull ? 64u / nfull : 1u) : 64u;
    const uint32_t tmask          = tail ? ((1u << (8 * tail)) - 1u) : 0u;
    uint32_t       a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint64_t       acc = 0;
    uint32_t       since = 0;
    for (uint32_t r = 0; r < bh; r++) {
        const uint32_t *wr = w + r * w_row_dw;
        const uint32_t *sr = s + r * s_row_dw;
        uint32_t        lo = wr[0];
        for (uint32_t i = 0; i < nfull; i++) {
            const uint32_t hi = wr[i + 1];
#ifdef SVT_HIP_NO_QSAD  // diagnostic build: same arithmetic with v_sad_u8 only
            a0 = __builtin_amdgcn_sad_u8(lo, sr[i], a0);
            a1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 1u), sr[i], a1);
            a2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 2u), sr[i], a2);
            a3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 3u), sr[i], a3);
#else
            acc = __builtin_amdgcn_qsad_pk_u16_u8(pair64(lo, hi), sr[i], acc);
#endif
            lo = hi;
        }
        if (tail) {
            const uint32_t hi = wr[nfull + 1];
            const uint32_t sv = sr[nfull] & tmask;
            a0 = __builtin_amdgcn_sad_u8(lo & tmask, sv, a0);
            a1 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 1u) & tmask, sv, a1);
            a2 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 2u) & tmask, sv, a2);
            a3 = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 3u) & tmask, sv, a3);
        }
        if (++since == rows_per_flush) {
            a0 += (uint32_t)(acc & 0xffff), a1 += (uint32_t)((acc >> 16) & 0xffff);
            a2 += (uint32_t)((acc >> 32) & 0xffff), a3 += (uint32_t)(acc >> 48);
            acc = 0, since = 0;
        }
    }
    a0 += (uint32_t)(acc & 0xffff), a1 += (uint32_t)((acc >> 16) & 0xffff);
    a2 += (uint32_t)((acc >> 32) & 0xffff), a3 += (uint32_t)(acc >> 48);
    out[0] = a0, out[1] = a1, out[2] = a2, out[3] = a3;
}

// Same result for a block of exactly BW = 16 / 32 / 64 samples per row whose source rows are 16-byte aligned in LDS:
// the row body is fully unrolled (b128 source reads, no per-dword loop control).
template <uint32_t BW>
__device__ __forceinline__ void quad_sad_fixed(const uint32_t *__restrict__ w, uint32_t w_row_dw, const uint32_t *__restrict__ s,
                                               uint32_t s_row_dw, uint32_t bh, uint32_t out[4]) {
    constexpr uint32_t NF = BW / 4, FLUSH = 64 / NF;  // rows before a 16-bit lane of the packed accumulator could overflow
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (uint32_t r0 = 0; r0 < bh; r0 += FLUSH) {
        uint64_t       acc = 0;
        const uint32_t r1  = r0 + FLUSH < bh ? r0 + FLUSH : bh;
#pragma unroll 1
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t *wr = w + r * w_row_dw;
            const uint32_t *sr = s + r * s_row_dw;
            uint32_t        d[NF + 1];
#pragma unroll
            for (uint32_t i = 0; i <= NF; i++) d[i] = wr[i];
#pragma unroll
            for (uint32_t i = 0; i < NF; i += 4) {
                const uint4 sv = *(const uint4 *)&sr[i];
                acc = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i], d[i + 1]), sv.x, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 1], d[i + 2]), sv.y, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 2], d[i + 3]), sv.z, acc);
                acc = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 3], d[i + 4]), sv.w, acc);
            }
        }
        a0 += (uint32_t)(acc & 0xffff), a1 += (uint32_t)((acc >> 16) & 0xffff);
        a2 += (uint32_t)((acc >> 32) & 0xffff), a3 += (uint32_t)(acc >> 48);
    }
    out[0] = a0, out[1] = a1, out[2] = a2, out[3] = a3;
}

// Two horizontally adjacent quads (8 positions, window dword `w` onwards) in one walk over the block: the window dwords of a
// row are read once for both (NF + 2 instead of 2 NF + 2), the 64-bit operand (d[j], d[j+1]) serves quad A against source
// dword j and quad B against source dword j - 1, and the source row is read once.  out[0..3] = quad A, out[4..7] = quad B.
template <uint32_t BW>
__device__ __forceinline__ void oct_sad_fixed(const uint32_t *__restrict__ w, uint32_t w_row_dw, const uint32_t *__restrict__ s,
                                              uint32_t s_row_dw, uint32_t bh, uint32_t out[8]) {
    constexpr uint32_t NF = BW / 4, FLUSH = 64 / NF;
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t r0 = 0; r0 < bh; r0 += FLUSH) {
        uint64_t       acc_a = 0, acc_b = 0;
        const uint32_t r1    = r0 + FLUSH < bh ? r0 + FLUSH : bh;
#pragma unroll 1
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t *wr = w + r * w_row_dw;
            const uint32_t *sr = s + r * s_row_dw;
            uint32_t        d[NF + 2];
#pragma unroll
            for (uint32_t i = 0; i <= NF + 1; i++) d[i] = wr[i];
#pragma unroll
            for (uint32_t i = 0; i < NF; i += 4) {
                const uint4 sv = *(const uint4 *)&sr[i];
                acc_a = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i], d[i + 1]), sv.x, acc_a);
                acc_b = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 1], d[i + 2]), sv.x, acc_b);
                acc_a = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 1], d[i + 2]), sv.y, acc_a);
                acc_b = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 2], d[i + 3]), sv.y, acc_b);
                acc_a = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 2], d[i + 3]), sv.z, acc_a);
                acc_b = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 3], d[i + 4]), sv.z, acc_b);
                acc_a = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 3], d[i + 4]), sv.w, acc_a);
                acc_b = __builtin_amdgcn_qsad_pk_u16_u8(pair64(d[i + 4], d[i + 5]), sv.w, acc_b);
            }
        }
        a[0] += (uint32_t)(acc_a & 0xffff), a[1] += (uint32_t)((acc_a >> 16) & 0xffff);
        a[2] += (uint32_t)((acc_a >> 32) & 0xffff), a[3] += (uint32_t)(acc_a >> 48);
        a[4] += (uint32_t)(acc_b & 0xffff), a[5] += (uint32_t)((acc_b >> 16) & 0xffff);
        a[6] += (uint32_t)((acc_b >> 32) & 0xffff), a[7] += (uint32_t)(acc_b >> 48);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = a[i];
}

__device__ __forceinline__ void search_plan_desc(SearchDesc &d, uint32_t bw, uint32_t bh, uint32_t win_cap_dw) {
    d.fast = 0;
    d.k    = 0;
    if (d.sa_w <= 0 || d.sa_h <= 0) {
        d.n_srows = 0, d.nq = 0, d.no = 0;
        return;
    }
    d.n_srows = d.skip ? (uint32_t)d.sa_h / 2u : (uint32_t)d.sa_h;
    d.nq      = ((uint32_t)d.sa_w + 3u) >> 2;
    d.no      = (d.nq + 1u) >> 1;
    if (d.raw_stride != 0 && bw <= 256) {
        // the two strides the reference passes are either equal or 2:1 (sub-sampled SAD); anything else: slow path
        d.k = d.ref_stride == d.raw_stride ? 1u : (d.ref_stride == 2 * d.raw_stride ? 2u : (d.ref_stride % d.raw_stride == 0 ? d.ref_stride / d.raw_stride : 0u));
        if (d.k) {
            d.rows_per_pos = (bh - 1) * d.k + 1;
            d.pitch_dw     = (d.nq + ((bw + 3) >> 2) + 1) | 1u;  // odd pitch: rows land on different banks
            d.fast         = (uint64_t)d.pitch_dw * d.rows_per_pos <= win_cap_dw && win_cap_dw <= 65536u;
            d.inv_no       = make_inv_small(d.no);
        }
    }
}

// segment that contains flat index `v` of a monotonic per-segment base (lds_dw or item_base)
template <bool ITEMS, class SH> __device__ __forceinline__ uint32_t find_seg(const SH &sh, uint32_t nseg, uint32_t v) {
    uint32_t lo = 0, hi = nseg;  // invariant: base[lo] <= v < base[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t b   = ITEMS ? sh.seg[mid].item_base : sh.seg[mid].lds_dw;
        if (v >= b)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Exhaustive search of n descriptors (same bw x bh block, `src` = block rows as dwords in LDS).
// On return sh.best[i] holds the winning key of descriptor i (KEY_NONE if nothing was searched): SAD << 32 | search row << 16 | column
// (search areas are int16 in every caller; (row, column) orders like the raster index and decodes without a division).
// Must be called by all WG_THREADS threads of the workgroup; contains barriers.  Descriptor i must have been
// written by thread i (or be visible through an earlier barrier).
template <class SH>
__device__ __forceinline__ void wg_multi_search(SH &sh, uint32_t n, const uint32_t *__restrict__ src,
                                       uint32_t src_row_dw, uint32_t bw, uint32_t bh, uint32_t *__restrict__ win,
                                       uint32_t win_cap_dw) {
    const uint32_t tid = threadIdx.x;
    // b128 source reads need 16-byte aligned rows (LDS addresses are 32-bit offsets: test the low bits)
    const bool src_aligned = (((uint32_t)(uintptr_t)src) & 15u) == 0 && (src_row_dw & 3u) == 0;
#ifdef SVT_HIP_ME_PROFILE
    unsigned long long ms_last = wall_clock64();
#endif
    if (tid < n) {
        search_plan_desc(sh.desc[tid], bw, bh, win_cap_dw);
        sh.best[tid] = KEY_NONE;
    }
    if (tid == 0) {
        sh.next_d = 0, sh.next_j = 0;
    }
    __syncthreads();

    // ---- plan ALL passes at once (first wave, lane = descriptor) when no descriptor needs to be split over passes: one prefix scan
    // and a ballot per pass boundary instead of a scan per pass, and one barrier less per pass ----
    if (tid < 64) {
        uint32_t need = 0, items = 0, nstage = 0;
        bool     valid = false;
        if (tid < n) {
            const SearchDesc &ds = sh.desc[tid];
            if (ds.fast && ds.n_srows) {
                nstage = (ds.n_srows - 1) * (ds.skip ? 2u : 1u) + ds.rows_per_pos;
                need   = (nstage * ds.pitch_dw + SEG_ALIGN - 1) & ~(SEG_ALIGN - 1);
                items  = ds.n_srows * ds.no;
                valid  = true;
            }
        }
        const bool plan_once = n <= 64 && __ballot(valid && need > win_cap_dw) == 0;
        uint32_t   npass = 0;
        if (plan_once) {
            uint32_t pn = need, pi = items;  // inclusive prefix sums
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t a = __shfl_up(pn, off, 64), b2 = __shfl_up(pi, off, 64);
                if ((int)tid >= off)
                    pn += a, pi += b2;
            }
            const uint64_t vmask = __ballot(valid);
            const uint32_t total_items = __shfl(pi, 63, 64);              // (every lane takes part: the source lane must be active)
            const uint32_t en = pn - need, ei = pi - items;               // exclusive
            const uint32_t sidx = (uint32_t)__popcll(vmask & ((1ull << tid) - 1ull));  // compact segment index
            uint64_t       starts = 0;                                    // lanes that begin a pass
            uint32_t       base = 0;                                      // exclusive prefix of the current pass's first descriptor
            if (vmask) {
                uint32_t first = (uint32_t)__ffsll((unsigned long long)vmask) - 1u;
                for (;;) {
                    starts |= 1ull << first, npass++;
                    base = __shfl(en, (int)first, 64);
                    const uint64_t over = __ballot(valid && tid > first && pn - base > win_cap_dw);
                    if (!over || npass == (uint32_t)MAX_PASS)
                        break;
                    first = (uint32_t)__ffsll((unsigned long long)over) - 1u;
                }
                // a plan with more passes than the table holds: leave it to the per-pass planner
                if (npass == (uint32_t)MAX_PASS && __ballot(valid && tid > first && pn - base > win_cap_dw))
                    npass = 0;
            }
            if (npass && valid) {
                const uint64_t below = starts & ((2ull << tid) - 1ull);    // pass starts at or before this lane
                const uint32_t mypass = (uint32_t)__popcll(below) - 1u, ps = 63u - (uint32_t)__clzll((unsigned long long)below);
                const uint32_t en0 = __shfl(en, (int)ps, 64), ei0 = __shfl(ei, (int)ps, 64);
                SearchSeg     &sg = sh.seg[sidx];
                sg.d = tid, sg.j0 = 0, sg.nj = sh.desc[tid].n_srows, sg.lds_dw = en - en0, sg.nstage = nstage, sg.item_base = ei - ei0;
                if (tid == ps)
                    sh.pass_seg0[mypass] = (uint8_t)sidx, sh.pass_item0[mypass] = ei;
            }
            if (npass && tid == 0)
                sh.pass_seg0[npass] = (uint8_t)__popcll(vmask), sh.pass_item0[npass] = total_items;
        }
        if (tid == 0)
            sh.npass = npass;
    }
    __syncthreads();
    const uint32_t npass_once = sh.npass;

    // ---- fast path: passes of (stage -> search) until every descriptor row has been searched ----
    for (uint32_t pass = 0;; pass++) {
      uint32_t s0 = 0, nseg, nitems_pass;
      bool     onepass;
      if (npass_once) {
        s0 = sh.pass_seg0[pass], nseg = sh.pass_seg0[pass + 1] - s0, nitems_pass = sh.pass_item0[pass + 1] - sh.pass_item0[pass];
        onepass = pass + 1 == npass_once;
      } else {

        // plan, by the first wave (lane = descriptor): starting at the continuation point, take as many whole descriptors
        // as fit the window buffer (prefix sum of their needs); a descriptor too big to fit even alone gets a pass of
        // its own for as many of its search rows as fit.
        if (tid < 64) {
            const uint32_t d0 = sh.next_d, j0 = sh.next_j;
            uint32_t       need = 0, items = 0, nstage = 0, nj = 0, jstart = 0;
            bool           part = false;
            if (tid >= d0 && tid < n) {
                const SearchDesc &ds = sh.desc[tid];
                jstart               = tid == d0 ? j0 : 0;
                if (ds.fast && ds.n_srows > jstart) {
                    const uint32_t step = ds.skip ? 2u : 1u;
                    nj                  = ds.n_srows - jstart;
                    nstage              = (nj - 1) * step + ds.rows_per_pos;
                    if (tid == d0 && nstage * ds.pitch_dw > win_cap_dw) {
                        // rows staged for nj searched rows: (nj-1)*step + rows_per_pos ('fast' guarantees nj >= 1)
                        nj     = (win_cap_dw / ds.pitch_dw - ds.rows_per_pos) / step + 1;  // (rare: a window that does not fit alone)
                        nstage = (nj - 1) * step + ds.rows_per_pos;
                        part   = true;
                    }
                    need  = (nstage * ds.pitch_dw + SEG_ALIGN - 1) & ~(SEG_ALIGN - 1);
                    items = nj * ds.no;
                }
            }
            const bool split = __shfl((int)part, (int)(d0 & 63u), 64) != 0;
            uint32_t   need_in = need, items_in = items, cnt_in = need ? 1u : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t a = __shfl_up(need_in, off, 64), b2 = __shfl_up(items_in, off, 64), c = __shfl_up(cnt_in, off, 64);
                if ((int)tid >= off)
                    need_in += a, items_in += b2, cnt_in += c;
            }
            const bool take = tid >= d0 && tid < n && need_in <= win_cap_dw && (!split || tid == d0);
            if (take && need) {
                SearchSeg &sg = sh.seg[cnt_in - 1];
                sg.d = tid, sg.j0 = jstart, sg.nj = nj, sg.lds_dw = need_in - need, sg.nstage = nstage;
                sg.item_base = items_in - items;
            }
            const uint32_t ntake = (uint32_t)__popcll(__ballot(take));
            // totals of the taken range live in its last lane
            const uint32_t last = d0 + (ntake ? ntake - 1 : 0);
            const uint32_t t_need = __shfl(need_in, (int)(last & 63u), 64), t_items = __shfl(items_in, (int)(last & 63u), 64),
                           t_cnt = __shfl(cnt_in, (int)(last & 63u), 64);
            const uint32_t nj_d0 = __shfl(nj, (int)(d0 & 63u), 64);
            if (tid == 0) {
                sh.nseg = ntake ? t_cnt : 0, sh.nitems = ntake ? t_items : 0, sh.nstage_dw = ntake ? t_need : 0;
                if (split)
                    sh.next_d = d0, sh.next_j = j0 + nj_d0;
                else
                    sh.next_d = d0 + ntake, sh.next_j = 0;
            }
        }
        __syncthreads();
        onepass = sh.next_d >= n;  // nothing left after this pass
        MS_PHASE(0);
        nseg = sh.nseg, nitems_pass = sh.nitems;
      }
        if (nseg == 0)
            break;
        // stage: 16-byte loads, 2..32 lanes per window row (stage_rows16).  With at least as many segments as waves every wave
        // takes whole segments, otherwise all waves split each segment's rows.
        {
            const uint32_t nwv = blockDim.x >> 6;
            const bool     per_wave = nseg >= nwv;
            const uint32_t lanes = per_wave ? 64u : blockDim.x, lid = per_wave ? (tid & 63u) : tid;
            for (uint32_t s = per_wave ? (tid >> 6) : 0u; s < nseg; s += per_wave ? nwv : 1u) {
                const SearchSeg   sg = sh.seg[s0 + s];
                const SearchDesc &ds = sh.desc[sg.d];
                const uint32_t    rstride = ds.raw_stride;
#ifdef SVT_HIP_ME_ABLATE
                if (g_ms_skip & 1)
                    continue;
#endif
                stage_rows16(win + sg.lds_dw, ds.pitch_dw, ds.ref + (size_t)sad_mul_u24(ds.skip ? 2 * sg.j0 + 1 : sg.j0, rstride), rstride,
                             sg.nstage, lanes, lid);
            }
        }
        __syncthreads();
        MS_PHASE(1);
        // search: one work item = eight horizontally adjacent positions (two quads) of one searched row
#ifdef SVT_HIP_ME_ABLATE
        const uint32_t nitems = (g_ms_skip & 4) ? 0u : nitems_pass;  // bit 2: no work items at all
#else
        const uint32_t nitems = nitems_pass;
#endif
        uint32_t s = 0;  // items grow from one iteration to the next, and so does the segment that holds them: advance, never search
        for (uint32_t item = tid; item < nitems; item += blockDim.x) {
            while (s + 1 < nseg && item >= sh.seg[s0 + s + 1].item_base) s++;
            const SearchSeg  sg = sh.seg[s0 + s];
            const SearchDesc &ds = sh.desc[sg.d];
            const uint32_t   li = item - sg.item_base;
            const uint32_t   jl = fast_div(li, ds.inv_no), q = 2u * (li - sad_mul_u24(jl, ds.no));  // q: first quad of the pair
            const uint32_t   step = ds.skip ? 2u : 1u;
            const uint32_t   wpitch = sad_mul_u24(ds.k, ds.pitch_dw);
            const uint32_t  *w    = win + sg.lds_dw + sad_mul_u24(jl * step, ds.pitch_dw) + q;
            uint32_t         sad[8];
            bool             plain = false;  // every SAD a real one (the fixed-width walks never leave a slot empty)
            // a half pair (odd quad count) still walks both quads: the second reads at most one dword past its row (staged
            // data or the next LDS object, never used: its positions fail the sa_w test below)
#ifdef SVT_HIP_ME_ABLATE
            if (g_ms_skip & 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) sad[i] = item + i;
            } else
#endif
#ifndef SVT_HIP_NO_QSAD
            if (src_aligned && bw == 16)
                oct_sad_fixed<16>(w, wpitch, src, src_row_dw, bh, sad), plain = true;
            else if (src_aligned && bw == 32)
                oct_sad_fixed<32>(w, wpitch, src, src_row_dw, bh, sad), plain = true;
            else if (src_aligned && bw == 64)
                oct_sad_fixed<64>(w, wpitch, src, src_row_dw, bh, sad), plain = true;
            else
#endif
            {
                quad_sad(w, wpitch, src, src_row_dw, bw, bh, sad);
                if (q + 1 < ds.nq)
                    quad_sad(w + 1, wpitch, src, src_row_dw, bw, bh, sad + 4);
                else
                    sad[4] = sad[5] = sad[6] = sad[7] = ~0u;
            }
            const uint32_t sy = ds.skip ? 2 * (sg.j0 + jl) + 1 : (sg.j0 + jl);
            const uint32_t saw = (uint32_t)ds.sa_w;
            // the eight positions are in raster order: a strict '<' on the SAD keeps the first minimum, and the
            // 64-bit (sad, raster index) key is built once
            // (a SAD is below 2^22: (sad << 3 | position) orders exactly like the pair, one v_min per position instead of a
            // compare and two selects)
            const uint32_t nvalid = saw - 4 * q;  // >= 1
            uint32_t       kmin   = ~0u;
            if (plain && __all(nvalid >= 8)) {  // the usual item: all eight positions inside the search area (uniform over the wave's active lanes)
#pragma unroll
                for (uint32_t p = 0; p < 8; p += 2) kmin = min(min((sad[p] << 3) | p, (sad[p + 1] << 3) | (p + 1)), kmin);  // v_min3
            } else {
#pragma unroll
                for (uint32_t p = 0; p < 8; p++) {
                    const uint32_t k = (p < nvalid && sad[p] != ~0u) ? ((sad[p] << 3) | p) : ~0u;
                    kmin             = k < kmin ? k : kmin;
                }
            }
            if (kmin != ~0u)
                atomicMin((unsigned long long *)&sh.best[sg.d], ((unsigned long long)(kmin >> 3) << 32) | ((sy << 16) | (4 * q + (kmin & 7u))));
        }
        __syncthreads();
        MS_PHASE(2);
        if (onepass)
            break;
    }

    // ---- slow path (window does not fit LDS, or strides are unrelated): straight from global memory ----
    for (uint32_t d = 0; d < n; d++) {
        const SearchDesc ds = sh.desc[d];
        if (ds.fast || ds.n_srows == 0)
            continue;
        const uint32_t npos = ds.n_srows * (uint32_t)ds.sa_w;
        const uint8_t *sb   = (const uint8_t *)src;
        uint64_t       key  = KEY_NONE;
        for (uint32_t p = tid; p < npos; p += blockDim.x) {
            const uint32_t j = p / (uint32_t)ds.sa_w, sx = p - j * (uint32_t)ds.sa_w;
            const uint32_t sy = ds.skip ? 2 * j + 1 : j;
            const uint8_t *g  = ds.ref + (size_t)sy * ds.raw_stride + sx;
            uint32_t       sad = 0;
            for (uint32_t r = 0; r < bh; r++)
                for (uint32_t c = 0; c < bw; c++) {
                    const int dv = (int)sb[r * src_row_dw * 4 + c] - (int)g[(size_t)r * ds.ref_stride + c];
                    sad += (uint32_t)(dv < 0 ? -dv : dv);
                }
            const uint64_t kk = ((uint64_t)sad << 32) | ((sy << 16) | sx);
            key               = kk < key ? kk : key;
        }
        if (key < KEY_NONE)
            atomicMin((unsigned long long *)&sh.best[d], (unsigned long long)key);
    }
    __syncthreads();
}

// Stage a bw x bh block (bytes) from global into LDS as rows of `row_dw` dwords (zero padded tail).
__device__ __forceinline__ void wg_stage_block(uint32_t *__restrict__ dst, uint32_t row_dw, const uint8_t *__restrict__ g,
                                               uint32_t stride, uint32_t bw, uint32_t bh) {
    const uint32_t total = row_dw * bh;
    for (uint32_t idx = threadIdx.x; idx < total; idx += WG_THREADS) {
        const uint32_t r = idx / row_dw, i = idx - r * row_dw;
        const uint8_t *p = g + (size_t)r * stride + 4 * i;
        uint32_t       v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++)
            if (4 * i + b < bw)
                v |= (uint32_t)p[b] << (8 * b);
        dst[idx] = v;
    }
}

// Wave64 sum reduction (all lanes get nothing in particular; lane 0 holds the total).
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Wave64 sum, every lane gets the total (butterfly).
__device__ __forceinline__ uint32_t wave_sum_all(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace dev
}  // namespace svthip
