// loopfilter_wiener.hip — Wiener restoration on gfx950.
// Replaces svt_av1_compute_stats_c / _highbd_c (restoration_pick.c:671-757) and svt_av1_wiener_convolve_add_src_c /
// svt_av1_highbd_wiener_convolve_add_src_c (convolve.c:57-200).
//
// Statistics.  The reference accumulates, per sample of the unit, y[k]*x and y[k]*y[l] for the win^2 taps
// y[k] = dgd(sample + tap k) - avg.  That is a Gram matrix: with Z = one row per sample and one column per tap (plus the source
// sample and a constant 1), everything the function returns is in Z^T Z.  Here Z^T Z is computed on the matrix cores with EXACT
// integer arithmetic: a sample minus half the range is split into two int8 digits c = 32 h + l (h = c >> 5 signed, l = c & 31), the
// matrix gets a column for every digit (2 x (win^2 + 1) + the ones column = 101 rows for win 7, 53 for win 5, padded to 128 / 64),
// and v_mfma_i32_32x32x32_i8 accumulates the products of 32 samples per instruction in int32.  Because both operands of a tile
// product are fragments of the same matrix, the fragment of a tile row serves as A and as B; only the upper triangle of tiles is
// computed.  The digits recombine as 1024 hh' + 32 (hl' + lh') + ll', the RAW moments follow from the ones column, and the mean is
// folded in at the end (exact integer algebra: sum (a - m)(b - m) = sum ab - m sum a - m sum b + N m^2), so no pre-pass over the
// unit is needed for `avg`.  One workgroup owns 64 x 256 samples of the unit (8 chunks of 32 rows staged in LDS as digit planes
// with their borders); a wave takes every fourth sample row and walks it in steps of 32 samples: per step a lane builds the 16
// bytes of each of its tile rows from aligned dword reads and v_alignbyte (the tap offset decides the byte phase, which is constant
// per lane), then 10 (win 7) or 3 (win 5) MFMAs.  The accumulators are combined in LDS (int64) once per workgroup.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_lf.h"
#include "common.hpp"
#include "lr_device.hpp"

using namespace svthip;

namespace {

constexpr int TW = 64, TH = 32;  // samples per tile
constexpr int W2MAX = 49;  // WIENER_WIN2

using svthip::lr::ldpx;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct StatsAux {  // raw first moments of one unit
    long long S[W2MAX];  // sum of dgd at tap k
    long long sum_src, n;
};


// eight horizontally adjacent samples from element `e` of a plane of bytes or 16-bit words (any alignment); `left` = samples that
// exist from there on (fewer than 8 at the end of a row: the others are never used and are not touched)
struct Px8 {
    uint16_t v[8];
};
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u128_any;
typedef uint32_t __attribute__((ext_vector_type(2), aligned(1))) u64_any;
__device__ __forceinline__ Px8 load_px8(const void *base, size_t e, int is16, int left) {
    Px8 r;
    if (left >= 8) {
        if (is16) {
            const u128_any q = *(const u128_any *)((const uint16_t *)base + e);
#pragma unroll
            for (int k = 0; k < 8; k++) r.v[k] = (uint16_t)(q[k >> 1] >> (16 * (k & 1)));
        } else {
            const u64_any q = *(const u64_any *)((const uint8_t *)base + e);
#pragma unroll
            for (int k = 0; k < 8; k++) r.v[k] = (uint16_t)((q[k >> 2] >> (8 * (k & 3))) & 0xff);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) r.v[k] = k < left ? (uint16_t)ldpx(base, e + k, is16) : (uint16_t)0;
    }
    return r;
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int WIN>
__global__ __launch_bounds__(256, 2) void wiener_stats_kernel(const SvtHipWienerUnit *__restrict__ units, int wgs_per_unit, int per, int gx, int gy, int is16, int bd,
                                                           long long *__restrict__ M, long long *__restrict__ H, StatsAux *__restrict__ aux) {
    constexpr int HALF = WIN / 2, W2 = WIN * WIN;
    constexpr int NCOL = W2 + 1;                    // columns with two digits: the taps and the source sample
    constexpr int ONES = 2 * NCOL, NROW = ONES + 1;  // rows of Z^T: high digits, low digits, the constant 1
    constexpr int NT = (NROW + 31) / 32, NTP = NT * (NT + 1) / 2;
    constexpr int P = 80;                           // bytes per row of a digit plane (64 samples + 2 * 3 border + the over-read of an aligned fragment)
    constexpr int DROWS = TH + 2 * HALF;
    constexpr int OFF_DH = 0, OFF_DL = OFF_DH + DROWS * P, OFF_SH = OFF_DL + DROWS * P, OFF_SL = OFF_SH + TH * P, OFF_ONE = OFF_SL + TH * P,
                  OFF_ZERO = OFF_ONE + 128, PL_BYTES = OFF_ZERO + 128;
    constexpr int HD = NCOL + 1;                    // Hx[a][b], a <= b: taps, source, ones
    __shared__ alignas(16) uint8_t pl[2 * PL_BYTES];  // two tiles: the next item is loaded and converted while this one is multiplied
    __shared__ long long Hx[HD * HD];
    // Work items of a unit = (chunk of 32 rows, 64-wide tile) of ITS grid, tile fastest.  The launch has about as many workgroups as the
    // GPU holds at once: `wgs_per_unit` per unit, each with a contiguous range of the unit's items, so that a workgroup combines its
    // accumulators and sends them to the unit's sums exactly once.  (`per`: the host's bound on the items of one workgroup.)
    const int              unit = __builtin_amdgcn_readfirstlane((int)blockIdx.x / wgs_per_unit), sub = (int)blockIdx.x - unit * wgs_per_unit;
    const SvtHipWienerUnit u    = units[unit];
    const int uw = u.h_end - u.h_start, uh = u.v_end - u.v_start;
    const int gxu = (uw + TW - 1) / TW, items = gxu * ((uh + TH - 1) / TH);
    const int peru = __builtin_amdgcn_readfirstlane(min(per, (items + wgs_per_unit - 1) / wgs_per_unit));
    const int i0 = sub * peru, i1 = min(items, i0 + peru);
    if (i0 >= i1)
        return;
    (void)gx, (void)gy;
    for (int i = threadIdx.x; i < HD * HD; i += 256) Hx[i] = 0;
    if (threadIdx.x < 128) {
        pl[OFF_ONE + threadIdx.x] = 1, pl[OFF_ZERO + threadIdx.x] = 0;
        pl[PL_BYTES + OFF_ONE + threadIdx.x] = 1, pl[PL_BYTES + OFF_ZERO + threadIdx.x] = 0;
    }
    const int mid = 1 << (bd - 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, kh = lane >> 5, r_lane = r, kh_lane = kh;
    // where this lane's row of each tile lives: byte offset of (sample row 0, sample 16 * kh) and the row pitch (0: a constant row)
    uint32_t addr[NT], pit[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int g = 32 * t + r;
        if (g < 2 * NCOL) {
            const int  col = g < NCOL ? g : g - NCOL;
            const bool low = g >= NCOL;
            if (col < W2)  // tap index = column * WIN + row (restoration_pick.c:686-691)
                addr[t] = (low ? OFF_DL : OFF_DH) + (col % WIN) * P + col / WIN;
            else
                addr[t] = low ? OFF_SL : OFF_SH;
            pit[t] = P;
        } else {
            addr[t] = g == ONES ? OFF_ONE : OFF_ZERO, pit[t] = 0;
        }
        addr[t] += 16 * kh;
    }
    auto flush = [&](int unit, const v16i(&acc)[NTP]) {
        __syncthreads();  // Hx is zero (start of the kernel) before the first sum arrives
        // (opaque copies: otherwise the 160 target addresses and weights below, which depend on the lane alone, are computed in front of
        // the item loop and held in registers through the whole kernel — next to 160 accumulators that is 220 spills)
        int r = r_lane, kh = kh_lane;
        asm volatile("" : "+v"(r), "+v"(kh));
        // digits -> columns: entry (a, b), a <= b, of Z^T Z goes to the column pair of its rows with the weight of its digits; a pair of
        // different digits of ONE column appears once in the upper triangle but twice in the product
        {
            int p = 0;
#pragma unroll
            for (int ta = 0; ta < NT; ta++)
#pragma unroll
                for (int tb = ta; tb < NT; tb++, p++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int a = 32 * ta + (i & 3) + 8 * (i >> 2) + 4 * kh, b = 32 * tb + r;
                        if (a <= b && b < NROW && acc[p][i]) {
                            const int ca = a < NCOL ? a : (a < ONES ? a - NCOL : NCOL), wa = a < NCOL ? 32 : 1;
                            const int cb = b < NCOL ? b : (b < ONES ? b - NCOL : NCOL), wb = b < NCOL ? 32 : 1;
                            const int w  = wa * wb * ((a < b && ca == cb) ? 2 : 1);
                            atomicAdd((unsigned long long *)&Hx[min(ca, cb) * HD + max(ca, cb)], (unsigned long long)((long long)acc[p][i] * w));
                        }
                    }
        }
        __syncthreads();
        // centred -> raw moments (sum (c + m)(c' + m) = sum cc' + m sum c + m sum c' + N m^2), then this workgroup's share goes to the unit
        long long *Hu = H + (size_t)unit * W2MAX * W2MAX, *Mu = M + (size_t)unit * W2MAX;
        StatsAux  &A  = aux[unit];
        const long long n = Hx[NCOL * HD + NCOL], m = mid;
        for (int e = threadIdx.x; e < W2 * W2; e += 256) {
            const int k = e / W2, l = e - k * W2;
            if (k <= l) {
                const long long v = Hx[k * HD + l] + m * (Hx[k * HD + NCOL] + Hx[l * HD + NCOL]) + n * m * m;
                if (v)
                    atomicAdd((unsigned long long *)&Hu[e], (unsigned long long)v);
            }
        }
        if (threadIdx.x < W2) {
            const int       k  = threadIdx.x;
            const long long sk = Hx[k * HD + NCOL], ss = Hx[W2 * HD + NCOL];
            atomicAdd((unsigned long long *)&Mu[k], (unsigned long long)(Hx[k * HD + W2] + m * (sk + ss) + n * m * m));
            atomicAdd((unsigned long long *)&A.S[k], (unsigned long long)(sk + n * m));
        }
        if (threadIdx.x == 0) {
            atomicAdd((unsigned long long *)&A.sum_src, (unsigned long long)(Hx[W2 * HD + NCOL] + n * m));
            atomicAdd((unsigned long long *)&A.n, (unsigned long long)n);
        }
    };
    v16i acc[NTP];
#pragma unroll
    for (int p = 0; p < NTP; p++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[p][i] = 0;
    // eight samples per load item, every load of a tile issued before the first digit is stored: one memory round trip per tile, and
    // that one under the multiplications of the tile before it
    constexpr int NI = 3;  // (32 + 6) rows x 9 groups of the degraded tile <= 2 x 256 load items, 32 x 8 of the source <= 256
    struct Geo {
        int x0, y0, tw, tv;
    };
    auto geo = [&](int item) {
        // (a division leaves its result in a vector register: back to a scalar one, or everything derived from it is held per lane)
        const int cy = __builtin_amdgcn_readfirstlane(item / gxu);
        Geo       g;
        g.x0 = (item - cy * gxu) * TW, g.y0 = cy * TH, g.tw = min(TW, uw - g.x0), g.tv = min(TH, uh - g.y0);
        return g;
    };
    auto issue = [&](const Geo &g, Px8(&px)[NI], int(&dst)[NI]) {
        const int gd = (g.tw + 2 * HALF + 7) >> 3, nd = (g.tv + 2 * HALF) * gd, gs = (g.tw + 7) >> 3, ns = g.tv * gs;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const bool deg = i < NI - 1;
            const int  li  = deg ? (int)threadIdx.x + 256 * i : (int)threadIdx.x;
            const int  gpr = deg ? gd : gs, n = deg ? nd : ns, cols = deg ? g.tw + 2 * HALF : g.tw;
            dst[i] = -1;
            if (li < n) {
                const int rr = li / gpr, c = 8 * (li - rr * gpr);
                dst[i]       = (deg ? OFF_DH : OFF_SH) + rr * P + c;
                const size_t e = deg ? (size_t)((ptrdiff_t)(u.v_start + g.y0 + rr - HALF) * u.dgd_stride + (u.h_start + g.x0 + c - HALF))
                                     : (size_t)((ptrdiff_t)(u.v_start + g.y0 + rr) * u.src_stride + (u.h_start + g.x0 + c));
                px[i] = load_px8(deg ? u.dgd : u.src, e, is16, cols - c);
            }
        }
    };
    auto commit = [&](const Px8(&px)[NI], const int(&dst)[NI], int buf) {
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (dst[i] >= 0) {
                uint32_t hi[2] = {0, 0}, lo[2] = {0, 0};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int v = (int)px[i].v[k] - mid;
                    hi[k >> 2] |= (uint32_t)((v >> 5) & 0xff) << (8 * (k & 3)), lo[k >> 2] |= (uint32_t)(v & 31) << (8 * (k & 3));
                }
                *(uint2 *)&pl[buf + dst[i]] = make_uint2(hi[0], hi[1]);
                *(uint2 *)&pl[buf + dst[i] + (i < NI - 1 ? OFF_DL - OFF_DH : OFF_SL - OFF_SH)] = make_uint2(lo[0], lo[1]);
            }
    };
    Px8 px[NI];
    int dst[NI];
    Geo gn = geo(i0);
    issue(gn, px, dst);
    commit(px, dst, 0);
    int buf = 0;
    for (int item = i0; item < i1; item++) {
        const Geo g  = gn;
        const int tw = g.tw, tv = g.tv;
        __syncthreads();  // this tile is complete, the other one (the item before) fully consumed
        const bool more = item + 1 < i1;
        if (more) {
            gn = geo(item + 1);
            issue(gn, px, dst);
        }
        // steps of 32 samples: this wave's rows y = wv, wv + 4, ..., two steps per row when the tile is wider than 32.
        const int nxs = tw > 32 ? 2 : 1, nsteps = ((tv - wv + 3) >> 2) * nxs;  // tv > wv or no step
        auto build = [&](int st, v4i(&f)[NT]) {
            const int y = wv + 4 * (nxs == 2 ? st >> 1 : st), xs = nxs == 2 ? 32 * (st & 1) : 0;
            // samples of the unit among this lane's 16 (the bytes behind them belong to the border or to an earlier chunk)
            const int  nv   = min(max(tw - xs - 16 * kh, 0), 16);
            const bool edge = xs + 32 > tw;  // uniform
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const uint32_t  a  = (uint32_t)buf + addr[t] + (uint32_t)y * pit[t] + (uint32_t)xs;
                const uint32_t *q  = (const uint32_t *)(pl + (a & ~3u));
                const uint32_t  sh = a & 3u;
                const uint32_t  d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                f[t][0] = (int)__builtin_amdgcn_alignbyte(d1, d0, sh), f[t][1] = (int)__builtin_amdgcn_alignbyte(d2, d1, sh);
                f[t][2] = (int)__builtin_amdgcn_alignbyte(d3, d2, sh), f[t][3] = (int)__builtin_amdgcn_alignbyte(d4, d3, sh);
                if (edge) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int vb = min(max(nv - 4 * i, 0), 4);
                        f[t][i] &= vb == 4 ? -1 : (int)((1u << (8 * vb)) - 1u);
                    }
                }
            }
        };
        if (wv < tv) {
            for (int st = 0; st < nsteps; st++) {
                v4i f[NT];
                build(st, f);
                int p = 0;
#pragma unroll
                for (int ta = 0; ta < NT; ta++)
#pragma unroll
                    for (int tb = ta; tb < NT; tb++, p++) acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[ta], f[tb], acc[p], 0, 0, 0);
            }
        }
        buf ^= PL_BYTES;
        if (more)
            commit(px, dst, buf);
    }
    flush(unit, acc);
}

__global__ __launch_bounds__(256) void wiener_zero_kernel(long long *__restrict__ M, long long *__restrict__ H, long long *__restrict__ A, size_t nM,
                                                          size_t nH, size_t nA) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // the grid covers nH, the largest of the three
    if (i < nH)
        H[i] = 0;
    if (i < nM)
        M[i] = 0;
    if (i < nA)
        A[i] = 0;
}

// raw moments -> the reference's M / H: fold the mean in, apply the high-bit-depth divider, mirror the lower triangle
__global__ __launch_bounds__(256) void wiener_finalize_kernel(int win, int divider, long long *__restrict__ M, long long *__restrict__ H,
                                                              const StatsAux *__restrict__ aux) {
    const int       w2 = win * win;
    long long      *Hu = H + (size_t)blockIdx.x * W2MAX * W2MAX, *Mu = M + (size_t)blockIdx.x * W2MAX;
    const StatsAux &A  = aux[blockIdx.x];
    const long long n = A.n, avg = n ? A.S[w2 / 2] / n : 0;  // the centre tap's sum is the sum of dgd over the unit (find_average)
    __shared__ long long Mraw[W2MAX];
    if ((int)threadIdx.x < w2)
        Mraw[threadIdx.x] = Mu[threadIdx.x];
    __syncthreads();
    if ((int)threadIdx.x < w2) {
        const int k = threadIdx.x;
        Mu[k] = (Mraw[k] - avg * A.sum_src - avg * A.S[k] + n * avg * avg) / divider;
    }
    for (int e = threadIdx.x; e < w2 * w2; e += 256) {
        const int k = e / w2, l = e - k * w2;
        if (k <= l)
            Hu[k * w2 + l] = (Hu[k * w2 + l] - avg * A.S[k] - avg * A.S[l] + n * avg * avg) / divider;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < w2 * w2; e += 256) {
        const int k = e / w2, l = e - k * w2;
        if (k > l)
            Hu[k * w2 + l] = Hu[l * w2 + k];
    }
}

// ---- separable 7-tap filter: one workgroup per 64 x 64 tile, intermediate rows in LDS ----
struct Taps {
    int16_t x[8], y[8];
};
__global__ __launch_bounds__(256) void wiener_convolve_kernel(const void *__restrict__ src, uint32_t src_stride, void *__restrict__ dst,
                                                              uint32_t dst_stride, int w, int h, Taps f, int is16, int bd, int r0, int r1) {
    __shared__ alignas(4) uint16_t in[(64 + 7) * lr::WIENER_IP + lr::WIENER_IN_SLACK];
    __shared__ uint16_t tmp[(64 + 7) * 64];
    constexpr int IP = lr::WIENER_IP;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const int tw = min(64, w - x0), th = min(64, h - y0);
    for (int idx = threadIdx.x; idx < (th + 7) * (tw + 7); idx += 256) {
        const int r = idx / (tw + 7), c = idx - r * (tw + 7);
        in[r * IP + c] = (uint16_t)ldpx(src, (size_t)((ptrdiff_t)(y0 + r - 3) * src_stride + (x0 + c - 3)), is16);
    }
    __syncthreads();
    lr::wiener_tile_filter<256>(in, tmp, threadIdx.x, tw, th, x0, y0, f.x, f.y, bd, r0, r1, is16, dst, dst_stride);
}

[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct AuxBuf {  // per-thread grow-only device buffer for the raw first moments
    StatsAux *dev = nullptr;
    size_t    cap = 0;
};

}  // namespace

// ------------------------------------------------------------------------------------------------ Tier B
extern "C" int32_t svt_hip_wiener_stats(const SvtHipWienerUnit *units, uint32_t n_units, int32_t wiener_win, int32_t is_16bit, int32_t bit_depth,
                                        int64_t *d_M, int64_t *d_H, void *stream) {
    if (!units || n_units == 0 || n_units > 65535 || (wiener_win != 7 && wiener_win != 5) || !d_M || !d_H ||
        (bit_depth != 8 && bit_depth != 10 && bit_depth != 12) || (bit_depth > 8 && !is_16bit)) {
        set_error("svt_hip_wiener_stats: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    int max_w = 0, max_h = 0;
    for (uint32_t i = 0; i < n_units; i++) {
        const SvtHipWienerUnit &u = units[i];
        if (!u.dgd || !u.src || u.h_end <= u.h_start || u.v_end <= u.v_start || u.h_end - u.h_start > 4096 || u.v_end - u.v_start > 4096) {
            set_error("svt_hip_wiener_stats: unit %u: bad limits", i);
            return SVT_HIP_ERR_BAD_PARAMETER;
        }
        max_w = u.h_end - u.h_start > max_w ? u.h_end - u.h_start : max_w;
        max_h = u.v_end - u.v_start > max_h ? u.v_end - u.v_start : max_h;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t                st = resolve_stream(stream);
    static thread_local AuxBuf ab;
    if (n_units > ab.cap) {
        if (ab.dev) {
            SVT_HIP_CHECK(hipDeviceSynchronize());
            SVT_HIP_CHECK(hipFree(ab.dev));
        }
        ab.dev = nullptr, ab.cap = 0;
        SVT_HIP_CHECK(hipMalloc((void **)&ab.dev, sizeof(StatsAux) * n_units * 2));
        ab.cap = n_units * 2;
    }
    const SvtHipWienerUnit *d_units = (const SvtHipWienerUnit *)stage_descriptors(units, sizeof(SvtHipWienerUnit) * n_units, st);
    if (!d_units)
        return SVT_HIP_ERR_RUNTIME;
    // the three sum arrays start from zero: one launch instead of three memsets (a launch costs more than it moves here)
    {
        const size_t nM = (size_t)W2MAX * n_units, nH = (size_t)W2MAX * W2MAX * n_units, nA = sizeof(StatsAux) / sizeof(long long) * n_units;
        hipLaunchKernelGGL(wiener_zero_kernel, dim3((unsigned)((nH + 255) / 256)), dim3(256), 0, st, (long long *)d_M, (long long *)d_H, (long long *)ab.dev, nM, nH, nA);
    }
    // about one workgroup per slot of the GPU (two per CU at this kernel's register count): every unit gets the same number of
    // workgroups, each with a contiguous range of the unit's (chunk, tile) items.  int32 accumulators: products of two digits are below
    // or equal to 2^12 and a wave sees a quarter of its workgroup's samples: at most 512 items (2^20 samples, 2^18 per wave) per workgroup.
    static int slots = 0;
    if (!slots) {
        int cus = 0, dev = 0;
        SVT_HIP_CHECK(hipGetDevice(&dev));
        SVT_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        slots = 2 * (cus > 0 ? cus : 256);
    }
    const int gx = (max_w + TW - 1) / TW, gy = (max_h + TH - 1) / TH, items = gx * gy;
    int       k  = slots / (int)n_units > 1 ? slots / (int)n_units : 1;
    k            = k > items ? items : k;
    k            = (items + k - 1) / k > 512 ? (items + 511) / 512 : k;
    const int  per = (items + k - 1) / k;
    const dim3 grid((unsigned)(n_units * k));
    if (wiener_win == 7)
        hipLaunchKernelGGL(wiener_stats_kernel<7>, grid, dim3(256), 0, st, d_units, k, per, gx, gy, is_16bit, bit_depth, (long long *)d_M, (long long *)d_H, ab.dev);
    else
        hipLaunchKernelGGL(wiener_stats_kernel<5>, grid, dim3(256), 0, st, d_units, k, per, gx, gy, is_16bit, bit_depth, (long long *)d_M, (long long *)d_H, ab.dev);
    const int divider = is_16bit ? (bit_depth == 12 ? 16 : (bit_depth == 10 ? 4 : 1)) : 1;  // restoration_pick.c:719-723
    hipLaunchKernelGGL(wiener_finalize_kernel, dim3(n_units), dim3(256), 0, st, wiener_win, divider, (long long *)d_M, (long long *)d_H,
                       (const StatsAux *)ab.dev);
    stage_commit(st);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_wiener_convolve(const void *d_src, uint32_t src_stride, void *d_dst, uint32_t dst_stride, uint32_t w, uint32_t h,
                                           const int16_t filter_x[8], const int16_t filter_y[8], int32_t is_16bit, int32_t bit_depth,
                                           void *stream) {
    if (!d_src || !d_dst || d_src == d_dst || !w || !h || w > 16384 || h > 16384 || !filter_x || !filter_y ||
        (bit_depth != 8 && bit_depth != 10 && bit_depth != 12) || (bit_depth > 8 && !is_16bit) || filter_x[7] != 0 || filter_y[7] != 0) {
        set_error("svt_hip_wiener_convolve: bad argument (7-tap kernels with a zero 8th coefficient, output must not alias input)");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    Taps f;
    memcpy(f.x, filter_x, 16), memcpy(f.y, filter_y, 16);
    int r0 = 3, r1 = 11;  // get_conv_params_wiener (convolve.h:70-86)
    const int range = bit_depth + 7 - r0 + 2;
    if (range > 16)
        r0 += range - 16, r1 -= range - 16;
    hipLaunchKernelGGL(wiener_convolve_kernel, dim3((w + 63) / 64, (h + 63) / 64), dim3(256), 0, resolve_stream(stream), d_src, src_stride, d_dst,
                       dst_stride, (int)w, (int)h, f, (int)is_16bit, (int)bit_depth, r0, r1);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------ Tier A
static void stats_tier_a(int32_t win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end,
                         int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H, int is16, int bit_depth) {
    if (!ensure_init())
        fatal("compute_stats");
    const int    half = win >> 1, w = h_end - h_start, h = v_end - v_start, px = is16 ? 2 : 1;
    const size_t dp = (size_t)w + 2 * half, dbytes = up256(dp * (h + 2 * half) * px), sbytes = up256((size_t)w * h * px);
    const size_t obytes = sizeof(int64_t) * (W2MAX + W2MAX * W2MAX);
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    uint8_t     *d = sc.device(dbytes + sbytes + obytes + 256), *hh = sc.host(dbytes + sbytes + obytes + 256);
    const uint8_t *db = is16 ? (const uint8_t *)((uintptr_t)dgd8 << 1) : dgd8, *sb = is16 ? (const uint8_t *)((uintptr_t)src8 << 1) : src8;
    for (int r = 0; r < h + 2 * half; r++)
        memcpy(hh + (size_t)r * dp * px, db + ((ptrdiff_t)(v_start + r - half) * dgd_stride + (h_start - half)) * px, dp * px);
    for (int r = 0; r < h; r++) memcpy(hh + dbytes + (size_t)r * w * px, sb + ((ptrdiff_t)(v_start + r) * src_stride + h_start) * px, (size_t)w * px);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, hh, dbytes + sbytes, hipMemcpyHostToDevice, st));
    SvtHipWienerUnit u{d + ((size_t)half * dp + half) * px, d + dbytes, (uint32_t)dp, (uint32_t)w, 0, w, 0, h};
    int64_t         *dM = (int64_t *)(d + dbytes + sbytes), *dH = dM + W2MAX;
    if (svt_hip_wiener_stats(&u, 1, win, is16, bit_depth, dM, dH, st) != SVT_HIP_OK)
        fatal("compute_stats");
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + dbytes + sbytes, dM, obytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const int w2 = win * win;
    memcpy(M, hh + dbytes + sbytes, sizeof(int64_t) * w2);
    memcpy(H, hh + dbytes + sbytes + sizeof(int64_t) * W2MAX, sizeof(int64_t) * w2 * w2);
}
static void svt_av1_compute_stats_hip_impl(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H);
extern "C" void svt_av1_compute_stats_hip(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H) { TIER_A_CALL(svt_av1_compute_stats, svt_av1_compute_stats_hip_impl(wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H), (wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H)); }
static void svt_av1_compute_stats_hip_impl(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H) {
    stats_tier_a(wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H, 0, 8);
}
static void svt_av1_compute_stats_highbd_hip_impl(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H, int32_t bit_depth);
extern "C" void svt_av1_compute_stats_highbd_hip(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H, int32_t bit_depth) { TIER_A_CALL(svt_av1_compute_stats_highbd, svt_av1_compute_stats_highbd_hip_impl(wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H, bit_depth), (wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H, bit_depth)); }
static void svt_av1_compute_stats_highbd_hip_impl(int32_t wiener_win, const uint8_t *dgd8, const uint8_t *src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end, int32_t dgd_stride, int32_t src_stride, int64_t *M, int64_t *H, int32_t bit_depth) {
    stats_tier_a(wiener_win, dgd8, src8, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H, 1, bit_depth);
}

static void convolve_tier_a(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *fx, const int16_t *fy,
                            int32_t w, int32_t h, int is16, int bd) {
    if (!ensure_init())
        fatal("wiener_convolve_add_src");
    const int    px = is16 ? 2 : 1;
    const size_t ip = (size_t)w + 8, ibytes = up256(ip * (h + 8) * px), obytes = up256((size_t)w * h * px);
    hipStream_t  st = resolve_stream(nullptr);
    Scratch     &sc = tls_scratch();
    uint8_t     *d = sc.device(ibytes + obytes + 256), *hh = sc.host(ibytes + obytes + 256);
    const uint8_t *sb = is16 ? (const uint8_t *)((uintptr_t)src << 1) : src;
    uint8_t       *ob = is16 ? (uint8_t *)((uintptr_t)dst << 1) : dst;
    for (int r = 0; r < h + 7; r++) memcpy(hh + (size_t)r * ip * px, sb + ((ptrdiff_t)(r - 3) * src_stride - 3) * px, (size_t)(w + 7) * px);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(d, hh, ibytes, hipMemcpyHostToDevice, st));
    // the reference reads its kernels through a 256-byte aligned table base + offset (convolve.c:45-54): with step 16 that is
    // the kernel at the pointer itself; the 8th coefficient of a Wiener kernel is always zero
    int16_t kx[8], ky[8];
    memcpy(kx, fx, 16), memcpy(ky, fy, 16);
    if (svt_hip_wiener_convolve(d + (3 * ip + 3) * px, (uint32_t)ip, d + ibytes, (uint32_t)w, (uint32_t)w, (uint32_t)h, kx, ky, is16, bd, st) != SVT_HIP_OK)
        fatal("wiener_convolve_add_src");
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(hh + ibytes, d + ibytes, obytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    for (int r = 0; r < h; r++) memcpy(ob + (size_t)r * dst_stride * px, hh + ibytes + (size_t)r * w * px, (size_t)w * px);
}
static void svt_av1_wiener_convolve_add_src_hip_impl(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params);
extern "C" void svt_av1_wiener_convolve_add_src_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params) { TIER_A_CALL(svt_av1_wiener_convolve_add_src, svt_av1_wiener_convolve_add_src_hip_impl(src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, conv_params), (src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, conv_params)); }
static void svt_av1_wiener_convolve_add_src_hip_impl(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params) {
    (void)conv_params;  // get_conv_params_wiener(8) is the only value the reference passes (restoration.c:443)
    convolve_tier_a(src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, 0, 8);
}
static void svt_av1_highbd_wiener_convolve_add_src_hip_impl(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params, int32_t bd);
extern "C" void svt_av1_highbd_wiener_convolve_add_src_hip(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params, int32_t bd) { TIER_A_CALL(svt_av1_highbd_wiener_convolve_add_src, svt_av1_highbd_wiener_convolve_add_src_hip_impl(src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, conv_params, bd), (src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, conv_params, bd)); }
static void svt_av1_highbd_wiener_convolve_add_src_hip_impl(const uint8_t *src, ptrdiff_t src_stride, uint8_t *dst, ptrdiff_t dst_stride, const int16_t *filter_x, const int16_t *filter_y, int32_t w, int32_t h, const SvtHipConvolveParams *conv_params, int32_t bd) {
    (void)conv_params;
    convolve_tier_a(src, src_stride, dst, dst_stride, filter_x, filter_y, w, h, 1, bd);
}

SVT_HIP_MODULE_WARMUP(loopfilter_wiener)
