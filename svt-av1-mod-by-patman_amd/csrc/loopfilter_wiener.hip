// loopfilter_wiener.hip — Wiener restoration on gfx950.
// Replaces svt_av1_compute_stats_c / _highbd_c (restoration_pick.c:671-757) and svt_av1_wiener_convolve_add_src_c /
// svt_av1_highbd_wiener_convolve_add_src_c (convolve.c:57-200).
//
// Statistics.  The reference accumulates, per sample of the unit, y[k]*x and y[k]*y[l] for the win^2 taps
// y[k] = dgd(sample + tap k) - avg.  Here the RAW second moments are accumulated instead and the mean is folded in at the
// end (exact integer algebra: sum (a - m)(b - m) = sum ab - m sum a - m sum b + N m^2), so no pre-pass over the unit is
// needed for `avg`.  One workgroup owns a 64 x 32 tile of the unit, staged in LDS with its border; a thread owns one
// pair of tap COLUMNS (c1 <= c2) — or a tap column and the source, which is the same sliding block — and walks DOWN a
// column of horizontally adjacent sample PAIRS: per step it reads one new row of each of its two columns (one aligned
// 32-bit LDS read each: the tile is kept twice, the second copy shifted by one sample; the other six vertical taps are
// the previous step's) and does the 49 multiply-accumulates of that 7 x 7 block of H as 49 v_dot2_u32_u16 in registers
// (uint32 partials, flushed per tile to int64).  28 column pairs + 7 column-by-source blocks x 7 column slices fill the
// 256 lanes; the first moments come from column sums of the tile.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/svt_hip_lf.h"
#include "common.hpp"
#include "lr_device.hpp"

using namespace svthip;

namespace {

constexpr int TW = 64, TH = 32;  // samples per tile (TH shrinks to 8 for 12-bit so that int32 partials cannot overflow)
constexpr int W2MAX = 49;  // WIENER_WIN2

using svthip::lr::ldpx;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct StatsAux {  // raw first moments of one unit
    long long S[W2MAX];  // sum of dgd at tap k
    long long sum_src, n;
};

constexpr int CHUNKS = 4;  // row chunks of one tile column that a workgroup accumulates before it touches global memory

template <int WIN>
__global__ __launch_bounds__(256, 4) void wiener_stats_kernel(const SvtHipWienerUnit *__restrict__ units, int is16, int th,
                                                           long long *__restrict__ M, long long *__restrict__ H, StatsAux *__restrict__ aux) {
    constexpr int HALF = WIN / 2, W2 = WIN * WIN;
    constexpr int NPAIR = WIN * (WIN + 1) / 2, NJOB = NPAIR + WIN;  // column pairs (blocks of H), column x source (rows of M)
    constexpr int NSL = 256 / NJOB;                                  // column slices per job
    constexpr int DP = TW + 2 * 3 + 2;
    // d1 is d shifted left by one sample: a thread whose column offset is odd reads its sample PAIRS from d1, so that every
    // pair is one aligned 32-bit LDS read.  s holds the source rows HALF rows down (rows above / below: zero), which makes a
    // column-by-source job the same sliding 7 x 7 block as a column pair: its middle column is the row of M.
    __shared__ __attribute__((aligned(4))) uint16_t d[(TH + 2 * 3) * DP];
    __shared__ __attribute__((aligned(4))) uint16_t d1[(TH + 2 * 3) * DP];
    __shared__ __attribute__((aligned(4))) uint16_t s[(TH + 2 * 3) * TW];
    __shared__ uint32_t  cs[WIN][TW + 2 * 3];                     // column sums of d over the tile rows, per vertical tap
    __shared__ long long Hl[W2 * W2], Ml[W2], Sl[W2], misc[2];    // this workgroup's totals (int64), flushed once at the end
    const SvtHipWienerUnit u = units[blockIdx.z];
    const int uw = u.h_end - u.h_start, uh = u.v_end - u.v_start;
    const int x0 = blockIdx.x * TW;
    if (x0 >= uw || (int)(blockIdx.y * CHUNKS * th) >= uh)
        return;
    const int tw = min(TW, uw - x0);
    for (int i = threadIdx.x; i < W2 * W2; i += 256) Hl[i] = 0;
    if (threadIdx.x < W2)
        Ml[threadIdx.x] = 0, Sl[threadIdx.x] = 0;
    if (threadIdx.x < 2)
        misc[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < (TH + 2 * 3) * TW; i += 256) s[i] = 0;
    const int job = threadIdx.x / NSL, sl = threadIdx.x - job * NSL;
    int       c1 = 0, c2 = 0;
    if (job < NPAIR) {  // column pair (c1 <= c2) from the triangular index
        int rem = job;
        while (rem >= WIN - c1) rem -= WIN - c1, c1++;
        c2 = c1 + rem;
    } else {
        c1 = job - NPAIR;
    }
    const bool is_m = job >= NPAIR;
    for (int chunk = 0; chunk < CHUNKS; chunk++) {
        const int y0 = (blockIdx.y * CHUNKS + chunk) * th;
        if (y0 >= uh)
            break;
        const int tv = min(th, uh - y0), npx = tv * tw;
        __syncthreads();  // previous chunk fully consumed (and the zeroing above done)
        for (int idx = threadIdx.x; idx < (tv + 2 * HALF) * (tw + 2 * HALF); idx += 256) {
            const int r = idx / (tw + 2 * HALF), c = idx - r * (tw + 2 * HALF);
            const uint16_t v = (uint16_t)ldpx(u.dgd, (size_t)((ptrdiff_t)(u.v_start + y0 + r - HALF) * u.dgd_stride + (u.h_start + x0 + c - HALF)), is16);
            d[r * DP + c] = v;
            if (c)
                d1[r * DP + c - 1] = v;
        }
        for (int idx = threadIdx.x; idx < (tv + HALF) * TW; idx += 256) {   // rows below a short last chunk read as zero again
            const int r = idx / TW, c = idx - r * TW;
            s[(r + HALF) * TW + c] = r < tv && c < tw
                ? (uint16_t)ldpx(u.src, (size_t)((ptrdiff_t)(u.v_start + y0 + r) * u.src_stride + (u.h_start + x0 + c)), is16) : (uint16_t)0;
        }
        __syncthreads();
        // Two horizontally adjacent samples per step: each multiply-accumulate of the block is one v_dot2_u32_u16 over the
        // pair (the partner of the last sample of an odd-width tile is masked to zero).  A thread walks DOWN its sample
        // columns: the WIN vertical taps of one row are WIN - 1 of the previous row's, so a step costs two LDS reads for the
        // WIN x WIN multiply-accumulates; the row loop is unrolled WIN times, which makes the rotation of the tap registers
        // a renaming.
        const int hw = (tw + 1) >> 1;
        if (job < NJOB) {
            uint32_t acc[WIN][WIN];
#pragma unroll
            for (int a = 0; a < WIN; a++)
#pragma unroll
                for (int b = 0; b < WIN; b++) acc[a][b] = 0;
            const uint16_t *ta = ((c1 & 1) ? d1 - 1 : d) + c1;
            const uint16_t *tb = is_m ? s : ((c2 & 1) ? d1 - 1 : d) + c2;
            const int       pb = is_m ? TW : DP;
            for (int cp = sl; cp < hw; cp += NSL) {
                const int      c    = 2 * cp;
                const uint32_t mask = c + 1 < tw ? 0xffffffffu : 0x0000ffffu;
                const uint16_t *qa = ta + c, *qb = tb + c;
                u16x2           va[WIN], vb[WIN];
#pragma unroll
                for (int k = 0; k < WIN - 1; k++) {
                    va[k] = __builtin_bit_cast(u16x2, *(const uint32_t *)&qa[k * DP] & mask);
                    vb[k] = __builtin_bit_cast(u16x2, *(const uint32_t *)&qb[k * pb]);
                }
                for (int r = 0; r < tv; r += WIN) {
#pragma unroll
                    for (int j = 0; j < WIN; j++) {
                        if (r + j < tv) {
                            constexpr int NEWEST = WIN - 1;
                            va[(j + NEWEST) % WIN] = __builtin_bit_cast(u16x2, *(const uint32_t *)&qa[(r + j + NEWEST) * DP] & mask);
                            vb[(j + NEWEST) % WIN] = __builtin_bit_cast(u16x2, *(const uint32_t *)&qb[(r + j + NEWEST) * pb]);
#pragma unroll
                            for (int a = 0; a < WIN; a++)
#pragma unroll
                                for (int b = 0; b < WIN; b++)
                                    acc[a][b] = __builtin_amdgcn_udot2(va[(j + a) % WIN], vb[(j + b) % WIN], acc[a][b], false);
                        }
                    }
                }
            }
            if (!is_m) {
                // tap index = column * WIN + row (restoration_pick.c:686-691); only the upper triangle k <= l is kept
#pragma unroll
                for (int a = 0; a < WIN; a++)
#pragma unroll
                    for (int b = 0; b < WIN; b++) {
                        const int k = c1 * WIN + a, l = c2 * WIN + b;
                        if (k <= l && acc[a][b])
                            atomicAdd((unsigned long long *)&Hl[k * W2 + l], (unsigned long long)acc[a][b]);
                    }
            } else {
#pragma unroll
                for (int a = 0; a < WIN; a++) atomicAdd((unsigned long long *)&Ml[c1 * WIN + a], (unsigned long long)acc[a][HALF]);
            }
        }
        // first moments: sum of dgd under every tap = column sums of the tile slid down the WIN vertical taps, then summed
        // over the tw columns behind each horizontal tap; the source sum likewise
        if ((int)threadIdx.x < tw + 2 * HALF) {
            const int x = threadIdx.x;
            uint32_t  w = 0;
            for (int r = 0; r < tv; r++) w += d[r * DP + x];
            cs[0][x] = w;
#pragma unroll
            for (int a = 1; a < WIN; a++) {
                w += (uint32_t)d[(tv + a - 1) * DP + x] - (uint32_t)d[(a - 1) * DP + x];
                cs[a][x] = w;
            }
        } else if (threadIdx.x >= 128 && (int)threadIdx.x < 128 + tw) {
            const int x = threadIdx.x - 128;
            long long ss = 0;
            for (int r = 0; r < tv; r++) ss += s[(r + HALF) * TW + x];
            atomicAdd((unsigned long long *)&misc[0], (unsigned long long)ss);
        }
        __syncthreads();
        if (threadIdx.x < W2) {  // k = column * WIN + row
            const int cc = threadIdx.x / WIN, a = threadIdx.x - cc * WIN;
            long long t = 0;
            for (int c = 0; c < tw; c++) t += cs[a][c + cc];
            Sl[threadIdx.x] += t;
        }
        if (threadIdx.x == 0)
            misc[1] += npx;
    }
    __syncthreads();
    long long *Hu = H + (size_t)blockIdx.z * W2MAX * W2MAX, *Mu = M + (size_t)blockIdx.z * W2MAX;
    StatsAux  &A  = aux[blockIdx.z];
    for (int e = threadIdx.x; e < W2 * W2; e += 256)
        if (e / W2 <= e % W2 && Hl[e])
            atomicAdd((unsigned long long *)&Hu[e], (unsigned long long)Hl[e]);
    if (threadIdx.x < W2) {
        atomicAdd((unsigned long long *)&Mu[threadIdx.x], (unsigned long long)Ml[threadIdx.x]);
        atomicAdd((unsigned long long *)&A.S[threadIdx.x], (unsigned long long)Sl[threadIdx.x]);
    }
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *)&A.sum_src, (unsigned long long)misc[0]);
        atomicAdd((unsigned long long *)&A.n, (unsigned long long)misc[1]);
    }
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
    SVT_HIP_CHECK(hipMemsetAsync(d_M, 0, sizeof(int64_t) * W2MAX * n_units, st));
    SVT_HIP_CHECK(hipMemsetAsync(d_H, 0, sizeof(int64_t) * W2MAX * W2MAX * n_units, st));
    SVT_HIP_CHECK(hipMemsetAsync(ab.dev, 0, sizeof(StatsAux) * n_units, st));
    // int32 partials: a thread sees at most th*64/NSL samples of one tile; 12-bit products need the smaller tile
    const int  th = bit_depth == 12 ? 8 : TH;
    const dim3 grid((max_w + TW - 1) / TW, (max_h + CHUNKS * th - 1) / (CHUNKS * th), n_units);
    if (wiener_win == 7)
        hipLaunchKernelGGL(wiener_stats_kernel<7>, grid, dim3(256), 0, st, d_units, is_16bit, th, (long long *)d_M, (long long *)d_H, ab.dev);
    else
        hipLaunchKernelGGL(wiener_stats_kernel<5>, grid, dim3(256), 0, st, d_units, is_16bit, th, (long long *)d_M, (long long *)d_H, ab.dev);
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
