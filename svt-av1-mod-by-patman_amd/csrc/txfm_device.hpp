// txfm_device.hpp — AV1 1-D transforms as fully unrolled register networks (gfx950).
//
// One LANE owns one row or column of a transform block; the whole butterfly network runs in its VGPRs.
// The networks are generated at compile time from their structure (the same derivation as
// oracle/src/orc_txfm.c): a DCT of size N = mirror butterfly + DCT(N/2) on the sums + an "odd part" of
// log2(N)-1 rotation levels on the differences; the 8/16-point ADSTs are rotation/butterfly ladders.  All loop
// bounds, indices and rotation angles are constant expressions, so after unrolling every array element
// is a register and every twiddle a scalar load from constant memory.
//
// Arithmetic contract (bit-exact with the reference, Source/Lib/Codec/inv_transforms.h:264-285): rotations use
// 32-bit WRAPPING products, a 64-bit sum and one rounding shift (half_btf); plain additions wrap; the inverse
// clamps every addition to the pass's stage range (inv_transforms.c:42-84).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svthip {
namespace txd {

__constant__ int32_t d_cospi[4][64];  // bits 10..13; uploaded by txfm_init_tables()
__constant__ int32_t d_sinpi[4][5];

struct Rot {
    const int32_t *c;  // d_cospi[bit-10]
    int            bit;
    int            clamp;  // inverse only: clamp bits for additions (0 = none)
};

constexpr int cbrev(int bits, int x) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int clog2(int n) {
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

__device__ __forceinline__ int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t rshift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }
__device__ __forceinline__ int32_t btf(int32_t w0, int32_t a, int32_t w1, int32_t b, int bit) {
    const int64_t r = (int64_t)mul32(w0, a) + (int64_t)mul32(w1, b);
    return (int32_t)((r + ((int64_t)1 << (bit - 1))) >> bit);
}
template <bool INV>
__device__ __forceinline__ int32_t clampv(int32_t v, int bit) {
    if (!INV)
        return v;
    const int32_t hi = (int32_t)(((int64_t)1 << (bit - 1)) - 1), lo = -hi - 1;  // bit is 16..20 here
    return v > hi ? hi : (v < lo ? lo : v);
}

// ---------------------------------------------------------------------------------------------- DCT
__device__ __forceinline__ void lvl_f1(int32_t *a, int lo, int hi, int S, int C, const Rot &r) {
    const int32_t x = a[lo], y = a[hi];
    a[lo] = btf(-r.c[S], x, r.c[C], y, r.bit);
    a[hi] = btf(r.c[S], y, r.c[C], x, r.bit);
}
__device__ __forceinline__ void lvl_f2(int32_t *a, int lo, int hi, int S, int C, const Rot &r) {
    const int32_t x = a[lo], y = a[hi];
    a[lo] = btf(-r.c[C], x, -r.c[S], y, r.bit);
    a[hi] = btf(r.c[C], y, -r.c[S], x, r.bit);
}
template <int M, int LV>
__device__ __forceinline__ void odd_level(int32_t *a, const Rot &r) {  // a points at the odd part base
    if constexpr (LV == 1) {
#pragma unroll
        for (int j = M / 4; j < M / 2; j++) lvl_f1(a, j, M - 1 - j, 32, 32, r);
    } else {
        constexpr int G = 1 << (LV - 2), gs = (M / 2) / G, q = gs / 4, unit = 64 >> LV;
#pragma unroll
        for (int k = 0; k < G; k++) {
            const int S = unit * (1 + 4 * cbrev(LV - 2, k)), C = 64 - S;
#pragma unroll
            for (int j = q; j < 2 * q; j++) lvl_f1(a, k * gs + j, M - 1 - (k * gs + j), S, C, r);
#pragma unroll
            for (int j = 2 * q; j < 3 * q; j++) lvl_f2(a, k * gs + j, M - 1 - (k * gs + j), S, C, r);
        }
    }
}
template <int M, int G, bool INV>
__device__ __forceinline__ void odd_bf(int32_t *a, const Rot &r) {
#pragma unroll
    for (int t = 0; t < M / G; t++)
#pragma unroll
        for (int i = 0; i < G / 2; i++) {
            const int     lo = t * G + i, hi = t * G + G - 1 - i;
            const int32_t x = a[lo], y = a[hi];
            if (!(t & 1)) {
                a[lo] = clampv<INV>(add32(x, y), r.clamp);
                a[hi] = clampv<INV>(sub32(x, y), r.clamp);
            } else {
                a[lo] = clampv<INV>(sub32(y, x), r.clamp);
                a[hi] = clampv<INV>(add32(y, x), r.clamp);
            }
        }
}
template <int M, bool INV>
__device__ __forceinline__ void odd_out(int32_t *a, const Rot &r) {
    constexpr int L = clog2(M), unit = 64 / (2 * M);
#pragma unroll
    for (int i = 0; i < M / 2; i++) {
        const int     B = unit * (1 + 4 * cbrev(L - 1, i)), A = 64 - B;
        const int     lo = i, hi = M - 1 - i;
        const int32_t x = a[lo], y = a[hi];
        if (!INV) {
            a[lo] = btf(r.c[A], x, r.c[B], y, r.bit);
            a[hi] = btf(r.c[A], y, -r.c[B], x, r.bit);
        } else {
            a[lo] = btf(r.c[A], x, -r.c[B], y, r.bit);
            a[hi] = btf(r.c[B], x, r.c[A], y, r.bit);
        }
    }
}
template <int M, bool INV>
__device__ __forceinline__ void odd_part(int32_t *a, const Rot &r) {
    constexpr int L = clog2(M);
    if constexpr (!INV) {
        if constexpr (L > 1) { odd_level<M, 1>(a, r); odd_bf<M, (M >> 1), false>(a, r); }
        if constexpr (L > 2) { odd_level<M, 2>(a, r); odd_bf<M, (M >> 2), false>(a, r); }
        if constexpr (L > 3) { odd_level<M, 3>(a, r); odd_bf<M, (M >> 3), false>(a, r); }
        if constexpr (L > 4) { odd_level<M, 4>(a, r); odd_bf<M, (M >> 4), false>(a, r); }
        odd_out<M, false>(a, r);
    } else {
        odd_out<M, true>(a, r);
        if constexpr (L > 4) { odd_bf<M, (M >> 4), true>(a, r); odd_level<M, 4>(a, r); }
        if constexpr (L > 3) { odd_bf<M, (M >> 3), true>(a, r); odd_level<M, 3>(a, r); }
        if constexpr (L > 2) { odd_bf<M, (M >> 2), true>(a, r); odd_level<M, 2>(a, r); }
        if constexpr (L > 1) { odd_bf<M, (M >> 1), true>(a, r); odd_level<M, 1>(a, r); }
    }
}
template <int N>
__device__ __forceinline__ void fdct_rec(int32_t *a, const Rot &r) {
    if constexpr (N == 2) {
        const int32_t x = a[0], y = a[1];
        a[0] = btf(r.c[32], x, r.c[32], y, r.bit);
        a[1] = btf(-r.c[32], y, r.c[32], x, r.bit);
    } else {
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const int32_t x = a[i], y = a[N - 1 - i];
            a[i] = add32(x, y), a[N - 1 - i] = sub32(x, y);
        }
        fdct_rec<N / 2>(a, r);
        odd_part<N / 2, false>(a + N / 2, r);
    }
}
template <int N>
__device__ __forceinline__ void idct_rec(int32_t *a, const Rot &r) {
    if constexpr (N == 2) {
        const int32_t x = a[0], y = a[1];
        a[0] = btf(r.c[32], x, r.c[32], y, r.bit);
        a[1] = btf(r.c[32], x, -r.c[32], y, r.bit);
    } else {
        idct_rec<N / 2>(a, r);
        odd_part<N / 2, true>(a + N / 2, r);
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const int32_t x = a[i], y = a[N - 1 - i];
            a[i] = clampv<true>(add32(x, y), r.clamp), a[N - 1 - i] = clampv<true>(sub32(x, y), r.clamp);
        }
    }
}
// in-place, natural order in and out
template <int N>
__device__ __forceinline__ void fdct(int32_t (&v)[N], const Rot &r) {
    fdct_rec<N>(v, r);
    int32_t t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = v[cbrev(clog2(N), k)];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
}
template <int N>
__device__ __forceinline__ void idct(int32_t (&v)[N], const Rot &r) {
    int32_t t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = v[cbrev(clog2(N), k)];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
    idct_rec<N>(v, r);
}

// --------------------------------------------------------------------------------------------- ADST
__device__ __forceinline__ void fadst4(int32_t (&v)[4], int bit) {
    const int32_t *s = d_sinpi[bit - 10];
    const int32_t  x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    if (!(x0 | x1 | x2 | x3))
        return;  // all zero stays all zero
    const int32_t s0 = mul32(s[1], x0), s1 = mul32(s[4], x0), s2 = mul32(s[2], x1), s3 = mul32(s[1], x1);
    const int32_t s4 = mul32(s[3], x2), s5 = mul32(s[4], x3), s6 = mul32(s[2], x3);
    const int32_t s7 = sub32(add32(x0, x1), x3);
    const int32_t y0 = add32(add32(s0, s2), s5), y1 = mul32(s[3], s7), y2 = add32(sub32(s1, s3), s6), y3 = s4;
    v[0] = rshift64(add32(y0, y3), bit);
    v[1] = rshift64(y1, bit);
    v[2] = rshift64(sub32(y2, y3), bit);
    v[3] = rshift64(add32(sub32(y2, y0), y3), bit);
}
__device__ __forceinline__ void iadst4(int32_t (&v)[4], int bit) {
    const int32_t *s = d_sinpi[bit - 10];
    const int32_t  x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    if (!(x0 | x1 | x2 | x3))
        return;
    int32_t       s0 = mul32(s[1], x0), s1 = mul32(s[2], x0), s2 = mul32(s[3], x1), s3 = mul32(s[4], x2);
    const int32_t s4 = mul32(s[1], x2), s5 = mul32(s[2], x3), s6 = mul32(s[4], x3);
    const int32_t s7 = add32(sub32(x0, x2), x3);
    s0 = add32(s0, s3), s1 = sub32(s1, s4), s3 = s2, s2 = mul32(s[3], s7);
    s0 = add32(s0, s5), s1 = sub32(s1, s6);
    v[0] = rshift64(add32(s0, s3), bit);
    v[1] = rshift64(add32(s1, s3), bit);
    v[2] = rshift64(s2, bit);
    v[3] = rshift64(sub32(add32(s0, s1), s3), bit);
}

template <int N> struct AdstPerm;
template <> struct AdstPerm<8> {
    static constexpr int in[8]  = {0, -7, -3, 4, -1, 6, 2, -5};  // a[j] = sign * x[index]   (transforms.c:1515-1522)
    static constexpr int out[8] = {1, 6, 3, 4, 5, 2, 7, 0};      // X[k] = a[index]          (transforms.c:1590-1597)
};
template <> struct AdstPerm<16> {
    static constexpr int in[16]  = {0, -15, -7, 8, -3, 12, 4, -11, -1, 14, 6, -9, 2, -13, -5, 10};
    static constexpr int out[16] = {1, 14, 3, 12, 5, 10, 7, 8, 9, 6, 11, 4, 13, 2, 15, 0};
};
__device__ __forceinline__ void adst_a(int32_t *a, int i, int X, const Rot &r) {  // (cX x + cY y, cY x - cX y)
    const int32_t x = a[i], y = a[i + 1];
    a[i]     = btf(r.c[X], x, r.c[64 - X], y, r.bit);
    a[i + 1] = btf(r.c[64 - X], x, -r.c[X], y, r.bit);
}
__device__ __forceinline__ void adst_b(int32_t *a, int i, int P, const Rot &r) {  // (-cP x + cQ y, cQ x + cP y)
    const int32_t x = a[i], y = a[i + 1];
    a[i]     = btf(-r.c[P], x, r.c[64 - P], y, r.bit);
    a[i + 1] = btf(r.c[64 - P], x, r.c[P], y, r.bit);
}
template <int N, int T>
__device__ __forceinline__ void adst_rot(int32_t *a, const Rot &r) {
    constexpr int G = 2 << T;
#pragma unroll
    for (int g = 0; g < N; g += G) {
        if constexpr (T == 1) {
            adst_a(a, g + 2, 32, r);
        } else {
            constexpr int np = G / 4, unit = 64 >> T;
#pragma unroll
            for (int p = 0; p < np / 2; p++) adst_a(a, g + G / 2 + 2 * p, unit * (1 + 4 * p), r);
#pragma unroll
            for (int p = 0; p < np / 2; p++) adst_b(a, g + G / 2 + np + 2 * p, 64 - unit * (1 + 4 * p), r);
        }
    }
}
template <int N, int SPAN, bool INV>
__device__ __forceinline__ void adst_bf(int32_t *a, const Rot &r) {
#pragma unroll
    for (int g = 0; g < N; g += 2 * SPAN)
#pragma unroll
        for (int i = 0; i < SPAN; i++) {
            const int32_t x = a[g + i], y = a[g + i + SPAN];
            a[g + i]        = clampv<INV>(add32(x, y), r.clamp);
            a[g + i + SPAN] = clampv<INV>(sub32(x, y), r.clamp);
        }
}
template <int N>
__device__ __forceinline__ void adst_final(int32_t *a, const Rot &r) {
    constexpr int unit = N == 8 ? 16 : 8, first = N == 8 ? 4 : 2;
#pragma unroll
    for (int j = 0; j < N / 2; j++) adst_a(a, 2 * j, first + unit * j, r);
}
template <int N>
__device__ __forceinline__ void fadst(int32_t (&v)[N], const Rot &r) {
    if constexpr (N == 4) {
        fadst4(v, r.bit);
    } else {
        int32_t a[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            constexpr auto &P = AdstPerm<N>::in;
            a[j] = P[j] < 0 ? (int32_t)(0u - (uint32_t)v[-P[j]]) : v[P[j]];
        }
        adst_rot<N, 1>(a, r);
        adst_bf<N, 2, false>(a, r);
        adst_rot<N, 2>(a, r);
        adst_bf<N, 4, false>(a, r);
        if constexpr (N == 16) {
            adst_rot<N, 3>(a, r);
            adst_bf<N, 8, false>(a, r);
        }
        adst_final<N>(a, r);
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = a[AdstPerm<N>::out[k]];
    }
}
template <int N>
__device__ __forceinline__ void iadst(int32_t (&v)[N], const Rot &r) {
    if constexpr (N == 4) {
        iadst4(v, r.bit);
    } else {
        int32_t a[N];
#pragma unroll
        for (int k = 0; k < N; k++) a[AdstPerm<N>::out[k]] = v[k];
        adst_final<N>(a, r);
        if constexpr (N == 16) {
            adst_bf<N, 8, true>(a, r);
            adst_rot<N, 3>(a, r);
        }
        adst_bf<N, 4, true>(a, r);
        adst_rot<N, 2>(a, r);
        adst_bf<N, 2, true>(a, r);
        adst_rot<N, 1>(a, r);
#pragma unroll
        for (int j = 0; j < N; j++) {
            constexpr auto &P = AdstPerm<N>::in;
            if (P[j] < 0)
                v[-P[j]] = (int32_t)(0u - (uint32_t)a[j]);
            else
                v[P[j]] = a[j];
        }
    }
}

// ----------------------------------------------------------------------------------------- identity
template <int N>
__device__ __forceinline__ void identity(int32_t (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (N == 4)
            v[i] = rshift64((int64_t)v[i] * 5793, 12);
        else if constexpr (N == 8)
            v[i] = (int32_t)((int64_t)v[i] * 2);
        else if constexpr (N == 16)
            v[i] = rshift64((int64_t)v[i] * 2 * 5793, 12);
        else if constexpr (N == 32)
            v[i] = (int32_t)((int64_t)v[i] * 4);
        else
            v[i] = rshift64((int64_t)v[i] * 4 * 5793, 12);
    }
}

// kind: 0 DCT, 1 ADST (also FLIPADST: the flip is applied by the caller), 3 identity
template <int N>
__device__ __forceinline__ void fwd1d(int32_t (&v)[N], int kind, int bit) {
    const Rot r{d_cospi[bit - 10], bit, 0};
    if (kind == 0) {
        fdct<N>(v, r);
    } else if (kind == 3) {
        identity<N>(v);
    } else {
        if constexpr (N <= 16)
            fadst<N>(v, r);
    }
}
template <int N>
__device__ __forceinline__ void inv1d(int32_t (&v)[N], int kind, int clamp) {
    const Rot r{d_cospi[2], 12, clamp};
    if (kind == 0) {
        idct<N>(v, r);
    } else if (kind == 3) {
        identity<N>(v);
    } else {
        if constexpr (N <= 16)
            iadst<N>(v, r);
    }
}

}  // namespace txd
}  // namespace svthip
