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

#include <type_traits>
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TXD_FN __device__ __forceinline__
#else
// host build: tools/txfm_bounds.cpp instantiates the same networks over intervals to derive the Fast-path limits
#define TXD_FN inline
#endif

namespace svthip {
namespace txd {

// Twiddle tables as compile-time constants: cospi[b][j] = round(cos(pi * j / 128) * 2^(10 + b)) (inv_transforms.c:3150-3224)
// and the sinpi constants of the 4-point ADST (inv_transforms.c:3226-3234), bits 10..13.  After inlining every table index
// is a constant expression, so each twiddle becomes an instruction literal instead of a scalar load the wave waits for.
constexpr double cx_cos(double x) {  // Taylor series, x in [0, pi/2]: error below 1e-15
    double sum = 1.0, term = 1.0;
    for (int k = 1; k <= 16; k++) {
        term *= -(x * x) / (double)((2 * k - 1) * (2 * k));
        sum += term;
    }
    return sum;
}
struct CosTable {
    int32_t v[4][64];
};
constexpr CosTable make_cospi() {
    CosTable t{};
    for (int b = 0; b < 4; b++)
        for (int j = 0; j < 64; j++) t.v[b][j] = (int32_t)(cx_cos(3.14159265358979323846 * j / 128.0) * (double)(1 << (10 + b)) + 0.5);
    return t;
}
struct SinTable {
    int32_t v[4][5];
};
#if defined(__HIPCC__)
#define TXD_CONST __device__ constexpr
#else
#define TXD_CONST constexpr
#endif
TXD_CONST CosTable COSPI = make_cospi();
TXD_CONST SinTable SINPI = {{{0, 330, 621, 836, 951}, {0, 660, 1241, 1672, 1901}, {0, 1321, 2482, 3344, 3803}, {0, 2642, 4964, 6689, 7606}}};
static_assert(make_cospi().v[2][32] == 2896 && make_cospi().v[3][32] == 5793 && make_cospi().v[0][0] == 1024 && make_cospi().v[3][63] == 201,
              "cospi table");

struct Rot {
    const int32_t *c;  // COSPI.v[bit - 10]
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

TXD_FN int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
TXD_FN int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
TXD_FN int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
TXD_FN int32_t rshift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }
TXD_FN int32_t clamp_bits(int32_t v, int bit) {
    const int32_t hi = (int32_t)(((int64_t)1 << (bit - 1)) - 1), lo = -hi - 1;  // bit is 16..20 here
#if defined(__HIPCC__)
    // one instruction; the compiler only forms a median-of-three for constant bounds
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#else
    return v > hi ? hi : (v < lo ? lo : v);
#endif
}
template <bool INV>
TXD_FN int32_t clampv(int32_t v, int bit) {
    return INV ? clamp_bits(v, bit) : v;
}

// ------------------------------------------------------------------------------------- arithmetic policies
// The networks below are written once over an arithmetic policy A (value type A::T):
//   Exact — the reference's arithmetic verbatim: 32-bit wrapping products, 64-bit sum, one rounding shift.  On gfx950 a
//           32-bit product is a quarter-rate instruction and the 64-bit tail costs six more.
//   Fast  — 24-bit multiplies (full rate) and a 32-bit sum.  Gives the same bits as Exact whenever every multiplied value is
//           below 2^23 in magnitude and no sum leaves 32 bits; the callers only take it when that is PROVEN for the vector at
//           hand: forward transforms compare the largest input magnitude with fast_fwd_limit() (derived by propagating
//           intervals through these same templates, tools/txfm_bounds.cpp), inverse transforms are covered for 8/10-bit
//           by their stage-range clamps.
//   (tools/txfm_bounds.cpp adds a third policy over intervals.)
struct Exact {
    using T = int32_t;
    static TXD_FN T add(T a, T b) { return add32(a, b); }
    static TXD_FN T sub(T a, T b) { return sub32(a, b); }
    static TXD_FN T neg(T a) { return (int32_t)(0u - (uint32_t)a); }
    static TXD_FN T mul(int32_t w, T a) { return mul32(w, a); }
    static TXD_FN T btf(int32_t w0, T a, int32_t w1, T b, int bit) {
        const int64_t r = (int64_t)mul32(w0, a) + (int64_t)mul32(w1, b);
        return (int32_t)((r + ((int64_t)1 << (bit - 1))) >> bit);
    }
    static TXD_FN T rs(T v, int bit) { return rshift64(v, bit); }                                   // rounding shift
    static TXD_FN T scale(T v, int32_t k, int bit) { return rshift64((int64_t)v * k, bit); }        // (v * k + half) >> bit
    static TXD_FN T times(T v, int32_t k) { return (int32_t)((int64_t)v * k); }
    template <bool INV> static TXD_FN T clamp(T v, int bit) { return clampv<INV>(v, bit); }
    static TXD_FN bool all_zero4(T a, T b, T c, T d) { return !(a | b | c | d); }
};
#if defined(__HIPCC__)
// v_mul_i32_i24 by its LLVM intrinsic: HIP's __mul24 is ((x << 8) >> 8) * ((y << 8) >> 8) in IR, and once the optimiser has proven the
// shifts redundant the instruction selector may not re-derive the 24-bit range and falls back to the quarter-rate v_mul_lo_u32
extern "C" __device__ int mul_i24(int, int) __asm("llvm.amdgcn.mul.i24");
extern "C" __device__ unsigned mul_u24(unsigned, unsigned) __asm("llvm.amdgcn.mul.u24");
extern "C" __device__ unsigned mulhi_u24(unsigned, unsigned) __asm("llvm.amdgcn.mulhi.u24");
struct Fast {
    using T = int32_t;
    static TXD_FN T add(T a, T b) { return a + b; }
    static TXD_FN T sub(T a, T b) { return a - b; }
    static TXD_FN T neg(T a) { return -a; }
    static TXD_FN T mul(int32_t w, T a) { return mul_i24(w, a); }
    static TXD_FN T btf(int32_t w0, T a, int32_t w1, T b, int bit) {
        return (mul_i24(w0, a) + mul_i24(w1, b) + (1 << (bit - 1))) >> bit;
    }
    static TXD_FN T rs(T v, int bit) { return (v + (1 << (bit - 1))) >> bit; }
    static TXD_FN T scale(T v, int32_t k, int bit) { return (mul_i24(v, k) + (1 << (bit - 1))) >> bit; }
    static TXD_FN T times(T v, int32_t k) { return mul_i24(v, k); }
    template <bool INV> static TXD_FN T clamp(T v, int bit) { return clampv<INV>(v, bit); }
    static TXD_FN bool all_zero4(T a, T b, T c, T d) { return !(a | b | c | d); }
};
#endif

// ------------------------------------------------------------------------------------- Fast-path limits
#if defined(__HIPCC__)
#define TXD_TABLE __device__ const
#else
#define TXD_TABLE static const
#endif
#if defined(__HIPCC__) || defined(TXD_WITH_TABLES)
// BEGIN GENERATED (tools/txfm_bounds.cpp)
// largest input magnitude for which Fast == Exact: [log2(N) - 2][kind: DCT, ADST, identity][cos bit - 10]
TXD_TABLE int32_t FWD_FAST_LIMIT[5][3][4] = {
    /* 4 */ {{741534, 370767, 185383, 92675}, {462819, 231509, 115729, 57863}, {370702, 370702, 370702, 370702}},
    /* 8 */ {{313229, 156559, 78279, 39142}, {267445, 133814, 66914, 33445}, {8388607, 8388607, 8388607, 8388607}},
    /* 16 */ {{133722, 66907, 33457, 16722}, {112837, 56386, 28195, 14096}, {185351, 185351, 185351, 185351}},
    /* 32 */ {{56418, 28193, 14097, 7048}, {0, 0, 0, 0}, {8388607, 8388607, 8388607, 8388607}},
    /* 64 */ {{23888, 11939, 5970, 2984}, {0, 0, 0, 0}, {92675, 92675, 92675, 92675}},
};
// stage clamps make Fast == Exact: [bit depth 8, 10, 12][pass: row, column][log2(N) - 2][kind: DCT, ADST, identity]
TXD_TABLE uint8_t INV_FAST_OK[3][2][5][3] = {
    {{{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 0, 1}, {1, 0, 1}}, {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 0, 1}, {1, 0, 1}}},
    {{{1, 0, 1}, {1, 1, 1}, {1, 1, 1}, {1, 0, 1}, {1, 0, 0}}, {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 0, 1}, {1, 0, 1}}},
    {{{0, 0, 0}, {0, 0, 1}, {0, 0, 0}, {0, 0, 1}, {0, 0, 0}}, {{1, 0, 1}, {1, 1, 1}, {1, 1, 1}, {1, 0, 1}, {1, 0, 0}}},
};
// END GENERATED
#endif
// index of a 1-D kernel kind (0 DCT, 1 ADST, 2 FLIPADST, 3 identity) in the tables above
TXD_FN int kind_index(int kind) { return kind == 0 ? 0 : (kind == 3 ? 2 : 1); }

// ---------------------------------------------------------------------------------------------- DCT
template <class A>
TXD_FN void lvl_f1(typename A::T *a, int lo, int hi, int S, int C, const Rot &r) {
    const typename A::T x = a[lo], y = a[hi];
    a[lo] = A::btf(-r.c[S], x, r.c[C], y, r.bit);
    a[hi] = A::btf(r.c[S], y, r.c[C], x, r.bit);
}
template <class A>
TXD_FN void lvl_f2(typename A::T *a, int lo, int hi, int S, int C, const Rot &r) {
    const typename A::T x = a[lo], y = a[hi];
    a[lo] = A::btf(-r.c[C], x, -r.c[S], y, r.bit);
    a[hi] = A::btf(r.c[C], y, -r.c[S], x, r.bit);
}
template <class A, int M, int LV>
TXD_FN void odd_level(typename A::T *a, const Rot &r) {  // a points at the odd part base
    if constexpr (LV == 1) {
#pragma unroll
        for (int j = M / 4; j < M / 2; j++) lvl_f1<A>(a, j, M - 1 - j, 32, 32, r);
    } else {
        constexpr int G = 1 << (LV - 2), gs = (M / 2) / G, q = gs / 4, unit = 64 >> LV;
#pragma unroll
        for (int k = 0; k < G; k++) {
            const int S = unit * (1 + 4 * cbrev(LV - 2, k)), C = 64 - S;
#pragma unroll
            for (int j = q; j < 2 * q; j++) lvl_f1<A>(a, k * gs + j, M - 1 - (k * gs + j), S, C, r);
#pragma unroll
            for (int j = 2 * q; j < 3 * q; j++) lvl_f2<A>(a, k * gs + j, M - 1 - (k * gs + j), S, C, r);
        }
    }
}
template <class A, int M, int G, bool INV>
TXD_FN void odd_bf(typename A::T *a, const Rot &r) {
#pragma unroll
    for (int t = 0; t < M / G; t++)
#pragma unroll
        for (int i = 0; i < G / 2; i++) {
            const int           lo = t * G + i, hi = t * G + G - 1 - i;
            const typename A::T x = a[lo], y = a[hi];
            if (!(t & 1)) {
                a[lo] = A::template clamp<INV>(A::add(x, y), r.clamp);
                a[hi] = A::template clamp<INV>(A::sub(x, y), r.clamp);
            } else {
                a[lo] = A::template clamp<INV>(A::sub(y, x), r.clamp);
                a[hi] = A::template clamp<INV>(A::add(y, x), r.clamp);
            }
        }
}
template <class A, int M, bool INV>
TXD_FN void odd_out(typename A::T *a, const Rot &r) {
    constexpr int L = clog2(M), unit = 64 / (2 * M);
#pragma unroll
    for (int i = 0; i < M / 2; i++) {
        const int           B = unit * (1 + 4 * cbrev(L - 1, i)), Aw = 64 - B;
        const int           lo = i, hi = M - 1 - i;
        const typename A::T x = a[lo], y = a[hi];
        if (!INV) {
            a[lo] = A::btf(r.c[Aw], x, r.c[B], y, r.bit);
            a[hi] = A::btf(r.c[Aw], y, -r.c[B], x, r.bit);
        } else {
            a[lo] = A::btf(r.c[Aw], x, -r.c[B], y, r.bit);
            a[hi] = A::btf(r.c[B], x, r.c[Aw], y, r.bit);
        }
    }
}
template <class A, int M, bool INV>
TXD_FN void odd_part(typename A::T *a, const Rot &r) {
    constexpr int L = clog2(M);
    if constexpr (!INV) {
        if constexpr (L > 1) { odd_level<A, M, 1>(a, r); odd_bf<A, M, (M >> 1), false>(a, r); }
        if constexpr (L > 2) { odd_level<A, M, 2>(a, r); odd_bf<A, M, (M >> 2), false>(a, r); }
        if constexpr (L > 3) { odd_level<A, M, 3>(a, r); odd_bf<A, M, (M >> 3), false>(a, r); }
        if constexpr (L > 4) { odd_level<A, M, 4>(a, r); odd_bf<A, M, (M >> 4), false>(a, r); }
        odd_out<A, M, false>(a, r);
    } else {
        odd_out<A, M, true>(a, r);
        if constexpr (L > 4) { odd_bf<A, M, (M >> 4), true>(a, r); odd_level<A, M, 4>(a, r); }
        if constexpr (L > 3) { odd_bf<A, M, (M >> 3), true>(a, r); odd_level<A, M, 3>(a, r); }
        if constexpr (L > 2) { odd_bf<A, M, (M >> 2), true>(a, r); odd_level<A, M, 2>(a, r); }
        if constexpr (L > 1) { odd_bf<A, M, (M >> 1), true>(a, r); odd_level<A, M, 1>(a, r); }
    }
}
template <class A, int N>
TXD_FN void fdct_rec(typename A::T *a, const Rot &r) {
    if constexpr (N == 2) {
        const typename A::T x = a[0], y = a[1];
        a[0] = A::btf(r.c[32], x, r.c[32], y, r.bit);
        a[1] = A::btf(-r.c[32], y, r.c[32], x, r.bit);
    } else {
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const typename A::T x = a[i], y = a[N - 1 - i];
            a[i] = A::add(x, y), a[N - 1 - i] = A::sub(x, y);
        }
        fdct_rec<A, N / 2>(a, r);
        odd_part<A, N / 2, false>(a + N / 2, r);
    }
}
template <class A, int N>
TXD_FN void idct_rec(typename A::T *a, const Rot &r) {
    if constexpr (N == 2) {
        const typename A::T x = a[0], y = a[1];
        a[0] = A::btf(r.c[32], x, r.c[32], y, r.bit);
        a[1] = A::btf(r.c[32], x, -r.c[32], y, r.bit);
    } else {
        idct_rec<A, N / 2>(a, r);
        odd_part<A, N / 2, true>(a + N / 2, r);
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            const typename A::T x = a[i], y = a[N - 1 - i];
            a[i] = A::template clamp<true>(A::add(x, y), r.clamp), a[N - 1 - i] = A::template clamp<true>(A::sub(x, y), r.clamp);
        }
    }
}
// in-place, natural order in and out
template <class A, int N>
TXD_FN void fdct(typename A::T (&v)[N], const Rot &r) {
    fdct_rec<A, N>(v, r);
    typename A::T t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = v[cbrev(clog2(N), k)];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
}
template <class A, int N>
TXD_FN void idct(typename A::T (&v)[N], const Rot &r) {
    typename A::T t[N];
#pragma unroll
    for (int k = 0; k < N; k++) t[k] = v[cbrev(clog2(N), k)];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = t[k];
    idct_rec<A, N>(v, r);
}

// --------------------------------------------------------------------------------------------- ADST
template <class A>
TXD_FN void fadst4(typename A::T (&v)[4], int bit) {
    using T = typename A::T;
    const int32_t *s = SINPI.v[bit - 10];
    const T        x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    if (A::all_zero4(x0, x1, x2, x3))
        return;  // all zero stays all zero
    const T s0 = A::mul(s[1], x0), s1 = A::mul(s[4], x0), s2 = A::mul(s[2], x1), s3 = A::mul(s[1], x1);
    const T s4 = A::mul(s[3], x2), s5 = A::mul(s[4], x3), s6 = A::mul(s[2], x3);
    const T s7 = A::sub(A::add(x0, x1), x3);
    const T y0 = A::add(A::add(s0, s2), s5), y1 = A::mul(s[3], s7), y2 = A::add(A::sub(s1, s3), s6), y3 = s4;
    v[0] = A::rs(A::add(y0, y3), bit);
    v[1] = A::rs(y1, bit);
    v[2] = A::rs(A::sub(y2, y3), bit);
    v[3] = A::rs(A::add(A::sub(y2, y0), y3), bit);
}
template <class A>
TXD_FN void iadst4(typename A::T (&v)[4], int bit) {
    using T = typename A::T;
    const int32_t *s = SINPI.v[bit - 10];
    const T        x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
    if (A::all_zero4(x0, x1, x2, x3))
        return;
    T       s0 = A::mul(s[1], x0), s1 = A::mul(s[2], x0), s2 = A::mul(s[3], x1), s3 = A::mul(s[4], x2);
    const T s4 = A::mul(s[1], x2), s5 = A::mul(s[2], x3), s6 = A::mul(s[4], x3);
    const T s7 = A::add(A::sub(x0, x2), x3);
    s0 = A::add(s0, s3), s1 = A::sub(s1, s4), s3 = s2, s2 = A::mul(s[3], s7);
    s0 = A::add(s0, s5), s1 = A::sub(s1, s6);
    v[0] = A::rs(A::add(s0, s3), bit);
    v[1] = A::rs(A::add(s1, s3), bit);
    v[2] = A::rs(s2, bit);
    v[3] = A::rs(A::sub(A::add(s0, s1), s3), bit);
}

// compile-time loop: the body receives the index as an integral constant, so a table entry read inside it is a constant
// expression (a run-time-indexed read of a static constexpr member array is emitted as a load from a device global)
template <int I, int N, class F>
TXD_FN void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
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
template <class A>
TXD_FN void adst_a(typename A::T *a, int i, int X, const Rot &r) {  // (cX x + cY y, cY x - cX y)
    const typename A::T x = a[i], y = a[i + 1];
    a[i]     = A::btf(r.c[X], x, r.c[64 - X], y, r.bit);
    a[i + 1] = A::btf(r.c[64 - X], x, -r.c[X], y, r.bit);
}
template <class A>
TXD_FN void adst_b(typename A::T *a, int i, int P, const Rot &r) {  // (-cP x + cQ y, cQ x + cP y)
    const typename A::T x = a[i], y = a[i + 1];
    a[i]     = A::btf(-r.c[P], x, r.c[64 - P], y, r.bit);
    a[i + 1] = A::btf(r.c[64 - P], x, r.c[P], y, r.bit);
}
template <class A, int N, int T>
TXD_FN void adst_rot(typename A::T *a, const Rot &r) {
    constexpr int G = 2 << T;
#pragma unroll
    for (int g = 0; g < N; g += G) {
        if constexpr (T == 1) {
            adst_a<A>(a, g + 2, 32, r);
        } else {
            constexpr int np = G / 4, unit = 64 >> T;
#pragma unroll
            for (int p = 0; p < np / 2; p++) adst_a<A>(a, g + G / 2 + 2 * p, unit * (1 + 4 * p), r);
#pragma unroll
            for (int p = 0; p < np / 2; p++) adst_b<A>(a, g + G / 2 + np + 2 * p, 64 - unit * (1 + 4 * p), r);
        }
    }
}
template <class A, int N, int SPAN, bool INV>
TXD_FN void adst_bf(typename A::T *a, const Rot &r) {
#pragma unroll
    for (int g = 0; g < N; g += 2 * SPAN)
#pragma unroll
        for (int i = 0; i < SPAN; i++) {
            const typename A::T x = a[g + i], y = a[g + i + SPAN];
            a[g + i]        = A::template clamp<INV>(A::add(x, y), r.clamp);
            a[g + i + SPAN] = A::template clamp<INV>(A::sub(x, y), r.clamp);
        }
}
template <class A, int N>
TXD_FN void adst_final(typename A::T *a, const Rot &r) {
    constexpr int unit = N == 8 ? 16 : 8, first = N == 8 ? 4 : 2;
#pragma unroll
    for (int j = 0; j < N / 2; j++) adst_a<A>(a, 2 * j, first + unit * j, r);
}
template <class A, int N>
TXD_FN void fadst(typename A::T (&v)[N], const Rot &r) {
    if constexpr (N == 4) {
        fadst4<A>(v, r.bit);
    } else {
        typename A::T a[N];
        static_for<0, N>([&](auto J) {
            constexpr int pj = AdstPerm<N>::in[J];
            if constexpr (pj < 0)
                a[J] = A::neg(v[-pj]);
            else
                a[J] = v[pj];
        });
        adst_rot<A, N, 1>(a, r);
        adst_bf<A, N, 2, false>(a, r);
        adst_rot<A, N, 2>(a, r);
        adst_bf<A, N, 4, false>(a, r);
        if constexpr (N == 16) {
            adst_rot<A, N, 3>(a, r);
            adst_bf<A, N, 8, false>(a, r);
        }
        adst_final<A, N>(a, r);
        static_for<0, N>([&](auto K) {
            constexpr int pk = AdstPerm<N>::out[K];
            v[K] = a[pk];
        });
    }
}
template <class A, int N>
TXD_FN void iadst(typename A::T (&v)[N], const Rot &r) {
    if constexpr (N == 4) {
        iadst4<A>(v, r.bit);
    } else {
        typename A::T a[N];
        static_for<0, N>([&](auto K) {
            constexpr int pk = AdstPerm<N>::out[K];
            a[pk] = v[K];
        });
        adst_final<A, N>(a, r);
        if constexpr (N == 16) {
            adst_bf<A, N, 8, true>(a, r);
            adst_rot<A, N, 3>(a, r);
        }
        adst_bf<A, N, 4, true>(a, r);
        adst_rot<A, N, 2>(a, r);
        adst_bf<A, N, 2, true>(a, r);
        adst_rot<A, N, 1>(a, r);
        static_for<0, N>([&](auto J) {
            constexpr int pj = AdstPerm<N>::in[J];
            if constexpr (pj < 0)
                v[-pj] = A::neg(a[J]);
            else
                v[pj] = a[J];
        });
    }
}

// ----------------------------------------------------------------------------------------- identity
template <class A, int N>
TXD_FN void identity(typename A::T (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (N == 4)
            v[i] = A::scale(v[i], 5793, 12);
        else if constexpr (N == 8)
            v[i] = A::times(v[i], 2);
        else if constexpr (N == 16)
            v[i] = A::scale(v[i], 2 * 5793, 12);
        else if constexpr (N == 32)
            v[i] = A::times(v[i], 4);
        else
            v[i] = A::scale(v[i], 4 * 5793, 12);
    }
}

// kind: 0 DCT, 1 ADST (also FLIPADST: the flip is applied by the caller), 3 identity
template <class A, int N>
TXD_FN void fwd1d(typename A::T (&v)[N], int kind, int bit) {
    const Rot r{COSPI.v[bit - 10], bit, 0};
    if (kind == 0) {
        fdct<A, N>(v, r);
    } else if (kind == 3) {
        identity<A, N>(v);
    } else {
        if constexpr (N <= 16)
            fadst<A, N>(v, r);
    }
}
template <class A, int N>
TXD_FN void inv1d(typename A::T (&v)[N], int kind, int clamp) {
    const Rot r{COSPI.v[2], 12, clamp};
    if (kind == 0) {
        idct<A, N>(v, r);
    } else if (kind == 3) {
        identity<A, N>(v);
    } else {
        if constexpr (N <= 16)
            iadst<A, N>(v, r);
    }
}

}  // namespace txd
}  // namespace svthip
