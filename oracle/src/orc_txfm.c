/*
 * oracle/src/orc_txfm.c — TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the reference's forward / inverse 2-D transforms:
 *   forward  Source/Lib/Codec/transforms.c:2259-2324 (av1_tranform_two_d_core_c) with the 1-D kernels
 *            svt_av1_fdct{4..64}_new (:50-1413), svt_av1_fadst{4,8,16}_new (:1415-1794), fidentity (:2205-2236)
 *   inverse  Source/Lib/Codec/inv_transforms.c:2459-2535 (inv_txfm2d_add_c) with the svt_av1_idct / iadst / iidentity kernels
 *            (:94-2361) and the 64-point zero-extension wrappers (:2567-2686)
 *
 * The reference unrolls every butterfly network stage by stage (fdct64 alone is ~770 lines).  Here the
 * SAME networks are generated from their structure: a DCT of size N is a mirror butterfly, a DCT of size
 * N/2 on the sums and an "odd part" of log2(N)-1 rotation levels on the differences; the ADSTs are
 * rotation / butterfly ladders.  Every value is produced by exactly the same sequence of integer
 * operations (32-bit wrapping products, one rounding per rotation — half_btf, inv_transforms.h:264-285 —
 * and, in the inverse, one clamp per addition), so results are bit-identical; pinned against every
 * size x type x bit-depth through oracle/_ref in tests/test_txfm_oracle.py.
 */
#include "orc_txfm.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants ------------------------------------------------------------------------------- */
static int32_t   COSPI[4][64]; /* bits 10..13: round(cos(pi*j/128) * 2^bit)  (inv_transforms.c:3196) */
static int       g_cos_ready;
/* sinpi: AV1 constants round(sqrt(2)*sin(j*pi/9)*2/3 * 2^bit) with [1]+[2]==[4] (inv_transforms.c:3226) */
static const int32_t SINPI[4][5] = {{0, 330, 621, 836, 951}, {0, 660, 1241, 1672, 1901},
                                    {0, 1321, 2482, 3344, 3803}, {0, 2642, 4964, 6689, 7606}};
#define NEW_SQRT2 5793      /* inv_transforms.h:250 */
#define NEW_INV_SQRT2 2896  /* inv_transforms.h:252 */
#define SQRT2_BITS 12

static void init_cos(void) {
    if (g_cos_ready)
        return;
    for (int b = 0; b < 4; b++)
        for (int j = 0; j < 64; j++) COSPI[b][j] = (int32_t)llround(cos(M_PI * j / 128.0) * (double)(1 << (10 + b)));
    g_cos_ready = 1;
}
const int32_t *orc_cospi(int bit) {
    init_cos();
    return COSPI[bit - 10];
}

static inline int32_t mul32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t add32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t sub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t rshift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }
/* half_btf (inv_transforms.h:264-285): 32-bit wrapping products, 64-bit sum, one rounding */
static inline int32_t btf(int32_t w0, int32_t a, int32_t w1, int32_t b, int bit) {
    const int64_t r = (int64_t)mul32(w0, a) + (int64_t)mul32(w1, b);
    return (int32_t)((r + ((int64_t)1 << (bit - 1))) >> bit);
}
/* clamp_value (inv_transforms.h): clamp to `bit` signed bits; bit <= 0 means no clamp */
static inline int32_t clampv(int32_t v, int bit) {
    if (bit <= 0)
        return v;
    const int64_t hi = ((int64_t)1 << (bit - 1)) - 1, lo = -((int64_t)1 << (bit - 1));
    return (int32_t)(v > hi ? hi : (v < lo ? lo : v));
}
static inline uint32_t brev(uint32_t bits, uint32_t x) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
static inline int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

/* ---- DCT ------------------------------------------------------------------------------------- */
typedef struct {
    const int32_t *c;
    int            bit, clamp; /* clamp bits (inverse only; 0 = forward, no clamp) */
} Rot;

/* reflection (-cS cC / cC cS): its own inverse — used by both directions */
static inline void lvl_f1(int32_t *a, int lo, int hi, int S, int C, const Rot *r) {
    const int32_t x = a[lo], y = a[hi];
    a[lo] = btf(-r->c[S], x, r->c[C], y, r->bit);
    a[hi] = btf(r->c[S], y, r->c[C], x, r->bit);
}
static inline void lvl_f2(int32_t *a, int lo, int hi, int S, int C, const Rot *r) {
    const int32_t x = a[lo], y = a[hi];
    a[lo] = btf(-r->c[C], x, -r->c[S], y, r->bit);
    a[hi] = btf(r->c[C], y, -r->c[S], x, r->bit);
}
/* rotation level l of the odd part [b, b+M) */
static void odd_level(int32_t *a, int b, int M, int l, const Rot *r) {
    if (l == 1) {
        for (int j = M / 4; j < M / 2; j++) lvl_f1(a, b + j, b + M - 1 - j, 32, 32, r);
        return;
    }
    const int G = 1 << (l - 2), gs = (M / 2) / G, q = gs / 4, unit = 64 >> l;
    for (int k = 0; k < G; k++) {
        const int S = unit * (1 + 4 * (int)brev((uint32_t)(l - 2), (uint32_t)k)), C = 64 - S;
        for (int j = q; j < 2 * q; j++) lvl_f1(a, b + k * gs + j, b + M - 1 - (k * gs + j), S, C, r);
        for (int j = 2 * q; j < 3 * q; j++) lvl_f2(a, b + k * gs + j, b + M - 1 - (k * gs + j), S, C, r);
    }
}
/* butterflies in groups of g over [b, b+M): even groups (x+y, x-y), odd groups (y-x, y+x) */
static void odd_bf(int32_t *a, int b, int M, int g, const Rot *r) {
    for (int t = 0; t < M / g; t++)
        for (int i = 0; i < g / 2; i++) {
            const int     lo = b + t * g + i, hi = b + t * g + g - 1 - i;
            const int32_t x = a[lo], y = a[hi];
            if (!(t & 1)) {
                a[lo] = clampv(add32(x, y), r->clamp);
                a[hi] = clampv(sub32(x, y), r->clamp);
            } else {
                a[lo] = clampv(sub32(y, x), r->clamp);
                a[hi] = clampv(add32(y, x), r->clamp);
            }
        }
}
static void odd_out(int32_t *a, int b, int M, int inverse, const Rot *r) {
    const int L = ilog2(M), unit = 64 / (2 * M);
    for (int i = 0; i < M / 2; i++) {
        const int     B = unit * (1 + 4 * (int)brev((uint32_t)(L - 1), (uint32_t)i)), A = 64 - B;
        const int     lo = b + i, hi = b + M - 1 - i;
        const int32_t x = a[lo], y = a[hi];
        if (!inverse) {
            a[lo] = btf(r->c[A], x, r->c[B], y, r->bit);
            a[hi] = btf(r->c[A], y, -r->c[B], x, r->bit);
        } else {
            a[lo] = btf(r->c[A], x, -r->c[B], y, r->bit);
            a[hi] = btf(r->c[B], x, r->c[A], y, r->bit);
        }
    }
}
static void fdct_rec(int32_t *a, int N, const Rot *r) {
    if (N == 2) {
        const int32_t x = a[0], y = a[1];
        a[0] = btf(r->c[32], x, r->c[32], y, r->bit);
        a[1] = btf(-r->c[32], y, r->c[32], x, r->bit);
        return;
    }
    for (int i = 0; i < N / 2; i++) {
        const int32_t x = a[i], y = a[N - 1 - i];
        a[i] = add32(x, y), a[N - 1 - i] = sub32(x, y);
    }
    fdct_rec(a, N / 2, r);
    const int M = N / 2, L = ilog2(M);
    for (int l = 1; l < L; l++) {
        odd_level(a, M, M, l, r);
        odd_bf(a, M, M, M >> l, r);
    }
    odd_out(a, M, M, 0, r);
}
static void idct_rec(int32_t *a, int N, const Rot *r) {
    if (N == 2) {
        const int32_t x = a[0], y = a[1];
        a[0] = btf(r->c[32], x, r->c[32], y, r->bit);
        a[1] = btf(r->c[32], x, -r->c[32], y, r->bit);
        return;
    }
    idct_rec(a, N / 2, r);
    const int M = N / 2, L = ilog2(M);
    odd_out(a, M, M, 1, r);
    for (int l = L - 1; l >= 1; l--) {
        odd_bf(a, M, M, M >> l, r);
        odd_level(a, M, M, l, r);
    }
    for (int i = 0; i < N / 2; i++) {
        const int32_t x = a[i], y = a[N - 1 - i];
        a[i] = clampv(add32(x, y), r->clamp), a[N - 1 - i] = clampv(sub32(x, y), r->clamp);
    }
}
static void fdct(const int32_t *in, int32_t *out, int N, int bit) {
    int32_t   a[64];
    const Rot r = {orc_cospi(bit), bit, 0};
    memcpy(a, in, sizeof(int32_t) * (size_t)N);
    fdct_rec(a, N, &r);
    const uint32_t n = (uint32_t)ilog2(N);
    for (int k = 0; k < N; k++) out[k] = a[brev(n, (uint32_t)k)];
}
static void idct(const int32_t *in, int32_t *out, int N, int bit, int clamp) {
    int32_t        a[64];
    const Rot      r = {orc_cospi(bit), bit, clamp};
    const uint32_t n = (uint32_t)ilog2(N);
    for (int k = 0; k < N; k++) a[k] = in[brev(n, (uint32_t)k)];
    idct_rec(a, N, &r);
    memcpy(out, a, sizeof(int32_t) * (size_t)N);
}

/* ---- ADST ------------------------------------------------------------------------------------ */
/* fadst4 (transforms.c:1415-1501) / iadst4 (inv_transforms.c:722-809): 32-bit wrapping arithmetic */
static void fadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = SINPI[bit - 10];
    const int32_t  x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    const int32_t s0 = mul32(s[1], x0), s1 = mul32(s[4], x0), s2 = mul32(s[2], x1), s3 = mul32(s[1], x1);
    const int32_t s4 = mul32(s[3], x2), s5 = mul32(s[4], x3), s6 = mul32(s[2], x3);
    const int32_t s7 = sub32(add32(x0, x1), x3);
    const int32_t y0 = add32(add32(s0, s2), s5), y1 = mul32(s[3], s7), y2 = add32(sub32(s1, s3), s6), y3 = s4;
    out[0] = rshift64(add32(y0, y3), bit);
    out[1] = rshift64(y1, bit);
    out[2] = rshift64(sub32(y2, y3), bit);
    out[3] = rshift64(add32(sub32(y2, y0), y3), bit);
}
static void iadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = SINPI[bit - 10];
    const int32_t  x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    if (!(x0 | x1 | x2 | x3)) {
        out[0] = out[1] = out[2] = out[3] = 0;
        return;
    }
    int32_t s0 = mul32(s[1], x0), s1 = mul32(s[2], x0), s2 = mul32(s[3], x1), s3 = mul32(s[4], x2);
    const int32_t s4 = mul32(s[1], x2), s5 = mul32(s[2], x3), s6 = mul32(s[4], x3);
    const int32_t s7 = add32(sub32(x0, x2), x3);
    s0 = add32(s0, s3), s1 = sub32(s1, s4), s3 = s2, s2 = mul32(s[3], s7);
    s0 = add32(s0, s5), s1 = sub32(s1, s6);
    out[0] = rshift64(add32(s0, s3), bit);
    out[1] = rshift64(add32(s1, s3), bit);
    out[2] = rshift64(s2, bit);
    out[3] = rshift64(sub32(add32(s0, s1), s3), bit);
}

/* stage-1 permutation of fadst8/16 (transforms.c:1515-1522, 1612-1627): a[j] = sign * in[index] */
static const int8_t ADST_IN8[8]   = {0, -7, -3, 4, -1, 6, 2, -5};
static const int8_t ADST_IN16[16] = {0, -15, -7, 8, -3, 12, 4, -11, -1, 14, 6, -9, 2, -13, -5, 10};
/* last-stage permutation: out[k] = a[index] (transforms.c:1590-1597, 1775-1790) */
static const int8_t ADST_OUT8[8]   = {1, 6, 3, 4, 5, 2, 7, 0};
static const int8_t ADST_OUT16[16] = {1, 14, 3, 12, 5, 10, 7, 8, 9, 6, 11, 4, 13, 2, 15, 0};

static inline void adst_a(int32_t *a, int i, int X, const Rot *r) { /* (cX x + cY y, cY x - cX y) */
    const int32_t x = a[i], y = a[i + 1];
    a[i]     = btf(r->c[X], x, r->c[64 - X], y, r->bit);
    a[i + 1] = btf(r->c[64 - X], x, -r->c[X], y, r->bit);
}
static inline void adst_b(int32_t *a, int i, int P, const Rot *r) { /* (-cP x + cQ y, cQ x + cP y) */
    const int32_t x = a[i], y = a[i + 1];
    a[i]     = btf(-r->c[P], x, r->c[64 - P], y, r->bit);
    a[i + 1] = btf(r->c[64 - P], x, r->c[P], y, r->bit);
}
static void adst_rot(int32_t *a, int N, int t, const Rot *r) {
    const int G = 2 << t; /* group size 2^(t+1); the upper half of each group is rotated pairwise */
    for (int g = 0; g < N; g += G) {
        if (t == 1) {
            adst_a(a, g + 2, 32, r);
            continue;
        }
        const int np = G / 4, unit = 64 >> t;
        for (int p = 0; p < np / 2; p++) adst_a(a, g + G / 2 + 2 * p, unit * (1 + 4 * p), r);
        for (int p = 0; p < np / 2; p++) adst_b(a, g + G / 2 + np + 2 * p, 64 - unit * (1 + 4 * p), r);
    }
}
static void adst_bf(int32_t *a, int N, int span, const Rot *r) {
    for (int g = 0; g < N; g += 2 * span)
        for (int i = 0; i < span; i++) {
            const int32_t x = a[g + i], y = a[g + i + span];
            a[g + i]        = clampv(add32(x, y), r->clamp);
            a[g + i + span] = clampv(sub32(x, y), r->clamp);
        }
}
static void adst_final(int32_t *a, int N, const Rot *r) {
    const int unit = N == 8 ? 16 : 8, first = N == 8 ? 4 : 2;
    for (int j = 0; j < N / 2; j++) adst_a(a, 2 * j, first + unit * j, r);
}
static void fadst(const int32_t *in, int32_t *out, int N, int bit) {
    if (N == 4) {
        fadst4(in, out, bit);
        return;
    }
    int32_t       a[16];
    const Rot     r = {orc_cospi(bit), bit, 0};
    const int8_t *pi = N == 8 ? ADST_IN8 : ADST_IN16, *po = N == 8 ? ADST_OUT8 : ADST_OUT16;
    const int     n = ilog2(N);
    for (int j = 0; j < N; j++) a[j] = pi[j] < 0 ? (int32_t)(0u - (uint32_t)in[-pi[j]]) : in[pi[j]];
    for (int t = 1; t < n; t++) {
        adst_rot(a, N, t, &r);
        adst_bf(a, N, 1 << t, &r);
    }
    adst_final(a, N, &r);
    for (int k = 0; k < N; k++) out[k] = a[po[k]];
}
static void iadst(const int32_t *in, int32_t *out, int N, int bit, int clamp) {
    if (N == 4) {
        iadst4(in, out, bit);
        return;
    }
    int32_t       a[16];
    const Rot     r = {orc_cospi(bit), bit, clamp};
    const int8_t *pi = N == 8 ? ADST_IN8 : ADST_IN16, *po = N == 8 ? ADST_OUT8 : ADST_OUT16;
    const int     n = ilog2(N);
    for (int k = 0; k < N; k++) a[po[k]] = in[k];
    adst_final(a, N, &r);
    for (int t = n - 1; t >= 1; t--) {
        adst_bf(a, N, 1 << t, &r);
        adst_rot(a, N, t, &r);
    }
    for (int j = 0; j < N; j++) {
        if (pi[j] < 0)
            out[-pi[j]] = (int32_t)(0u - (uint32_t)a[j]);
        else
            out[pi[j]] = a[j];
    }
}

/* ---- identity (transforms.c:2205-2236, inv_transforms.c:2331-2361) --------------------------- */
static void identity(const int32_t *in, int32_t *out, int N) {
    for (int i = 0; i < N; i++) {
        switch (N) {
        case 4: out[i] = rshift64((int64_t)in[i] * NEW_SQRT2, SQRT2_BITS); break;
        case 8: out[i] = (int32_t)((int64_t)in[i] * 2); break;
        case 16: out[i] = rshift64((int64_t)in[i] * 2 * NEW_SQRT2, SQRT2_BITS); break;
        case 32: out[i] = (int32_t)((int64_t)in[i] * 4); break;
        default: out[i] = rshift64((int64_t)in[i] * 4 * NEW_SQRT2, SQRT2_BITS); break;
        }
    }
}

/* ---- 2-D configuration ----------------------------------------------------------------------- */
/* 1-D kinds per tx_type (vtx_tab / htx_tab, inv_transforms.h:52-87): 0 DCT, 1 ADST, 2 FLIPADST, 3 IDTX */
static const uint8_t VTX[16] = {0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3};
static const uint8_t HTX[16] = {0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2};
/* indexed [log2(w)-2][log2(h)-2]; 0 = size does not exist */
static const int8_t FWD_SHIFT[5][5][3] = {
    /* w=4  */ {{2, 0, 0}, {2, -1, 0}, {2, -1, 0}, {0, 0, 0}, {0, 0, 0}},
    /* w=8  */ {{2, -1, 0}, {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, 0, 0}},
    /* w=16 */ {{2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, 0}},
    /* w=32 */ {{0, 0, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0}, {0, -2, -2}},
    /* w=64 */ {{0, 0, 0}, {0, 0, 0}, {2, -4, 0}, {2, -4, -2}, {0, -2, -2}}};
static const int8_t FWD_COS_COL[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13},
                                         {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t FWD_COS_ROW[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12},
                                         {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
static const int8_t INV_SHIFT0[5][5]  = {{0, 0, -1, 0, 0}, {0, -1, -1, -2, 0}, {-1, -1, -2, -1, -2},
                                         {0, -2, -1, -2, -1}, {0, 0, -2, -1, -2}};
#define INV_SHIFT1 (-4)
#define INV_COS_BIT 12

static void txfm1d_fwd(int kind, const int32_t *in, int32_t *out, int N, int bit) {
    if (kind == 0)
        fdct(in, out, N, bit);
    else if (kind == 3)
        identity(in, out, N);
    else
        fadst(in, out, N, bit);
}
static void txfm1d_inv(int kind, const int32_t *in, int32_t *out, int N, int bit, int clamp) {
    if (kind == 0)
        idct(in, out, N, bit, clamp);
    else if (kind == 3)
        identity(in, out, N);
    else
        iadst(in, out, N, bit, clamp);
}
/* svt_av1_round_shift_array_c with bit = -shift (transforms.c / inv_transforms.c:2423-2434) */
static void shift_array(int32_t *a, int n, int shift) {
    if (shift == 0)
        return;
    if (shift < 0)
        for (int i = 0; i < n; i++) a[i] = rshift64(a[i], -shift);
    else
        for (int i = 0; i < n; i++) a[i] = (int32_t)((uint32_t)a[i] * (1u << shift));
}

int orc_txfm_valid(int w, int h, int tx_type) {
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    if (wi < 0 || wi > 4 || hi < 0 || hi > 4 || (1 << (wi + 2)) != w || (1 << (hi + 2)) != h || tx_type < 0 || tx_type > 15)
        return 0;
    if (FWD_COS_COL[wi][hi] == 0)
        return 0;
    /* ADST exists for 4/8/16 only, identity for every size, DCT for every size */
    if ((VTX[tx_type] == 1 || VTX[tx_type] == 2) && h > 16)
        return 0;
    if ((HTX[tx_type] == 1 || HTX[tx_type] == 2) && w > 16)
        return 0;
    return 1;
}

/* transforms.c:2259-2324.  shape: 0 full, 1 N2 (keep top-left w/2 x h/2), 2 N4 (w/4 x h/4); the pruned
 * variants (transforms.c:5131-5354, 6698-6918) compute the kept coefficients with the same arithmetic and
 * write zeros elsewhere. */
void orc_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int w, int h, int tx_type, int bd,
                    int shape) {
    (void)bd; /* only feeds the (unused) stage ranges of the reference */
    const int     wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    const int8_t *sh = FWD_SHIFT[wi][hi];
    const int     cos_col = FWD_COS_COL[wi][hi], cos_row = FWD_COS_ROW[wi][hi];
    const int     vk = VTX[tx_type], hk = HTX[tx_type];
    const int     ud = vk == 2, lr = hk == 2;
    const int     rect = (w == 2 * h || h == 2 * w);
    int32_t      *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * h);
    int32_t       tin[64], tout[64];
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = input[(size_t)(ud ? h - 1 - r : r) * stride + c];
        shift_array(tin, h, sh[0]);
        txfm1d_fwd(vk, tin, tout, h, cos_col);
        shift_array(tout, h, sh[1]);
        for (int r = 0; r < h; r++) buf[r * w + (lr ? w - 1 - c : c)] = tout[r];
    }
    for (int r = 0; r < h; r++) {
        txfm1d_fwd(hk, buf + r * w, output + r * w, w, cos_row);
        shift_array(output + r * w, w, sh[2]);
        if (rect)
            for (int c = 0; c < w; c++) output[r * w + c] = rshift64((int64_t)output[r * w + c] * NEW_SQRT2, SQRT2_BITS);
    }
    if (shape) {
        const int kw = w >> shape, kh = h >> shape;
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++)
                if (r >= kh || c >= kw)
                    output[r * w + c] = 0;
    }
    free(buf);
}

/* svt_handle_transform{64x64,64x32,32x64,64x16,16x64}_c (transforms.c:2374-2542): energy of the discarded
 * area, then repack the kept 32-wide area contiguously.  Returns the energy. */
uint64_t orc_handle_transform64(int32_t *output, int w, int h) {
    uint64_t e = 0;
    if (w == 64) { /* right half of the kept rows */
        const int kr = h == 64 ? 32 : h;
        for (int r = 0; r < kr; r++)
            for (int c = 32; c < 64; c++) e += (uint64_t)((int64_t)output[r * 64 + c] * (int64_t)output[r * 64 + c]);
    }
    if (h == 64) /* bottom half, full width */
        for (int r = 32; r < 64; r++)
            for (int c = 0; c < w; c++) e += (uint64_t)((int64_t)output[r * w + c] * (int64_t)output[r * w + c]);
    if (w == 64) {
        const int kr = h == 64 ? 32 : h;
        for (int r = 1; r < kr; r++) memmove(output + r * 32, output + r * 64, 32 * sizeof(int32_t));
    }
    return e;
}

static inline uint16_t clip_pixel_add(uint16_t dest, int64_t trans, int bd) {
    /* highbd_clip_pixel_add / check_range (inv_transforms.c:2400-2421) */
    const int64_t mx = ((int64_t)1 << (7 + bd)) - 1 + ((int64_t)914 << (bd - 7)), mn = -mx - 1;
    trans            = trans > mx ? mx : (trans < mn ? mn : trans);
    const int32_t v  = (int32_t)dest + (int32_t)trans;
    const int32_t hi = (1 << bd) - 1;
    return (uint16_t)(v < 0 ? 0 : (v > hi ? hi : v));
}

/* inv_txfm2d_add_c (inv_transforms.c:2459-2535) incl. the 64-point wrappers: `input` holds
 * min(w,32) x min(h,32) coefficients (row stride min(w,32)); everything beyond is zero. */
void orc_inv_txfm2d_add(const int32_t *input, const uint16_t *pred, int32_t stride_r, uint16_t *recon,
                        int32_t stride_w, int w, int h, int tx_type, int bd) {
    const int wi = ilog2(w) - 2, hi = ilog2(h) - 2;
    const int sh0 = INV_SHIFT0[wi][hi], sh1 = INV_SHIFT1;
    const int vk = VTX[tx_type], hk = HTX[tx_type];
    const int ud = vk == 2, lr = hk == 2;
    const int rect = (w == 2 * h || h == 2 * w);
    const int iw = w > 32 ? 32 : w, ih = h > 32 ? 32 : h;
    const int range_row = bd == 8 ? 16 : (bd == 10 ? 18 : 20), range_col = bd == 12 ? 18 : 16;
    int32_t  *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * h);
    int32_t   tin[64], tout[64];
    for (int r = 0; r < h; r++) {
        for (int c = 0; c < w; c++) {
            const int32_t v = (r < ih && c < iw) ? input[r * iw + c] : 0;
            tin[c]          = rect ? rshift64((int64_t)v * NEW_INV_SQRT2, SQRT2_BITS) : v;
        }
        for (int c = 0; c < w; c++) tin[c] = clampv(tin[c], bd + 8);
        txfm1d_inv(hk, tin, buf + r * w, w, INV_COS_BIT, range_row);
        shift_array(buf + r * w, w, sh0);
    }
    const int col_clamp = bd + 6 > 16 ? bd + 6 : 16;
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = clampv(buf[r * w + (lr ? w - 1 - c : c)], col_clamp);
        txfm1d_inv(vk, tin, tout, h, INV_COS_BIT, range_col);
        shift_array(tout, h, sh1);
        for (int r = 0; r < h; r++)
            recon[(size_t)r * stride_w + c] = clip_pixel_add(pred[(size_t)r * stride_r + c], tout[ud ? h - 1 - r : r], bd);
    }
    free(buf);
}

/* svt_av1_inv_txfm_add_c (inv_transforms.c:3177-3193): the 8-bit entry widens to 16 bits and narrows back */
void orc_inv_txfm2d_add_8bit(const int32_t *input, const uint8_t *pred, int32_t stride_r, uint8_t *recon,
                             int32_t stride_w, int w, int h, int tx_type) {
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * 64 * 64);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) tmp[r * 64 + c] = pred[(size_t)r * stride_r + c];
    orc_inv_txfm2d_add(input, tmp, 64, tmp, 64, w, h, tx_type, 8);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) recon[(size_t)r * stride_w + c] = (uint8_t)tmp[r * 64 + c];
    free(tmp);
}

/* ---- quantizers ------------------------------------------------------------
 * All four are per-coefficient functions + eob = 1 + last scan position with a non-zero level; the
 * trailing-zero pre-scan of the reference (full_loop.c:36-46) only skips work.  qm/iqm may be NULL (flat 32). */
#define QM_BITS 5
#define RPOT(v, n) (((v) + (((1 << (n)) >> 1))) >> (n)) /* ROUND_POWER_OF_TWO */
static inline int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* svt_aom_quantize_b_c_ii (full_loop.c:25-75) */
void orc_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                    const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr,
                    const int16_t *scan, const uint8_t *qm, const uint8_t *iqm, int log_scale) {
    const int32_t zbins[2] = {RPOT(zbin[0], log_scale), RPOT(zbin[1], log_scale)};
    intptr_t      eob      = -1;
    memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
    memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
    for (intptr_t i = 0; i < n; i++) {
        const int32_t rc = scan[i], c = coeff[rc], sign = c < 0 ? -1 : 0;
        const int32_t abs_c = (c ^ sign) - sign;
        const int32_t wt    = qm ? qm[rc] : (1 << QM_BITS);
        if (mul32(abs_c, wt) >= (zbins[rc != 0] << QM_BITS)) {
            int32_t t0 = add32(abs_c, RPOT(round[rc != 0], log_scale));
            int64_t tmp = t0 < INT16_MIN ? INT16_MIN : (t0 > INT16_MAX ? INT16_MAX : t0); /* clamp() is an int function */
            tmp *= wt;
            const int32_t t32 = (int32_t)(((((tmp * quant[rc != 0]) >> 16) + tmp) * quant_shift[rc != 0]) >>
                                          (16 - log_scale + QM_BITS));
            qcoeff[rc]        = (t32 ^ sign) - sign;
            const int32_t iwt = iqm ? iqm[rc] : (1 << QM_BITS);
            const int32_t dq  = (dequant[rc != 0] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
            const int32_t adq = mul32(t32, dq) >> log_scale;
            dqcoeff[rc]       = (adq ^ sign) - sign;
            if (t32)
                eob = i;
        }
    }
    *eob_ptr = (uint16_t)(eob + 1);
}

/* svt_aom_highbd_quantize_b_c (full_loop.c:145-194) */
void orc_highbd_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                           const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                           const int16_t *dequant, uint16_t *eob_ptr, const int16_t *scan, const uint8_t *qm,
                           const uint8_t *iqm, int log_scale) {
    const int32_t zbins[2] = {RPOT(zbin[0], log_scale), RPOT(zbin[1], log_scale)};
    intptr_t      eob      = -1;
    memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
    memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
    for (intptr_t i = 0; i < n; i++) {
        const int32_t rc = scan[i], c = coeff[rc], sign = c < 0 ? -1 : 0;
        const int32_t wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
        const int32_t cw = mul32(c, wt);
        if (!(cw >= (zbins[rc != 0] * (1 << QM_BITS)) || cw <= (-zbins[rc != 0] * (1 << QM_BITS))))
            continue;
        const int32_t abs_c = (c ^ sign) - sign;
        const int64_t tmp1  = (int64_t)abs_c + RPOT(round[rc != 0], log_scale);
        const int64_t tmpw  = tmp1 * wt;
        const int64_t tmp2  = ((tmpw * quant[rc != 0]) >> 16) + tmpw;
        const int32_t aq    = (int32_t)((tmp2 * quant_shift[rc != 0]) >> (16 - log_scale + QM_BITS));
        qcoeff[rc]          = (aq ^ sign) - sign;
        const int32_t dq    = (dequant[rc != 0] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
        const int32_t adq   = mul32(aq, dq) >> log_scale;
        dqcoeff[rc]         = (adq ^ sign) - sign;
        if (aq)
            eob = i;
    }
    *eob_ptr = (uint16_t)(eob + 1);
}

/* quantize_fp_helper_c (full_loop.c:278-338) */
void orc_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round, const int16_t *quant, int32_t *qcoeff,
                     int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr, const int16_t *scan, const uint8_t *qm,
                     const uint8_t *iqm, int log_scale) {
    const int rounding[2] = {RPOT(round[0], log_scale), RPOT(round[1], log_scale)};
    int       eob         = -1;
    memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
    memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
    for (int i = 0; i < n; i++) {
        const int rc = scan[i], c = coeff[rc], sign = c < 0 ? -1 : 0;
        int64_t   abs_c = (c ^ sign) - sign;
        int       t32   = 0;
        if (!qm && !iqm) {
            if ((abs_c << (1 + log_scale)) >= (int32_t)dequant[rc != 0]) {
                abs_c = clamp64(abs_c + rounding[rc != 0], INT16_MIN, INT16_MAX);
                t32   = (int)((abs_c * quant[rc != 0]) >> (16 - log_scale));
                if (t32) {
                    qcoeff[rc]        = (t32 ^ sign) - sign;
                    const int32_t adq = mul32(t32, dequant[rc != 0]) >> log_scale;
                    dqcoeff[rc]       = (adq ^ sign) - sign;
                }
            }
        } else {
            const int wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
            const int dq = (dequant[rc != 0] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
            if (abs_c * wt >= (dequant[rc != 0] << (QM_BITS - (1 + log_scale)))) {
                abs_c             = clamp64(abs_c + rounding[rc != 0], INT16_MIN, INT16_MAX);
                t32               = (int)((abs_c * wt * quant[rc != 0]) >> (16 - log_scale + QM_BITS));
                qcoeff[rc]        = (t32 ^ sign) - sign;
                const int32_t adq = mul32(t32, dq) >> log_scale;
                dqcoeff[rc]       = (adq ^ sign) - sign;
            }
        }
        if (t32)
            eob = i;
    }
    *eob_ptr = (uint16_t)(eob + 1);
}

/* highbd_quantize_fp_helper_c (full_loop.c:383-449) */
void orc_highbd_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round, const int16_t *quant, int32_t *qcoeff,
                            int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr, const int16_t *scan,
                            const uint8_t *qm, const uint8_t *iqm, int log_scale) {
    const int shift = 16 - log_scale;
    int       eob   = -1;
    for (int i = 0; i < n; i++) {
        const int rc = scan[i], c = coeff[rc], sign = c < 0 ? -1 : 0, rc01 = rc != 0;
        qcoeff[rc] = dqcoeff[rc] = 0;
        if (qm || iqm) {
            const int     wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
            const int     dq = (dequant[rc01] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
            const int64_t abs_c = (c ^ sign) - sign;
            if (abs_c * wt >= (dequant[rc01] << (QM_BITS - (1 + log_scale)))) {
                const int64_t tmp = abs_c + RPOT(round[rc01], log_scale);
                const int     aq  = (int)((tmp * quant[rc01] * wt) >> (shift + QM_BITS));
                qcoeff[rc]        = (aq ^ sign) - sign;
                const int32_t adq = mul32(aq, dq) >> log_scale;
                dqcoeff[rc]       = (adq ^ sign) - sign;
                if (aq)
                    eob = i;
            }
        } else {
            const int abs_c = (c ^ sign) - sign;
            if ((int32_t)((uint32_t)abs_c << (1 + log_scale)) >= dequant[rc01]) {
                const int64_t tmp = (int64_t)abs_c + RPOT(round[rc01], log_scale);
                const int     aq  = (int)((tmp * quant[rc01]) >> shift);
                qcoeff[rc]        = (aq ^ sign) - sign;
                const int32_t adq = mul32(aq, dequant[rc01]) >> log_scale;
                dqcoeff[rc]       = (adq ^ sign) - sign;
                if (aq)
                    eob = i;
            }
        }
    }
    *eob_ptr = (uint16_t)(eob + 1);
}
