/*
 * svt_hip_bind_txt.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 3a of INTEGRATION.md): the
 * transform-type search of mode decision (tx_type_search, Source/Lib/Codec/product_coding_loop.c:4459-4860) evaluates up to 16
 * transform types of ONE transform block, each starting with svt_aom_estimate_transform on the same residual — a natural batch.
 * The patch computes the set of types the loop can reach (its static filters) in front of the loop and calls
 *     cache = svt_hip_bind_txt_prepare(residual, stride, tx_size, bit_depth, pf_shape, type_mask);
 * which runs the forward transforms of all of them in ONE svt_hip_txfm_quant_batch call (SVT_HIP_TX_FWD, SVT_HIP_QUANT_NONE: the
 * quantiser of the loop is svt_aom_quantize_inv_quantize with its RDOQ, which stays where it is); inside the loop
 *     if (svt_hip_bind_txt_take(cache, tx_type, coeff, &three_quad_energy)) <the reference's own svt_aom_estimate_transform call>;
 * hands the coefficients over.  The loop's data-dependent skips (rate-cost threshold, SATD early exit, early group exit) only
 * leave some of the prepared types unused.  Declined (returns NULL / 1, the reference's call runs): sizes with a 64-point side (one
 * type only, and their 64 -> 32 repack leaves stale rows behind that a later SATD would read), ONLY_DC shape, the 32x32 one-dimensional
 * types the reference computes with a C function outside the RTCD table (transforms.c:3065-3069).
 * This is the encoder call site of the transform batches (row h); it is NOT fast — one PCIe round trip per transform block — and
 * exists to gate the batched entry point inside the running encoder by bitstream md5.  Active with SVTAV1_HIP_TIERB_TXT=1.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "definitions.h"

#include "svt_hip.h"
#include "svt_hip_txfm.h"
#include "svt_hip_bind.h"
#include "svt_hip_bind_dev.h"

static int32_t (*p_batch)(uint8_t *, const SvtHipTxfmDesc *, SvtHipTxfmResult *, uint32_t, uint32_t, uint32_t, void *);
static int           g_active;
static unsigned long g_blocks, g_transforms, g_taken;

static void report(void) {
    fprintf(stderr, "svt_hip_bind_txt: %lu transform blocks / %lu forward transforms through svt_hip_txfm_quant_batch, %lu of them used by the search\n",
            g_blocks, g_transforms, g_taken);
}

void svt_hip_bind_txt_setup(void *(*sym)(const char *)) {
    p_batch  = (int32_t(*)(uint8_t *, const SvtHipTxfmDesc *, SvtHipTxfmResult *, uint32_t, uint32_t, uint32_t, void *))sym("svt_hip_txfm_quant_batch");
    g_active = hd_env_on("SVTAV1_HIP_TIERB_TXT") && g_hd.ok && p_batch;
    if (g_active)
        atexit(report);
}

#define TXT_MAX 16
typedef struct TxtCache {
    int              n, w, h;
    uint8_t          type[TXT_MAX];
    SvtHipTxfmResult res[TXT_MAX];
    int32_t         *coeff; /* [n][w * h], pinned staging */
    size_t           coeff_cap;
} TxtCache;
static __thread TxtCache t_cache;

void *svt_hip_bind_txt_prepare(const int16_t *residual, uint32_t stride, int tx_size, int bit_depth, int pf_shape, uint32_t type_mask) {
    if (!g_active || tx_size < 0 || tx_size >= TX_SIZES_ALL || pf_shape < 0 || pf_shape > 2 || !type_mask)
        return NULL;
    const int w = tx_size_wide[tx_size], h = tx_size_high[tx_size];
    if (w == 64 || h == 64)
        return NULL;
    if (tx_size == TX_32X32 && (type_mask & ((1u << V_DCT) | (1u << H_DCT) | (1u << V_ADST) | (1u << H_ADST) | (1u << V_FLIPADST) | (1u << H_FLIPADST))))
        return NULL;
    TxtCache *c = &t_cache;
    c->n = 0, c->w = w, c->h = h;
    for (int t = 0; t < TX_TYPES && c->n < TXT_MAX; t++)
        if (type_mask & (1u << t))
            c->type[c->n++] = (uint8_t)t;
    const int    n = c->n;
    const size_t n_res = (size_t)w * h * sizeof(int16_t), n_co = (size_t)w * h * sizeof(int32_t);
    const size_t o_res = 0, o_co = (n_res + 255) & ~(size_t)255, o_desc = o_co + (((size_t)n * n_co + 255) & ~(size_t)255);
    const size_t o_out = o_desc + (((size_t)n * sizeof(SvtHipTxfmDesc) + 255) & ~(size_t)255), total = o_out + n * sizeof(SvtHipTxfmResult) + 256;
    if (c->coeff_cap < (size_t)n * n_co) {
        hd_host_free(c->coeff);
        c->coeff     = (int32_t *)hd_host_alloc((size_t)TXT_MAX * 32 * 32 * sizeof(int32_t));
        c->coeff_cap = c->coeff ? (size_t)TXT_MAX * 32 * 32 * sizeof(int32_t) : 0;
        if (!c->coeff)
            return NULL;
    }
    uint8_t *dev = hd_alloc(total);
    if (!dev)
        return NULL;
    /* residual rows packed (stride w), descriptors, one call */
    int16_t        packed[32 * 32];
    SvtHipTxfmDesc desc[TXT_MAX];
    for (int r = 0; r < h; r++) memcpy(packed + (size_t)r * w, residual + (size_t)r * stride, (size_t)w * sizeof(int16_t));
    memset(desc, 0, sizeof(desc));
    for (int i = 0; i < n; i++) {
        SvtHipTxfmDesc *d = &desc[i];
        d->residual_off = o_res, d->residual_stride = (uint32_t)w;
        d->coeff_off = o_co + (size_t)i * n_co;
        d->qcoeff_off = d->dqcoeff_off = d->pred_off = d->recon_off = d->iscan_off = d->qm_off = d->iqm_off = SVT_HIP_NO_OFFSET;
        d->tx_type = c->type[i], d->shape = (uint8_t)pf_shape, d->bit_depth = (uint8_t)bit_depth, d->quant_mode = SVT_HIP_QUANT_NONE;
        d->flags = SVT_HIP_TX_FWD;
    }
    int rc = hd_upload(dev + o_res, packed, n_res) | hd_upload(dev + o_desc, desc, (size_t)n * sizeof(SvtHipTxfmDesc));
    if (rc == 0)
        rc = p_batch(dev, (const SvtHipTxfmDesc *)(dev + o_desc), (SvtHipTxfmResult *)(dev + o_out), (uint32_t)n, (uint32_t)w, (uint32_t)h, NULL);
    if (rc == 0)
        rc = hd_download(c->coeff, dev + o_co, (size_t)n * n_co) | hd_download(c->res, dev + o_out, (size_t)n * sizeof(SvtHipTxfmResult));
    rc |= hd_sync();
    hd_free(dev);
    if (rc != 0) {
        fprintf(stderr, "svt_hip_bind_txt: a transform block stays on the CPU (%s)\n", hd_error());
        c->n = 0;
        return NULL;
    }
    __atomic_add_fetch(&g_blocks, 1, __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_transforms, (unsigned long)n, __ATOMIC_RELAXED);
    return c;
}

/* 0: coeff / three_quad_energy of `tx_type` filled in from the batch; 1: the caller runs svt_aom_estimate_transform */
int svt_hip_bind_txt_take(void *cache, int tx_type, int32_t *coeff, uint64_t *three_quad_energy) {
    TxtCache *c = (TxtCache *)cache;
    if (!c)
        return 1;
    for (int i = 0; i < c->n; i++)
        if (c->type[i] == tx_type) {
            memcpy(coeff, c->coeff + (size_t)i * c->w * c->h, (size_t)c->w * c->h * sizeof(int32_t));
            *three_quad_energy = c->res[i].three_quad_energy;
            __atomic_add_fetch(&g_taken, 1, __ATOMIC_RELAXED);
            return 0;
        }
    return 1;
}
