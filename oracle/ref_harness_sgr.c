/*
 * oracle/ref_harness_sgr.c — TEST INFRASTRUCTURE.  search_selfguided_restoration (restoration_pick.c:550-652) and
 * finer_search_pixel_proj_error (:320-411) are `static` in the reference.  To pin the oracle's restatement of that
 * driver against the REAL code, this translation unit compiles the reference's restoration_pick.c in place (by
 * #include from where it lies, nothing is copied) and exposes one entry point; oracle/Makefile then makes every
 * other symbol of this object local (objcopy --keep-global-symbol) so it does not clash with restoration_pick.o of
 * the reference archive, whose RTCD pointers and tables it shares.
 */
#include "restoration_pick.c"

__attribute__((visibility("default"))) int ref_sgr_search_unit(const uint8_t *dat8, int32_t width, int32_t height, int32_t dat_stride,
                                                                const uint8_t *src8, int32_t src_stride, int32_t highbd,
                                                                int32_t bit_depth, int32_t pu_w, int32_t pu_h, int32_t start_ep,
                                                                int32_t end_ep, int32_t ep_inc, int32_t do_refine, int32_t out[3]) {
    int32_t      *rstbuf = (int32_t *)malloc(SGRPROJ_TMPBUF_SIZE);
    int8_t        refs[2] = {-1, -1};
    int32_t       cnt[SGRPROJ_PARAMS] = {0};
    SgFilterCtrls ctrls;
    if (!rstbuf)
        return -1;
    memset(&ctrls, 0, sizeof(ctrls));
    ctrls.enabled = 1, ctrls.step_range = 16;
    for (int p = 0; p < PLANE_TYPES; p++)
        ctrls.start_ep[p] = (int8_t)start_ep, ctrls.end_ep[p] = (int8_t)end_ep, ctrls.ep_inc[p] = (int8_t)ep_inc, ctrls.refine[p] = (int8_t)do_refine;
    const SgrprojInfo r = search_selfguided_restoration(
        dat8, width, height, dat_stride, src8, src_stride, highbd, bit_depth, pu_w, pu_h, rstbuf, refs, cnt, &ctrls, 0, 16);
    out[0] = r.ep, out[1] = r.xqd[0], out[2] = r.xqd[1];
    free(rstbuf);
    return 0;
}
