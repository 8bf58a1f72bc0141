/*
 * svt_hip.h — common part of the C-ABI of libsvtav1_hip (MI355X / gfx950 hot path for SVT-AV1).
 *
 * The library is a drop-in for the encoder's per-superblock DSP hot path.  It replaces
 * entries of the reference's process-global function-pointer tables ("RTCD"):
 *   /root/reference/Source/Lib/Codec/aom_dsp_rtcd.h:24-29   (RTCD_EXTERN pointers, encoder only)
 *   /root/reference/Source/Lib/Codec/common_dsp_rtcd.h      (shared pointers)
 * which are filled once from svt_av1_enc_init (Source/Lib/Globals/enc_handle.c:1475-1476).
 *
 * Two tiers:
 *   Tier A  *_hip functions with EXACTLY the RTCD signature of the pointer they replace.
 *           Host pointers in, host pointers out; one call = one block.  Used for parity and
 *           as the literal drop-in (svt_hip_install_rtcd).
 *   Tier B  svt_hip_*_frame / *_batch functions working on DEVICE-resident pictures; one call =
 *           all blocks of one or more pictures.  Called from the kernel-process loops
 *           (me_process.c:174-290, pic_analysis_process.c:2126,2137, ...) for throughput.
 *
 * Plain C, no torch / C++ types.  All functions are thread-safe and re-entrant (the reference
 * calls the pointers concurrently from many worker threads, enc_handle.c:2283-2331).
 */
#ifndef SVT_HIP_H
#define SVT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVT_HIP_API __attribute__((visibility("default")))

/* Error convention mirrors EbErrorType (Source/API/EbSvtAv1.h:126-132): 0 = EB_ErrorNone. */
typedef enum SvtHipStatus {
    SVT_HIP_OK                = 0,
    SVT_HIP_ERR_NO_DEVICE     = (int32_t)0x80001000, /* == EB_ErrorInsufficientResources */
    SVT_HIP_ERR_BAD_PARAMETER = (int32_t)0x80001005, /* == EB_ErrorBadParameter */
    SVT_HIP_ERR_RUNTIME       = (int32_t)0x80001001  /* == EB_ErrorUndefined: a HIP call failed */
} SvtHipStatus;

/* Library / device life-cycle.  svt_hip_init must succeed before any other call; on failure the
 * caller keeps its CPU pointers installed (SURVEY §8b "Error convention"). */
SVT_HIP_API int32_t     svt_hip_init(int32_t device_ordinal);
SVT_HIP_API void        svt_hip_shutdown(void);
SVT_HIP_API int32_t     svt_hip_device_count(void);
SVT_HIP_API const char *svt_hip_last_error(void);   /* thread-local message of the last failure */
SVT_HIP_API const char *svt_hip_version(void);

/* Device memory + streams for hosts that do not bring their own (the C encoder).  A `stream`
 * argument anywhere in this API is a hipStream_t passed as void*; NULL = the calling thread's
 * private stream owned by the library. */
SVT_HIP_API int32_t svt_hip_malloc(void **dptr, size_t bytes);
SVT_HIP_API int32_t svt_hip_free(void *dptr);
SVT_HIP_API int32_t svt_hip_memset(void *dptr, int value, size_t bytes, void *stream);
SVT_HIP_API int32_t svt_hip_upload(void *dptr, const void *hptr, size_t bytes, void *stream);
SVT_HIP_API int32_t svt_hip_download(void *hptr, const void *dptr, size_t bytes, void *stream);
SVT_HIP_API int32_t svt_hip_upload_2d(void *dptr, size_t dpitch, const void *hptr, size_t hpitch,
                                      size_t width_bytes, size_t height, void *stream);
SVT_HIP_API int32_t svt_hip_download_2d(void *hptr, size_t hpitch, const void *dptr, size_t dpitch,
                                        size_t width_bytes, size_t height, void *stream);
SVT_HIP_API int32_t svt_hip_copy(void *d_dst, const void *d_src, size_t bytes, void *stream); /* device to device */
/* page-locked host memory for staging buffers (transfers from / to it run at the full PCIe rate and truly asynchronously) */
SVT_HIP_API int32_t svt_hip_host_alloc(void **hptr, size_t bytes);
SVT_HIP_API int32_t svt_hip_host_free(void *hptr);
SVT_HIP_API int32_t svt_hip_stream_create(void **stream);
SVT_HIP_API int32_t svt_hip_stream_destroy(void *stream);
SVT_HIP_API int32_t svt_hip_stream_sync(void *stream);

/* ---- installing the Tier A functions into the reference's RTCD dispatch -------------------------------
 * Every Tier A export is named <reference pointer name>_hip and has that pointer's exact signature
 * (aom_dsp_rtcd.h / common_dsp_rtcd.h).  svt_hip_rtcd_lookup("svt_sad_loop_kernel") returns
 * &svt_sad_loop_kernel_hip, or NULL when this library has no replacement for that pointer.  (Two aliases, for the
 * pointers without an svt_ prefix: `downsample_2d`, aom_dsp_rtcd.h:838, resolves to svt_aom_downsample_2d_hip and
 * `sad_16b_kernel`, :861, to svt_aom_sad_16b_kernel_hip.)
 * svt_hip_install_rtcd assigns all bindings it can resolve and reports how many; it installs nothing and
 * returns SVT_HIP_ERR_NO_DEVICE when no gfx950 device can be initialised, so the caller keeps the CPU
 * functions selected by svt_aom_setup_rtcd_internal (aom_dsp_rtcd.c:187).  See INTEGRATION.md. */
typedef struct SvtHipRtcdBinding {
    const char *name; /* the reference's pointer variable, e.g. "svt_av1_fwd_txfm2d_16x16" */
    void      **slot; /* its address, e.g. (void **)&svt_av1_fwd_txfm2d_16x16 */
} SvtHipRtcdBinding;
SVT_HIP_API void   *svt_hip_rtcd_lookup(const char *reference_pointer_name);
SVT_HIP_API int32_t svt_hip_install_rtcd(const SvtHipRtcdBinding *bindings, uint32_t n, uint32_t *n_installed);

/* Tier A leaves have the reference's `void` signatures and cannot report an error: when a HIP call fails inside one, the library
 * (once per process) logs "HIP hot path disabled", puts back the CPU pointers svt_hip_install_rtcd had replaced and completes
 * the call with the host's own function (SURVEY 8b "never abort").  svt_hip_tier_a_failed_over() returns 1 from then on: a
 * host reports it; a parity test that compares "C" with "HIP" MUST assert it is 0, or it may have compared C with C. */
SVT_HIP_API int32_t svt_hip_tier_a_failed_over(void);

/* ---- test hooks ---------------------------------------------------------------------------------------------------------
 * Inert unless the process was started with SVTAV1_HIP_TEST_HOOKS=1 in its environment.  svt_hip_debug_inject_failure(n):
 * the n-th device check inside a Tier A leaf from now on fails (0 = the next one, negative = off).
 * svt_hip_debug_tier_a_broken(reset): returns the fail-over latch (like svt_hip_tier_a_failed_over) and, with reset != 0,
 * clears it so that a test suite can go on (the CPU pointers stay restored until svt_hip_install_rtcd runs again). */
SVT_HIP_API void    svt_hip_debug_inject_failure(int32_t n);
SVT_HIP_API int32_t svt_hip_debug_tier_a_broken(int32_t reset);

/* One padded 8-bit plane (mirror of the luma part of EbPictureBufferDesc,
 * Source/Lib/Codec/pic_buffer_desc.h:34-75).  `buf` points at the first byte of the padded
 * buffer (buffer_y); sample (x,y) of the picture is buf[(org_y+y)*stride + org_x + x]. */
typedef struct SvtHipPlane8 {
    uint8_t *buf;
    uint32_t stride;
    uint16_t org_x, org_y;   /* left / top padding */
    uint16_t width, height;  /* picture size without padding */
} SvtHipPlane8;

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_H */
