/*
 * svt_hip_bind_dev.h — what the Tier B glue files (svt_hip_bind_{me,tf,tpl,lf,pa}.c) share: the resolved device API of
 * libsvtav1_hip.so, PCIe byte counters, and the DEVICE-RESIDENT PICTURE MIRRORS of SURVEY.md 8b "Ownership" ("the shim owns
 * device mirrors ... per-picture device planes keyed by picture number").
 *
 * A mirror is the device copy of one host buffer of the encoder (a plane of an EbPictureBufferDesc, a gathered mode-info
 * grid ...), identified by the host address and a TAG the caller derives from what the buffer holds (picture number, which
 * processing stage wrote it last).  hd_mirror_get() uploads only when the cache has no copy with that tag; a source picture
 * that used to cross PCIe once as ME source, up to five times as ME reference, once per temporal-filter window it is part of
 * and again for the TPL dispenser now crosses once.  Entries are dropped at the reference's own hand-over points (the hooks
 * call hd_mirror_drop when a stage is about to rewrite a buffer on the CPU) and by LRU above a byte budget
 * (SVTAV1_HIP_MIRROR_MB, default 6144).  SVTAV1_HIP_MIRROR_VERIFY=1 (the tests set it) downloads every hit and compares it
 * with the host buffer: a stale mirror is reported ("STALE mirror") and replaced — the tests assert the message never appears.
 */
#ifndef SVT_HIP_BIND_DEV_H
#define SVT_HIP_BIND_DEV_H

#include <stddef.h>
#include <stdint.h>

typedef struct HipDev {
    int32_t (*malloc_)(void **, size_t);
    int32_t (*free_)(void *);
    int32_t (*upload)(void *, const void *, size_t, void *);
    int32_t (*download)(void *, const void *, size_t, void *);
    int32_t (*memset_)(void *, int32_t, size_t, void *);
    int32_t (*sync)(void *);
    const char *(*last_error)(void);
    int ok; /* every pointer above resolved */
} HipDev;
extern HipDev g_hd;

void        svt_hip_bind_dev_setup(void *(*sym)(const char *));
const char *hd_error(void);
int         hd_env_on(const char *name); /* getenv(name) is a non-zero number */

/* counted transfers (NULL stream = the calling thread's private stream; hd_sync waits for it) */
int hd_upload(void *d, const void *h, size_t n);
int hd_download(void *h, const void *d, size_t n);
int hd_sync(void);
/* device scratch from a recycling pool (no hipMalloc / hipFree on the hooks' paths once the sizes of a sequence have been seen) */
uint8_t *hd_alloc(size_t n);
void     hd_free(void *d);
/* page-locked host staging from a recycling pool (falls back to malloc when the library has no svt_hip_host_alloc) */
void *hd_host_alloc(size_t n);
void  hd_host_free(void *h);

/* ---- mirrors ---------------------------------------------------------------------------------------------------------- */
/* content tags: (picture number << 8) | stage.  A buffer's stage changes whenever somebody rewrites it. */
enum {
    HD_ST_SOURCE    = 1,  /* input picture planes as the picture-analysis stage sees them (before the temporal filter) */
    HD_ST_FILTERED  = 2,  /* after produce_temporally_filtered_pic (or pictures that are never filtered, once ME sees them) */
    HD_ST_RECON     = 3,  /* reconstruction after EncDec, before deblocking */
    HD_ST_DEBLOCKED = 4,
    HD_ST_CDEF      = 5,
    HD_ST_RESTORED  = 6,
    HD_ST_TPL_RECON = 7,  /* mc_flow_rec_picture_buffer of a picture, after its dispenser */
    HD_ST_MI_LF     = 8,  /* gathered SvtHipLfMi grid of a picture */
    HD_ST_MI_SKIP   = 9,  /* gathered 8x8 skip bitmap (CDEF) */
    HD_ST_SOURCE16  = 10, /* pcs->input_frame16bit as the CDEF stage sees it */
    HD_ST_CDEF_EXT  = 11, /* reconstruction after CDEF with the borders the restoration search extended (restoration_pick.c:1511) */
    HD_ST_SOURCE16_LR = 12, /* pcs->input_frame16bit after set_unscaled_input_16bit (cdef_process.c:418) */
};
#define HD_TAG(picture_number, stage) (((uint64_t)(picture_number) << 8) | (uint64_t)(stage))

/* Device copy of host[0 .. bytes) whose content is `tag`; uploaded if the cache holds none.  The entry is PINNED (never evicted,
 * never dropped under the caller) until hd_mirror_unpin(host).  NULL on failure (hd_error()). */
uint8_t *hd_mirror_get(const void *host, size_t bytes, uint64_t tag);
/* A device buffer of `bytes` bytes the caller will fill on the device and that then holds what host will hold under `tag`
 * (the caller downloads it into host, or knows host already equals it).  Replaces any other entry of `host`.  Pinned. */
uint8_t *hd_mirror_new(const void *host, size_t bytes, uint64_t tag);
void     hd_mirror_unpin(const void *host);
/* The host buffer is about to change (or was released): forget its mirror (deferred until unpinned). */
void hd_mirror_drop(const void *host);
/* Re-tag: the entry of `host` (if any, with tag `from`) now describes content `to` — the caller knows both are the same bytes. */
void hd_mirror_retag(const void *host, uint64_t from, uint64_t to);

/* ---- "the first caller computes the picture, the others wait" ----------------------------------------------------------
 * The reference's kernels are called per segment / block from several threads; a whole-picture entry point runs once.  An
 * entry is identified by (owner, key); `total` calls are expected per entry, after which it is recycled.  Never full: entries
 * are allocated on demand. */
typedef struct HdOnce HdOnce;
typedef struct HdOnceTable {
    HdOnce *head;
} HdOnceTable;
/* Returns the entry; *first = 1 for exactly one caller, which must call hd_once_done(entry, ok, payload) when it has finished;
 * every other caller blocks until then.  hd_once_result gives ok / payload; hd_once_release counts the caller out and frees
 * the payload (with free_payload) after the last one. */
HdOnce *hd_once_enter(HdOnceTable *t, const void *owner, uint64_t key, uint32_t total, int *first);
void    hd_once_done(HdOnce *e, int ok, void *payload);
int     hd_once_ok(const HdOnce *e);
void   *hd_once_payload(const HdOnce *e);
void    hd_once_release(HdOnceTable *t, HdOnce *e, void (*free_payload)(void *));

/* wall-clock spent inside each hook (summed over threads and calls; printed at exit): hd_timer_add(name, hd_now_ns() - t0) */
uint64_t hd_now_ns(void);
void     hd_timer_add(const char *name, uint64_t ns);

/* statistics line of the glue at exit (svt_hip_bind_dev.c prints the PCIe totals and the mirror hit rate) */
void hd_count_picture(void);

#endif
