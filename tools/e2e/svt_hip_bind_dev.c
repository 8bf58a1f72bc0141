/*
 * svt_hip_bind_dev.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch: the part every Tier B hook
 * shares (see svt_hip_bind_dev.h): resolved device API, PCIe byte counters, device-resident picture mirrors, and the
 * "first caller computes the picture" table.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "svt_hip_bind_dev.h"

HipDev g_hd;

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  g_cv = PTHREAD_COND_INITIALIZER;
static unsigned long long g_up_bytes, g_down_bytes, g_hits, g_misses, g_hit_bytes, g_stale, g_evicted, g_pictures;
static unsigned long long g_pool_bytes, g_pool_allocs, g_pool_reuses;
static size_t             g_budget = (size_t)6144 << 20, g_resident;
static int                g_verify;
static int32_t (*p_host_alloc)(void **, size_t);

#include <time.h>
typedef struct HdTimer {
    const char        *name;
    unsigned long long ns, calls;
} HdTimer;
static HdTimer g_timers[24];
uint64_t hd_now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
void hd_timer_add(const char *name, uint64_t ns) { /* takes g_mu: never call it with g_mu held */
    pthread_mutex_lock(&g_mu);
    for (int i = 0; i < 24; i++) {
        if (!g_timers[i].name)
            g_timers[i].name = name;
        if (g_timers[i].name == name || strcmp(g_timers[i].name, name) == 0) {
            g_timers[i].ns += ns, g_timers[i].calls++;
            break;
        }
    }
    pthread_mutex_unlock(&g_mu);
}

int hd_env_on(const char *name) {
    const char *e = getenv(name);
    return e && atoi(e) != 0;
}
const char *hd_error(void) { return g_hd.last_error ? g_hd.last_error() : "?"; }

static void report(void) {
    fprintf(stderr,
            "svt_hip_bind_dev: PCIe %.1f MB up / %.1f MB down over %llu pictures; mirrors: %llu hits (%.1f MB not uploaded again), %llu uploads, "
            "%llu evicted, %llu STALE; device pool %.1f MB in %llu slabs, %llu block reuses\n",
            g_up_bytes / 1048576.0, g_down_bytes / 1048576.0, g_pictures, g_hits, g_hit_bytes / 1048576.0, g_misses, g_evicted, g_stale,
            g_pool_bytes / 1048576.0, g_pool_allocs, g_pool_reuses);
    for (int i = 0; i < 24 && g_timers[i].name; i++)
        fprintf(stderr, "svt_hip_bind_dev: hook %-22s %6llu calls %9.2f ms total %8.3f ms per call\n", g_timers[i].name, g_timers[i].calls,
                g_timers[i].ns * 1e-6, g_timers[i].calls ? g_timers[i].ns * 1e-6 / g_timers[i].calls : 0.0);
}

void svt_hip_bind_dev_setup(void *(*sym)(const char *)) {
    g_hd.malloc_    = (int32_t(*)(void **, size_t))sym("svt_hip_malloc");
    g_hd.free_      = (int32_t(*)(void *))sym("svt_hip_free");
    g_hd.upload     = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_upload");
    g_hd.download   = (int32_t(*)(void *, const void *, size_t, void *))sym("svt_hip_download");
    g_hd.memset_    = (int32_t(*)(void *, int32_t, size_t, void *))sym("svt_hip_memset");
    g_hd.sync       = (int32_t(*)(void *))sym("svt_hip_stream_sync");
    g_hd.last_error = (const char *(*)(void))sym("svt_hip_last_error");
    p_host_alloc    = (int32_t(*)(void **, size_t))sym("svt_hip_host_alloc");
    g_hd.ok         = g_hd.malloc_ && g_hd.free_ && g_hd.upload && g_hd.download && g_hd.memset_ && g_hd.sync;
    const char *mb  = getenv("SVTAV1_HIP_MIRROR_MB");
    if (mb)
        g_budget = (size_t)strtoull(mb, NULL, 10) << 20; /* 0: no caching at all (every get uploads) */
    g_verify = hd_env_on("SVTAV1_HIP_MIRROR_VERIFY");
    if (g_hd.ok)
        atexit(report);
}

void hd_count_picture(void) { __atomic_add_fetch(&g_pictures, 1, __ATOMIC_RELAXED); }

/* The encoder's buffers are pageable memory: hipMemcpyAsync from them ran at 0.5 GB/s here (2.4 MB of a 1080p luma plane: 4.7 ms, most
 * of the picture-analysis hook).  Through a pinned staging block from the pool (a memcpy at memory speed + a DMA) the same upload takes
 * 0.3 ms.  The staged copy is complete on return (the block goes back to the pool). */
#define HD_STAGE_MIN ((size_t)16 << 10)
int hd_upload(void *d, const void *h, size_t n) {
    __atomic_add_fetch(&g_up_bytes, n, __ATOMIC_RELAXED);
    static int no_staging = -1;
    if (no_staging < 0)
        no_staging = hd_env_on("SVTAV1_HIP_NO_STAGING");
    if (n < HD_STAGE_MIN || no_staging)
        return g_hd.upload(d, h, n, NULL);
    void *st = hd_host_alloc(n);
    if (!st)
        return g_hd.upload(d, h, n, NULL);
    memcpy(st, h, n);
    const int rc = g_hd.upload(d, st, n, NULL) | g_hd.sync(NULL);
    hd_host_free(st);
    return rc;
}
int hd_download(void *h, const void *d, size_t n) {
    __atomic_add_fetch(&g_down_bytes, n, __ATOMIC_RELAXED);
    return g_hd.download(h, d, n, NULL);
}
int hd_sync(void) { return g_hd.sync(NULL); }

/* Device scratch comes from a recycling pool: hipMalloc / hipFree take hundreds of microseconds and hipFree waits for the whole device,
 * which serialises every encoder thread that has a kernel in flight -- with a dozen allocations per hook call that was most of a hook's
 * cost.  Blocks are rounded up to 64 KiB multiples (the sizes repeat from picture to picture) and never go back to the driver. */
typedef struct PoolBlk {
    struct PoolBlk *next;
    uint8_t        *dev;
    size_t          bytes;
    int             busy;
} PoolBlk;
static PoolBlk        *g_pool;
static uint8_t        *g_slab;
static size_t          g_slab_left;
static pthread_mutex_t g_pool_mu = PTHREAD_MUTEX_INITIALIZER;

uint8_t *hd_alloc(size_t n) {
    const size_t want = ((n ? n : 1) + 65535) & ~(size_t)65535;
    pthread_mutex_lock(&g_pool_mu);
    PoolBlk *best = NULL;
    for (PoolBlk *b = g_pool; b; b = b->next)
        if (!b->busy && b->bytes >= want && b->bytes <= want + (want >> 2) && (!best || b->bytes < best->bytes))
            best = b;
    if (best) {
        best->busy = 1, g_pool_reuses++;
        pthread_mutex_unlock(&g_pool_mu);
        return best->dev;
    }
    /* a new block: carved from a slab (one hipMalloc per 128 MiB: a hipMalloc per picture plane cost ~8 ms each at start-up) */
    PoolBlk *b = (PoolBlk *)calloc(1, sizeof(*b));
    if (!b) {
        pthread_mutex_unlock(&g_pool_mu);
        return NULL;
    }
    if (g_slab_left < want) {
        const size_t slab = want > ((size_t)128 << 20) ? want : ((size_t)128 << 20);
        void        *p    = NULL;
        if (g_hd.malloc_(&p, slab) != 0) {
            pthread_mutex_unlock(&g_pool_mu);
            free(b);
            return NULL;
        }
        g_slab = (uint8_t *)p, g_slab_left = slab, g_pool_bytes += slab, g_pool_allocs++;
    }
    b->dev = g_slab, b->bytes = want, b->busy = 1;
    g_slab += want, g_slab_left -= want;
    b->next = g_pool, g_pool = b;
    pthread_mutex_unlock(&g_pool_mu);
    return b->dev;
}
void hd_free(void *d) {
    if (!d)
        return;
    pthread_mutex_lock(&g_pool_mu);
    for (PoolBlk *b = g_pool; b; b = b->next)
        if (b->dev == (uint8_t *)d) {
            b->busy = 0;
            break;
        }
    pthread_mutex_unlock(&g_pool_mu);
}

typedef struct HostBlk {
    struct HostBlk *next;
    void           *p;
    size_t          bytes;
    int             busy, pinned;
} HostBlk;
static HostBlk *g_hpool;

void *hd_host_alloc(size_t n) {
    const size_t want = ((n ? n : 1) + 65535) & ~(size_t)65535;
    pthread_mutex_lock(&g_pool_mu);
    HostBlk *best = NULL;
    for (HostBlk *b = g_hpool; b; b = b->next)
        if (!b->busy && b->bytes >= want && b->bytes <= want + (want >> 2) && (!best || b->bytes < best->bytes))
            best = b;
    if (best) {
        best->busy = 1;
        pthread_mutex_unlock(&g_pool_mu);
        return best->p;
    }
    pthread_mutex_unlock(&g_pool_mu);
    HostBlk *b = (HostBlk *)calloc(1, sizeof(*b));
    if (!b)
        return NULL;
    if (p_host_alloc && p_host_alloc(&b->p, want) == 0)
        b->pinned = 1;
    else
        b->p = malloc(want);
    if (!b->p) {
        free(b);
        return NULL;
    }
    b->bytes = want, b->busy = 1;
    pthread_mutex_lock(&g_pool_mu);
    b->next = g_hpool, g_hpool = b;
    pthread_mutex_unlock(&g_pool_mu);
    return b->p;
}
void hd_host_free(void *h) {
    if (!h)
        return;
    pthread_mutex_lock(&g_pool_mu);
    for (HostBlk *b = g_hpool; b; b = b->next)
        if (b->p == h) {
            b->busy = 0;
            break;
        }
    pthread_mutex_unlock(&g_pool_mu);
}

/* ---- mirrors ---------------------------------------------------------------------------------------------------------- */
typedef struct Mirror {
    struct Mirror *next;
    const void    *host;
    size_t         bytes;
    uint64_t       tag, last_use;
    uint8_t       *dev;
    int            pins, loading, doomed;
} Mirror;
static Mirror  *g_mirrors;
static uint64_t g_clock;

static Mirror *find(const void *host) {
    for (Mirror *m = g_mirrors; m; m = m->next)
        if (m->host == host)
            return m;
    return NULL;
}
static void unlink_free(Mirror *m) { /* g_mu held, m unpinned */
    for (Mirror **pp = &g_mirrors; *pp; pp = &(*pp)->next)
        if (*pp == m) {
            *pp = m->next;
            break;
        }
    g_resident -= m->bytes;
    hd_free(m->dev);
    free(m);
}
static void make_room(size_t need) { /* g_mu held */
    while (g_resident + need > g_budget) {
        Mirror *lru = NULL;
        for (Mirror *m = g_mirrors; m; m = m->next)
            if (!m->pins && !m->loading && (!lru || m->last_use < lru->last_use))
                lru = m;
        if (!lru)
            return; /* everything resident is in use: go over budget rather than fail */
        unlink_free(lru), g_evicted++;
    }
}
static Mirror *insert(const void *host, size_t bytes, uint64_t tag) { /* g_mu held; returns a pinned, loading entry */
    make_room(bytes);
    Mirror *m = (Mirror *)calloc(1, sizeof(*m));
    if (!m)
        return NULL;
    m->dev = hd_alloc(bytes + 256);
    if (!m->dev) {
        free(m);
        return NULL;
    }
    m->host = host, m->bytes = bytes, m->tag = tag, m->pins = 1, m->loading = 1, m->last_use = ++g_clock;
    m->next = g_mirrors, g_mirrors = m;
    g_resident += bytes;
    return m;
}

uint8_t *hd_mirror_get(const void *host, size_t bytes, uint64_t tag) {
    pthread_mutex_lock(&g_mu);
    for (;;) {
        Mirror *m = find(host);
        if (m && m->loading) { /* somebody is uploading / producing it right now */
            pthread_cond_wait(&g_cv, &g_mu);
            continue;
        }
        if (m && !m->doomed && m->tag == tag && m->bytes == bytes) {
            m->pins++, m->last_use = ++g_clock;
            int stale = 0;
            if (g_verify) {
                pthread_mutex_unlock(&g_mu);
                uint8_t *tmp = (uint8_t *)malloc(bytes);
                if (tmp && g_hd.download(tmp, m->dev, bytes, NULL) == 0 && g_hd.sync(NULL) == 0 && memcmp(tmp, host, bytes) != 0) {
                    size_t k = 0;
                    while (tmp[k] == ((const uint8_t *)host)[k]) k++;
                    fprintf(stderr, "svt_hip_bind_dev: STALE mirror of %p (tag %llx, %zu bytes): first difference at byte %zu\n", host,
                            (unsigned long long)tag, bytes, k);
                    stale = 1;
                }
                free(tmp);
                pthread_mutex_lock(&g_mu);
            }
            if (!stale) {
                g_hits++, g_hit_bytes += bytes;
                pthread_mutex_unlock(&g_mu);
                return m->dev;
            }
            g_stale++;
            m->pins--; /* fall through: replace it */
        }
        if (m) {
            if (m->pins) { /* in use with other content: wait for its users, then replace */
                m->doomed = 1;
                pthread_cond_wait(&g_cv, &g_mu);
                continue;
            }
            unlink_free(m);
        }
        m = insert(host, bytes, tag);
        if (!m) {
            pthread_mutex_unlock(&g_mu);
            return NULL;
        }
        g_misses++;
        pthread_mutex_unlock(&g_mu);
        const int rc = hd_upload(m->dev, host, bytes) | hd_sync();
        pthread_mutex_lock(&g_mu);
        m->loading = 0;
        pthread_cond_broadcast(&g_cv);
        if (rc != 0) {
            m->pins = 0;
            unlink_free(m);
            pthread_mutex_unlock(&g_mu);
            return NULL;
        }
        pthread_mutex_unlock(&g_mu);
        return m->dev;
    }
}

uint8_t *hd_mirror_new(const void *host, size_t bytes, uint64_t tag) {
    pthread_mutex_lock(&g_mu);
    for (;;) {
        Mirror *m = find(host);
        if (m && (m->loading || m->pins)) {
            m->doomed = 1;
            pthread_cond_wait(&g_cv, &g_mu);
            continue;
        }
        if (m)
            unlink_free(m);
        m = insert(host, bytes, tag);
        if (m)
            m->loading = 0; /* the caller fills it before anybody else can know the tag it will be asked for */
        pthread_mutex_unlock(&g_mu);
        return m ? m->dev : NULL;
    }
}

void hd_mirror_unpin(const void *host) {
    pthread_mutex_lock(&g_mu);
    Mirror *m = find(host);
    if (m && m->pins > 0 && --m->pins == 0) {
        if (m->doomed || g_budget == 0)
            unlink_free(m);
        pthread_cond_broadcast(&g_cv);
    }
    pthread_mutex_unlock(&g_mu);
}

void hd_mirror_drop(const void *host) {
    pthread_mutex_lock(&g_mu);
    Mirror *m = find(host);
    if (m) {
        if (m->pins || m->loading)
            m->doomed = 1;
        else
            unlink_free(m);
    }
    pthread_mutex_unlock(&g_mu);
}

void hd_mirror_retag(const void *host, uint64_t from, uint64_t to) {
    pthread_mutex_lock(&g_mu);
    Mirror *m = find(host);
    if (m && m->tag == from && !m->doomed)
        m->tag = to;
    pthread_mutex_unlock(&g_mu);
}

/* ---- first caller computes ------------------------------------------------------------------------------------------ */
struct HdOnce {
    struct HdOnce *next;
    const void    *owner;
    uint64_t       key;
    uint32_t       total, seen;
    int            state; /* 1 being computed, 2 done ok, 3 done, not ok */
    void          *payload;
};

HdOnce *hd_once_enter(HdOnceTable *t, const void *owner, uint64_t key, uint32_t total, int *first) {
    pthread_mutex_lock(&g_mu);
    HdOnce *e = t->head;
    while (e && !(e->owner == owner && e->key == key)) e = e->next;
    *first = 0;
    if (!e) {
        e = (HdOnce *)calloc(1, sizeof(*e));
        if (!e) {
            pthread_mutex_unlock(&g_mu);
            return NULL;
        }
        e->owner = owner, e->key = key, e->total = total ? total : 1, e->state = 1;
        e->next = t->head, t->head = e;
        *first = 1;
        pthread_mutex_unlock(&g_mu);
        return e;
    }
    while (e->state == 1) pthread_cond_wait(&g_cv, &g_mu);
    pthread_mutex_unlock(&g_mu);
    return e;
}
void hd_once_done(HdOnce *e, int ok, void *payload) {
    pthread_mutex_lock(&g_mu);
    e->payload = payload, e->state = ok ? 2 : 3;
    pthread_cond_broadcast(&g_cv);
    pthread_mutex_unlock(&g_mu);
}
int   hd_once_ok(const HdOnce *e) { return e->state == 2; }
void *hd_once_payload(const HdOnce *e) { return e->payload; }
void  hd_once_release(HdOnceTable *t, HdOnce *e, void (*free_payload)(void *)) {
    pthread_mutex_lock(&g_mu);
    if (++e->seen >= e->total) {
        for (HdOnce **pp = &t->head; *pp; pp = &(*pp)->next)
            if (*pp == e) {
                *pp = e->next;
                break;
            }
        pthread_mutex_unlock(&g_mu);
        if (e->payload && free_payload)
            free_payload(e->payload);
        free(e);
        return;
    }
    pthread_mutex_unlock(&g_mu);
}
