// runtime.cpp — device life-cycle, streams, memory and error plumbing of libsvtav1_hip.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"

namespace svthip {

static thread_local char tls_err[512] = "";
static std::atomic<int>  g_device{-1};   // the device every entry point works on; -1 until svt_hip_init succeeds
static std::atomic<int>  g_sticky{-1};   // first device ever bound: per-thread streams / scratch / once-uploaded tables live there
static std::mutex        g_mutex;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tls_err, sizeof(tls_err), fmt, ap);
    va_end(ap);
}

bool ensure_init() {
    if (g_device.load() >= 0) {
        // each host thread must select the device once
        static thread_local bool bound = false;
        if (!bound) {
            if (hipSetDevice(g_device.load()) != hipSuccess)
                return false;
            bound = true;
        }
        return true;
    }
    // not initialised (or shut down): never pick a device silently — a rank that forgot svt_hip_init(local_rank) would
    // otherwise run on GPU 0
    set_error("library not initialised: call svt_hip_init(device_ordinal) first");
    return false;
}

// The "calling thread's private stream" (stream == NULL in the API): a host such as the encoder has dozens of worker threads, and
// creating a HIP stream costs milliseconds (a hardware queue) -- 25 ms per first call of a thread inside the patched encoder.  Threads
// draw from a small pool instead: thread k uses stream k mod POOL.  Two threads that share a stream only wait for each other's work in
// svt_hip_stream_sync; order inside one thread is kept, which is all the API promises.
#ifndef SVT_HIP_STREAM_POOL
#define SVT_HIP_STREAM_POOL 8
#endif
namespace {
constexpr int     STREAM_POOL = SVT_HIP_STREAM_POOL;
hipStream_t       g_pool_streams[STREAM_POOL];
std::mutex        g_pool_mutex;
std::atomic<int>  g_next_thread{0};
}  // namespace

namespace {
WarmupFn g_warmups[64];
int      g_n_warmups = 0;
}  // namespace
WarmupRegistrar::WarmupRegistrar(WarmupFn fn) {
    if (g_n_warmups < 64)
        g_warmups[g_n_warmups++] = fn;
}
void run_module_warmups(hipStream_t st) {
    for (int i = 0; i < g_n_warmups; i++) g_warmups[i](st);
}

hipStream_t resolve_stream(void *stream) {
    if (stream)
        return (hipStream_t)stream;
    static thread_local int slot = -1;
    if (slot < 0)
        slot = g_next_thread.fetch_add(1) % STREAM_POOL;
    hipStream_t s = g_pool_streams[slot];
    if (!s) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        s = g_pool_streams[slot];
        if (!s) {
            // streams belong to the device that is current when they are created: bind this thread first
            (void)ensure_init();
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
                set_error("cannot create a stream for the calling thread");
                return nullptr;  // the legacy default stream: callers carry on, correct but serialised
            }
            g_pool_streams[slot] = s;
        }
    }
    return s;
}

namespace {
struct StageSlot {
    uint8_t   *dev = nullptr, *pinned = nullptr;
    size_t     cap = 0;
    hipEvent_t done = nullptr;
    bool       pending = false;
};
struct StageRing {
    StageSlot slot[4];
    int       next = 0, last = -1;
};
thread_local StageRing g_ring;
}  // namespace

void *stage_descriptors(const void *host, size_t bytes, hipStream_t st) {
    StageRing &r = g_ring;
    StageSlot &s = r.slot[r.next];
    auto       fail = [&](hipError_t e, const char *what) {
        set_error("stage_descriptors: %s: %s", what, hipGetErrorString(e));
        return (void *)nullptr;
    };
    hipError_t e;
    if (s.pending) {
        if ((e = hipEventSynchronize(s.done)) != hipSuccess)
            return fail(e, "hipEventSynchronize");
        s.pending = false;
    }
    if (!s.done && (e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess)
        return fail(e, "hipEventCreate");
    if (bytes > s.cap) {
        if (s.dev)
            (void)hipFree(s.dev);
        if (s.pinned)
            (void)hipHostFree(s.pinned);
        s.dev = s.pinned = nullptr, s.cap = 0;
        const size_t cap = bytes < 65536 ? 65536 : bytes * 2;
        if ((e = hipMalloc((void **)&s.dev, cap)) != hipSuccess)
            return fail(e, "hipMalloc");
        if ((e = hipHostMalloc((void **)&s.pinned, cap, hipHostMallocDefault)) != hipSuccess)
            return fail(e, "hipHostMalloc");
        s.cap = cap;
    }
    memcpy(s.pinned, host, bytes);
    if ((e = hipMemcpyAsync(s.dev, s.pinned, bytes, hipMemcpyHostToDevice, st)) != hipSuccess)
        return fail(e, "hipMemcpyAsync");
    r.last = r.next;
    r.next = (r.next + 1) % 4;
    return s.dev;
}
void stage_commit(hipStream_t st) {
    StageRing &r = g_ring;
    if (r.last < 0)
        return;
    StageSlot &s = r.slot[r.last];
    if (hipEventRecord(s.done, st) == hipSuccess)
        s.pending = true;
    r.last = -1;
}

// ---- Tier A failure handling (common.hpp) ----
namespace {
struct SavedSlot {
    char   name[96];
    void **slot;
    void  *cpu_fn;
};
std::mutex        g_slots_mutex;
SavedSlot         g_slots[512];
int               g_n_slots = 0;
std::atomic<bool> g_tier_a_broken{false};
std::atomic<int>  g_inject{-1};  // test hook: the n-th SVT_HIP_CHECK_FATAL from now fails (svt_hip_debug_inject_failure)
}  // namespace

[[noreturn]] void tier_a_throw(const char *fmt, ...) {
    TierAError e;
    va_list    ap;
    va_start(ap, fmt);
    vsnprintf(e.what, sizeof(e.what), fmt, ap);
    va_end(ap);
    throw e;
}
bool tier_a_broken() { return g_tier_a_broken.load(std::memory_order_relaxed); }
void tier_a_fail(const char *leaf, const char *what) {
    std::lock_guard<std::mutex> lk(g_slots_mutex);
    if (g_tier_a_broken.exchange(true))
        return;
    int restored = 0;
    for (int i = 0; i < g_n_slots; i++)
        if (g_slots[i].slot && g_slots[i].cpu_fn)
            *g_slots[i].slot = g_slots[i].cpu_fn, restored++;
    fprintf(stderr, "libsvtav1_hip: %s: %s -- HIP hot path disabled, %d RTCD pointers restored to the CPU kernels\n", leaf, what,
            restored);
}
void *tier_a_cpu(const char *leaf) {
    {
        std::lock_guard<std::mutex> lk(g_slots_mutex);
        for (int i = 0; i < g_n_slots; i++)
            if (strcmp(g_slots[i].name, leaf) == 0 && g_slots[i].cpu_fn)
                return g_slots[i].cpu_fn;
    }
    fprintf(stderr, "libsvtav1_hip fatal: %s_hip failed and no CPU function was saved for it (svt_hip_install_rtcd was not used): %s\n",
            leaf, svt_hip_last_error());
    abort();
}
static void remember_slot(const char *stem, void **slot, void *cpu_fn) {
    std::lock_guard<std::mutex> lk(g_slots_mutex);
    for (int i = 0; i < g_n_slots; i++)
        if (strcmp(g_slots[i].name, stem) == 0) {
            g_slots[i].slot = slot, g_slots[i].cpu_fn = cpu_fn;
            return;
        }
    if (g_n_slots < (int)(sizeof(g_slots) / sizeof(g_slots[0]))) {
        snprintf(g_slots[g_n_slots].name, sizeof(g_slots[g_n_slots].name), "%s", stem);
        g_slots[g_n_slots].slot = slot, g_slots[g_n_slots].cpu_fn = cpu_fn;
        g_n_slots++;
    }
}
bool tier_a_inject_now() {
    int v = g_inject.load();
    while (v >= 0) {
        if (g_inject.compare_exchange_weak(v, v - 1))
            return v == 0;
    }
    return false;
}

uint8_t *Scratch::device(size_t bytes) {
    if (bytes > dev_cap) {
        if (dev)
            (void)hipFree(dev);
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
        if (tier_a_inject_now() || hipMalloc((void **)&dev, cap + 256) != hipSuccess) {
            dev = nullptr, dev_cap = 0;
            tier_a_throw("hipMalloc(%zu) for the Tier A scratch buffer failed", cap);
        }
        dev_cap = cap;
    }
    return dev;
}
uint8_t *Scratch::host(size_t bytes) {
    if (bytes > pinned_cap) {
        if (pinned)
            (void)hipHostFree(pinned);
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
        if (hipHostMalloc((void **)&pinned, cap + 256, hipHostMallocDefault) != hipSuccess) {
            pinned = nullptr, pinned_cap = 0;
            tier_a_throw("hipHostMalloc(%zu) for the Tier A scratch buffer failed", cap);
        }
        pinned_cap = cap;
    }
    return pinned;
}
Scratch &tls_scratch() {
    static thread_local Scratch s;
    return s;
}

}  // namespace svthip

using namespace svthip;

extern "C" {

int32_t svt_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int32_t svt_hip_init(int32_t device_ordinal) {
    std::lock_guard<std::mutex> lk(g_mutex);
    int                         n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return SVT_HIP_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        set_error("device ordinal %d out of range (%d devices)", device_ordinal, n);
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (g_sticky.load() >= 0 && g_sticky.load() != device_ordinal) {
        // per-thread streams, scratch buffers and the once-uploaded constant tables stay on the first device: one process
        // = one GPU (the multi-GPU layout is one process per GPU)
        set_error("already bound to device %d: one process drives one GPU", g_sticky.load());
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    SVT_HIP_CHECK(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    SVT_HIP_CHECK(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device_ordinal,
                  prop.gcnArchName);
        return SVT_HIP_ERR_NO_DEVICE;
    }
    const bool first = g_sticky.load() < 0;
    g_device.store(device_ordinal);
    g_sticky.store(device_ordinal);
    if (first) {  // load every translation unit's code object now instead of inside the first calls of the host's worker threads,
                  // and create the stream pool here: a worker thread's first call created its slot's stream under the pool lock, and the
                  // six picture-analysis threads of the encoder, all starting at once, paid 16 ms each for it
        {
            std::lock_guard<std::mutex> pk(g_pool_mutex);
            for (int i = 0; i < STREAM_POOL; i++)
                if (!g_pool_streams[i] && hipStreamCreateWithFlags(&g_pool_streams[i], hipStreamNonBlocking) != hipSuccess)
                    g_pool_streams[i] = nullptr;  // created on demand by resolve_stream
        }
        run_module_warmups(nullptr);
        SVT_HIP_CHECK(hipDeviceSynchronize());
    }
    return SVT_HIP_OK;
}

void svt_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mutex);
    g_device.store(-1);
}

const char *svt_hip_last_error(void) { return tls_err; }
const char *svt_hip_version(void) { return "svtav1-hip 0.1 (gfx950)"; }

int32_t svt_hip_malloc(void **dptr, size_t bytes) {
    if (!dptr)
        return SVT_HIP_ERR_BAD_PARAMETER;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    // +256 B beyond what was asked for: a defensive margin, not part of any contract (the documented contracts — e.g. the
    // plane contract of include/svt_hip_me.h — hold for caller-allocated memory of the exact size)
    SVT_HIP_CHECK(hipMalloc(dptr, bytes + 256));
    return SVT_HIP_OK;
}
int32_t svt_hip_free(void *dptr) {
    SVT_HIP_CHECK(hipFree(dptr));
    return SVT_HIP_OK;
}
int32_t svt_hip_memset(void *dptr, int value, size_t bytes, void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemsetAsync(dptr, value, bytes, resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_upload(void *dptr, const void *hptr, size_t bytes, void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_download(void *hptr, const void *dptr, size_t bytes, void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_upload_2d(void *dptr, size_t dpitch, const void *hptr, size_t hpitch, size_t width_bytes,
                          size_t height, void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemcpy2DAsync(dptr, dpitch, hptr, hpitch, width_bytes, height, hipMemcpyHostToDevice,
                                   resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_download_2d(void *hptr, size_t hpitch, const void *dptr, size_t dpitch, size_t width_bytes, size_t height,
                            void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemcpy2DAsync(hptr, hpitch, dptr, dpitch, width_bytes, height, hipMemcpyDeviceToHost, resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_host_alloc(void **hptr, size_t bytes) {
    if (!hptr)
        return SVT_HIP_ERR_BAD_PARAMETER;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SVT_HIP_OK;
}
int32_t svt_hip_host_free(void *hptr) {
    if (hptr)
        SVT_HIP_CHECK(hipHostFree(hptr));
    return SVT_HIP_OK;
}
int32_t svt_hip_copy(void *d_dst, const void *d_src, size_t bytes, void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, resolve_stream(stream)));
    return SVT_HIP_OK;
}
int32_t svt_hip_stream_create(void **stream) {
    if (!stream)
        return SVT_HIP_ERR_BAD_PARAMETER;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    hipStream_t s;
    SVT_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return SVT_HIP_OK;
}
int32_t svt_hip_stream_destroy(void *stream) {
    SVT_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return SVT_HIP_OK;
}
int32_t svt_hip_stream_sync(void *stream) {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipStreamSynchronize(resolve_stream(stream)));
    return SVT_HIP_OK;
}

}  // extern "C"

// ---- RTCD installation: <reference pointer name> -> this library's <name>_hip export, resolved through the
// dynamic symbol table of the library itself so that the list can never drift from what is exported.
#include <dlfcn.h>

extern "C" void *svt_hip_rtcd_lookup(const char *name) {
    // the two RTCD pointers of the path whose variables have no svt_ prefix (aom_dsp_rtcd.h:838, :861)
    if (name && strcmp(name, "downsample_2d") == 0)
        name = "svt_aom_downsample_2d";
    if (name && strcmp(name, "sad_16b_kernel") == 0)  // aom_dsp_rtcd.h:861
        name = "svt_aom_sad_16b_kernel";
    if (!name || strncmp(name, "svt_", 4) != 0 || strncmp(name, "svt_hip_", 8) == 0 || strlen(name) > 200)
        return nullptr;
    static void *self = [] {
        Dl_info info;
        if (!dladdr((void *)&svt_hip_rtcd_lookup, &info) || !info.dli_fname)
            return (void *)nullptr;
        return dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }();
    if (!self)
        return nullptr;
    char sym[256];
    snprintf(sym, sizeof(sym), "%s_hip", name);
    return dlsym(self, sym);
}

// Whether a Tier A leaf has failed over (public, read-only: a host reports it; tests assert it stays 0).
extern "C" int32_t svt_hip_tier_a_failed_over(void) { return svthip::g_tier_a_broken.load() ? 1 : 0; }

// Test hooks (declared in svt_hip.h under "test hooks"): the n-th Tier A device check from now on reports a failure (n = 0: the
// next one; n < 0: off); and a way to clear the latch again between tests.  Inert unless the process was started with
// SVTAV1_HIP_TEST_HOOKS=1 -- a production host cannot trip them by accident (reading the latch always works).
static bool test_hooks_enabled() {
    static const bool on = [] {
        const char *e = getenv("SVTAV1_HIP_TEST_HOOKS");
        return e && e[0] == '1';
    }();
    return on;
}
extern "C" void svt_hip_debug_inject_failure(int32_t n) {
    if (test_hooks_enabled())
        svthip::g_inject.store(n);
}
extern "C" int32_t svt_hip_debug_tier_a_broken(int32_t reset) {
    const int32_t was = svthip::g_tier_a_broken.load() ? 1 : 0;
    if (reset && test_hooks_enabled())
        svthip::g_tier_a_broken.store(false);
    return was;
}

extern "C" int32_t svt_hip_install_rtcd(const SvtHipRtcdBinding *b, uint32_t n, uint32_t *n_installed) {
    if (n_installed)
        *n_installed = 0;
    if (!b && n) {
        svthip::set_error("svt_hip_install_rtcd: null binding table");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!svthip::ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;  // nothing installed: the caller keeps its CPU pointers
    if (svthip::tier_a_broken()) {
        svthip::set_error("svt_hip_install_rtcd: the HIP path was disabled after a device failure");
        return SVT_HIP_ERR_RUNTIME;
    }
    uint32_t done = 0;
    for (uint32_t i = 0; i < n; i++) {
        void *fn = b[i].slot ? svt_hip_rtcd_lookup(b[i].name) : nullptr;
        if (fn) {
            // the export's stem is what TIER_A_CALL looks the CPU function up by (two pointers have no svt_ prefix)
            const char *stem = strcmp(b[i].name, "downsample_2d") == 0 ? "svt_aom_downsample_2d"
                : (strcmp(b[i].name, "sad_16b_kernel") == 0 ? "svt_aom_sad_16b_kernel" : b[i].name);
            if (*b[i].slot != fn)
                svthip::remember_slot(stem, b[i].slot, *b[i].slot);
            *b[i].slot = fn, done++;
        }
    }
    if (n_installed)
        *n_installed = done;
    return SVT_HIP_OK;
}
