// common.hpp — shared host-side helpers of libsvtav1_hip (gfx950 only; no CUDA / multi-backend paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/svt_hip.h"
#include "../../include/svt_hip_me.h"

namespace svthip {

// Thread-local error text returned by svt_hip_last_error().
void set_error(const char *fmt, ...);

// Calling thread's private stream (created lazily) unless the caller passed its own.
hipStream_t resolve_stream(void *stream);

// Grow-only per-thread device + pinned scratch used by the Tier A (host-pointer) entry points.
struct Scratch {
    uint8_t *dev    = nullptr;
    uint8_t *pinned = nullptr;
    size_t   dev_cap = 0, pinned_cap = 0;
    uint8_t *device(size_t bytes);
    uint8_t *host(size_t bytes);
};
Scratch &tls_scratch();

bool ensure_init();

// Small descriptor arrays that a Tier B entry point receives in HOST memory travel to the device through a per-thread
// ring of pinned + device staging slots.  stage_descriptors() copies and returns the device address (nullptr + error
// set on failure); stage_commit() must be called after the kernels that read it have been launched: it records the
// event that guards the slot against reuse.  No allocation happens on the launch path once the slots have grown.
void *stage_descriptors(const void *host, size_t bytes, hipStream_t st);
void  stage_commit(hipStream_t st);

// The HIP runtime loads the code object of a translation unit when the first of its kernels is launched (hundreds of milliseconds for
// the whole library, paid by whichever encoder threads come first: 23 ms per call of the first 35 calls inside the patched encoder).
// Every .hip file registers one empty kernel; svt_hip_init launches them all, so the cost is paid once, at initialisation.
typedef void (*WarmupFn)(hipStream_t);
struct WarmupRegistrar {
    explicit WarmupRegistrar(WarmupFn fn);
};
void run_module_warmups(hipStream_t st);

}  // namespace svthip

#define SVT_HIP_MODULE_WARMUP(tag)                                                                                  \
    namespace {                                                                                                     \
    __global__ void svt_hip_warmup_##tag() {}                                                                       \
    void svt_hip_warmup_launch_##tag(hipStream_t st) { hipLaunchKernelGGL(svt_hip_warmup_##tag, dim3(1), dim3(64), 0, st); } \
    svthip::WarmupRegistrar svt_hip_warmup_registrar_##tag(svt_hip_warmup_launch_##tag);                             \
    }

#define SVT_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            svthip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SVT_HIP_ERR_RUNTIME;                                                            \
        }                                                                                          \
    } while (0)

// ---- Tier A failure handling (SURVEY.md 8b "never abort") ---------------------------------------------------------------
// A Tier A leaf has the reference's signature and cannot report an error.  When a HIP call fails inside one (device lost,
// out of memory ...) it throws TierAError; the exported wrapper (TIER_A_CALL) catches it, and tier_a_fail() then, ONCE per
// process: logs the cause, puts the CPU function pointers that svt_hip_install_rtcd() had replaced back into the encoder's
// RTCD slots and latches the library as broken.  The failing call itself — and any call that was already on its way into a
// leaf on another thread — is completed by the restored CPU function with the same arguments, so the encoder carries on with
// its own kernels and an intact bitstream.  Without an installer-saved CPU pointer (the leaf was called directly, e.g.
// through ctypes) there is nothing to fall back on: the process stops with the error message, as before.
namespace svthip {
struct TierAError {
    char what[256];
};
[[noreturn]] void tier_a_throw(const char *fmt, ...);
bool              tier_a_broken();
void              tier_a_fail(const char *leaf, const char *what);
void             *tier_a_cpu(const char *leaf);  // saved CPU function of <leaf> (never NULL: stops the process otherwise)
bool              tier_a_inject_now();           // test hook (svt_hip_debug_inject_failure)
}  // namespace svthip

#define SVT_HIP_CHECK_FATAL(expr)                                                                            \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess || svthip::tier_a_inject_now())                                                 \
            svthip::tier_a_throw("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Body of every exported Tier A function: NAME = the reference's pointer name, IMPL_EXPR = the call that does the work,
// ARGS = the parenthesised argument list for the CPU function.
#define TIER_A_CALL(NAME, IMPL_EXPR, ARGS)                             \
    do {                                                               \
        typedef decltype(&NAME##_hip) TierAFn_;                        \
        if (__builtin_expect(!svthip::tier_a_broken(), 1)) {           \
            try {                                                      \
                return IMPL_EXPR;                                      \
            } catch (const svthip::TierAError &e_) {                   \
                svthip::tier_a_fail(#NAME, e_.what);                   \
            }                                                          \
        }                                                              \
        return ((TierAFn_)svthip::tier_a_cpu(#NAME))ARGS;              \
    } while (0)
