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

}  // namespace svthip

#define SVT_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            svthip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SVT_HIP_ERR_RUNTIME;                                                            \
        }                                                                                          \
    } while (0)

// For void Tier A functions: they cannot report; they record the error and abort loudly — a silent
// wrong result would corrupt the bitstream (and there is deliberately no CPU fallback in this library).
#define SVT_HIP_CHECK_FATAL(expr)                                                                  \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "libsvtav1_hip fatal: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                           \
            abort();                                                                               \
        }                                                                                          \
    } while (0)
