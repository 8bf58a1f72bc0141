/*
 * svt_hip_bind.c — OUR glue compiled into the reference encoder by tools/reference_hip.patch (Step 1 of
 * INTEGRATION.md): the table of the reference's RTCD pointers (aom_dsp_rtcd.h, common_dsp_rtcd.h) that have a HIP
 * leaf, and the installer called from svt_av1_enc_init right behind svt_aom_setup_rtcd_internal
 * (Source/Lib/Globals/enc_handle.c:1475-1476).  The library is loaded with dlopen so that an encoder built with
 * this file still runs (on its C / SIMD kernels) on a machine without ROCm.
 */
#include <dlfcn.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"

#include "svt_hip.h" /* include/ of the svtav1-hip repository */
#include "svt_hip_bind.h"
#include "svt_hip_bind_dev.h"

#define HIP_SLOT(p) {#p, (void **)&p},
static const SvtHipRtcdBinding hip_bindings[] = {
#include "svt_hip_bind_table.inc"
};
#undef HIP_SLOT

static int name_in_list(const char *name, const char *list) {
    /* comma-separated prefixes */
    while (list && *list) {
        const char *e = strchr(list, ',');
        size_t      n = e ? (size_t)(e - list) : strlen(list);
        if (n && strncmp(name, list, n) == 0)
            return 1;
        list = e ? e + 1 : NULL;
    }
    return 0;
}

static void *open_library(void) {
    const char *env = getenv("SVTAV1_HIP_LIB");
    if (env && *env)
        return dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    char    path[PATH_MAX];
    ssize_t n = readlink("/proc/self/exe", path, sizeof(path) - 32);
    if (n > 0) {
        path[n]  = 0;
        char *sl = strrchr(path, '/');
        if (sl) {
            strcpy(sl + 1, "libsvtav1_hip.so");
            void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
            if (h)
                return h;
        }
    }
    return dlopen("libsvtav1_hip.so", RTLD_NOW | RTLD_GLOBAL);
}

static void *g_lib;
static void *lib_sym(const char *name) { return g_lib ? dlsym(g_lib, name) : NULL; }

/* e2e build only (tools/e2e/svt_hip_bind_simd.c): the reference's x86 intrinsics ladder as an encoder-level CPU baseline */
int svt_hip_bind_simd_install(void) __attribute__((weak));

int svt_hip_bind_install(char *msg, unsigned msg_len) {
    const char *simd = getenv("SVTAV1_E2E_SIMD");
    int n_simd = 0;
    if (simd && atoi(simd) && svt_hip_bind_simd_install) {
        n_simd = svt_hip_bind_simd_install();
        if (atoi(simd) == 1) { /* 2: intrinsics table first, then the HIP library on top (the batched hooks, SVTAV1_HIP_ONLY leaves) */
            snprintf(msg, msg_len, "no HIP library: SVTAV1_E2E_SIMD=1, %d RTCD pointers now point at the reference's x86 intrinsics kernels", n_simd);
            return n_simd;
        }
    }
    void *h = open_library();
    if (!h) {
        snprintf(msg, msg_len, "cannot load libsvtav1_hip.so: %s", dlerror());
        return -1;
    }
    int32_t (*p_init)(int32_t)                                             = (int32_t(*)(int32_t))dlsym(h, "svt_hip_init");
    int32_t (*p_install)(const SvtHipRtcdBinding *, uint32_t, uint32_t *) =
        (int32_t(*)(const SvtHipRtcdBinding *, uint32_t, uint32_t *))dlsym(h, "svt_hip_install_rtcd");
    const char *(*p_err)(void) = (const char *(*)(void))dlsym(h, "svt_hip_last_error");
    if (!p_init || !p_install || !p_err) {
        snprintf(msg, msg_len, "libsvtav1_hip.so lacks svt_hip_init / svt_hip_install_rtcd");
        return -1;
    }
    const char *dev = getenv("SVTAV1_HIP_DEVICE");
    if (p_init(dev ? atoi(dev) : 0) != SVT_HIP_OK) {
        snprintf(msg, msg_len, "%s", p_err());
        return -1;
    }
    const char       *only = getenv("SVTAV1_HIP_ONLY"), *skip = getenv("SVTAV1_HIP_SKIP");
    const uint32_t    n_all = sizeof(hip_bindings) / sizeof(hip_bindings[0]);
    SvtHipRtcdBinding sel[sizeof(hip_bindings) / sizeof(hip_bindings[0])];
    uint32_t          n = 0;
    for (uint32_t i = 0; i < n_all; i++) {
        if (only && *only && !name_in_list(hip_bindings[i].name, only))
            continue;
        if (skip && *skip && name_in_list(hip_bindings[i].name, skip))
            continue;
        sel[n++] = hip_bindings[i];
    }
    uint32_t done = 0;
    if (n && p_install(sel, n, &done) != SVT_HIP_OK) { /* n == 0: SVTAV1_HIP_ONLY matched nothing — batched entry points only */
        snprintf(msg, msg_len, "%s", p_err());
        return -1;
    }
    g_lib = h;
    svt_hip_bind_dev_setup(lib_sym); /* device API, PCIe counters, device-resident picture mirrors shared by the hooks below */
    svt_hip_bind_me_setup(lib_sym); /* Step 2b: batched open-loop ME (SVTAV1_HIP_TIERB_ME=1) */
    svt_hip_bind_tf_setup(lib_sym); /* Step 6b: whole-picture temporal filter (SVTAV1_HIP_TIERB_TF=1) */
    svt_hip_bind_tpl_setup(lib_sym); /* Step 3c: whole-picture TPL dispenser (SVTAV1_HIP_TIERB_TPL=1) */
    svt_hip_bind_pa_setup(lib_sym);  /* Step 2a: pyramid + block variances of the picture-analysis kernel (SVTAV1_HIP_TIERB_PA=1) */
    svt_hip_bind_txt_setup(lib_sym); /* Step 3a: the transform-type search's forward transforms as one batch per block (SVTAV1_HIP_TIERB_TXT=1) */
    svt_hip_bind_lf_setup(lib_sym);  /* Step 4: deblocking / CDEF / restoration of whole pictures (SVTAV1_HIP_TIERB_DLF / _CDEF / _LR=1) */
    snprintf(msg, msg_len, "%u of %u RTCD pointers now point at HIP leaves", done, n_all);
    return (int)done;
}
