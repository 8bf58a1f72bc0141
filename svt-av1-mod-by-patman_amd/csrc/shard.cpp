// shard.cpp — multi-GPU layer of libsvtav1_hip (include/svt_hip_shard.h): picture -> GPU assignment and the one exchange
// step of the encoder, the publication of a reconstructed reference picture over RCCL (xGMI inside a node).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include <rccl/rccl.h>

#include "../../include/svt_hip_shard.h"
#include "common.hpp"

using namespace svthip;

namespace {

// RCCL entry points, resolved on first use: the library proper has no link-time dependency on librccl, and a process that
// already carries an RCCL (PyTorch does) shares it through the soname.
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl           g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.h)
                break;
        }
        if (!g_rccl.h) {
            set_error("cannot load librccl.so: %s", dlerror());
            return;
        }
#define SYM(field, sym) g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.h, sym)
        SYM(GetUniqueId, "ncclGetUniqueId");
        SYM(CommInitRank, "ncclCommInitRank");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(Broadcast, "ncclBroadcast");
        SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Broadcast && g_rccl.GetErrorString;
        if (!g_rccl.ok)
            set_error("librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast");
    });
    return g_rccl.ok ? &g_rccl : nullptr;
}

#define RCCL_CHECK(r, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t rc_ = (expr);                                                                 \
        if (rc_ != ncclSuccess) {                                                                  \
            set_error("%s failed: %s", #expr, (r)->GetErrorString(rc_));                           \
            return SVT_HIP_ERR_RUNTIME;                                                            \
        }                                                                                          \
    } while (0)

}  // namespace

extern "C" {

uint32_t svt_hip_shard_layer(uint32_t pic, uint32_t minigop) {
    if (minigop == 0 || (minigop & (minigop - 1)) || pic == 0 || pic > minigop)
        return 0;
    uint32_t layer = 0, step = minigop;
    while (pic % step)
        step >>= 1, layer++;
    return layer;
}

uint32_t svt_hip_shard_slot(uint32_t pic, uint32_t minigop) {
    // position of the picture in the order the closed-loop stages can start them: base picture, then layer by layer,
    // display order inside a layer: 0 | 1 | 2 3 | 4..7 | 8..15 | 16..31
    const uint32_t layer = svt_hip_shard_layer(pic, minigop);
    if (layer == 0)
        return 0;
    const uint32_t step = minigop >> layer;  // the pictures of this layer are the odd multiples of `step`
    return (1u << (layer - 1)) + (pic / step) / 2;
}

uint32_t svt_hip_shard_owner(uint32_t pic, uint32_t minigop, uint32_t n_gpus) {
    return n_gpus ? svt_hip_shard_slot(pic, minigop) % n_gpus : 0;
}

uint32_t svt_hip_shard_owner_gop(uint64_t picture_number, uint32_t minigop, uint32_t n_gpus) {
    if (n_gpus == 0 || minigop == 0 || (minigop & (minigop - 1)))
        return 0;
    if (picture_number == 0)
        return 0;                                                 // the key picture
    const uint64_t mg  = (picture_number - 1) / minigop;          // mini-GOP index: pictures mg*minigop+1 .. (mg+1)*minigop
    const uint32_t pic = (uint32_t)(picture_number - mg * minigop);
    // rotate by the mini-GOP index: base pictures (slot 0) alternate over the GPUs, and so does every other slot
    return (uint32_t)((svt_hip_shard_slot(pic, minigop) + mg + 1) % n_gpus);
}

void svt_hip_shard_segment(uint32_t n, uint32_t world, uint32_t rank, uint32_t lookback, uint32_t lookahead, uint32_t out[4]) {
    const uint32_t lo = lookback, hi = n > lookahead ? n - lookahead : 0;
    const uint32_t cnt = hi > lo ? hi - lo : 0;
    if (world == 0)
        world = 1;
    const uint32_t base = cnt / world, rem = cnt % world;
    const uint32_t first = lo + rank * base + (rank < rem ? rank : rem);
    const uint32_t last  = first + base + (rank < rem ? 1 : 0);
    if (first >= last) {
        out[0] = out[1] = out[2] = out[3] = first;
        return;
    }
    out[0] = first, out[1] = last, out[2] = first - lookback, out[3] = last + lookahead;
}

int32_t svt_hip_comm_get_unique_id(uint8_t id[SVT_HIP_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == SVT_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id)
        return SVT_HIP_ERR_BAD_PARAMETER;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const Rccl *r = rccl();
    if (!r)
        return SVT_HIP_ERR_RUNTIME;
    ncclUniqueId u;
    RCCL_CHECK(r, r->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SVT_HIP_OK;
}

int32_t svt_hip_comm_create(const uint8_t id[SVT_HIP_COMM_ID_BYTES], int32_t world, int32_t rank, void **comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) {
        set_error("svt_hip_comm_create: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const Rccl *r = rccl();
    if (!r)
        return SVT_HIP_ERR_RUNTIME;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    RCCL_CHECK(r, r->CommInitRank(&c, world, u, rank));
    *comm = (void *)c;
    return SVT_HIP_OK;
}

int32_t svt_hip_comm_destroy(void *comm) {
    if (!comm)
        return SVT_HIP_OK;
    const Rccl *r = rccl();
    if (!r)
        return SVT_HIP_ERR_RUNTIME;
    RCCL_CHECK(r, r->CommDestroy((ncclComm_t)comm));
    return SVT_HIP_OK;
}

int32_t svt_hip_publish_reference(void *d_picture, size_t bytes, int32_t owner, void *comm, void *producer_stream, void *side_stream,
                                  void **done) {
    if (!d_picture || !bytes || !comm || !side_stream || owner < 0) {
        set_error("svt_hip_publish_reference: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    const Rccl *r = rccl();
    if (!r)
        return SVT_HIP_ERR_RUNTIME;
    hipStream_t prod = resolve_stream(producer_stream), side = (hipStream_t)side_stream;
    // order the broadcast behind the kernels that wrote the picture: an event of the producer stream, awaited by the side stream
    static thread_local hipEvent_t produced = nullptr;
    if (!produced)
        SVT_HIP_CHECK(hipEventCreateWithFlags(&produced, hipEventDisableTiming));
    SVT_HIP_CHECK(hipEventRecord(produced, prod));
    SVT_HIP_CHECK(hipStreamWaitEvent(side, produced, 0));
    RCCL_CHECK(r, r->Broadcast(d_picture, d_picture, bytes, ncclUint8, owner, (ncclComm_t)comm, side));
    if (done) {
        if (!*done) {
            hipEvent_t e;
            SVT_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            *done = (void *)e;
        }
        SVT_HIP_CHECK(hipEventRecord((hipEvent_t)*done, side));
    }
    return SVT_HIP_OK;
}

int32_t svt_hip_wait_reference(void *done, void *consumer_stream) {
    if (!done)
        return SVT_HIP_ERR_BAD_PARAMETER;
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    SVT_HIP_CHECK(hipStreamWaitEvent(resolve_stream(consumer_stream), (hipEvent_t)done, 0));
    return SVT_HIP_OK;
}

int32_t svt_hip_event_destroy(void *done) {
    if (done)
        SVT_HIP_CHECK(hipEventDestroy((hipEvent_t)done));
    return SVT_HIP_OK;
}

}  // extern "C"
