// txfm.hip — fused forward transform -> (64-pt energy/repack) -> quantise -> [inverse transform + reconstruction]
// for batches of transform blocks, plus the ABI-identical per-call entry points (include/svt_hip_txfm.h).
//
// Mapping: a transform block (TB) of W x H uses L = max(W, H) adjacent lanes of one wavefront.  Column pass: lane c
// owns column c (H values in VGPRs, 1-D network fully unrolled, txfm_device.hpp); the W x H intermediate goes
// through LDS (row pitch W+1: conflict-free both ways); row pass: lane r owns row r and — still in registers —
// applies the 64-point energy/zero-out, the quantiser (scan position via the caller's iscan table, eob by a
// lane-group max-reduction) and, if requested, feeds the de-quantised row straight into the inverse row pass.
// The per-TB HBM traffic is therefore exactly the algorithmic minimum: residual in, qcoeff/dqcoeff (and recon) out.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/svt_hip_txfm.h"
#include "common.hpp"
#include "txfm_block.hpp"

using namespace svthip;
using namespace svthip::txd;
using namespace svthip::txb;

namespace {

template <int W, int H>
__global__ __launch_bounds__((Geo<W, H>::NT * Geo<W, H>::L), (Geo<W, H>::MINW)) void txfm_kernel(uint8_t *__restrict__ base,
                                                                            const SvtHipTxfmDesc *__restrict__ descs,
                                                                            SvtHipTxfmResult *__restrict__ results,
                                                                            uint32_t n, const uint32_t *__restrict__ perm) {
    using G = Geo<W, H>;
    constexpr int L = G::L, PW = G::PW, NT = G::NT;
    __shared__ int32_t buf[NT][H * PW];
    const int          t = threadIdx.x % L, slot = threadIdx.x / L;
    const uint32_t     i    = blockIdx.x * NT + slot;
    const bool         live = i < n;
    const uint32_t     tb   = live ? (perm ? perm[i] : i) : 0;  // perm: blocks grouped by transform type (group_by_type_kernel)
    txfm_block<W, H>(base, descs[tb], results + tb, live, t, buf[slot]);
}

// A wave that holds blocks of different 1-D kernel kinds runs every kind it holds.  For the sizes where several blocks share a
// wave, the batch call first groups the descriptor indices by tx_type inside chunks of GROUP_CHUNK blocks (one workgroup per
// chunk: LDS histogram -> ranks -> scatter; no global atomics, one extra launch): all but at most 15 waves per chunk then hold a
// single type.  Where a block's results go is unchanged (they are addressed through its descriptor / its index).
constexpr uint32_t GROUP_CHUNK = 4096, GROUP_MIN_BLOCKS = 2048;
__global__ __launch_bounds__(1024) void group_by_type_kernel(const SvtHipTxfmDesc *__restrict__ descs, uint32_t n, uint32_t *__restrict__ perm) {
    __shared__ uint32_t hist[16], start[16];
    const uint32_t base = blockIdx.x * GROUP_CHUNK;
    if (threadIdx.x < 16)
        hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t key[GROUP_CHUNK / 1024], rank[GROUP_CHUNK / 1024];
#pragma unroll
    for (uint32_t k = 0; k < GROUP_CHUNK / 1024; k++) {
        const uint32_t i = base + k * 1024 + threadIdx.x;
        key[k] = i < n ? (descs[i].tx_type & 15u) : 16u;
        if (key[k] < 16)
            rank[k] = atomicAdd(&hist[key[k]], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int q = 0; q < 16; q++) start[q] = acc, acc += hist[q];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < GROUP_CHUNK / 1024; k++)
        if (key[k] < 16)
            perm[base + start[key[k]] + rank[k]] = base + k * 1024 + threadIdx.x;
}

struct GroupWs {  // grow-only per-thread index buffer, guarded by an event against reuse from another stream
    uint32_t  *perm = nullptr;
    size_t     cap  = 0;
    hipEvent_t ev   = nullptr;
};
GroupWs &tls_group_ws() {
    static thread_local GroupWs w;
    return w;
}

// Stand-alone quantiser: one workgroup per block of n coefficients.
__global__ __launch_bounds__(256) void quantize_kernel(uint8_t *__restrict__ base, const SvtHipTxfmDesc *__restrict__ descs,
                                                       SvtHipTxfmResult *__restrict__ results, uint32_t n_coeffs) {
    __shared__ uint32_t   s_eob;
    const SvtHipTxfmDesc &d = descs[blockIdx.x];
    if (threadIdx.x == 0)
        s_eob = 0;
    __syncthreads();
    QP q;
    load_qp(q, d, base);
    const int32_t *ci    = (const int32_t *)(base + d.coeff_off);
    const int16_t *iscan = (const int16_t *)(base + d.iscan_off);
    int32_t       *qo = (int32_t *)(base + d.qcoeff_off), *dqo = (int32_t *)(base + d.dqcoeff_off);
    uint32_t       eob = 0;
    for (uint32_t rc = threadIdx.x; rc < n_coeffs; rc += 256) {
        int32_t qc, dqc;
        const bool small = q.simple && (uint32_t)(ci[rc] + 32767) <= 65534u;
        if (small && (q.mode == SVT_HIP_QUANT_B || q.mode == SVT_HIP_QUANT_B_HBD))
            quant_small<true>(q, ci[rc], rc != 0, qc, dqc);
        else if (small)
            quant_small<false>(q, ci[rc], rc != 0, qc, dqc);
        else
            quant_one(q, ci[rc], rc, qc, dqc);
        if (qc) {
            const uint32_t pos = (uint32_t)(uint16_t)iscan[rc] + 1u;
            eob                = pos > eob ? pos : eob;
        }
        qo[rc] = qc, dqo[rc] = dqc;
    }
    eob = group_max<64>(eob);
    if ((threadIdx.x & 63) == 0)
        atomicMax(&s_eob, eob);
    __syncthreads();
    if (threadIdx.x == 0) {
        SvtHipTxfmResult r;
        r.three_quad_energy = 0, r.eob = (uint16_t)s_eob, r.pad_ = 0, r.satd = 0;
        results[blockIdx.x] = r;
    }
}

// svt_handle_transformWxH on a coefficient buffer that already lives in memory (Tier A only).
__global__ __launch_bounds__(256) void handle64_kernel(int32_t *__restrict__ co, int w, int h, int energy_on, uint64_t *__restrict__ out) {
    __shared__ unsigned long long s_e;
    if (threadIdx.x == 0)
        s_e = 0;
    __syncthreads();
    const int iw = w < 32 ? w : 32, ih = h < 32 ? h : 32;
    uint64_t  e  = 0;
    if (energy_on)
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int r = i / w, c = i - r * w;
            if (r >= ih || c >= iw)
                e += (uint64_t)((int64_t)co[i] * (int64_t)co[i]);
        }
    e = group_sum64<64>(e);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(&s_e, (unsigned long long)e);
    __syncthreads();
    if (w == 64) {  // repack rows 1..ih-1 to a pitch of 32 (rows move towards lower addresses: go row by row)
        for (int r = 1; r < ih; r++) {
            int32_t v = 0;
            if (threadIdx.x < 32)
                v = co[r * 64 + threadIdx.x];
            __syncthreads();
            if (threadIdx.x < 32)
                co[r * 32 + threadIdx.x] = v;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0)
        *out = s_e;
}

std::once_flag g_tables_once;
int32_t        g_tables_rc = SVT_HIP_OK;
void           upload_tables() {
    // the twiddles are compile-time constants (txfm_device.hpp); check the constexpr cosine against libm once
    constexpr CosTable tab = make_cospi();
    for (int b = 0; b < 4; b++)
        for (int j = 0; j < 64; j++)
            if (tab.v[b][j] != (int32_t)llround(cos(M_PI * j / 128.0) * (double)(1 << (10 + b)))) {
                set_error("transform constant table self-check failed");
                g_tables_rc = SVT_HIP_ERR_RUNTIME;
            }
}
int32_t txfm_ready() {
    if (!ensure_init())
        return SVT_HIP_ERR_NO_DEVICE;
    std::call_once(g_tables_once, upload_tables);
    return g_tables_rc;
}

template <int W, int H>
void launch_txfm(uint8_t *base, const SvtHipTxfmDesc *descs, SvtHipTxfmResult *res, uint32_t n, hipStream_t st) {
    using G = Geo<W, H>;
    const uint32_t *perm = nullptr;
    // Measured on the 4K 10-bit workload (DCT / ADST alternating between neighbours): the grouped launch is NOT faster — the
    // mixed-type waves cost 0-3 %, the grouping pass 13-30 us per launch — so it is off unless SVTAV1_HIP_GROUP_TX is set.
    static const bool group = getenv("SVTAV1_HIP_GROUP_TX") != nullptr;
    if (G::NT > 1 && n >= GROUP_MIN_BLOCKS && group) {
        GroupWs &w = tls_group_ws();
        bool     ok = true;
        if (!w.ev)
            ok = hipEventCreateWithFlags(&w.ev, hipEventDisableTiming) == hipSuccess;
        if (ok && w.cap < n) {
            if (w.perm)
                (void)hipFree(w.perm);  // synchronises: no launch still reads it
            w.perm = nullptr, w.cap = 0;
            ok = hipMalloc((void **)&w.perm, (size_t)n * sizeof(uint32_t)) == hipSuccess;
            if (ok)
                w.cap = n;
        } else if (ok) {
            ok = hipStreamWaitEvent(st, w.ev, 0) == hipSuccess;  // the previous user of the buffer may sit on another stream
        }
        if (ok) {
            hipLaunchKernelGGL(group_by_type_kernel, dim3((n + GROUP_CHUNK - 1) / GROUP_CHUNK), dim3(1024), 0, st, descs, n, w.perm);
            perm = w.perm;
        }
        hipLaunchKernelGGL((txfm_kernel<W, H>), dim3((n + G::NT - 1) / G::NT), dim3(G::NT * G::L), 0, st, base, descs, res, n, perm);
        if (perm)
            (void)hipEventRecord(w.ev, st);
        return;
    }
    hipLaunchKernelGGL((txfm_kernel<W, H>), dim3((n + G::NT - 1) / G::NT), dim3(G::NT * G::L), 0, st, base, descs, res, n, perm);
}

bool dispatch_txfm(uint32_t w, uint32_t h, uint8_t *base, const SvtHipTxfmDesc *descs, SvtHipTxfmResult *res, uint32_t n,
                   hipStream_t st) {
#define CASE(W, H)                                \
    if (w == W && h == H) {                       \
        launch_txfm<W, H>(base, descs, res, n, st); \
        return true;                              \
    }
    CASE(4, 4) CASE(8, 8) CASE(16, 16) CASE(32, 32) CASE(64, 64) CASE(4, 8) CASE(8, 4) CASE(8, 16) CASE(16, 8) CASE(16, 32)
    CASE(32, 16) CASE(32, 64) CASE(64, 32) CASE(4, 16) CASE(16, 4) CASE(8, 32) CASE(32, 8) CASE(16, 64) CASE(64, 16)
#undef CASE
    return false;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ Tier B
extern "C" int32_t svt_hip_txfm_quant_batch(uint8_t *d_base, const SvtHipTxfmDesc *d_desc, SvtHipTxfmResult *d_result,
                                            uint32_t n_blocks, uint32_t w, uint32_t h, void *stream) {
    if (!d_base || !d_desc || !d_result) {
        set_error("svt_hip_txfm_quant_batch: NULL pointer");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    const int32_t rc = txfm_ready();
    if (rc != SVT_HIP_OK)
        return rc;
    if (n_blocks == 0)
        return SVT_HIP_OK;
    if (!dispatch_txfm(w, h, d_base, d_desc, d_result, n_blocks, resolve_stream(stream))) {
        set_error("svt_hip_txfm_quant_batch: %ux%u is not an AV1 transform size", w, h);
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

extern "C" int32_t svt_hip_quantize_batch(uint8_t *d_base, const SvtHipTxfmDesc *d_desc, SvtHipTxfmResult *d_result,
                                          uint32_t n_blocks, uint32_t n_coeffs, void *stream) {
    if (!d_base || !d_desc || !d_result || n_coeffs == 0) {
        set_error("svt_hip_quantize_batch: bad argument");
        return SVT_HIP_ERR_BAD_PARAMETER;
    }
    const int32_t rc = txfm_ready();
    if (rc != SVT_HIP_OK)
        return rc;
    if (n_blocks == 0)
        return SVT_HIP_OK;
    hipLaunchKernelGGL(quantize_kernel, dim3(n_blocks), dim3(256), 0, resolve_stream(stream), d_base, d_desc, d_result, n_coeffs);
    SVT_HIP_CHECK(hipGetLastError());
    return SVT_HIP_OK;
}

// ------------------------------------------------------------------------------------------------ Tier A
namespace {

[[noreturn]] void fatal(const char *what) { svthip::tier_a_throw("%s: %s", what, svt_hip_last_error()); }
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// One-block arena in the per-thread scratch: [desc][result][buffers...]
struct Arena {
    uint8_t *h, *d;
    size_t   used;
    Arena(size_t bytes) {
        Scratch &sc = tls_scratch();
        h = sc.host(bytes), d = sc.device(bytes), used = 512;
    }
    size_t put(const void *src, size_t bytes) {
        const size_t off = used;
        if (src)
            memcpy(h + off, src, bytes);
        used += up256(bytes + 16);
        return off;
    }
};

void run_one(int w, int h, Arena &a, size_t upload_bytes, hipStream_t st) {
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.d, a.h, upload_bytes, hipMemcpyHostToDevice, st));
    if (!dispatch_txfm((uint32_t)w, (uint32_t)h, a.d, (const SvtHipTxfmDesc *)a.d, (SvtHipTxfmResult *)(a.d + 256), 1, st))
        fatal("bad transform size");
    SVT_HIP_CHECK_FATAL(hipGetLastError());
}

void fwd_tier_a(int w, int h, int shape, int16_t *input, int32_t *output, uint32_t stride, int32_t tx_type, uint8_t bd) {
    if (txfm_ready() != SVT_HIP_OK)
        fatal("forward transform");
    const size_t in_bytes = ((size_t)(h - 1) * stride + w) * 2, out_bytes = (size_t)w * h * 4;
    Arena        a(1024 + up256(in_bytes + 16) + up256(out_bytes + 16));
    SvtHipTxfmDesc *dsc = (SvtHipTxfmDesc *)a.h;
    memset(dsc, 0, sizeof(*dsc));
    dsc->residual_off = a.put(input, in_bytes);
    dsc->coeff_off    = a.put(nullptr, out_bytes);
    dsc->qcoeff_off = dsc->dqcoeff_off = dsc->qm_off = dsc->iqm_off = dsc->iscan_off = dsc->pred_off = dsc->recon_off = SVT_HIP_NO_OFFSET;
    dsc->residual_stride = stride;
    dsc->tx_type = (uint8_t)tx_type, dsc->shape = (uint8_t)shape, dsc->bit_depth = bd;
    dsc->quant_mode = SVT_HIP_QUANT_NONE, dsc->flags = SVT_HIP_TX_FWD | SVT_HIP_TX_FULLCOEFF;
    hipStream_t st = resolve_stream(nullptr);
    run_one(w, h, a, dsc->coeff_off, st);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + dsc->coeff_off, a.d + dsc->coeff_off, out_bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(output, a.h + dsc->coeff_off, out_bytes);
}

template <class PIX>
void inv_tier_a(int w, int h, const int32_t *input, const PIX *out_r, int32_t stride_r, PIX *out_w, int32_t stride_w, int32_t tx_type,
                int32_t bd) {
    if (txfm_ready() != SVT_HIP_OK)
        fatal("inverse transform");
    const int    iw = w < 32 ? w : 32, ih = h < 32 ? h : 32;
    const size_t co_bytes = (size_t)iw * ih * 4, pr_bytes = ((size_t)(h - 1) * stride_r + w) * sizeof(PIX),
                 rc_bytes = ((size_t)(h - 1) * stride_w + w) * sizeof(PIX);
    Arena           a(1024 + up256(co_bytes + 16) + up256(pr_bytes + 16) + up256(rc_bytes + 16));
    SvtHipTxfmDesc *dsc = (SvtHipTxfmDesc *)a.h;
    memset(dsc, 0, sizeof(*dsc));
    dsc->dqcoeff_off = a.put(input, co_bytes);
    dsc->pred_off    = a.put(out_r, pr_bytes);
    dsc->recon_off   = a.put(nullptr, rc_bytes);
    dsc->residual_off = dsc->coeff_off = dsc->qcoeff_off = dsc->qm_off = dsc->iqm_off = dsc->iscan_off = SVT_HIP_NO_OFFSET;
    dsc->pred_stride = (uint32_t)stride_r, dsc->recon_stride = (uint32_t)stride_w;
    dsc->tx_type = (uint8_t)tx_type, dsc->bit_depth = (uint8_t)bd;
    dsc->quant_mode = SVT_HIP_QUANT_NONE, dsc->flags = SVT_HIP_TX_INV | (sizeof(PIX) == 2 ? SVT_HIP_TX_PIXEL16 : 0);
    hipStream_t st = resolve_stream(nullptr);
    run_one(w, h, a, dsc->recon_off, st);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + dsc->recon_off, a.d + dsc->recon_off, rc_bytes, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    const PIX *rc = (const PIX *)(a.h + dsc->recon_off);
    for (int r = 0; r < h; r++) memcpy(out_w + (size_t)r * stride_w, rc + (size_t)r * stride_w, (size_t)w * sizeof(PIX));
}

void quant_tier_a(int mode, const int32_t *coeff_ptr, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                  const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                  const int16_t *iscan, const uint8_t *qm, const uint8_t *iqm, int log_scale) {
    if (txfm_ready() != SVT_HIP_OK)
        fatal("quantize");
    const size_t    cb = (size_t)n * 4;
    Arena           a(1024 + 3 * up256(cb + 16) + up256((size_t)n * 2 + 16) + 2 * up256((size_t)n + 16));
    SvtHipTxfmDesc *dsc = (SvtHipTxfmDesc *)a.h;
    memset(dsc, 0, sizeof(*dsc));
    dsc->coeff_off = a.put(coeff_ptr, cb);
    dsc->iscan_off = a.put(iscan, (size_t)n * 2);
    dsc->qm_off    = qm ? a.put(qm, (size_t)n) : SVT_HIP_NO_OFFSET;
    dsc->iqm_off   = iqm ? a.put(iqm, (size_t)n) : SVT_HIP_NO_OFFSET;
    const size_t upload = a.used;
    dsc->qcoeff_off     = a.put(nullptr, cb);
    dsc->dqcoeff_off    = a.put(nullptr, cb);
    dsc->residual_off = dsc->pred_off = dsc->recon_off = SVT_HIP_NO_OFFSET;
    for (int i = 0; i < 2; i++) {
        dsc->zbin[i] = zbin ? zbin[i] : 0, dsc->round[i] = round[i], dsc->quant[i] = quant[i];
        dsc->quant_shift[i] = quant_shift ? quant_shift[i] : 0, dsc->dequant[i] = dequant[i];
    }
    dsc->quant_mode = (uint8_t)mode, dsc->log_scale = (uint8_t)log_scale;
    hipStream_t st = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.d, a.h, upload, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(quantize_kernel, dim3(1), dim3(256), 0, st, a.d, (const SvtHipTxfmDesc *)a.d, (SvtHipTxfmResult *)(a.d + 256),
                       (uint32_t)n);
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + 256, a.d + 256, sizeof(SvtHipTxfmResult), hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + dsc->qcoeff_off, a.d + dsc->qcoeff_off, a.used - dsc->qcoeff_off, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    memcpy(qcoeff, a.h + dsc->qcoeff_off, cb);
    memcpy(dqcoeff, a.h + dsc->dqcoeff_off, cb);
    *eob = ((const SvtHipTxfmResult *)(a.h + 256))->eob;
}

uint64_t handle_tier_a(int w, int h, int energy_on, int32_t *output) {
    if (txfm_ready() != SVT_HIP_OK)
        fatal("handle_transform");
    const size_t bytes = (size_t)w * h * 4;
    Arena        a(1024 + up256(bytes + 16));
    const size_t off = a.put(output, bytes);
    hipStream_t  st  = resolve_stream(nullptr);
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.d + off, a.h + off, bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(handle64_kernel, dim3(1), dim3(256), 0, st, (int32_t *)(a.d + off), w, h, energy_on, (uint64_t *)(a.d + 256));
    SVT_HIP_CHECK_FATAL(hipGetLastError());
    const int    iw = w < 32 ? w : 32, ih = h < 32 ? h : 32;
    SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + 256, a.d + 256, 8, hipMemcpyDeviceToHost, st));
    if (w == 64)
        SVT_HIP_CHECK_FATAL(hipMemcpyAsync(a.h + off, a.d + off, (size_t)iw * ih * 4, hipMemcpyDeviceToHost, st));
    SVT_HIP_CHECK_FATAL(hipStreamSynchronize(st));
    if (w == 64)  // rows 1..ih-1 were repacked; everything beyond the packed area is left as the C function leaves it
        for (int r = 1; r < ih; r++) memmove(output + r * 32, a.h + off + (size_t)r * 32 * 4, 32 * 4);
    return *(const uint64_t *)(a.h + 256);
}

}  // namespace

#define FWD_DEF(W, H)                                                                                                          \
    extern "C" void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t *i, int32_t *o, uint32_t s, int32_t t, uint8_t b) { TIER_A_CALL(svt_av1_fwd_txfm2d_##W##x##H, fwd_tier_a(W, H, 0, i, o, s, t, b), (i, o, s, t, b)); } \
    extern "C" void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t *i, int32_t *o, uint32_t s, int32_t t, uint8_t b) { TIER_A_CALL(svt_av1_fwd_txfm2d_##W##x##H##_N2, fwd_tier_a(W, H, 1, i, o, s, t, b), (i, o, s, t, b)); } \
    extern "C" void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t *i, int32_t *o, uint32_t s, int32_t t, uint8_t b) { TIER_A_CALL(svt_av1_fwd_txfm2d_##W##x##H##_N4, fwd_tier_a(W, H, 2, i, o, s, t, b), (i, o, s, t, b)); }
FWD_DEF(4, 4) FWD_DEF(8, 8) FWD_DEF(16, 16) FWD_DEF(32, 32) FWD_DEF(64, 64) FWD_DEF(4, 8) FWD_DEF(8, 4) FWD_DEF(8, 16) FWD_DEF(16, 8)
FWD_DEF(16, 32) FWD_DEF(32, 16) FWD_DEF(32, 64) FWD_DEF(64, 32) FWD_DEF(4, 16) FWD_DEF(16, 4) FWD_DEF(8, 32) FWD_DEF(32, 8)
FWD_DEF(16, 64) FWD_DEF(64, 16)

#define INV_SQ(W, H)                                                                                                         \
    extern "C" void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *i, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, \
                                                           int32_t t, int32_t bd) { TIER_A_CALL(svt_av1_inv_txfm2d_add_##W##x##H, inv_tier_a(W, H, i, r, sr, w, sw, t, bd), (i, r, sr, w, sw, t, bd)); }
#define INV_TS(W, H)                                                                                                         \
    extern "C" void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *i, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, \
                                                           int32_t t, int32_t txs, int32_t bd) { TIER_A_CALL(svt_av1_inv_txfm2d_add_##W##x##H, inv_tier_a(W, H, i, r, sr, w, sw, t, bd), (i, r, sr, w, sw, t, txs, bd)); }
#define INV_EOB(W, H)                                                                                                        \
    extern "C" void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t *i, uint16_t *r, int32_t sr, uint16_t *w, int32_t sw, \
                                                           int32_t t, int32_t txs, int32_t eob, int32_t bd) { TIER_A_CALL(svt_av1_inv_txfm2d_add_##W##x##H, inv_tier_a(W, H, i, r, sr, w, sw, t, bd), (i, r, sr, w, sw, t, txs, eob, bd)); }
INV_SQ(4, 4) INV_SQ(8, 8) INV_SQ(16, 16) INV_SQ(32, 32) INV_SQ(64, 64) INV_TS(4, 8) INV_TS(8, 4) INV_TS(4, 16) INV_TS(16, 4)
INV_EOB(8, 16) INV_EOB(16, 8) INV_EOB(16, 32) INV_EOB(32, 16) INV_EOB(32, 64) INV_EOB(64, 32) INV_EOB(8, 32) INV_EOB(32, 8)
INV_EOB(16, 64) INV_EOB(64, 16)

#define HANDLE_DEF(W, H)                                                                                                  \
    extern "C" uint64_t svt_handle_transform##W##x##H##_hip(int32_t *o) { TIER_A_CALL(svt_handle_transform##W##x##H, handle_tier_a(W, H, 1, o), (o)); }             \
    extern "C" uint64_t svt_handle_transform##W##x##H##_N2_N4_hip(int32_t *o) { TIER_A_CALL(svt_handle_transform##W##x##H##_N2_N4, handle_tier_a(W, H, 0, o), (o)); }
HANDLE_DEF(16, 64) HANDLE_DEF(32, 64) HANDLE_DEF(64, 16) HANDLE_DEF(64, 32) HANDLE_DEF(64, 64)

#define QA SVT_HIP_QARGS
#define QP_ coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, iscan
#define QC_ coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan  /* the reference's full argument list */
// svt_av1_inv_txfm_add (common_dsp_rtcd.h:150; inv_transforms.c:3177-3193): the 8-bit destination form, dispatched on
// txfm_param->tx_size like highbd_inv_txfm_add (:3112-3146).  The reference widens the prediction to 16 bits, transforms at
// txfm_param->bd and narrows; with bd == 8 (the only value its callers pass, :3101,3163) that equals the 8-bit pixel path of
// the kernel.  Lossless (the 4x4 Walsh-Hadamard form) is never requested by this reference version (full_loop.c:1703,1715
// and src_ops_process.c:1157 pass 0) and is refused loudly.
static void svt_av1_inv_txfm_add_hip_impl(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w, const SvtHipTxfmParam *txfm_param);
extern "C" void svt_av1_inv_txfm_add_hip(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w, const SvtHipTxfmParam *txfm_param) { TIER_A_CALL(svt_av1_inv_txfm_add, svt_av1_inv_txfm_add_hip_impl(dqcoeff, dst_r, stride_r, dst_w, stride_w, txfm_param), (dqcoeff, dst_r, stride_r, dst_w, stride_w, txfm_param)); }
static void svt_av1_inv_txfm_add_hip_impl(const int32_t *dqcoeff, uint8_t *dst_r, int32_t stride_r, uint8_t *dst_w, int32_t stride_w, const SvtHipTxfmParam *txfm_param) {
    static const uint8_t wide[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
    static const uint8_t high[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
    if (!txfm_param || txfm_param->tx_size >= 19 || txfm_param->lossless || txfm_param->bd != 8) {
        set_error("svt_av1_inv_txfm_add: unsupported parameters (tx_size %d, lossless %d, bd %d)", txfm_param ? txfm_param->tx_size : -1,
                  txfm_param ? txfm_param->lossless : -1, txfm_param ? txfm_param->bd : -1);
        fatal("inverse transform");
    }
    inv_tier_a<uint8_t>(wide[txfm_param->tx_size], high[txfm_param->tx_size], dqcoeff, dst_r, stride_r, dst_w, stride_w,
                        txfm_param->tx_type, 8);
}

extern "C" void svt_aom_quantize_b_hip(QA, const uint8_t *qm, const uint8_t *iqm, int32_t ls) { TIER_A_CALL(svt_aom_quantize_b, quant_tier_a(SVT_HIP_QUANT_B, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }
extern "C" void svt_av1_quantize_b_qm_hip(QA, const uint8_t *qm, const uint8_t *iqm, int32_t ls) { TIER_A_CALL(svt_av1_quantize_b_qm, quant_tier_a(SVT_HIP_QUANT_B, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }
extern "C" void svt_aom_highbd_quantize_b_hip(QA, const uint8_t *qm, const uint8_t *iqm, int32_t ls) { TIER_A_CALL(svt_aom_highbd_quantize_b, quant_tier_a(SVT_HIP_QUANT_B_HBD, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }
extern "C" void svt_av1_highbd_quantize_b_qm_hip(QA, const uint8_t *qm, const uint8_t *iqm, int32_t ls) { TIER_A_CALL(svt_av1_highbd_quantize_b_qm, quant_tier_a(SVT_HIP_QUANT_B_HBD, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }
extern "C" void svt_av1_quantize_fp_hip(QA) { TIER_A_CALL(svt_av1_quantize_fp, quant_tier_a(SVT_HIP_QUANT_FP, QP_, nullptr, nullptr, 0), (QC_)); }
extern "C" void svt_av1_quantize_fp_32x32_hip(QA) { TIER_A_CALL(svt_av1_quantize_fp_32x32, quant_tier_a(SVT_HIP_QUANT_FP, QP_, nullptr, nullptr, 1), (QC_)); }
extern "C" void svt_av1_quantize_fp_64x64_hip(QA) { TIER_A_CALL(svt_av1_quantize_fp_64x64, quant_tier_a(SVT_HIP_QUANT_FP, QP_, nullptr, nullptr, 2), (QC_)); }
extern "C" void svt_av1_quantize_fp_qm_hip(QA, const uint8_t *qm, const uint8_t *iqm, int16_t ls) { TIER_A_CALL(svt_av1_quantize_fp_qm, quant_tier_a(SVT_HIP_QUANT_FP, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }
extern "C" void svt_av1_highbd_quantize_fp_hip(QA, int16_t ls) { TIER_A_CALL(svt_av1_highbd_quantize_fp, quant_tier_a(SVT_HIP_QUANT_FP_HBD, QP_, nullptr, nullptr, ls), (QC_, ls)); }
extern "C" void svt_av1_highbd_quantize_fp_qm_hip(QA, const uint8_t *qm, const uint8_t *iqm, int16_t ls) { TIER_A_CALL(svt_av1_highbd_quantize_fp_qm, quant_tier_a(SVT_HIP_QUANT_FP_HBD, QP_, qm, iqm, ls), (QC_, qm, iqm, ls)); }

SVT_HIP_MODULE_WARMUP(txfm)
