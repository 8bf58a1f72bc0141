/*
 * svt_hip_shard.h — C-ABI of the multi-GPU layer of the hot path: one encoder process per GPU,
 * pictures sharded across them, ONE exchange step — the GPU that reconstructed and loop-filtered a reference picture
 * publishes it to the GPUs that will predict from it.
 *
 * Reference hook points (paths relative to /root/reference/Source/Lib/Codec):
 *   rest_process.c:659-660, 732-744   the restoration kernel hands the finished reference picture to the picture manager
 *                                     (svt_aom_pad_ref_and_set_flags + the PictureDemuxResults post): where
 *                                     svt_hip_publish_reference is called
 *   me_process.c:174-290, pic_analysis_process.c:2126  open-loop stages: read SOURCE pictures only -> no exchange,
 *                                     svt_hip_shard_segment decides which pictures a GPU analyses
 *   enc_handle.c:4379-4387            hierarchical_levels 5 -> 6 temporal layers, mini-GOP 32: svt_hip_shard_layer / _owner
 *
 * The transport is RCCL (librccl.so is loaded on first use; nothing else in this library needs it), collectives run on a
 * side stream so that they overlap the owner's next picture, and xGMI being point to point the whole padded picture
 * (Y, U, V back to back in one allocation, about 27 MB at 4K 10-bit) travels as ONE broadcast.
 */
#ifndef SVT_HIP_SHARD_H
#define SVT_HIP_SHARD_H

#include "svt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Temporal layer of picture `pic_in_minigop` (1 .. minigop_size, display order; minigop_size = the base-layer picture) of
 * a hierarchical mini-GOP whose size is a power of two: 0 for the base picture, 1 for minigop_size/2, ... */
SVT_HIP_API uint32_t svt_hip_shard_layer(uint32_t pic_in_minigop, uint32_t minigop_size);

/* Layer-aware owner for the CLOSED-loop stages: a mini-GOP can only run as many pictures at once as the
 * current layer holds (1, 1, 2, 4, 8, 16 for 32).  svt_hip_shard_slot numbers the pictures of a mini-GOP in the order they
 * become startable (base picture 0, layer 1 -> 1, layer 2 -> 2..3, layer 3 -> 4..7, ...; display order inside a layer);
 * svt_hip_shard_owner = slot mod n_gpus, so the pictures of one layer sit on different GPUs AND consecutive layers continue
 * the round-robin (no GPU collects the first picture of every layer; 32 pictures over 8 GPUs = 4 each).
 * svt_hip_shard_owner_gop takes the picture number in display order over the whole sequence (0 = key picture, mini-GOP m
 * = pictures m*size+1 .. (m+1)*size) and additionally rotates by the mini-GOP index, so base-layer pictures alternate. */
SVT_HIP_API uint32_t svt_hip_shard_slot(uint32_t pic_in_minigop, uint32_t minigop_size);
SVT_HIP_API uint32_t svt_hip_shard_owner(uint32_t pic_in_minigop, uint32_t minigop_size, uint32_t n_gpus);
SVT_HIP_API uint32_t svt_hip_shard_owner_gop(uint64_t picture_number, uint32_t minigop_size, uint32_t n_gpus);

/* Contiguous segment for the OPEN-loop stages: out = {first, last, ctx_first, ctx_last}: pictures [first, last) are
 * analysed by `rank`, [ctx_first, ctx_last) must be resident on it (its references).  Only pictures with a full set of
 * references are analysed (lookback .. n_pictures - lookahead - 1), split as evenly as possible. */
SVT_HIP_API void svt_hip_shard_segment(uint32_t n_pictures, uint32_t world, uint32_t rank, uint32_t lookback, uint32_t lookahead,
                                       uint32_t out[4]);

/* ---- communicator (RCCL) ---------------------------------------------------------------------------------------------
 * Rank 0 calls svt_hip_comm_get_unique_id and hands the 128 bytes to the other processes by whatever channel the host
 * has (a file, a socket, MPI ...); every rank then calls svt_hip_comm_create (collective: it returns when all `world`
 * ranks have joined).  svt_hip_init(device) must have succeeded first. */
#define SVT_HIP_COMM_ID_BYTES 128
SVT_HIP_API int32_t svt_hip_comm_get_unique_id(uint8_t id[SVT_HIP_COMM_ID_BYTES]);
SVT_HIP_API int32_t svt_hip_comm_create(const uint8_t id[SVT_HIP_COMM_ID_BYTES], int32_t world, int32_t rank, void **comm);
SVT_HIP_API int32_t svt_hip_comm_destroy(void *comm);

/* Broadcast `bytes` bytes at device address d_picture from rank `owner` to every rank of `comm` (collective: every rank
 * calls it with its own copy of the buffer).  The broadcast is enqueued on `side_stream` AFTER everything that is on
 * `producer_stream` at the time of the call (the owner's in-loop filter kernels), and `*done` (created on first use when
 * *done == NULL, reused otherwise) is recorded behind it: consumers call svt_hip_wait_reference(done, their stream) before
 * the first kernel that reads the picture.  Asynchronous; streams are hipStream_t as void*, NULL = the calling thread's
 * private stream for producer_stream (side_stream must be given). */
SVT_HIP_API int32_t svt_hip_publish_reference(void *d_picture, size_t bytes, int32_t owner, void *comm, void *producer_stream,
                                              void *side_stream, void **done);
SVT_HIP_API int32_t svt_hip_wait_reference(void *done, void *consumer_stream);
SVT_HIP_API int32_t svt_hip_event_destroy(void *done);

#ifdef __cplusplus
}
#endif
#endif /* SVT_HIP_SHARD_H */
