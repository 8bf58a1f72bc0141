"""Multi-GPU layout of the hot path: pictures are sharded across ranks, nothing else.

The open-loop analysis of one picture (pyramid, variance, ME against its reference pictures) reads only that
picture and its references' source pictures, so a clip splits into contiguous segments, one per
rank (= one per GPU), each carrying `lookback` / `lookahead` context pictures so that every reference is resident
locally.  There is no exchange step in the open-loop data path; the process group is used for the start/stop barrier, the
max-over-ranks timing and (optionally) collecting the small per-picture results on rank 0.

The one real exchange step of the encoder sits behind the in-loop filters of this path: the GPU that
reconstructed and filtered a REFERENCE picture publishes it to the GPUs that will predict from it (inter-prediction
interpolation, temporal filter).  `publish_reference()` is that step: one broadcast of the whole padded picture.
"""
import torch
import torch.distributed as dist


def segment(n_pictures, world, rank, lookback=2, lookahead=2):
    """Pictures [first, last) are analysed by `rank`; [ctx_first, ctx_last) must be resident on it.
    Only pictures with a full set of references are analysed: indices lookback .. n_pictures - lookahead - 1."""
    lo, hi = lookback, n_pictures - lookahead
    n = max(hi - lo, 0)
    base, rem = divmod(n, world)
    first = lo + rank * base + min(rank, rem)
    last = first + base + (1 if rank < rem else 0)
    if first >= last:
        return first, first, first, first
    return first, last, first - lookback, last + lookahead


def max_over_ranks(value, device="cpu"):
    """The timing reduction of the bench contract (slowest rank defines the step time)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def analyse_segment(analyse_picture, n_pictures, world, rank, lookback=2, lookahead=2):
    """Run `analyse_picture(i)` for this rank's pictures; returns {picture index: result}."""
    first, last, _, _ = segment(n_pictures, world, rank, lookback, lookahead)
    return {i: analyse_picture(i) for i in range(first, last)}


def gather_on_root(local_results):
    """Collect the per-picture results of all ranks on rank 0 (control-plane sized data: a few KB per picture)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(local_results)
    parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(local_results, parts, dst=0)
    if dist.get_rank() != 0:
        return None
    merged = {}
    for p in parts:
        merged.update(p)
    return merged


def publish_reference(picture, owner, stream=None, async_op=True, producer=None):
    """Broadcast one reconstructed reference picture from its owner rank to every other rank (the reference publishes it at
    rest_process.c:659-660, 732-744 once restoration has finished).

    `picture` is ONE contiguous tensor holding the padded Y/U/V planes back to back (the planes the kernels use are views of
    it): a single large collective per picture instead of three small ones — xGMI rings are per-link bound (≈27 MB for a
    padded 4K 10-bit 4:2:0 picture ≈ 0.2 ms per link).  Only `is_ref` pictures travel.

    Ordering: the in-loop filter kernels that wrote the picture run on a stream the caller handed to the library, not on
    torch's current stream, so nothing orders the broadcast behind them by itself.  Pass that stream (a torch.cuda.Stream or
    ExternalStream) as `producer`: the side `stream` waits for it before the broadcast is enqueued.  With `async_op` the work
    handle is returned; consumers must `work.wait()` (CPU tensors) or make their stream wait for the side stream
    (`consumer.wait_stream(stream)`) before predicting from the picture.  Returns None on a single rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    assert picture.is_contiguous()
    raw = picture.view(-1).view(torch.uint8)      # bytes: every backend moves them, whatever the sample type
    if stream is not None:
        if producer is not None:
            stream.wait_stream(producer)
        with torch.cuda.stream(stream):
            return dist.broadcast(raw, src=owner, async_op=async_op)
    return dist.broadcast(raw, src=owner, async_op=async_op)


def layer_of(pic_in_minigop, minigop=32):
    """Temporal layer of a picture inside a hierarchical mini-GOP (display order 1..minigop; `minigop` = the base-layer
    picture): 0 for the base picture, then 1 for minigop/2, 2 for the odd multiples of minigop/4, ...  (the reference's 6-layer
    random-access structure for hierarchical_levels 5, enc_handle.c:4379-4387)."""
    assert 1 <= pic_in_minigop <= minigop and minigop & (minigop - 1) == 0
    layer, step = 0, minigop
    while pic_in_minigop % step:
        step //= 2
        layer += 1
    return layer


def slot_of(pic_in_minigop, minigop=32):
    """Position of a picture in the order the closed-loop stages can start them: base picture 0, then layer by layer
    (display order inside a layer): 0 | 1 | 2 3 | 4..7 | 8..15 | 16..31  (svt_hip_shard_slot)."""
    layer = layer_of(pic_in_minigop, minigop)
    if layer == 0:
        return 0
    step = minigop >> layer               # pictures of this layer: odd multiples of step
    return (1 << (layer - 1)) + (pic_in_minigop // step) // 2


def layer_aware_owner(pic_in_minigop, world, minigop=32):
    """the closed-loop stages of a mini-GOP can only run as many pictures at once as the current layer holds
    (1, 1, 2, 4, 8, 16 for a 32-picture mini-GOP), so the pictures of ONE layer go to DIFFERENT GPUs, and consecutive layers
    continue the round-robin: owner = slot mod world.  Open-loop stages ignore this (any GPU can take any picture)."""
    return slot_of(pic_in_minigop, minigop) % world


def owner_in_sequence(picture_number, world, minigop=32):
    """svt_hip_shard_owner_gop: display-order picture number over the whole sequence (0 = key picture); rotated by the
    mini-GOP index so that base-layer pictures alternate over the GPUs."""
    if picture_number == 0:
        return 0
    mg = (picture_number - 1) // minigop
    return (slot_of(picture_number - mg * minigop, minigop) + mg + 1) % world


class ReferencePublisher:
    """bench.py --gpus N: every rank owns one reconstructed 4K reference picture (padded Y/U/V in one allocation) and the
    ranks take turns publishing theirs, one broadcast per step on a side stream that overlaps the next step's kernels.
    transport "torch": torch.distributed.broadcast (RCCL through PyTorch, the default); "c": the library's own
    svt_hip_publish_reference over its own RCCL communicator (include/svt_hip_shard.h), falling back to "torch" if the
    communicator cannot be created."""

    def __init__(self, width, height, bit_depth, device, rank, world, pad=160, transport="torch", lib=None):
        bps = 1 if bit_depth == 8 else 2
        luma = (width + 2 * pad) * (height + 2 * pad)
        chroma = (width // 2 + pad) * (height // 2 + pad)
        self.nbytes = (luma + 2 * chroma) * bps
        self.picture = torch.full((self.nbytes,), rank + 1, dtype=torch.uint8, device=device)
        self.side = torch.cuda.Stream(device=device)
        self.rank, self.world, self.pending = rank, world, None
        self.transport, self.lib, self.comm, self.done = "torch", lib, None, None
        if transport == "c" and lib is not None:
            import ctypes as C
            # Every rank walks the same sequence of collectives whatever fails locally: rank 0 always broadcasts (None when
            # it could not make an id), every rank that got an id joins the communicator, and the transport is AGREED by an
            # all-reduce(min) of the per-rank success flags -- a rank never ends up in a different collective than the others.
            ident = None
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                if lib.svt_hip_comm_get_unique_id(buf) == 0:
                    ident = bytes(buf)
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            ok, comm = 0, C.c_void_p()
            if box[0] is not None:
                ok = int(lib.svt_hip_comm_create((C.c_uint8 * 128).from_buffer_copy(box[0]), world, rank, C.byref(comm)) == 0)
            flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.comm, self.done, self.transport = comm, C.c_void_p(), "c"
            elif ok:
                lib.svt_hip_comm_destroy(comm)

    def publish(self, producer, owner):
        if self.transport == "c":
            import ctypes as C
            rc = self.lib.svt_hip_publish_reference(C.c_void_p(self.picture.data_ptr()), C.c_size_t(self.nbytes), owner, self.comm,
                                                    C.c_void_p(producer.cuda_stream), C.c_void_p(self.side.cuda_stream), C.byref(self.done))
            assert rc == 0, self.lib.svt_hip_last_error().decode()
            return
        if self.pending is not None:
            self.pending.wait()               # the previous broadcast: its buffer is about to be reused
        self.pending = publish_reference(self.picture, owner, stream=self.side, async_op=True, producer=producer)

    def finish(self):
        if self.transport == "c":
            self.side.synchronize()
            return
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
