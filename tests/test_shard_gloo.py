"""CPU, world_size 2 over gloo: the picture sharding of the multi-GPU path gives exactly the single-process result.
The compute inside each rank is the oracle (test-only) — what is under test is the host-side sharding logic that
bench.py / a multi-GPU integration uses: segment boundaries, context pictures, result gathering, timing reduction."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import me_cases as M
from svtav1_hip import frames, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H, N = 192, 128, 11
L0, L1 = (-1, -2), (1, 2)


def analyse_factory():
    import pyorc
    orc = pyorc.oracle()
    clip = M.make_clip("pan", W, H, N, seed=3)

    def analyse(i, lo=0, hi=N):
        # a rank only builds pyramids for its resident window [lo, hi)
        pyrs = {k: None for k in range(N)}
        built = M.build_pyramids(orc, clip[lo:hi])
        for k, p in zip(range(lo, hi), built):
            pyrs[k] = p
        prm = M.scenario_params("m8_360p_tl2", i, [i + o for o in L0], [i + o for o in L1])
        return M.run_cpu(orc.orc_me_frame_range, prm, pyrs, i, [i + o for o in L0], [i + o for o in L1], W, H)
    return analyse


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        analyse = analyse_factory()
        first, last, c0, c1 = shard.segment(N, world, rank)
        # the C-ABI form of the same decisions (include/svt_hip_shard.h), what a C host calls
        import ctypes as C
        from svtav1_hip import abi
        lib = abi.load()
        seg = (C.c_uint32 * 4)()
        lib.svt_hip_shard_segment(N, world, rank, 2, 2, seg)
        assert tuple(seg) == (first, last, c0, c1)
        lib.svt_hip_shard_owner.restype = C.c_uint32
        assert [lib.svt_hip_shard_owner(i, 32, world) for i in range(1, 33)] == [shard.layer_aware_owner(i, world) for i in range(1, 33)]
        local = shard.analyse_segment(lambda i: analyse(i, c0, c1), N, world, rank)
        assert sorted(local) == list(range(first, last))
        shard.barrier()
        slowest = shard.max_over_ranks(1.0 + rank)
        merged = shard.gather_on_root(local)
        # the exchange step: rank 1 owns a reconstructed reference picture (one flat buffer, planes are views of it)
        import torch
        rng = np.random.default_rng(5)
        ref_pic = rng.integers(0, 1024, size=3 * 96 * 64 // 2, dtype=np.int64).astype(np.int16)
        pic = torch.from_numpy(ref_pic.copy() if rank == 1 else np.zeros_like(ref_pic))
        work = shard.publish_reference(pic, owner=1)
        work.wait()
        assert np.array_equal(pic.numpy(), ref_pic)
        luma = pic[:96 * 64].view(64, 96)       # a plane view of the published picture
        assert int(luma[3, 5]) == int(ref_pic[3 * 96 + 5])
        if rank == 0:
            q.put((slowest, {i: {k: v.copy() for k, v in r.items()} for i, r in merged.items()}))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_c_abi_assignment_matches_python():
    """svt_hip_shard_layer / _owner / _segment (C) == shard.layer_of / layer_aware_owner / segment (Python)."""
    import ctypes as C
    from svtav1_hip import abi
    lib = abi.load()
    lib.svt_hip_shard_layer.restype = lib.svt_hip_shard_owner.restype = C.c_uint32
    for mg in (8, 16, 32):
        assert [lib.svt_hip_shard_layer(i, mg) for i in range(1, mg + 1)] == [shard.layer_of(i, mg) for i in range(1, mg + 1)]
        for world in (1, 2, 3, 4, 8):
            own = [lib.svt_hip_shard_owner(i, mg, world) for i in range(1, mg + 1)]
            assert own == [shard.layer_aware_owner(i, world, mg) for i in range(1, mg + 1)]
            # pictures of one layer sit on different GPUs as long as the layer has at most `world` of them
            for layer in range(1, mg.bit_length()):
                members = [own[i - 1] for i in range(1, mg + 1) if shard.layer_of(i, mg) == layer]
                assert len(set(members)) == min(len(members), world)
            # balanced: no GPU collects the first picture of every layer (the round-2 assignment put 6 of 32 on GPU 0)
            if mg % world == 0:
                assert [own.count(g) for g in range(world)] == [mg // world] * world
        # whole sequence: base-layer pictures (multiples of the mini-GOP size) alternate over the GPUs
        lib.svt_hip_shard_owner_gop.restype = C.c_uint32
        lib.svt_hip_shard_owner_gop.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        for world in (2, 4, 8):
            seq = [lib.svt_hip_shard_owner_gop(n, mg, world) for n in range(0, 8 * mg + 1)]
            assert seq == [shard.owner_in_sequence(n, world, mg) for n in range(0, 8 * mg + 1)]
            bases = [seq[m * mg] for m in range(1, 9)]
            assert all(bases[i] != bases[i + 1] for i in range(7)), bases
            if mg % world == 0:
                assert [seq[1:].count(g) for g in range(world)] == [8 * mg // world] * world
    seg = (C.c_uint32 * 4)()
    for n in (5, 11, 20, 64):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lib.svt_hip_shard_segment(n, world, r, 2, 2, seg)
                assert tuple(seg) == shard.segment(n, world, r)
    if lib.svt_hip_device_count() == 0:      # no GPU: the communicator entry points fail loudly instead of pretending
        ident = (C.c_uint8 * 128)()
        assert lib.svt_hip_comm_get_unique_id(ident) == abi.SVT_HIP_ERR_NO_DEVICE


def test_segments_partition_the_clip():
    for n in (5, 11, 20, 64):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, last, c0, c1 = shard.segment(n, world, r)
                seen += list(range(first, last))
                if last > first:
                    assert c0 == first - 2 and c1 == last + 2 and c0 >= 0 and c1 <= n
            assert seen == list(range(2, n - 2))
            sizes = [shard.segment(n, world, r)[1] - shard.segment(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_ranks_equal_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    slowest, merged = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest == 2.0
    analyse = analyse_factory()
    assert sorted(merged) == list(range(2, N - 2))
    for i in sorted(merged):
        M.assert_same(analyse(i), merged[i], f"picture {i}")


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the way the driver starts it) must start 2 ranks itself, before any GPU
    call, and print ONE line with n_gpus = 2, every rank having seen 2 ranks in the communicator.  SVTAV1_BENCH_REHEARSAL=launch
    stops each rank after the rank plumbing (no GPU here)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SVTAV1_BENCH_REHEARSAL"] = "launch"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_in_communicator"] == 2 and line["max_over_ranks_check"] == 2.0
    # a rank that dies takes the launcher down with a non-zero exit code instead of hanging the others
    env["SVTAV1_BENCH_REHEARSAL"] = "launch"
    env["SVTAV1_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "rank 1 exited" in r.stderr
