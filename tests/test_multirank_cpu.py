"""world_size-2 (and 3) gloo tests of the N > 1 path on CPU: the tile -> rank split, the dense
record layout (tile block + ray count), the one all-gather of a frame
(rays1bench_amd.sharding.gather_records, the same function bench.py calls with RCCL) and the
assembly.  The device kernels cannot run
here, so each rank fills its block with the ORACLE's render of exactly its shard; the checker
is the oracle's render of the whole frame."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

import rays1bench_amd as r1  # noqa: E402
from rays1bench_amd import binding, sharding  # noqa: E402
import r1o  # noqa: E402

W, H, SPP, SEED = 150, 100, 2, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = r1.create_medium_scene(W, H)
        sa = r1o.SceneArrays.from_c(sc.spheres, sc.camera)
        # this rank's tiles only (oracle honours shard/num_shards exactly like r1_render)
        part, part_rays, _ = r1o.render_frame(sa, r1o.make_params(W, H, SPP, SEED, shard=rank, num_shards=world), nthreads=2)
        p = r1.make_params(W, H, SPP, SEED, shard=rank, num_shards=world)
        nbytes = binding.shard_block_bytes(p)
        assert nbytes == sharding.block_bytes(W, H, world)
        record = torch.from_numpy(sharding.make_record(sharding.pack_block(part, rank, world), part_rays))
        assert record.numel() == nbytes + sharding.RECORD_TRAILER == sharding.record_bytes(W, H, world)
        gathered = torch.zeros(world * record.numel(), dtype=torch.uint8)
        sharding.gather_records(dist, record, gathered)  # the frame's one exchange step
        img, rays = sharding.assemble_records(gathered.numpy(), W, H, world)
        full, full_rays, _ = r1o.render_frame(sa, r1o.make_params(W, H, SPP, SEED), nthreads=2)
        assert rays == full_rays == sharding.total_rays(gathered, world)
        assert img.tobytes() == full.tobytes()
        # every pixel belongs to exactly one rank
        mine = np.zeros((H, W), np.int64)
        tx, _ = sharding.tiles(W, H)
        for t in sharding.shard_tiles(W, H, rank, world):
            mine[(t // tx) * 32:(t // tx) * 32 + 32, (t % tx) * 32:(t % tx) * 32 + 32] += 1
        cover = torch.from_numpy(mine)
        dist.all_reduce(cover)
        assert (cover.numpy() == 1).all()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _rank_batch(rank, world, port, out_dir):
    """Frame batches: every rank contributes n_frames records to ONE all-gather (bench.py's rank path, r1_render_shard_device_batch):
    the gathered buffer is [rank][frame][record]; frame f is seeded SEED + f * STRIDE."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_frames, stride = 3, 5
        sc = r1.create_small_scene(W, H)
        sa = r1o.SceneArrays.from_c(sc.spheres, sc.camera)
        recs = []
        for f in range(n_frames):
            part, part_rays, _ = r1o.render_frame(sa, r1o.make_params(W, H, SPP, SEED + f * stride, shard=rank, num_shards=world), nthreads=2)
            recs.append(sharding.make_record(sharding.pack_block(part, rank, world), part_rays))
        records = torch.from_numpy(np.concatenate(recs))
        assert records.numel() == n_frames * binding.shard_record_bytes(r1.make_params(W, H, SPP, SEED, shard=rank, num_shards=world))
        gathered = torch.zeros(world * records.numel(), dtype=torch.uint8)
        sharding.gather_records(dist, records, gathered)  # the batch's one exchange step
        frames = sharding.assemble_records_batch(gathered.numpy(), n_frames, W, H, world)
        for f, (img, rays) in enumerate(frames):
            full, full_rays, _ = r1o.render_frame(sa, r1o.make_params(W, H, SPP, SEED + f * stride), nthreads=2)
            assert rays == full_rays and img.tobytes() == full.tobytes(), f
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_frame_batches_one_gather_per_batch_gloo(tmp_path):
    mp.spawn(_rank_batch, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_gather_assemble_gloo(world, tmp_path):
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]


def test_sharding_helpers_match_c_abi():
    for (w, h, n) in [(1200, 800, 8), (1200, 800, 1), (77, 45, 3), (31, 33, 2), (1920, 1080, 8)]:
        p = r1.make_params(w, h, 1, shard=0, num_shards=n)
        total, per = binding.tile_count(p)
        tx, ty = sharding.tiles(w, h)
        assert total == tx * ty and per == sharding.tiles_per_shard(w, h, n)
        assert binding.shard_block_bytes(p) == sharding.block_bytes(w, h, n)
        # pack / assemble round trip
        rng = np.random.default_rng(w + h + n)
        img = rng.integers(1, 255, (h, w, 3), dtype=np.uint8)
        blocks = np.concatenate([sharding.pack_block(img, s, n) for s in range(n)])
        assert sharding.assemble(blocks, w, h, n).tobytes() == img.tobytes()


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("w,h,tw,th,frames", [(1200, 800, 32, 32, 1), (300, 200, 32, 32, 4), (77, 45, 5, 7, 2), (1920, 1080, 32, 32, 8)])
def test_in_process_multi_gpu_layout_equals_the_rank_path_mirror(n, w, h, tw, th, frames):
    """r1_multi_layout is the arithmetic csrc/r1_multi.cpp sizes and addresses its buffers with (records, the gathered
    [device][frame][record] buffer, the strided column of counts, the frame records that go to the host).  No device is involved, so
    N = 2, 3, 8 are checked here against the host mirror of the rank path (sharding.py) — the N > 1 run itself is the driver's."""
    from rays1bench_amd import binding
    p = r1.make_params(w, h, 4, 1, tile_w=tw, tile_h=th)
    L = binding.multi_layout(p, n, frames)
    assert L["block_bytes"] == sharding.block_bytes(w, h, n, tw, th)
    assert L["record_bytes"] == sharding.record_bytes(w, h, n, tw, th) and L["record_bytes"] % 8 == 0
    assert L["count_offset"] == L["record_bytes"] - sharding.RECORD_TRAILER and L["count_offset"] % 8 == 0
    assert L["send_bytes"] == frames * L["record_bytes"] and L["gathered_bytes"] == n * L["send_bytes"]
    assert L["frame_record_bytes"] == ((w * h * 3 + 7) & ~7) + 8 and L["frame_count_offset"] == L["frame_record_bytes"] - 8
    assert L["host_bytes"] == frames * L["frame_record_bytes"] and L["counts_pitch"] == L["send_bytes"]
    # a synthetic gathered buffer laid out by these numbers reassembles through the mirror: every tile in its place, counts summed
    rng = np.random.default_rng(n * 1000 + w)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(frames)]
    gathered = np.zeros(L["gathered_bytes"], np.uint8)
    for d in range(n):
        for f in range(frames):
            rec = sharding.make_record(sharding.pack_block(imgs[f], d, n, tw, th), 1000 * f + d)
            assert rec.size == L["record_bytes"]
            at = (d * frames + f) * L["record_bytes"]
            gathered[at:at + rec.size] = rec
    # the strided column of counts of frame 0, as hipMemcpy2D reads it (pitch = counts_pitch, 8 bytes wide, n rows)
    col = np.stack([gathered[d * L["counts_pitch"] + L["count_offset"]:d * L["counts_pitch"] + L["count_offset"] + 8] for d in range(n)])
    assert [int(x) for x in np.ascontiguousarray(col).view(np.uint64).reshape(-1)] == list(range(n))
    for f, (img, rays) in enumerate(sharding.assemble_records_batch(gathered, frames, w, h, n, tw, th)):
        assert img.tobytes() == imgs[f].tobytes() and rays == sum(1000 * f + d for d in range(n))
