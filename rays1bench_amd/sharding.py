"""Host-side index arithmetic of the multi-device decomposition (no compute).

The frame is cut into tile_w x tile_h tiles (the reference's 32 x 32 work items,
rayweek1.cpp:855-856 / tiles_required :61-68); tile t belongs to shard t % num_shards
(interleaved, because cost is spatially uneven — SURVEY.md §8e).  Each shard produces one
dense block of `tiles_per_shard` padded tiles (include/rays1.h r1_render_shard_device); the
blocks of all shards, concatenated in shard order, are what the RCCL all-gather returns, and
`assemble` is the host mirror of r1_assemble_device.  bench.py and the world_size-2 gloo test
share these functions."""
import numpy as np


def tiles(width, height, tile_w=32, tile_h=32):
    tx = (width + tile_w - 1) // tile_w
    ty = (height + tile_h - 1) // tile_h
    return tx, ty


def tiles_per_shard(width, height, num_shards, tile_w=32, tile_h=32):
    tx, ty = tiles(width, height, tile_w, tile_h)
    return (tx * ty + num_shards - 1) // num_shards


def block_bytes(width, height, num_shards, tile_w=32, tile_h=32):
    return tiles_per_shard(width, height, num_shards, tile_w, tile_h) * tile_w * tile_h * 3


def shard_tiles(width, height, shard, num_shards, tile_w=32, tile_h=32):
    tx, ty = tiles(width, height, tile_w, tile_h)
    return list(range(shard, tx * ty, num_shards))


def pack_block(image, shard, num_shards, tile_w=32, tile_h=32):
    """(h, w, 3) uint8 image -> this shard's dense block (flat uint8), padding zeroed."""
    h, w, _ = image.shape
    tx, _ = tiles(w, h, tile_w, tile_h)
    per = tiles_per_shard(w, h, num_shards, tile_w, tile_h)
    block = np.zeros((per, tile_h, tile_w, 3), np.uint8)
    for lt, t in enumerate(shard_tiles(w, h, shard, num_shards, tile_w, tile_h)):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        sub = image[y0:y0 + tile_h, x0:x0 + tile_w]
        block[lt, :sub.shape[0], :sub.shape[1]] = sub
    return block.reshape(-1)


def assemble(blocks, width, height, num_shards, tile_w=32, tile_h=32):
    """Concatenated shard blocks (flat uint8, shard-major) -> (h, w, 3) image."""
    tx, ty = tiles(width, height, tile_w, tile_h)
    per = tiles_per_shard(width, height, num_shards, tile_w, tile_h)
    b = np.asarray(blocks, np.uint8).reshape(num_shards, per, tile_h, tile_w, 3)
    out = np.zeros((height, width, 3), np.uint8)
    for t in range(tx * ty):
        x0, y0 = (t % tx) * tile_w, (t // tx) * tile_h
        th, tw = min(tile_h, height - y0), min(tile_w, width - x0)
        out[y0:y0 + th, x0:x0 + tw] = b[t % num_shards, t // num_shards, :th, :tw]
    return out


RECORD_TRAILER = 8  # bytes at the end of a shard's record: its uint64 ray count


def record_bytes(width, height, num_shards, tile_w=32, tile_h=32):
    """One rank's gather record (include/rays1.h r1_shard_record_bytes): the dense tile block, padded to a
    multiple of 8 bytes so that the count is an aligned 64-bit word for any tile size, then the 8-byte ray count."""
    return ((block_bytes(width, height, num_shards, tile_w, tile_h) + 7) & ~7) + RECORD_TRAILER


def gather_records(dist, record, gathered):
    """The one exchange step of a frame: ONE all-gather (RCCL on the GPU box, gloo in the CPU
    tests) of every rank's record = tile block + ray count, so pixels and counts travel together
    (the reference sums `out_num_rays` over threads after the join, rayweek1.cpp:809-813).
    `record`/`gathered` are uint8 torch tensors on the rank's device."""
    dist.all_gather_into_tensor(gathered, record)


def make_record(block, rays):
    """numpy: dense tile block (flat uint8) + ray count -> one gather record."""
    block = np.asarray(block, np.uint8)
    pad = np.zeros((-block.size) % 8, np.uint8)
    return np.concatenate([block, pad, np.array([rays], np.uint64).view(np.uint8)])


def assemble_records(gathered, width, height, num_shards, tile_w=32, tile_h=32):
    """Host mirror of r1_assemble_device_strided over gathered records; returns (image, total rays)."""
    g = np.asarray(gathered, np.uint8).reshape(num_shards, -1)
    nblock = block_bytes(width, height, num_shards, tile_w, tile_h)
    assert g.shape[1] == record_bytes(width, height, num_shards, tile_w, tile_h)
    blocks = g[:, :nblock].reshape(-1)
    rays = int(np.ascontiguousarray(g[:, -RECORD_TRAILER:]).view(np.uint64).sum())
    return assemble(blocks, width, height, num_shards, tile_w, tile_h), rays


def assemble_records_batch(gathered, n_frames, width, height, num_shards, tile_w=32, tile_h=32):
    """Host mirror of r1_assemble_device_records_batch: `gathered` is what ONE all-gather of every rank's n_frames records
    returns, [rank][frame][record]; returns [(image, total rays)] per frame."""
    rec = record_bytes(width, height, num_shards, tile_w, tile_h)
    g = np.asarray(gathered, np.uint8).reshape(num_shards, n_frames, rec)
    return [assemble_records(np.ascontiguousarray(g[:, f, :]).reshape(-1), width, height, num_shards, tile_w, tile_h) for f in range(n_frames)]


def total_rays(gathered, num_shards):
    """Sum of the ray counts in the trailers of a gathered buffer (torch uint8 tensor)."""
    import torch
    rec = gathered.numel() // num_shards
    return int(gathered.view(num_shards, rec)[:, rec - RECORD_TRAILER:].contiguous().view(torch.int64).sum().item())
