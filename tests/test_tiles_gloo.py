"""N > 1 path on CPU: world_size-2 gloo.  Exercises what bench.py does per frame: round-robin tile ownership,
padded per-rank tile buffers, ONE gather to rank 0, de-tiling on the root, max-over-ranks timing reduction."""
import os
import socket

import numpy as np

from tests.tilehelp import detile_numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from strelka_amd import tiles

W, H, T = 200, 120, 32


def pixel_value(px, py):
    return np.stack([px * 1.0, py * 1.0, px * 1000.0 + py, np.ones_like(px, dtype=np.float64)], -1).astype(np.float32)


def fill_tiles(tile_xy, max_tiles):
    """what a rank's accumulator holds: f(pixel) in Morton slot order, zero padding past its tile count"""
    buf = np.zeros((max_tiles, T * T, 4), np.float32)
    m = np.arange(T * T, dtype=np.uint32)

    def compact(v):
        v = v & 0x55555555
        v = (v ^ (v >> 1)) & 0x33333333
        v = (v ^ (v >> 2)) & 0x0F0F0F0F
        v = (v ^ (v >> 4)) & 0x00FF00FF
        v = (v ^ (v >> 8)) & 0x0000FFFF
        return v

    xl, yl = compact(m).astype(np.int64), compact(m >> 1).astype(np.int64)
    for k, (x0, y0) in enumerate(tile_xy.astype(np.int64)):
        buf[k] = pixel_value(x0 + xl, y0 + yl)
    return buf


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = tiles.assign_tiles(W, H, T, world, rank)
    mt = tiles.max_tiles_per_rank(W, H, T, world)
    local = torch.from_numpy(fill_tiles(mine, mt))
    parts = tiles.gather_tiles(local, world, rank, dist)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py takes the max time over ranks
    if rank == 0:
        img = np.zeros((H, W, 4), np.float32)
        for r, part in enumerate(parts):
            tr = tiles.assign_tiles(W, H, T, world, r)
            detile_numpy(part.numpy()[: len(tr)], tr, T, W, H, out=img)
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        ok = np.array_equal(img, pixel_value(xx, yy)) and float(t.item()) == float(world)
        ret.put(bool(ok))
    else:
        assert parts is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_tile_gather_reassembles_the_frame(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True
