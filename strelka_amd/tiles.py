"""Multi-GPU pixel-tile sharding (new functionality: the reference is single-process, single-GPU).

Every pixel is independent (sampler state depends only on (x, y, sampleIndex, sppTotal): RandomSampler.h:130-137;
accumulation is per pixel: OptixRender.cu:60-78), so the frame is cut into square tiles and rank r of W renders
tiles t = r (mod W) of the row-major tile list for ALL sub-frames.  Sharding by SAMPLES would change the image (the
accumulator is a non-linear, order-dependent LDR-space lerp), sharding by tiles reproduces the single-GPU image bit
for bit.  One collective per frame: a gather of the float4 tile accumulators to rank 0 (RCCL over xGMI; every sender
uses its own link into the root), followed by a de-tiling scatter on the root.
"""
import numpy as np


def tile_grid(width, height, tile_size):
    """(x0, y0) of every tile, row-major; partial tiles at the right/top edge are included."""
    xs = np.arange(0, width, tile_size, dtype=np.uint32)
    ys = np.arange(0, height, tile_size, dtype=np.uint32)
    return np.stack([np.tile(xs, len(ys)), np.repeat(ys, len(xs))], 1).astype(np.uint32)


def assign_tiles(width, height, tile_size, world_size, rank):
    """Tiles owned by `rank`: interleaved round-robin so that expensive image regions spread over all GPUs."""
    return np.ascontiguousarray(tile_grid(width, height, tile_size)[rank::world_size])


def max_tiles_per_rank(width, height, tile_size, world_size):
    n = len(tile_grid(width, height, tile_size))
    return (n + world_size - 1) // world_size


def gather_tiles(local_tiles, world_size, rank, dist=None, dst=0, out=None):
    """One gather of equally sized (padded) tile-accumulator tensors to `dst`.
    local_tiles: torch tensor [max_tiles, tile*tile, 4] float32 (rows past this rank's tile count are padding).
    out: optional preallocated [world_size, max_tiles, tile*tile, 4] tensor on dst (the parts are written into its slices,
    so the root can de-tile everything with one launch).  Returns the list of per-rank tensors on dst, None elsewhere."""
    import torch

    if world_size == 1 or dist is None:
        return [local_tiles]
    if local_tiles.is_cuda and dist.get_backend() == "gloo":
        # (test configuration: several ranks on one GPU, no RCCL) stage through the host
        host = local_tiles.cpu()
        parts = [torch.empty_like(host) for _ in range(world_size)] if rank == dst else None
        dist.gather(host, gather_list=parts, dst=dst)
        if rank != dst:
            return None
        if out is not None:
            for r, p in enumerate(parts):
                out[r].copy_(p)
            return [out[r] for r in range(world_size)]
        return [t.to(local_tiles.device) for t in parts]
    if rank == dst:
        parts = [out[r] for r in range(world_size)] if out is not None else [torch.empty_like(local_tiles) for _ in range(world_size)]
    else:
        parts = None
    dist.gather(local_tiles, gather_list=parts, dst=dst)
    if local_tiles.is_cuda:
        # the collective runs on RCCL's stream and the consumer (skh_scatter_tiles) on the renderer's own stream:
        # finish the gather before handing the buffers over
        torch.cuda.current_stream().synchronize()
    return parts
