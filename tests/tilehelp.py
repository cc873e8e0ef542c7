"""Host restatement of the de-tiling scatter, for the tests only (on the GPU the root uses skh_scatter_tiles).
Slot order inside a tile is Morton(xl, yl) (strelka_amd/csrc/skh_kernels.h: slot_to_pixel)."""
import numpy as np


def detile_numpy(tiles_rgba, tile_xy, tile_size, width, height, out=None):
    """tiles_rgba [n_tiles, tile*tile, 4] + tile origins -> H x W x 4 image"""
    t = np.asarray(tiles_rgba).reshape(len(tile_xy), tile_size * tile_size, 4)
    if out is None:
        out = np.zeros((height, width, 4), np.float32)
    m = np.arange(tile_size * tile_size, dtype=np.uint32)

    def compact(v):
        v = v & 0x55555555
        v = (v ^ (v >> 1)) & 0x33333333
        v = (v ^ (v >> 2)) & 0x0F0F0F0F
        v = (v ^ (v >> 4)) & 0x00FF00FF
        v = (v ^ (v >> 8)) & 0x0000FFFF
        return v

    xl, yl = compact(m), compact(m >> 1)
    for k, (x0, y0) in enumerate(np.asarray(tile_xy, np.int64)):
        px, py = x0 + xl, y0 + yl
        ok = (px < width) & (py < height)
        out[py[ok], px[ok]] = t[k][ok]
    return out
