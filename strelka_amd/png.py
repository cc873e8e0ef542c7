"""Minimal PNG reader / writer (zlib only).

Reader: what the glTF loader needs to fetch `diffuse_texture` / `normalmap_texture` images the way the reference does with
`stbi_load(file, &w, &h, &c, STBI_rgb_alpha)` (OptixRender.cpp:1191-1198): 8-bit grey / grey+alpha / RGB / palette / RGBA,
non-interlaced, returned as HxWx4 uint8 with rows top to bottom (grey -> rgb replicated, missing alpha = 255).
Writer: hdRunner's screenshot (src/hdRunner/main.cpp:407-440) hands a float4 image to Hio with `flipped = true` (row 0 of the
render buffer is the BOTTOM of the picture); `save_png(path, image, flipped=True)` does the same with 8-bit quantisation.
"""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


class PngError(ValueError):
    pass


def _paeth(a, b, c):
    p = a.astype(np.int32) + b.astype(np.int32) - c.astype(np.int32)
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def load_png(path):
    with open(path, "rb") as f:
        return decode_png(f.read(), path)


def decode_png(blob, path="<memory>"):
    """PNG file content (bytes) -> HxWx4 uint8.  Every malformed input ends in PngError (a ValueError): truncated chunks
    (struct.error) and corrupt deflate streams (zlib.error) included, so that one bad texture cannot abort a scene load."""
    try:
        return _decode_png(blob, path)
    except (struct.error, zlib.error, IndexError) as e:
        raise PngError(f"{path}: corrupt PNG ({e})") from e


def _decode_png(blob, path):
    if blob[:8] != _SIG:
        raise PngError(f"{path}: not a PNG file")
    off, idat, palette, trns, hdr = 8, [], None, None, None
    while off + 8 <= len(blob):
        n, tag = struct.unpack_from(">I4s", blob, off)
        data = blob[off + 8:off + 8 + n]
        off += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"PLTE":
            palette = np.frombuffer(data, np.uint8).reshape(-1, 3)
        elif tag == b"tRNS":
            trns = np.frombuffer(data, np.uint8)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
    if hdr is None:
        raise PngError(f"{path}: no IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise PngError(f"{path}: only 8-bit non-interlaced PNGs are supported (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    stride = w * ch
    if len(raw) != h * (stride + 1):
        raise PngError(f"{path}: unexpected amount of image data")
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].copy()
        if ft == 0:
            pass
        elif ft == 2:
            line = (line + prev).astype(np.uint8)
        elif ft == 1:  # Sub: cur[x] = raw[x] + cur[x-1] (mod 256) = a running sum per channel
            line = (np.cumsum(line.reshape(w, ch).astype(np.uint32), axis=0) & 0xFF).astype(np.uint8).reshape(-1)
        elif ft in (3, 4):  # Average / Paeth: the predictor needs the DECODED left neighbour (floor / select: no closed form), so
            # these rows are a per-pixel scan, vectorised over channels only: ~3 us per pixel, i.e. fine for the 256^2 ... 1 K
            # textures of glTF samples, minutes for a 4 K one whose encoder picked these filters for every row
            px, pv = line.reshape(w, ch), prev.reshape(w, ch)
            cur = np.zeros((w, ch), np.uint8)
            left = np.zeros(ch, np.uint8)
            ul = np.zeros(ch, np.uint8)
            for x in range(w):
                if ft == 1:
                    pred = left
                elif ft == 3:
                    pred = ((left.astype(np.uint16) + pv[x].astype(np.uint16)) // 2).astype(np.uint8)
                else:
                    pred = _paeth(left, pv[x], ul)
                cur[x] = (px[x] + pred).astype(np.uint8)
                left, ul = cur[x], pv[x]
            line = cur.reshape(-1)
        else:
            raise PngError(f"{path}: bad filter type {ft}")
        out[y] = line
        prev = line
    px = out.reshape(h, w, ch)
    rgba = np.full((h, w, 4), 255, np.uint8)
    if ctype == 0:
        rgba[..., :3] = px
    elif ctype == 2:
        rgba[..., :3] = px
    elif ctype == 3:
        if palette is None:
            raise PngError(f"{path}: palette image without PLTE")
        rgba[..., :3] = palette[px[..., 0]]
        if trns is not None:
            a = np.full(256, 255, np.uint8)
            a[:len(trns)] = trns
            rgba[..., 3] = a[px[..., 0]]
    elif ctype == 4:
        rgba[..., :3] = px[..., :1]
        rgba[..., 3] = px[..., 1]
    else:
        rgba[...] = px
    return rgba


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def save_png(path, image, flipped=False):
    """image: HxWx3 / HxWx4, uint8 or float (clamped to [0, 1], x 255, rounded).  flipped=True writes row 0 at the bottom."""
    a = np.asarray(image)
    if a.dtype != np.uint8:
        a = np.clip(np.nan_to_num(a.astype(np.float64)), 0.0, 1.0)
        a = (a * 255.0 + 0.5).astype(np.uint8)
    if a.ndim != 3 or a.shape[2] not in (3, 4):
        raise PngError("image must be HxWx3 or HxWx4")
    if flipped:
        a = a[::-1]
    h, w, ch = a.shape
    raw = np.zeros((h, 1 + w * ch), np.uint8)
    raw[:, 1:] = a.reshape(h, w * ch)
    with open(path, "wb") as f:
        f.write(_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 6, 0, 0, 0)) +
                _chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + _chunk(b"IEND", b""))
