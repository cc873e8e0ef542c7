"""Baseline JPEG reader (and a small writer for the tests), numpy + standard library only.

The reference loads every texture with `stbi_load(file, &w, &h, &c, STBI_rgb_alpha)` (OptixRender.cpp:1191-1198); most glTF
assets ship JPEG colour maps.  This module decodes what those files are in practice: baseline sequential DCT (SOF0), 8-bit,
Huffman, 1 or 3 components with 1x1 / 2x1 / 1x2 / 2x2 chroma subsampling, restart intervals, JFIF YCbCr.  Progressive,
arithmetic-coded, 12-bit and CMYK files are rejected with a clear error.  Decoding follows ITU T.81; the inverse DCT is done
in floating point and chroma is upsampled by replication, so a pixel may differ by +-1..2 from stb_image's integer IDCT and
smoothed upsampling (stb_image itself is not in the reference tree, so that arithmetic cannot be pinned here).
Returns HxWx4 uint8, alpha = 255, rows top to bottom -- the layout `skh_set_textures` takes.
"""
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                   62, 63])
_k = np.arange(8)
_C = np.cos((2 * _k[:, None] + 1) * _k[None, :] * np.pi / 16) * np.where(_k == 0, np.sqrt(1 / 8), np.sqrt(2 / 8))[None, :]  # [x][u]


class JpegError(ValueError):
    pass


class _Huff:
    """canonical Huffman table from the 16 code-length counts + symbols (T.81 annex C); lookup by (length, code)"""

    def __init__(self, counts, symbols):
        self.table = {}
        code, k = 0, 0
        for length in range(1, 17):
            for _ in range(counts[length - 1]):
                self.table[(length, code)] = symbols[k]
                code += 1
                k += 1
            code <<= 1


class _Bits:
    def __init__(self, data):
        self.data, self.pos, self.acc, self.n = data, 0, 0, 0

    def _fill(self):
        d = self.data
        while self.n <= 24:
            if self.pos >= len(d):
                b = 0  # past the end of the segment: zeros (a well-formed scan never needs them)
            else:
                b = d[self.pos]
                self.pos += 1
                if b == 0xFF and self.pos < len(d) and d[self.pos] == 0:
                    self.pos += 1  # stuffed zero after a data byte 0xFF
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8

    def bit(self):
        if self.n == 0:
            self._fill()
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        if k == 0:
            return 0
        while self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def decode(self, huff):
        code = 0
        for length in range(1, 17):
            code = (code << 1) | self.bit()
            s = huff.table.get((length, code))
            if s is not None:
                return s
        raise JpegError("bad Huffman code")


def _extend(v, t):
    return v if t == 0 or v >= (1 << (t - 1)) else v - (1 << t) + 1


def decode_jpeg(blob, name="<memory>"):
    if blob[:2] != b"\xff\xd8":
        raise JpegError(f"{name}: not a JPEG file")
    pos = 2
    qt, hdc, hac = {}, {}, {}
    frame, restart = None, 0
    while pos + 4 <= len(blob):
        if blob[pos] != 0xFF:
            raise JpegError(f"{name}: marker expected at byte {pos}")
        marker = blob[pos + 1]
        pos += 2
        if marker == 0xFF:
            pos -= 1
            continue
        if marker in (0xD8, 0x01) or 0xD0 <= marker <= 0xD7:
            continue
        if marker == 0xD9:
            break
        (length,) = struct.unpack_from(">H", blob, pos)
        seg = blob[pos + 2:pos + length]
        if marker == 0xDB:
            k = 0
            while k < len(seg):
                pq, tq = seg[k] >> 4, seg[k] & 15
                if pq:
                    vals = np.frombuffer(seg[k + 1:k + 129], ">u2").astype(np.float64)
                    k += 129
                else:
                    vals = np.frombuffer(seg[k + 1:k + 65], np.uint8).astype(np.float64)
                    k += 65
                q = np.zeros(64)
                q[ZIGZAG] = vals
                qt[tq] = q
        elif marker == 0xC4:
            k = 0
            while k < len(seg):
                tc, th = seg[k] >> 4, seg[k] & 15
                counts = list(seg[k + 1:k + 17])
                n = sum(counts)
                (hac if tc else hdc)[th] = _Huff(counts, list(seg[k + 17:k + 17 + n]))
                k += 17 + n
        elif marker == 0xC0 or marker == 0xC1:
            prec, h, w, nc = struct.unpack_from(">BHHB", seg, 0)
            if prec != 8:
                raise JpegError(f"{name}: {prec}-bit samples are not supported")
            comps = [(seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15, seg[8 + 3 * i]) for i in range(nc)]
            frame = (w, h, comps)
        elif marker in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError(f"{name}: only baseline sequential JPEG is supported (SOF marker 0x{marker:02x})")
        elif marker == 0xDD:
            (restart,) = struct.unpack_from(">H", seg, 0)
        elif marker == 0xDA:
            if frame is None:
                raise JpegError(f"{name}: scan before frame header")
            ns = seg[0]
            sel = {seg[1 + 2 * i]: (seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)}
            return _decode_scan(blob, pos + length, frame, sel, qt, hdc, hac, restart, name)
        pos += length
    raise JpegError(f"{name}: no scan found")


def _decode_scan(blob, start, frame, sel, qt, hdc, hac, restart, name):
    w, h, comps = frame
    if len(comps) not in (1, 3):
        raise JpegError(f"{name}: {len(comps)}-component JPEG (CMYK?) is not supported")
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux, mcuy = (w + 8 * hmax - 1) // (8 * hmax), (h + 8 * vmax - 1) // (8 * vmax)
    planes = [np.zeros((mcuy * c[2] * 8, mcux * c[1] * 8)) for c in comps]
    # entropy-coded data: cut at restart markers
    data = blob[start:]
    segments, cur, i = [], bytearray(), 0
    while i < len(data):
        b = data[i]
        if b == 0xFF and i + 1 < len(data):
            n = data[i + 1]
            if n == 0:
                cur += b"\xff\x00"
                i += 2
                continue
            if 0xD0 <= n <= 0xD7:
                segments.append(bytes(cur))
                cur = bytearray()
                i += 2
                continue
            if n == 0xFF:
                i += 1
                continue
            break  # EOI or another marker
        cur.append(b)
        i += 1
    segments.append(bytes(cur))
    pred = [0] * len(comps)
    seg_i, bits, in_seg = 0, _Bits(segments[0]), 0
    for my in range(mcuy):
        for mx in range(mcux):
            if restart and in_seg == restart:
                seg_i += 1
                if seg_i >= len(segments):
                    raise JpegError(f"{name}: missing restart segment")
                bits, in_seg, pred = _Bits(segments[seg_i]), 0, [0] * len(comps)
            in_seg += 1
            for ci, (cid, ch, cv, tq) in enumerate(comps):
                td, ta = sel[cid]
                for by in range(cv):
                    for bx in range(ch):
                        coef = np.zeros(64)
                        t = bits.decode(hdc[td])
                        pred[ci] += _extend(bits.bits(t), t)
                        coef[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = bits.decode(hac[ta])
                            r, s = rs >> 4, rs & 15
                            if s == 0:
                                if r != 15:
                                    break
                                k += 16
                                continue
                            k += r
                            if k > 63:
                                raise JpegError(f"{name}: coefficient index out of range")
                            coef[ZIGZAG[k]] = _extend(bits.bits(s), s)
                            k += 1
                        block = (_C @ (coef * qt[tq]).reshape(8, 8) @ _C.T) + 128.0  # rows = y (v), cols = x (u)
                        y0, x0 = (my * cv + by) * 8, (mx * ch + bx) * 8
                        planes[ci][y0:y0 + 8, x0:x0 + 8] = block
    out = np.full((h, w, 4), 255, np.uint8)
    full = []
    for (cid, ch, cv, tq), p in zip(comps, planes):
        p = np.repeat(np.repeat(p, vmax // cv, axis=0), hmax // ch, axis=1)[:h, :w]
        full.append(p)
    if len(comps) == 1:
        g = np.clip(np.floor(full[0] + 0.5), 0, 255).astype(np.uint8)
        out[..., 0] = out[..., 1] = out[..., 2] = g
    else:
        y, cb, cr = full[0], full[1] - 128.0, full[2] - 128.0
        rgb = np.stack([y + 1.402 * cr, y - 0.344136 * cb - 0.714136 * cr, y + 1.772 * cb], -1)
        out[..., :3] = np.clip(np.floor(rgb + 0.5), 0, 255).astype(np.uint8)
    return out


def load_jpeg(path):
    with open(path, "rb") as f:
        return decode_jpeg(f.read(), path)


# ---- a small baseline encoder (4:4:4 or 4:2:0, the Annex K tables), used by the tests to make input files ----
_QY = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112,
                100, 103, 99], np.float64)
_QC = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] +
               [99] * 32, np.float64)
_DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
_DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
_AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
         [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
          0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
          0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
          0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
          0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
          0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
          0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
          0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
_AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
         [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
          0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
          0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
          0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
          0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
          0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
          0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
          0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def _codes(counts, symbols):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            out[symbols[k]] = (length, code)
            code += 1
            k += 1
        code <<= 1
    return out


def save_jpeg(path, image, quality_scale=0.5, subsample=False, restart_interval=0):
    """image: HxWx3 (or HxWx4, alpha dropped) uint8.  quality_scale multiplies the Annex K quantisation tables."""
    a = np.asarray(image)[..., :3].astype(np.float64)
    h, w = a.shape[:2]
    y = 0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2]
    cb = 128 - 0.168736 * a[..., 0] - 0.331264 * a[..., 1] + 0.5 * a[..., 2]
    cr = 128 + 0.5 * a[..., 0] - 0.418688 * a[..., 1] - 0.081312 * a[..., 2]
    hs = 2 if subsample else 1
    mw, mh = 8 * hs, 8 * hs
    W2, H2 = -(-w // mw) * mw, -(-h // mh) * mh

    def pad(p):
        return np.pad(p, ((0, H2 - h), (0, W2 - w)), mode="edge")

    y, cb, cr = pad(y), pad(cb), pad(cr)
    if subsample:
        cb = cb.reshape(H2 // 2, 2, W2 // 2, 2).mean((1, 3))
        cr = cr.reshape(H2 // 2, 2, W2 // 2, 2).mean((1, 3))
    qy = np.clip(np.floor(_QY * quality_scale + 0.5), 1, 255)
    qc = np.clip(np.floor(_QC * quality_scale + 0.5), 1, 255)
    tabs = [(_codes(*_DC_L), _codes(*_AC_L)), (_codes(*_DC_C), _codes(*_AC_C))]
    acc, nbits, out = 0, 0, bytearray()

    def put(code, length):
        nonlocal acc, nbits
        acc = (acc << length) | code
        nbits += length
        while nbits >= 8:
            b = (acc >> (nbits - 8)) & 0xFF
            out.append(b)
            if b == 0xFF:
                out.append(0)
            nbits -= 8
        acc &= (1 << nbits) - 1

    def flush():
        nonlocal acc, nbits
        if nbits:
            put((1 << (8 - nbits)) - 1, 8 - nbits)

    def category(v):
        av = abs(int(v))
        return av.bit_length()

    def block(p, y0, x0, q, tab, pred):
        b = p[y0:y0 + 8, x0:x0 + 8] - 128.0
        coef = np.floor((_C.T @ b @ _C) / q.reshape(8, 8) + 0.5).astype(np.int64).reshape(64)[ZIGZAG]
        dc = int(coef[0]) - pred
        t = category(dc)
        put(*reversed(tab[0][t]))
        if t:
            put(dc if dc >= 0 else dc + (1 << t) - 1, t)
        run = 0
        last = int(np.max(np.nonzero(coef)[0])) if np.any(coef[1:]) else 0
        for k in range(1, last + 1):
            v = int(coef[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                put(*reversed(tab[1][0xF0]))
                run -= 16
            s = category(v)
            put(*reversed(tab[1][(run << 4) | s]))
            put(v if v >= 0 else v + (1 << s) - 1, s)
            run = 0
        if last < 63:
            put(*reversed(tab[1][0x00]))
        return int(coef[0])

    preds, count, rst = [0, 0, 0], 0, 0
    for my in range(H2 // mh):
        for mx in range(W2 // mw):
            if restart_interval and count and count % restart_interval == 0:
                flush()
                out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                preds = [0, 0, 0]
            count += 1
            for by in range(hs):
                for bx in range(hs):
                    preds[0] = block(y, my * mh + 8 * by, mx * mw + 8 * bx, qy, tabs[0], preds[0])
            preds[1] = block(cb, my * 8, mx * 8, qc, tabs[1], preds[1])
            preds[2] = block(cr, my * 8, mx * 8, qc, tabs[1], preds[2])
    flush()

    def seg(marker, payload):
        return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload

    def dqt(i, q):
        return bytes([i]) + bytes(int(v) for v in q.reshape(64)[ZIGZAG])

    def dht(tc, th, spec):
        return bytes([(tc << 4) | th]) + bytes(spec[0]) + bytes(spec[1])

    hdr = b"\xff\xd8" + seg(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0") + seg(0xDB, dqt(0, qy)) + seg(0xDB, dqt(1, qc))
    hdr += seg(0xC0, struct.pack(">BHHB", 8, h, w, 3) + bytes([1, (hs << 4) | hs, 0, 2, 0x11, 1, 3, 0x11, 1]))
    hdr += seg(0xC4, dht(0, 0, _DC_L)) + seg(0xC4, dht(1, 0, _AC_L)) + seg(0xC4, dht(0, 1, _DC_C)) + seg(0xC4, dht(1, 1, _AC_C))
    if restart_interval:
        hdr += seg(0xDD, struct.pack(">H", restart_interval))
    hdr += seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    with open(path, "wb") as f:
        f.write(hdr + bytes(out) + b"\xff\xd9")
