#!/usr/bin/env python3
"""Decode the reference's only published output at its REAL precision (16 bits per channel).

Input : /root/reference/demo.png -- demo2.yml at 16384 spp, 800x600 (README.md:1-3), written as a P3 PPM by
        Image::write (fluxcore/src/image.rs:43-61: `(c * 65535.99) as u16`, maxval 65535) and converted to PNG
        by ImageMagick (tEXt date chunks; the gAMA/cHRM chunks are tags added by that conversion, the samples
        are the PPM's linear values).
Output: tests/golden/demo2_ref_800x600_u16.npy -- uint16 [600][800][3], the file's samples untouched.

Pure zlib + PNG unfiltering (colour type 2, bit depth 16, no interlace): PIL's `.convert("RGB")` silently drops
such a file to 8 bits, which is what the round-1 fixture (demo2_ref_100x75.npy) was made from.
This is data derived from a data file, not reference source.  Run once in the build container (the GPU box has
no /root/reference); the .npy is committed.
"""
import os
import struct
import zlib

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/demo.png"


def decode_png16(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, hdr = 8, [], None
    while p < len(d):
        n, t = struct.unpack(">I4s", d[p:p + 8])
        body = d[p + 8:p + 8 + n]
        assert zlib.crc32(t + body) == struct.unpack(">I", d[p + 8 + n:p + 12 + n])[0], "chunk CRC"
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT":
            idat.append(body)
        p += 12 + n
    w, h, depth, ctype, comp, flt, interlace = hdr
    assert (depth, ctype, comp, flt, interlace) == (16, 2, 0, 0, 0), hdr
    raw = zlib.decompress(b"".join(idat))
    bpp, stride = 6, w * 6
    assert len(raw) == h * (stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=y * (stride + 1) + 1).astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:  # Up
            cur = (line + prev) & 255
        elif ft in (1, 3, 4):  # Sub / Average / Paeth need the left neighbour: sequential per byte lane
            cur = np.zeros(stride, dtype=np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        else:
            raise ValueError(f"filter {ft}")
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, 3, 2).astype(np.uint16)
    return (img[..., 0] << 8) | img[..., 1]  # big-endian samples


if __name__ == "__main__":
    img = decode_png16(SRC)
    assert img.shape == (600, 800, 3) and img.dtype == np.uint16
    np.save(os.path.join(here, "demo2_ref_800x600_u16.npy"), img)
    lo = img & 255
    print("shape", img.shape, "distinct values", len(np.unique(img)), "distinct low bytes", len(np.unique(lo)),
          "mean", (img / 65535.99).mean(axis=(0, 1)))
