"""Tiny dependency-free PNG writer for eyeballing renders (sRGB tonemap)."""
import struct
import zlib

import numpy as np


def write_png(path, rgb, scale=1.0):
    a = np.clip(np.asarray(rgb, np.float64) * scale, 0, None)
    a = np.where(a <= 0.0031308, 12.92 * a, 1.055 * np.power(np.maximum(a, 1e-12), 1 / 2.4) - 0.055)
    a = (np.clip(a, 0, 1) * 255 + 0.5).astype(np.uint8)
    h, w, _ = a.shape
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
