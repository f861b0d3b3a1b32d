#!/usr/bin/env python3
"""Small valid files of every format the scene loader reads, as seeds for fuzz_loaders."""
import io
import os
import struct
import sys
import zlib

import numpy as np
from PIL import Image

out = sys.argv[1]
rng = np.random.RandomState(1)
img = rng.randint(0, 256, size=(19, 23, 4)).astype(np.uint8)
rgba = Image.fromarray(img, "RGBA")
rgb = rgba.convert("RGB")


def save(name, im, fmt, **kw):
    im.save(os.path.join(out, name), fmt, **kw)


save("rgb.png", rgb, "PNG")
save("rgba.png", rgba, "PNG")
save("pal.png", rgb.quantize(16), "PNG")
save("grey1.png", rgb.convert("1"), "PNG")
save("grey16.png", Image.fromarray((rng.rand(9, 11) * 65535).astype(np.uint16)), "PNG")
save("base.jpg", rgb, "JPEG", quality=85, subsampling=2)
save("rst.jpg", rgb, "JPEG", quality=85, subsampling=1, restart_marker_blocks=2)
save("prog.jpg", rgb, "JPEG", quality=85, subsampling=2, progressive=True)
save("grey.jpg", rgb.convert("L"), "JPEG", progressive=True)
save("rgb.bmp", rgb, "BMP")
save("pal.bmp", rgb.quantize(16), "BMP")
save("rgba.bmp", rgba, "BMP")
save("rgb.tga", rgb, "TGA")
save("rle.tga", rgba, "TGA", compression="tga_rle")
save("pal.tga", rgb.quantize(16), "TGA")
save("rgb.ppm", rgb, "PPM")
# PFM and a ZIP-compressed half-float EXR
w, h = 8, 4
pix = rng.rand(h, w, 3).astype("<f4")
open(os.path.join(out, "env.pfm"), "wb").write(b"PF\n%d %d\n-1.0\n" % (w, h) + pix.tobytes())


def exr(compression):
    def attr(name, typ, body):
        return name + b"\0" + typ + b"\0" + struct.pack("<I", len(body)) + body
    chans = b"".join(c + b"\0" + struct.pack("<IBBBBII", 1, 0, 0, 0, 0, 1, 1) for c in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2) + attr(b"channels", b"chlist", chans) + attr(b"compression", b"compression", bytes([compression])) \
        + attr(b"dataWindow", b"box2i", box) + attr(b"displayWindow", b"box2i", box) + attr(b"lineOrder", b"lineOrder", b"\0") \
        + attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1)) + attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0, 0)) \
        + attr(b"screenWindowWidth", b"float", struct.pack("<f", 1)) + b"\0"
    per = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, h, per):
        raw = b"".join(pix[y, :, c].astype("<f2").tobytes() for y in range(y0, min(h, y0 + per)) for c in (2, 1, 0))
        if compression:
            t = np.frombuffer(raw, np.uint8)
            half = (len(t) + 1) // 2
            re = np.concatenate([t[0::2], t[1::2]]).astype(np.int32)
            d = re.copy()
            d[1:] = (re[1:] - re[:-1] + 128 + 256) % 256
            z = zlib.compress(d.astype(np.uint8).tobytes())
            data = z if len(z) < len(raw) else raw
        else:
            data = raw
        blocks.append(struct.pack("<iI", y0, len(data)) + data)
    table_at = len(head)
    offs, at = [], table_at + 8 * len(blocks)
    for b in blocks:
        offs.append(at)
        at += len(b)
    return head + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blocks)


open(os.path.join(out, "none.exr"), "wb").write(exr(0))
open(os.path.join(out, "zip.exr"), "wb").write(exr(3))
open(os.path.join(out, "tri.obj"), "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1\nf 2 4 3\n")
ply = b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n" \
      b"element face 1\nproperty list uchar int vertex_indices\nend_header\n" + np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], "<f4").tobytes() + b"\x03" + np.array([0, 1, 2], "<i4").tobytes()
open(os.path.join(out, "tri.ply"), "wb").write(ply)
print("seeds in", out)
