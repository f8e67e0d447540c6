"""Build assets/scenes/sponza_textures_256.tar.xz from the reference's Data/Sponza/textures.

Runs in the development container only (the GPU box has no /root/reference). The 19 diffuse maps
Data/Sponza/scene.xml references (5 more are missing upstream and stay missing -> pink fallback,
SURVEY.md 8c) are box-filtered 4x4 -> 1 in 8-bit space and written as uncompressed TGAs of a
quarter of the side length. scene_path("sponza") unpacks them and replicates every texel 4x4
again, so the texture set the renderer loads has the reference's dimensions (1024x1024 etc.)
and memory footprint; only the finest two mip levels carry less detail than upstream.
"""
import io
import os
import re
import struct
import sys
import tarfile

import numpy as np

SRC = "/root/reference/Data/Sponza"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTOR = 4


def read_tga(path):
    data = open(path, "rb").read()
    id_len, cmap, kind = data[0], data[1], data[2]
    w, h, bpp, desc = struct.unpack("<HHBB", data[12:18])
    assert kind == 2 and cmap == 0 and bpp in (24, 32), (path, kind, bpp)
    px = np.frombuffer(data, np.uint8, w * h * (bpp // 8), 18 + id_len).reshape(h, w, bpp // 8)
    return px, desc


def write_tga(px, desc):
    h, w, c = px.shape
    header = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, c * 8, desc)
    return header + px.tobytes()


def main():
    xml = open(os.path.join(SRC, "scene.xml")).read()
    names = sorted(set(re.findall(r"textures[\\/]+([A-Za-z0-9_]+\.tga)", xml)))
    out = os.path.join(ROOT, "assets", "scenes", "sponza_textures_256.tar.xz")
    kept = 0
    with tarfile.open(out, "w:xz", preset=9) as tar:
        for name in names:
            path = os.path.join(SRC, "textures", name)
            if not os.path.exists(path):
                print("missing upstream (pink fallback):", name)
                continue
            px, desc = read_tga(path)
            h, w, c = px.shape
            small = px.reshape(h // FACTOR, FACTOR, w // FACTOR, FACTOR, c).astype(np.float32).mean(axis=(1, 3))
            small = np.clip(np.floor(small + 0.5), 0, 255).astype(np.uint8)
            blob = write_tga(small, desc)
            info = tarfile.TarInfo("Sponza/textures_quarter/" + name)
            info.size = len(blob)
            tar.addfile(info, io.BytesIO(blob))
            kept += 1
    print("packed %d textures -> %s (%.1f MB)" % (kept, out, os.path.getsize(out) / 1e6))


if __name__ == "__main__":
    sys.exit(main())
