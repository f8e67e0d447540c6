#!/usr/bin/env python3
"""Regenerates tests/golden/jpeg/: small JPEG files of every flavour the decoder handles, and
jpeg_golden.json with the sha256 of the RGBA bytes that the REFERENCE'S OWN stb_image (oracle/_ref,
compiled verbatim from /root/reference/include) decodes from each of them.

The files are written with Pillow, which the system interpreter does not have; the build container's
/opt/conda interpreter does, so that part runs there:   python tests/golden/make_jpeg_fixtures.py
"""
import hashlib, json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "jpeg")
PIL_PYTHON = "/opt/conda/bin/python"

WRITER = r'''
import sys, numpy as np
from PIL import Image
out = sys.argv[1]
rng = np.random.default_rng(21)
def picture(w, h, mode):
    """Smooth gradients + some noise + a hard edge: exercises DC prediction, AC runs and clamping."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    ch = []
    for k in range({"L": 1, "RGB": 3, "CMYK": 4}[mode]):
        c = 127 + 100 * np.sin(x / (3.0 + k) + k) * np.cos(y / (4.0 + k)) + rng.normal(0, 12, (h, w))
        c[:, w // 2:] += 60 * ((k % 2) * 2 - 1)
        ch.append(np.clip(c, 0, 255).astype(np.uint8))
    return Image.fromarray(ch[0] if mode == "L" else np.dstack(ch), mode)
cases = [
    ("base_444_q90",      (37, 29), "RGB",  dict(quality=90, subsampling=0)),
    ("base_422_q75",      (37, 29), "RGB",  dict(quality=75, subsampling=1)),
    ("base_420_q50",      (37, 29), "RGB",  dict(quality=50, subsampling=2)),
    ("base_420_q100",     (64, 48), "RGB",  dict(quality=100, subsampling=2)),
    ("base_420_q5",       (33, 17), "RGB",  dict(quality=5, subsampling=2)),
    ("base_420_optimize", (40, 40), "RGB",  dict(quality=80, subsampling=2, optimize=True)),
    ("base_420_restart",  (70, 50), "RGB",  dict(quality=70, subsampling=2, restart_marker_blocks=3)),
    ("base_444_restart",  (70, 50), "RGB",  dict(quality=85, subsampling=0, restart_marker_rows=1)),
    ("base_grey",         (31, 23), "L",    dict(quality=80)),
    ("base_1x1",          (1, 1),   "RGB",  dict(quality=90, subsampling=2)),
    ("base_2x3_422",      (2, 3),   "RGB",  dict(quality=90, subsampling=1)),
    ("base_8x8",          (8, 8),   "RGB",  dict(quality=95, subsampling=0)),
    ("base_17x16_420",    (17, 16), "RGB",  dict(quality=60, subsampling=2)),
    ("prog_444",          (37, 29), "RGB",  dict(quality=85, subsampling=0, progressive=True)),
    ("prog_420",          (45, 38), "RGB",  dict(quality=60, subsampling=2, progressive=True)),
    ("prog_422_restart",  (70, 50), "RGB",  dict(quality=75, subsampling=1, progressive=True, restart_marker_blocks=4)),
    ("prog_grey",         (31, 23), "L",    dict(quality=70, progressive=True)),
    ("prog_420_q100",     (24, 40), "RGB",  dict(quality=100, subsampling=2, progressive=True)),
    ("cmyk",              (21, 19), "CMYK", dict(quality=85)),
    ("cmyk_prog",         (21, 19), "CMYK", dict(quality=85, progressive=True)),
]
for name, (w, h), mode, options in cases:
    picture(w, h, mode).save("%s/%s.jpg" % (out, name), "JPEG", **options)
print(len(cases), "files written")
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.run([PIL_PYTHON, "-c", WRITER, OUT], check=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import binding as oracle
    assert oracle.ref_lib() is not None, "oracle/_ref missing: run `make -C oracle ref` where /root/reference exists"
    import numpy as np

    def srgb_to_linear_u8(rgba8):   # what the texture loader stores: gamma_to_linear in float32, truncated to 8 bits
        c = rgba8.astype(np.float32) / np.float32(255.0)
        lin = np.where(c <= np.float32(0.04045), c / np.float32(12.92), ((c + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4))
        return np.clip(lin * np.float32(255.0), 0, 255).astype(np.uint8)

    golden = {}
    for name in sorted(os.listdir(OUT)):
        if not name.endswith(".jpg"):
            continue
        rgba = oracle.ref_stbi_load(os.path.join(OUT, name))
        assert rgba is not None, name
        golden[name] = {"width": int(rgba.shape[1]), "height": int(rgba.shape[0]), "sha256": hashlib.sha256(rgba.tobytes()).hexdigest(),
                        "sha256_linear": hashlib.sha256(srgb_to_linear_u8(rgba).tobytes()).hexdigest()}
    json.dump({"source": "stb_image v2.19 as vendored by the reference, compiled verbatim (oracle/_ref)", "files": golden},
              open(os.path.join(HERE, "jpeg_golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(golden), "digests")


if __name__ == "__main__":
    main()
