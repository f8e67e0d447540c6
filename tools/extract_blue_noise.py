#!/usr/bin/env python3
"""Convert the blue-noise tile table to the binary asset the host library loads.

The 16 tiles (128x128, two 8-bit channels each) are Christoph Peters' CC0 "free blue
noise textures" (LDR_RG01_0..15.png); the reference carries them as a C array
(Src/Util/BlueNoise.cpp, uploaded verbatim at Integrator.cpp:301-303).  This script
re-encodes that DATA as raw little-endian uint16 (low byte = channel x, high byte =
channel y, exactly the bytes the device reads as uchar2) into
assets/blue_noise_16x128x128_rg8.bin.  Run only where /root/reference is mounted.
"""
import re, sys, numpy as np
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Src/Util/BlueNoise.cpp"
dst = sys.argv[2] if len(sys.argv) > 2 else "assets/blue_noise_16x128x128_rg8.bin"
vals = re.findall(r"0x([0-9a-fA-F]{4})", open(src).read())
a = np.array([int(v, 16) for v in vals], dtype="<u2")
assert a.size == 16 * 128 * 128, a.size
a.tofile(dst)
print("wrote", dst, a.size * 2, "bytes")
