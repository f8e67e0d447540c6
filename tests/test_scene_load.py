"""The scene-loading side against the reference's own loaders.

oracle/_ref/libref_scene.so is the reference's Scene / AssetManager / MitsubaLoader / XMLParser / OBJ, PLY, serialized
and hair loaders / TextureLoader (stb_image, stb_dxt, mip maps) / Sky, compiled verbatim from /root/reference/Src
(oracle/ref/ref_scene_harness.cpp). Every test loads the same files with it and with the product's host library and
requires the same scene: the listing of camera / meshes / materials / media (floats compared by bit pattern), every
triangle of every mesh, every texel of every texture level, the sky. Where oracle/_ref was not built (no reference
mount) the digests in tests/golden/scene_golden.json, written from the reference's output by
tests/golden/make_scene_golden.py, stand in for it.
"""
import hashlib
import json
import os
import shutil
import struct

import numpy as np
import pytest

from test_loaders import _ply_bytes, _png_bytes, _serialized_archive

GOLDEN_PATH = os.path.join(os.path.dirname(__file__), "golden", "scene_golden.json")
GOLDEN = json.load(open(GOLDEN_PATH)) if os.path.exists(GOLDEN_PATH) else {}


def write_sky(path, seed=1, width=4, height=2):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (height, width) + rng.integers(1, 255, width * height * 4, dtype=np.uint8).tobytes())
    return str(path)


def decode_bc1(blocks, bw, bh):
    """(bw*bh, 8) BC1 blocks -> (bh*4, bw*4, 4) RGBA8; interpolated colours are the exact thirds rounded to nearest."""
    b = blocks.astype(np.uint32)
    c0 = b[:, 0] | (b[:, 1] << 8); c1 = b[:, 2] | (b[:, 3] << 8)
    idx = b[:, 4] | (b[:, 5] << 8) | (b[:, 6] << 16) | (b[:, 7] << 24)

    def expand(c):
        r, g, bl = (c >> 11) & 31, (c >> 5) & 63, c & 31
        return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (bl << 3) | (bl >> 2), np.full_like(r, 255)], -1).astype(np.int32)
    e0, e1 = expand(c0), expand(c1)
    four = (c0 > c1)[:, None]
    t2 = np.where(four, (2 * e0 + e1 + 1) // 3, (e0 + e1) // 2); t3 = np.where(four, (e0 + 2 * e1 + 1) // 3, 0)
    t2[:, 3] = 255; t3[:, 3] = np.where(four[:, 0], 255, 0)
    palette = np.stack([e0, e1, t2, t3], 1)
    select = ((idx[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3).astype(np.int64)
    texels = np.take_along_axis(palette, select[:, :, None].repeat(4, 2), 1)
    return texels.reshape(bh, bw, 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(bh * 4, bw * 4, 4).astype(np.uint8)


def reference_texture_as_rgba8(r):
    """A reference texture (RGBA8 or BC1 blocks, mip_offsets in bytes) -> list of (h, w, 4) levels + the size its LOD bias sees."""
    levels = []
    if r["format"] == 3:
        w, h = r["width"], r["height"]
        for l, offset in enumerate(r["mip_offsets"]):
            lw, lh = max(w >> l, 1), max(h >> l, 1)
            levels.append(r["data"][offset: offset + lw * lh * 4].reshape(lh, lw, 4))
        return levels, (0, 0)
    assert r["format"] == 0, "only BC1 and RGBA come out of load_stb"
    for l, offset in enumerate(r["mip_offsets"]):
        bw, bh = max(r["width"] >> l, 1), max(r["height"] >> l, 1)
        levels.append(decode_bc1(r["data"][offset: offset + bw * bh * 8].reshape(-1, 8), bw, bh))
    return levels, (r["width"], r["height"])


def product_scene_digest(scene):
    """sha256 over everything compared below, in a form both sides can produce."""
    h = hashlib.sha256()
    h.update(scene.describe().encode())
    for m in range(scene.mesh_data_count):
        h.update(np.ascontiguousarray(scene.mesh_data_array(m, "triangles", np.float32)).tobytes())
    for i in range(scene.describe().count("\ntexture ")):
        t = scene.texture(i)
        h.update(struct.pack("<4i", t["width"], t["height"], t["lod_width"], t["lod_height"]))
        h.update(t["texels"].tobytes())
    h.update(scene.sky().tobytes())
    return h.hexdigest()


def assert_same_scene(grt, oracle, path, sky, key=None, **config):
    """Loads `path` with both sides and compares everything; with `key`, also checks / (when REGENERATE is set) records
    the golden digest."""
    grt.config_reset()
    grt.config_set(enable_block_compression=int(config.get("enable_block_compression", 1)), mipmap_filter=config.get("mipmap_filter", 0),
                   enable_mipmapping=int(config.get("enable_mipmapping", 1)))
    scene = grt.Scene(str(path), sky=sky); scene.wait_until_loaded()
    if key is not None and key in GOLDEN:
        assert product_scene_digest(scene) == GOLDEN[key], key
    if oracle.ref_scene_lib() is None:
        scene.close(); grt.config_reset()
        if key is None or key not in GOLDEN:
            pytest.skip("oracle/_ref/libref_scene.so not built (no /root/reference on this machine) and no golden digest")
        return None
    ref = oracle.ReferenceScene(str(path), sky=sky, **config)
    want, got = ref.description.splitlines(), scene.describe().splitlines()
    for a, b in zip(want, got):
        assert a == b
    assert len(want) == len(got)
    for m in range(ref.count("mesh_data")):
        assert np.array_equal(ref.triangles(m), scene.mesh_data_array(m, "triangles", np.float32).reshape(-1, 24), equal_nan=True), "mesh data %d" % m
    for i in range(ref.count("texture")):
        levels, lod_size = reference_texture_as_rgba8(ref.texture(i))
        mine = scene.texture(i)
        assert (mine["lod_width"], mine["lod_height"]) == lod_size and len(levels) == len(mine["mip_offsets"]), "texture %d" % i
        for l, level in enumerate(levels):
            lw, lh = max(mine["width"] >> l, 1), max(mine["height"] >> l, 1)
            texels = mine["texels"][mine["mip_offsets"][l]: mine["mip_offsets"][l] + lw * lh].reshape(lh, lw, 4)
            assert np.array_equal(level[:lh, :lw], texels), "texture %d level %d" % (i, l)
    assert np.array_equal(ref.sky(), scene.sky())
    digest = product_scene_digest(scene)
    listing = ref.description
    ref.close(); scene.close(); grt.config_reset()
    return digest, listing


def posix_copy_of(grt, name, tmp_path):
    """A scratch copy of a bundled scene (the reference writes .bvh caches next to the meshes it loads) whose texture
    paths use '/' -- the files use the reference's Windows separators, which its loader hands to fopen as they are."""
    src = os.path.dirname(grt.scene_path(name))
    dst = tmp_path / name
    shutil.copytree(src, dst)
    xml = (dst / "scene.xml").read_text().replace("\\\\", "/").replace("\\", "/")
    (dst / "scene.xml").write_text(xml)
    return dst / "scene.xml"


def test_cornell_box_loads_like_the_reference(grt, oracle, tmp_path):
    assert_same_scene(grt, oracle, posix_copy_of(grt, "cornellbox", tmp_path), write_sky(tmp_path / "sky.hdr"), key="cornellbox")


@pytest.mark.parametrize("block_compression", [1, 0])
def test_sponza_loads_like_the_reference(grt, oracle, tmp_path, block_compression):
    """383 OBJ meshes, 25 materials, 19 TGA maps (+5 missing ones: the fallback texel) through gamma, box mip chain,
    8-bit quantisation and -- the reference's default -- BC1 encoding."""
    result = assert_same_scene(grt, oracle, posix_copy_of(grt, "sponza", tmp_path), write_sky(tmp_path / "sky.hdr"),
                               key="sponza_bc%d" % block_compression, enable_block_compression=block_compression)
    if result:
        assert result[1].count("\nmesh ") >= 383 and result[1].count("\ntexture ") == 24


def write_feature_scene(tmp_path):
    """One Mitsuba file that walks through MitsubaLoader.cpp: every bsdf type and wrapper, named and nested ids, texture
    scale nodes, media, all five primitive shapes, obj / ply / serialized / hair shapes, shape groups and instances,
    every transform node, an include, the three sensor types' parameters, area / point emitters."""
    rng = np.random.default_rng(21)
    (tmp_path / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0.5\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\nf 2 4 3\nf -1 -2 -3 -4\n")
    positions = np.round(rng.random((6, 3)) * 2 - 1, 3).astype(np.float32)
    (tmp_path / "m.ply").write_bytes(_ply_bytes("binary_little_endian", positions, None, np.round(rng.random((6, 2)), 3), [[0, 1, 2], [2, 3, 4, 5]], "int", False))
    quad = dict(name="quad", double=False, positions=[[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], normals=[[0, 0, 1]] * 4,
                uvs=[[0, 0], [1, 0], [1, 1], [0, 1]], faces=[[0, 1, 2], [0, 2, 3]])
    fan = dict(name="fan", double=True, positions=np.round(rng.random((5, 3)) * 2, 3), faces=[[0, 1, 2], [0, 2, 3], [0, 3, 4]], face_normals=True)
    (tmp_path / "meshes.serialized").write_bytes(_serialized_archive([quad, fan], 4))
    strands = [np.array([[0, 0, 0], [0, 1, 0], [0.2, 2, 0], [0.2, 3, 0.1]], np.float32), np.array([[1, 0, 0], [1, 0.5, 0.5], [1, 1, 1]], np.float32)]
    (tmp_path / "h.hair").write_bytes("".join("".join("%g %g %g\n" % tuple(v) for v in s) + "\n" for s in strands).encode())
    pixels = rng.integers(0, 256, (16, 32, 3))
    (tmp_path / "wood.png").write_bytes(_png_bytes(pixels, 2, 8))
    (tmp_path / "odd.png").write_bytes(_png_bytes(rng.integers(0, 256, (5, 12, 4)), 6, 8))
    (tmp_path / "extra.xml").write_text('<scene version="0.5.0"><bsdf type="diffuse" id="included"><rgb name="reflectance" value="0.1, 0.9, 0.1"/></bsdf>'
                                        '<shape type="disk"><ref id="included"/><transform name="toWorld"><translate x="-3"/></transform></shape></scene>')
    xml = tmp_path / "features.xml"
    xml.write_text("""<?xml version="1.0" encoding="utf-8"?>
<!-- every construct the loader knows -->
<scene version="0.6.0">
	<integrator type="path"><integer name="maxDepth" value="9"/></integrator>
	<sensor type="perspective">
		<float name="fov" value="42.5"/>
		<transform name="toWorld"><matrix value="-1 0 0 0.5 0 1 0 1.25 0 0 -1 6 0 0 0 1"/></transform>
		<sampler type="independent"><integer name="sampleCount" value="64"/></sampler>
		<film type="hdrfilm"><integer name="width" value="320"/><integer name="height" value="200"/></film>
	</sensor>
	<texture type="bitmap" id="wood"><string name="filename" value="wood.png"/></texture>
	<bsdf type="diffuse" id="matte"><srgb name="reflectance" value="0.8, 0.4, 0.2"/></bsdf>
	<bsdf type="diffuse" id="textured"><ref name="reflectance" id="wood"/></bsdf>
	<bsdf type="diffuse" id="scaled"><texture name="reflectance" type="scale"><float name="scale" value="0.5"/><texture type="bitmap"><string name="filename" value="odd.png"/></texture></texture></bsdf>
	<bsdf type="twosided" id="wrapped"><bsdf type="roughplastic"><rgb name="diffuseReflectance" value="0.2, 0.3, 0.4"/><float name="alpha" value="0.25"/><float name="intIOR" value="1.6"/></bsdf></bsdf>
	<bsdf type="plastic" id="smooth_plastic"><rgb name="diffuseReflectance" value="0.9, 0.1, 0.1"/></bsdf>
	<bsdf type="roughdiffuse" id="rough_diffuse"><rgb name="reflectance" value="0.3, 0.3, 0.7"/><float name="alpha" value="0.4"/></bsdf>
	<bsdf type="phong" id="phong"><rgb name="diffuseReflectance" value="0.3, 0.6, 0.3"/><float name="exponent" value="40"/></bsdf>
	<bsdf type="conductor" id="mirror"><string name="material" value="none"/></bsdf>
	<bsdf type="roughconductor" id="copper"><rgb name="eta" value="0.2, 0.92, 1.1"/><rgb name="k" value="3.9, 2.45, 2.14"/><float name="alpha" value="0.15"/></bsdf>
	<bsdf type="mask" id="masked"><bsdf type="bumpmap"><bsdf type="coating"><bsdf type="conductor"><rgb name="eta" value="1.5, 1.0, 0.5"/><rgb name="k" value="2, 2, 2"/></bsdf></bsdf></bsdf></bsdf>
	<bsdf type="dielectric" id="glass"><float name="intIOR" value="1.5"/><float name="extIOR" value="1.0"/></bsdf>
	<bsdf type="thindielectric" id="pane"><string name="intIOR" value="bk7"/></bsdf>
	<bsdf type="roughdielectric" id="frosted"><string name="intIOR" value="diamond"/><string name="extIOR" value="water"/><float name="alpha" value="0.2"/></bsdf>
	<bsdf type="difftrans" id="sheet"><rgb name="transmittance" value="0.5, 0.6, 0.7"/></bsdf>
	<shape type="obj"><string name="filename" value="tri.obj"/><ref id="matte"/>
		<transform name="toWorld"><scale value="2"/><rotate x="0" y="1" z="0" angle="30"/><translate x="1" y="2" z="3"/></transform></shape>
	<shape type="obj"><string name="filename" value="tri.obj"/><ref id="textured"/>
		<transform name="toWorld"><scale x="1.5" y="1.5" z="1.5"/><rotate x="1" angle="-45"/><translate y="-1"/></transform></shape>
	<shape type="ply"><string name="filename" value="m.ply"/><ref id="scaled"/></shape>
	<shape type="serialized"><string name="filename" value="meshes.serialized"/><ref id="wrapped"/></shape>
	<shape type="serialized"><string name="filename" value="meshes.serialized"/><integer name="shapeIndex" value="1"/><ref id="smooth_plastic"/>
		<transform name="toWorld"><lookat origin="1, 1, 1" target="0, 0, 0" up="0, 1, 0"/></transform></shape>
	<shape type="hair"><string name="filename" value="h.hair"/><float name="radius" value="0.05"/><ref id="rough_diffuse"/></shape>
	<shape type="rectangle"><ref id="phong"/><transform name="toWorld"><rotate y="1" angle="90"/><translate x="4"/></transform></shape>
	<shape type="cube"><ref id="mirror"/><transform name="toWorld"><scale value="0.5"/><translate x="-2" y="0.5"/></transform></shape>
	<shape type="disk"><ref id="copper"/></shape>
	<shape type="cylinder"><point name="p0" x="0" y="0" z="0"/><point name="p1" x="0" y="2" z="0"/><float name="radius" value="0.3"/><ref id="masked"/></shape>
	<shape type="sphere"><point name="center" x="1" y="1" z="-1"/><float name="radius" value="0.75"/><ref id="glass"/>
		<medium type="homogeneous" name="interior"><rgb name="sigmaA" value="0.1, 0.2, 0.3"/><rgb name="sigmaS" value="1, 1.5, 2"/><phase type="hg"><float name="g" value="0.3"/></phase></medium></shape>
	<shape type="sphere"><float name="radius" value="0.4"/><ref id="frosted"/>
		<medium type="homogeneous" name="interior"><rgb name="albedo" value="0.8, 0.7, 0.6"/><rgb name="sigmaT" value="2, 2, 2"/><float name="scale" value="3"/><phase type="isotropic"/></medium></shape>
	<shape type="rectangle"><ref id="pane"/><transform name="toWorld"><translate z="-4"/></transform></shape>
	<shape type="rectangle"><ref id="sheet"/><transform name="toWorld"><translate z="-5"/></transform></shape>
	<shape type="shapegroup" id="group">
		<shape type="cube"><ref id="matte"/></shape>
		<shape type="obj"><string name="filename" value="tri.obj"/><ref id="copper"/></shape>
	</shape>
	<shape type="instance"><ref id="group"/><transform name="toWorld"><scale value="0.25"/><translate x="3" y="3"/></transform></shape>
	<shape type="instance"><ref id="group"/><transform name="toWorld"><rotate z="1" angle="180"/><translate x="-3" y="3"/></transform></shape>
	<shape type="rectangle"><emitter type="area"><rgb name="radiance" value="12, 11, 10"/></emitter><transform name="toWorld"><rotate x="1" angle="90"/><translate y="5"/></transform></shape>
	<emitter type="point"><point name="position" x="2" y="4" z="-2"/><rgb name="intensity" value="30, 30, 30"/></emitter>
	<include filename="extra.xml"/>
</scene>
""")
    return xml


def test_every_mitsuba_construct_loads_like_the_reference(grt, oracle, tmp_path, monkeypatch):
    write_feature_scene(tmp_path)
    write_sky(tmp_path / "sky.hdr", seed=2)
    monkeypatch.chdir(tmp_path)           # relative names: the ribbon angle of a hair file is seeded from its file name
    result = assert_same_scene(grt, oracle, "features.xml", "sky.hdr", key="features")
    if result:
        listing = result[1]
        assert "num_bounces=9" in listing and "width=320 height=200" in listing
        assert listing.count("\nmaterial ") >= 16 and listing.count("\nmedium ") == 3 and listing.count("\ntexture ") == 2


@pytest.mark.parametrize("sensor", ["thinlens", "perspective_rdist"])
def test_sensor_variants_and_mip_filters_load_like_the_reference(grt, oracle, tmp_path, sensor):
    rng = np.random.default_rng(5)
    (tmp_path / "map.png").write_bytes(_png_bytes(rng.integers(0, 256, (64, 32, 3)), 2, 8))
    extra = '<float name="apertureRadius" value="0.125"/><float name="focusDistance" value="7.5"/>' if sensor == "thinlens" else '<string name="fovAxis" value="x"/>'
    xml = tmp_path / "s.xml"
    xml.write_text('<scene version="0.5.0"><sensor type="%s"><float name="fov" value="35"/>%s'
                   '<transform name="toWorld"><lookat origin="3, 4, 5" target="0, 1, 0" up="0, 1, 0"/></transform></sensor>'
                   '<shape type="rectangle"><bsdf type="diffuse"><texture name="reflectance" type="bitmap"><string name="filename" value="map.png"/></texture></bsdf></shape>'
                   '<emitter type="envmap"><string name="filename" value="sky.hdr"/></emitter></scene>' % (sensor, extra))
    sky = write_sky(tmp_path / "sky.hdr", seed=3, width=6, height=3)
    for mipmap_filter in (1, 2):
        for block_compression in (1, 0):
            assert_same_scene(grt, oracle, xml, sky, enable_block_compression=block_compression, mipmap_filter=mipmap_filter)
    assert_same_scene(grt, oracle, xml, sky, enable_mipmapping=0)
