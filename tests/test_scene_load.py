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


def test_bvh_caches_are_interchangeable_with_the_references(grt, oracle, tmp_path):
    """`<mesh>.bvh` (BVHLoader.cpp:19-249): the files the reference writes carry the same header and the same payload --
    triangles, BVHNode2s, indices -- as ours; it loads ours without rebuilding, and we load its."""
    from test_loaders import _read_bvh_cache
    if oracle.ref_scene_lib() is None:
        pytest.skip("oracle/_ref/libref_scene.so not built (no /root/reference on this machine)")
    rng = np.random.default_rng(8)
    n = 400
    p0 = rng.random((n, 3)) * 4; p1 = p0 + rng.random((n, 3)) * 3 - 1.5; p2 = p0 + rng.random((n, 3)) * 0.4

    def write_obj(directory):
        directory.mkdir()
        with open(directory / "m.obj", "w") as f:
            for a, b, c in zip(p0, p1, p2):
                f.write("v %.6f %.6f %.6f\nv %.6f %.6f %.6f\nv %.6f %.6f %.6f\n" % (*a, *b, *c))
            for i in range(n):
                f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
        (directory / "s.xml").write_text('<scene version="0.5.0"><shape type="obj"><string name="filename" value="m.obj"/></shape></scene>')
        return directory / "s.xml", str(directory / "m.obj") + ".bvh"

    sky = write_sky(tmp_path / "sky.hdr")
    for bvh_type, name in ((3, "bvh8"), (0, "sah"), (1, "sbvh")):          # the reference's BVHType order: BVH, SBVH, BVH4, BVH8
        theirs_xml, theirs_cache = write_obj(tmp_path / ("theirs_" + name))
        ours_xml, ours_cache = write_obj(tmp_path / ("ours_" + name))

        ref = oracle.ReferenceScene(str(theirs_xml), sky=sky, bvh_type=bvh_type); ref.close()      # writes theirs
        grt.config_reset(); grt.config_set(bvh_type=name, enable_bvh_cache=1)
        scene = grt.Scene(str(ours_xml), sky=sky); pt = grt.Pathtracer(scene, 8, 8, device=-1)      # writes ours
        built = {k: scene.mesh_data_array(0, k, dt).copy() for k, dt in (("triangles", np.float32), ("bvh2_nodes", np.uint8), ("bvh2_indices", np.int32))}
        pt.close(); scene.close()

        a, b = _read_bvh_cache(theirs_cache), _read_bvh_cache(ours_cache)
        assert all(a[k] == b[k] for k in ("ident", "version", "bvh_type", "optimized", "cost_node", "cost_leaf")), name
        assert all(np.array_equal(a[k], b[k]) for k in ("triangles", "nodes", "indices")), name
        assert a["bvh_type"] == (1 if name == "sbvh" else 0)

        # the reference accepts our file: it neither rebuilds nor rewrites it, and ends up with our triangles
        before = (os.stat(ours_cache).st_mtime_ns, open(ours_cache, "rb").read())
        ref = oracle.ReferenceScene(str(ours_xml), sky=sky, bvh_type=bvh_type)
        assert np.array_equal(ref.triangles(0), b["triangles"]); ref.close()
        assert (os.stat(ours_cache).st_mtime_ns, open(ours_cache, "rb").read()) == before

        # and we accept the reference's: mark one coordinate in its file (recompressed here) and see it arrive
        from test_loaders import _write_bvh_cache
        marked = dict(a); marked["triangles"] = a["triangles"].copy(); marked["triangles"][7, 1] += 0.25
        raw_reference_file = open(theirs_cache, "rb").read()
        grt.config_reset(); grt.config_set(bvh_type=name, enable_bvh_cache=1)
        scene = grt.Scene(str(theirs_xml), sky=sky); scene.wait_until_loaded()                     # reads the reference's own bytes
        assert np.array_equal(scene.mesh_data_array(0, "triangles", np.float32).reshape(-1, 24), a["triangles"])
        scene.close()
        assert open(theirs_cache, "rb").read() == raw_reference_file                                # loaded, not rebuilt and saved again
        _write_bvh_cache(theirs_cache, marked)
        scene = grt.Scene(str(theirs_xml), sky=sky); scene.wait_until_loaded()
        assert scene.mesh_data_array(0, "triangles", np.float32).reshape(-1, 24)[7, 1] == marked["triangles"][7, 1]
        scene.close()
    grt.config_reset()


def test_exporters_write_the_same_files_as_the_references(grt, oracle, tmp_path):
    """PPMExporter.cpp / EXRExporter.cpp (tinyexr with the reference's zero-initialised header): same bytes."""
    import ctypes
    if oracle.ref_scene_lib() is None:
        pytest.skip("oracle/_ref/libref_scene.so not built (no /root/reference on this machine)")
    rng = np.random.default_rng(12)
    lib = grt.host_lib()
    lib.grt_export_ppm_display.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for h, w in ((5, 7), (1, 1), (33, 64), (130, 17)):
        img = (rng.random((h, w, 3)) * 3 - 0.5).astype(np.float32)
        img[0, 0] = (-1.0, 0.0, 70000.0)                  # negative, zero, beyond the half range
        img[h - 1, w - 1] = (1e-8, 6.1e-5, 0.333)         # flushes to zero, the smallest normal half, an inexact value
        oracle.ref_export_image(tmp_path / "ref.exr", img)
        grt.export_image(tmp_path / "ours.exr", img)
        assert open(tmp_path / "ref.exr", "rb").read() == open(tmp_path / "ours.exr", "rb").read(), (h, w)
        oracle.ref_export_image(tmp_path / "ref.ppm", img)
        assert lib.grt_export_ppm_display(str(tmp_path / "ours.ppm").encode(), w, w, h, np.ascontiguousarray(img).ctypes.data) == 0
        assert open(tmp_path / "ref.ppm", "rb").read() == open(tmp_path / "ours.ppm", "rb").read(), (h, w)


def test_command_lines_configure_like_the_references_argument_parser(grt, oracle):
    """Args.cpp:51-184 against `pathtracer --print-config`: short and long names, `-b` meaning bounces (the reference
    gives --bvh the same short name, after --bounce), bare arguments as scene files, boolean spellings, unknown options
    skipped, the bounce count clamped. (-c and -S are always given: those two defaults differ, see DESIGN.md.)"""
    import subprocess
    if oracle.ref_scene_lib() is None:
        pytest.skip("oracle/_ref/libref_scene.so not built (no /root/reference on this machine)")
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-raytracer_amd", "host", "pathtracer")
    common = ["-c", "true", "-S", "sky.hdr"]
    cases = [
        ["a.xml"],
        ["-s", "a.xml", "b.obj", "--scene", "c.ply"],
        ["-W", "640", "-H", "360", "-b", "5", "-N", "16", "-o", "out.exr", "a.xml"],
        ["--width", "33", "--height", "17", "--bounce", "1000", "--samples", "0", "--output", "x.ppm", "a.xml"],
        ["--bounce", "-3", "a.xml"],
        ["-I", "ao", "a.xml"], ["--integrator", "pathtracer", "a.xml"],
        ["--bvh", "sah", "a.xml"], ["--bvh", "sbvh", "a.xml"], ["--bvh", "bvh4", "a.xml"], ["--bvh", "bvh8", "a.xml"],
        ["--nee", "false", "--mis", "0", "a.xml"], ["--nee", "TRUE", "--mis", "False", "a.xml"], ["--nee", "maybe", "a.xml"],
        ["--force-rebuild", "a.xml"],
        ["-O", "true", "-Ot", "1500", "-Ob", "7", "a.xml"], ["--optimize", "1", "--opt-time", "20", "--opt-batches", "3", "a.xml"],
        ["--sah-node", "2.5", "--sah-leaf", "0.75", "--sbvh-alpha", "0.001", "a.xml"], ["--sbvh-alpha", "1e-3", "a.xml"],
        ["--mipmap", "false", "--mip-filter", "kaiser", "a.xml"], ["--mip-filter", "lanczos", "a.xml"], ["--mip-filter", "box", "a.xml"],
        ["--frobnicate", "-x", "a.xml", "--width"],
        ["-c", "false", "a.xml"], ["--compress", "0", "a.xml"],
        ["-Ot", "5", "-O", "false", "a.xml", "-W", "12"],
    ]
    for arguments in cases:
        arguments = common + arguments
        ours = subprocess.run([cli, "--print-config", *arguments], capture_output=True, text=True, timeout=60)
        assert ours.returncode == 0, (arguments, ours.stderr)
        assert ours.stdout.splitlines()[-1] + "\n" == oracle.ref_args_parse(arguments), arguments


def product_mesh_file(grt, path, xml=None):
    grt.config_reset()
    scene = grt.Scene(str(xml or path)); scene.wait_until_loaded()
    tris = scene.mesh_data_array(0, "triangles", np.float32).reshape(-1, 24).copy()
    scene.close()
    return tris


def test_mesh_file_loaders_equal_the_references(grt, oracle, tmp_path, monkeypatch):
    """OBJLoader / PLYLoader / SerializedLoader / MitshairLoader on their own, on files that lean on their corners:
    OBJ with negative and partial indices, polygons, missing normals, odd whitespace and comments; PLY in all three
    encodings with and without normals, extra properties and short index types; serialized archives of both dictionary
    widths, single and double precision; ascii and binary hair."""
    if oracle.ref_scene_lib() is None:
        pytest.skip("oracle/_ref/libref_scene.so not built (no /root/reference on this machine)")
    monkeypatch.chdir(tmp_path)              # relative names: the ribbon angle of a hair file is seeded from its file name
    rng = np.random.default_rng(31)

    (tmp_path / "a.obj").write_text(
        "# comment\n\nv 0 0 0\nv 1 0 0\nv   0 1 0  \nv 1 1 0.5\nv 2 1 0.25\nv -1.5e0 2 1\n"
        "vt 0 0\nvt 1 0\nvt 0 1\nvt 0.25 0.75\nvn 0 0 1\nvn 0 1 0\nvn 1 0 0\n"
        "o thing\ng part\nusemtl none\ns off\n"
        "f 1/1/1 2/2/1 3/3/1\nf 2 4 3\nf -1 -2 -3 -4\nf 1//2 2//2 4//3 5//1 6//2\nf 1/1 2/2 4/4\nf 3/3/3 2/2/2 1/1/1\n")
    n = 200
    p = rng.random((n, 3, 3)) * 4
    with open(tmp_path / "b.obj", "w") as f:
        for t in p:
            for v in t: f.write("v %.7g %.7g %.7g\n" % tuple(v))
        for i in range(n): f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    for name in ("a.obj", "b.obj"):
        assert np.array_equal(oracle.ref_load_mesh_file("obj", name), product_mesh_file(grt, name), equal_nan=True), name
    write_sky(tmp_path / "sky.hdr")
    assert_same_scene(grt, oracle, "a.obj", "sky.hdr")        # a mesh file named as the scene (Scene.cpp:29-32): default camera and material

    positions = np.round(rng.random((9, 3)) * 4 - 2, 3).astype(np.float32)
    normals = rng.random((9, 3)).astype(np.float32); normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    uvs = np.round(rng.random((9, 2)), 3)
    faces = [[0, 1, 2], [2, 3, 4, 5], [1, 6, 5, 4, 3], [6, 7, 8]]
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        for variant, (with_normals, extras, index_type) in enumerate(((True, False, "int"), (False, True, "ushort" if fmt != "ascii" else "uint"), (True, True, "uchar" if fmt != "ascii" else "int"))):
            name = "m_%s_%d.ply" % (fmt, variant)
            # (extra properties on vertices only: the reference stops reading a face at the first property that is not the
            # index list, which ends an ascii load with an error and silently desynchronises a binary one)
            (tmp_path / name).write_bytes(_ply_bytes(fmt, positions, normals if with_normals else None, uvs, faces, index_type, extras, face_extras=False))
            assert np.array_equal(oracle.ref_load_mesh_file("ply", name), product_mesh_file(grt, name), equal_nan=True), name
    assert_same_scene(grt, oracle, "m_ascii_0.ply", "sky.hdr")

    quad = dict(name="quad", positions=[[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], normals=[[0, 0, 1]] * 4, uvs=[[0, 0], [1, 0], [1, 1], [0, 1]],
                colours=[[1, 0, 0]] * 4, faces=[[0, 1, 2], [0, 2, 3]])
    fan = dict(name="fan", positions=np.round(rng.random((5, 3)) * 2, 3), faces=[[0, 1, 2], [0, 2, 3], [0, 3, 4]], face_normals=True)
    soup_positions = rng.random((60, 3)) * 3
    soup = dict(name="soup", positions=soup_positions, normals=rng.random((60, 3)) - 0.5, faces=[[3 * i, 3 * i + 1, 3 * i + 2] for i in range(20)])
    for version in (3, 4):
        for double in ((False, True) if version > 3 else (False,)):          # version 3 archives are single precision by definition
            meshes = [dict(m, double=double) for m in (quad, fan, soup)]
            name = "meshes_v%d_%d.serialized" % (version, double)
            (tmp_path / name).write_bytes(_serialized_archive(meshes, version))
            for index in range(3):
                xml = tmp_path / "s.xml"
                xml.write_text('<scene version="0.5.0"><shape type="serialized"><string name="filename" value="%s"/><integer name="shapeIndex" value="%d"/></shape></scene>' % (name, index))
                assert np.array_equal(oracle.ref_load_mesh_file("serialized", name, index), product_mesh_file(grt, name, "s.xml"), equal_nan=True), (name, index)

    strands = [np.cumsum(rng.random((k, 3)).astype(np.float32) * 0.3, axis=0) for k in (4, 1, 3, 2, 9)]
    ascii_hair = "".join("".join("%.6g %.6g %.6g\n" % tuple(v) for v in s) + "\n" for s in strands).encode()
    binary_hair = b"BINARY_HAIR" + struct.pack("<I", sum(len(s) for s in strands)) + b"".join(s.tobytes() + struct.pack("<f", np.inf) for s in strands)
    for name, data in (("a.hair", ascii_hair), ("b.hair", binary_hair)):
        (tmp_path / name).write_bytes(data)
        for radius in (0.05, 0.5):
            xml = tmp_path / "h.xml"
            xml.write_text('<scene version="0.5.0"><shape type="hair"><string name="filename" value="%s"/><float name="radius" value="%g"/></shape></scene>' % (name, radius))
            # ("./": the name as the Mitsuba loader composes it from the scene's directory, which is what seeds the ribbon angle)
            assert np.array_equal(oracle.ref_load_mesh_file("hair", "./" + name, radius), product_mesh_file(grt, name, "h.xml"), equal_nan=True), (name, radius)


def test_dds_textures_load_like_the_reference(grt, oracle, tmp_path):
    """TextureLoader::load_dds (TextureLoader.cpp:19-106): DXT1 blocks used as stored; the chain ends where halving the
    block counts reaches zero, whatever the file holds beyond that."""
    rng = np.random.default_rng(17)
    for name, (w, h) in (("a", (32, 16)), ("b", (64, 64)), ("c", (8, 8))):
        levels, lw, lh = 0, w, h
        blocks = b""
        while True:
            blocks += rng.integers(0, 256, ((lw + 3) // 4) * ((lh + 3) // 4) * 8, dtype=np.uint8).tobytes()
            levels += 1
            if lw == 1 and lh == 1: break
            lw, lh = max(lw // 2, 1), max(lh // 2, 1)
        header = b"DDS " + struct.pack("<IIIIIII", 124, 0x1007 | 0x20000, h, w, 0, 0, levels) + b"\0" * 44 + struct.pack("<II4sIIIII", 32, 4, b"DXT1", 0, 0, 0, 0, 0) + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
        (tmp_path / (name + ".dds")).write_bytes(header + blocks)
    (tmp_path / "s.xml").write_text('<scene version="0.5.0">' + "".join(
        '<shape type="rectangle"><bsdf type="diffuse"><texture name="reflectance" type="bitmap"><string name="filename" value="%s.dds"/></texture></bsdf></shape>' % n for n in "abc") + '</scene>')
    assert_same_scene(grt, oracle, tmp_path / "s.xml", write_sky(tmp_path / "sky.hdr"))
    grt.config_reset()
    scene = grt.Scene(str(tmp_path / "s.xml")); scene.wait_until_loaded()
    assert [len(scene.texture(i)["mip_offsets"]) for i in range(3)] == [3, 5, 2]
    scene.close()
