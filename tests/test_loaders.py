"""Host side: Mitsuba XML / OBJ loading, primitive tessellation, device data layout (CPU only)."""
import os

import numpy as np
import pytest

from conftest import make_pathtracer


def test_cornell_scene_inventory(grt):
    scene, pt = make_pathtracer(grt, "cornellbox", 512, 512, -1)
    # reference Data/cornellbox/scene.xml: 5 walls + light (rectangles, 2 tris) + 2 cubes (12 tris)
    assert scene.mesh_count == 8
    tris = pt.array("triangles").reshape(-1, 24)
    assert tris.shape[0] == 36
    # XML overrides: film 1024x1024 and maxDepth 65 (MitsubaLoader.cpp:610-616)
    assert grt.config_get("initial_width") == 1024
    assert pt.device_config().num_bounces == 65
    types = pt.array("material_types")
    assert (types == grt.MATERIAL_LIGHT).sum() == 1 and (types == grt.MATERIAL_DIFFUSE).sum() >= 8
    mats = pt.array("materials").reshape(-1, 8)
    light = mats[types == grt.MATERIAL_LIGHT][0]
    assert np.allclose(light[:3], [17, 12, 4])
    assert pt.lights_total_weight > 0
    pt.close(); scene.close()


def test_device_layout_rules(grt):
    """reference Integrator.cpp:101-283,399-430: TLAS slots first, MSB = identity, triangles as edges."""
    scene, pt = make_pathtracer(grt, "cornellbox", 64, 64, -1)
    roots = pt.array("mesh_bvh_root_indices").view(np.uint32)
    assert (roots >> 31).all()                      # primitives have baked transforms -> identity
    assert ((roots & 0x7fffffff) >= 2 * scene.mesh_count).all()
    nodes = pt.array("bvh8_nodes").reshape(-1, 80)
    assert nodes.shape[0] == 2 * scene.mesh_count + 8   # one CWBVH node per tiny mesh
    order = pt.array("tlas_indices")
    assert sorted(order.tolist()) == list(range(scene.mesh_count))
    xf = pt.array("mesh_transforms").reshape(-1, 12)
    assert np.allclose(xf, np.tile(np.eye(4, dtype=np.float32)[:3].reshape(-1), (scene.mesh_count, 1)))
    pt.close(); scene.close()


def test_sponza_inventory(grt):
    scene, pt = make_pathtracer(grt, "sponza", 64, 64, -1)
    assert scene.mesh_count == 384 and scene.mesh_data_count == 383
    tris = pt.array("triangles").reshape(-1, 24)
    assert tris.shape[0] == 262687                     # unique Sponza triangles
    nodes = pt.array("bvh8_nodes")
    assert nodes.size // 80 == 32291 + 2 * 384         # BASELINE.md section 2 + reserved TLAS slots
    roots = pt.array("mesh_bvh_root_indices").view(np.uint32)
    assert ((roots >> 31) == 0).sum() == 2             # the two translated icosphere lights
    assert pt.array("light_mesh_transform_indices").size == 2
    pt.close(); scene.close()


def test_obj_loader_fan_triangulation_and_negative_indices(grt, tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# quad\no q\nv 0 0 0\nv 1 0 0 1.0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4//-1 -3//-1 -2//-1\n")
    grt.config_reset()
    scene = grt.Scene(str(obj))
    scene.wait_until_loaded()
    tris = scene.mesh_data_array(0, "triangles", np.float32).reshape(-1, 24)
    assert tris.shape[0] == 3
    assert np.allclose(tris[0, 0:9], [0, 0, 0, 1, 0, 0, 1, 1, 0])
    assert np.allclose(tris[1, 0:9], [0, 0, 0, 1, 1, 0, 0, 1, 0])     # fan: (v0, prev, curr)
    assert np.allclose(tris[0, 18:24], [0, 1, 1, 1, 1, 0])            # v flipped: t.y = 1 - t.y
    assert np.allclose(tris[2, 0:9], [0, 0, 0, 1, 0, 0, 1, 1, 0])     # negative indices
    scene.close()


def test_unsupported_scene_format_is_an_error(grt, tmp_path):
    bad = tmp_path / "scene.ply"
    bad.write_text("ply\n")
    grt.config_reset()
    with pytest.raises(RuntimeError, match="not supported"):
        grt.Scene(str(bad))


def test_mitsuba_materials_media_and_instances(grt, tmp_path):
    (tmp_path / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    xml = tmp_path / "scene.xml"
    xml.write_text("""<?xml version="1.0"?>
<!-- comment -->
<scene version="0.5.0">
  <integrator type="path"><integer name="maxDepth" value="7"/></integrator>
  <sensor type="thinlens"><float name="fov" value="60"/><float name="apertureRadius" value="0.1"/>
    <transform name="toWorld"><lookat origin="0, 0, 5" target="0, 0, 0" up="0, 1, 0"/></transform>
    <film type="hdrfilm"><integer name="width" value="320"/><integer name="height" value="200"/></film></sensor>
  <bsdf type="roughplastic" id="pl"><rgb name="diffuseReflectance" value="0.2, 0.3, 0.4"/><float name="alpha" value="0.3"/></bsdf>
  <bsdf type="roughconductor" id="cu"><rgb name="eta" value="1.25, 1.02, 0.3"/><rgb name="k" value="2.48, 2.58, 3.2"/><float name="alpha" value="0.1"/></bsdf>
  <shape type="obj"><string name="filename" value="tri.obj"/><ref id="pl"/>
    <transform name="toWorld"><scale value="2"/><translate x="1" y="2" z="3"/></transform></shape>
  <shape type="sphere"><float name="radius" value="0.5"/><bsdf type="roughdielectric"><string name="intIOR" value="water"/><float name="alpha" value="0.1"/></bsdf>
    <medium type="homogeneous" name="interior"><rgb name="sigmaA" value="0.1, 0.2, 0.3"/><rgb name="sigmaS" value="1, 1, 1"/><phase type="hg"><float name="g" value="0.2"/></phase></medium></shape>
  <shape type="rectangle"><ref id="cu"/></shape>
  <shape type="cube"><emitter type="area"><rgb name="radiance" value="5, 5, 5"/></emitter></shape>
</scene>""")
    grt.config_reset()
    scene = grt.Scene(str(xml))
    pt = grt.Pathtracer(scene, 320, 200, device=-1)
    pt.update()
    assert pt.device_config().num_bounces == 7
    types = pt.array("material_types").tolist()
    assert types.count(grt.MATERIAL_PLASTIC) == 1 and types.count(grt.MATERIAL_CONDUCTOR) == 1
    assert types.count(grt.MATERIAL_DIELECTRIC) == 1 and types.count(grt.MATERIAL_LIGHT) == 1
    mats = pt.array("materials").reshape(-1, 8)
    diel = mats[types.index(grt.MATERIAL_DIELECTRIC)]
    assert diel[:1].view(np.int32)[0] == 1 and abs(diel[1] - 1.333) < 1e-6 and abs(diel[2] - 0.1) < 1e-7   # medium id, ior, roughness
    media = pt.array("media").reshape(-1, 8)
    assert media.shape[0] == 2 and abs(media[1, 3] - 0.2) < 1e-7
    # Van de Hulst round trip is approximate; sigma_t = 1/mfp is exact
    assert np.allclose(media[1, 0:3] + media[1, 4:7], [1.1, 1.2, 1.3], rtol=1e-5)
    # the OBJ instance keeps a transform: scale 2 then translate
    order = pt.array("tlas_indices").tolist()
    roots = pt.array("mesh_bvh_root_indices").view(np.uint32)
    xf = pt.array("mesh_transforms").reshape(-1, 12)
    i = order.index(0)
    assert (roots[i] >> 31) == 0
    assert np.allclose(xf[i], [2, 0, 0, 1, 0, 2, 0, 2, 0, 0, 2, 3], atol=1e-6)
    inv = pt.array("mesh_transforms_inv").reshape(-1, 12)
    assert np.allclose(inv[i], [0.5, 0, 0, -0.5, 0, 0.5, 0, -1, 0, 0, 0.5, -1.5], atol=1e-6)
    cam = pt.camera()
    assert abs(cam.aperture_radius - 0.1) < 1e-7 and np.allclose(list(cam.position), [0, 0, 5])
    assert scene.mesh_count == 4 and pt.array("triangles").size // 24 == 1 + 20 * 64 + 2 + 12
    pt.close(); scene.close()
