"""rt_set_node_format: the merged wavefront's traversal launches read either the uploaded 80-byte CWBVH nodes
(RT_NODES_REFERENCE, the bytes of CUDA/Raytracing/BVH8.h:19-25) or the library's 96-byte decoded copy of them
(RT_NODES_DECODED, the default: kernels_trace.hip "decoded nodes"). The decoded node holds the same numbers with the
exponent / meta bytes pre-expanded, and the slab tests run as packed fused multiply-adds: every float is the one the
80-byte walk computes, so frames, AOVs and per-bounce queue sizes must be IDENTICAL -- in the flattened layout (the
engine without TLAS code), in the reference's layout (TLAS slots decoded again whenever a new TLAS arrives) and while
instances move. Every other GPU test renders with the default (decoded) and meets the oracle, which walks the 80 bytes."""
import ctypes

import numpy as np
import pytest

from conftest import make_pathtracer
from test_gpu_parity import _render_plan, compare_frames

pytestmark = pytest.mark.gpu


def _with_format(node_format):
    def prepare(lib, pt):
        from gpu_raytracer_amd import set_node_format
        set_node_format(pt.ctx, node_format)
    return prepare


@pytest.mark.parametrize("merge_static", [1, 0])
def test_decoded_nodes_render_what_the_reference_nodes_render(grt, merge_static):
    plan = [(0, 4), (0, 4), (4, 2)]
    config = dict(num_bounces=6, merge_static=merge_static)
    aovs = (grt.AOV_ALBEDO, grt.AOV_NORMAL)
    decoded = _render_plan(grt, "sponza", 640, 360, "merged", plan, config, _with_format("decoded"), aovs)
    reference = _render_plan(grt, "sponza", 640, 360, "merged", plan, config, _with_format("reference"), aovs)
    assert np.array_equal(decoded[0], reference[0]) and decoded[0][..., :3].max() > 0.0
    for a, b in zip(decoded[1], reference[1]):
        assert np.array_equal(a, b)
    assert decoded[2] == reference[2] and sum(decoded[2][0]) > 640 * 360
    grt.config_reset()


def test_decoded_nodes_follow_a_moving_instance_and_a_format_switch(grt, oracle, tmp_path):
    """40 instances beside a flattened floor: the TLAS changes every frame, its node slots are decoded again each time; the
    format is switched in the middle of the sequence (the wavefront drains, the copy is rebuilt). Frames against the oracle."""
    from test_tlas import instanced_scene_file
    grt.config_reset(); grt.config_set(num_bounces=4, static_mesh_copy_limit_mb=1)   # (the blob stays instanced: 40 TLAS leaves beside the flattened tree's)
    scene = grt.Scene(instanced_scene_file(str(tmp_path / "s"), count=40)); grt.config_set(num_bounces=4, static_mesh_copy_limit_mb=1)
    pt = grt.Pathtracer(scene, 192, 128, device=0); pt.update()
    compare_frames(grt, oracle, pt, 2, 192, 128)
    for step, node_format in enumerate(("decoded", "reference", "decoded")):
        grt.set_node_format(pt.ctx, node_format)
        position, rotation, scale = scene.mesh_transform(7)
        scene.set_mesh_transform(7, (position[0], position[1] + 0.25, position[2]), rotation, scale)
        pt.invalidate("scene"); pt.update()
        compare_frames(grt, oracle, pt, 2, 192, 128)
    pt.close(); scene.close(); grt.config_reset()


def test_the_node_cache_does_not_change_a_frame(grt):
    """rt_set_node_cache: the flattened scene's traversal launch reads the top three levels of its tree from an LDS copy (the same
    80 bytes per node). Frames, AOVs and queue sizes are those of the launch that reads every node from global memory; a scene
    whose tree has fewer nodes than the cache holds (Cornell box: 2 nodes) included."""
    for scene_name, w, h in (("sponza", 640, 360), ("cornellbox", 320, 240)):
        results = []
        for node_cache in (1, 0):
            plan = [(0, 4), (4, 4)]
            results.append(_render_plan(grt, scene_name, w, h, "merged", plan, dict(num_bounces=6, node_cache=node_cache), None, (grt.AOV_ALBEDO,)))
        assert np.array_equal(results[0][0], results[1][0]) and results[0][0][..., :3].max() > 0.0, scene_name
        assert np.array_equal(results[0][1][0], results[1][1][0]) and results[0][2] == results[1][2], scene_name
    scene, pt = make_pathtracer(grt, "sponza", 64, 36, 0)
    assert pt.static_geometry_whole_scene and 9 < pt.static_geometry_node_cache[1] <= 64
    pt.close(); scene.close(); grt.config_reset()
