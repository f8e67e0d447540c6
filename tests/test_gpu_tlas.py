"""GPU part of the device TLAS build (see tests/test_tlas.py for the CPU part): the kernel reproduces the bytes of its
one-thread restatement; rays traced through the device-built TLAS hit what they hit through the host-built one; frames
rendered with it match the oracle, also while instances move and several frames are in flight."""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import make_pathtracer
from test_tlas import check_tlas, instanced_scene_file, world_boxes_of

pytestmark = pytest.mark.gpu

REL_L1_TOL = 1e-4


def read_tlas(grt, pt, n):
    lib = grt.device_lib()
    lib.rt_read_tlas.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    order = np.zeros(n, np.int32); nodes = np.zeros((2 * n, 80), np.uint8); count = ctypes.c_int32(0)
    assert lib.rt_read_tlas(pt.ctx, order.ctypes.data, nodes.ctypes.data, 2 * n, ctypes.byref(count)) == 0, lib.rt_last_error(pt.ctx)
    return nodes[:count.value], order


@pytest.mark.reference_layout
@pytest.mark.parametrize("count", [1, 5, 60, 1500])
def test_device_tlas_equals_its_restatement_and_traces_like_the_host_tlas(grt, oracle, tmp_path, count):
    path = instanced_scene_file(str(tmp_path / "s"), count=count)
    w, h = 160, 100
    hits, orders = {}, {}
    for device_tlas in (1, 0):
        grt.config_reset(); grt.config_set(device_tlas=device_tlas, num_bounces=3)
        scene = grt.Scene(path)
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
        n = scene.mesh_count
        if device_tlas:
            nodes, order = read_tlas(grt, pt, n)
            transforms = pt.array("scene_order_transforms").reshape(-1, 12).copy(); boxes = pt.array("scene_order_boxes").reshape(-1, 6).copy()
            want_nodes, want_order = oracle.tlas_build(transforms, boxes)
            assert np.array_equal(order, want_order) and np.array_equal(nodes, want_nodes)       # byte for byte
            check_tlas(nodes, order, world_boxes_of(transforms, boxes))
            orders[1] = order
        else:
            orders[0] = pt.array("tlas_indices").copy()
        view = oracle.SceneView(pt)
        o, d, _ = view.generate(0, 0, w * h)
        rng = np.random.default_rng(7)
        extra_o = rng.uniform(-14, 14, (3, 20000)).astype(np.float32); extra_d = rng.normal(size=(3, 20000)).astype(np.float32); extra_d /= np.linalg.norm(extra_d, axis=0)
        o = np.concatenate([o, extra_o], axis=1); d = np.concatenate([d, extra_d], axis=1)
        hits[device_tlas], _ = grt.trace_rays(pt.ctx, o, d)
        if device_tlas:   # the oracle walks the very nodes the device built (host view of the device TLAS): bit-exact, instance ids included
            want, _ = view.trace(o, d)
            assert np.array_equal(hits[1], want)
        pt.close(); scene.close()
    hit = hits[0][:, 1] != 0xffffffff
    assert hit.mean() > 0.3 and np.array_equal(hits[0][:, 1:], hits[1][:, 1:])
    assert np.array_equal(orders[0][hits[0][hit, 0].astype(np.int64)], orders[1][hits[1][hit, 0].astype(np.int64)])
    grt.config_reset()


def test_frames_with_a_device_tlas_match_the_oracle_while_instances_move(grt, oracle, tmp_path):
    """enable_scene_update: every frame the instances move and Integrator::build_tlas runs -- on the device (device_tlas = 1;
    the default -1 picks it for such scenes from 1024 instances on). (a) each frame against the oracle, which reads the device-built TLAS back; the light tables
    name instances by scene index and the device maps them; (b) six frames accumulated with 1 and with 3 frames in flight
    (slot scheduler: every chain reads the TLAS version it was submitted with) are bit-identical."""
    path = instanced_scene_file(str(tmp_path / "s"), count=40)
    w, h = 200, 120
    lib = grt.device_lib()

    def move(scene, base, frame):
        for m in range(3, scene.mesh_count):
            pos, _, scale = base[m]
            a = 0.35 * frame + 0.2 * m
            scene.set_mesh_transform(m, [pos[0] + 0.6 * np.sin(a), pos[1], pos[2] + 0.6 * np.cos(a)], [0.0, float(np.sin(a / 2)), 0.0, float(np.cos(a / 2))], scale)

    grt.config_reset(); grt.config_set(num_bounces=3, enable_scene_update=1, device_tlas=1)
    scene = grt.Scene(path)
    base = [scene.mesh_transform(m) for m in range(scene.mesh_count)]
    pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
    for frame in range(3):
        move(scene, base, frame); pt.update()
        lib.rt_render_sample.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert lib.rt_render_sample(pt.ctx, 0) == 0
        nodes, order = read_tlas(grt, pt, scene.mesh_count)                # built on the device: rt_read_tlas has something to read
        view = oracle.SceneView(pt); ref = oracle.Frame(view)
        oc = ref.render_sample(0); c = pt.counters()
        assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(list(c.trace[:3]), list(oc.trace[:3]))), (list(c.trace[:3]), list(oc.trace[:3]))
        assert all(abs(a - b) <= 2 + 0.002 * b for a, b in zip(list(c.shadow[:3]), list(oc.shadow[:3])))
        got, want = pt.read_framebuffer()[:, :w, :3], ref.final[:, :w, :3]
        assert np.abs(got - want).sum() / want.sum() < REL_L1_TOL, frame
    pt.close(); scene.close()

    images = []
    for in_flight in (1, 3):
        grt.config_reset(); grt.config_set(num_bounces=3, enable_scene_update=1, device_tlas=1)
        scene = grt.Scene(path)
        pt = grt.Pathtracer(scene, w, h, device=0); pt.update()
        grt.set_samples_in_flight(pt.ctx, in_flight)
        for frame in range(40):      # more frames than the scene ring has versions (12)
            move(scene, base, frame); pt.update()
            assert lib.rt_render_sample(pt.ctx, frame) == 0
        images.append(pt.read_framebuffer().copy())
        pt.close(); scene.close()
    assert np.array_equal(images[0], images[1]) and images[0][..., :3].max() > 0.0
    grt.config_reset()


def test_merged_wavefront_on_a_scene_with_more_instances_than_its_lds_root_table(grt, tmp_path):
    """The fused traversal launch keeps the BLAS roots of up to 1 024 instances in LDS and fetches every node -- TLAS nodes
    included -- from the BLAS node array, into whose reserved slots the TLAS is copied. Beyond 1 024 instances the roots
    come from global memory: frames of a 1 203-instance scene (host TLAS and device TLAS) are bit-identical under the
    merged wavefront and under the per-submission chains, whose kernels read the TLAS from its own buffer."""
    import ctypes
    path = instanced_scene_file(str(tmp_path / "s"), count=1200)
    lib = grt.device_lib()
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    for device_tlas in (0, 1):
        images = []
        for scheduler in ("merged", "slots"):
            grt.config_reset(); grt.config_set(device_tlas=device_tlas, num_bounces=4)
            scene = grt.Scene(path)
            pt = grt.Pathtracer(scene, 160, 100, device=0); pt.update()
            assert scene.mesh_count > 1024
            grt.set_scheduler(pt.ctx, scheduler)
            for first in (0, 2, 4):
                assert lib.rt_render_samples(pt.ctx, first, 2) == 0
            images.append(pt.read_framebuffer().copy())
            pt.close(); scene.close()
        assert np.array_equal(images[0], images[1]) and images[0][..., :3].max() > 0, device_tlas
    grt.config_reset()


def test_device_tlas_switched_on_after_the_scene_was_flattened(grt, oracle, tmp_path):
    """A scene is staged with its static instances flattened (the default; here the whole scene: rays start inside node 0). Then
    device_tlas is switched on at run time: the TLAS the device builds has a leaf per scene instance and node 0 becomes its
    root -- the integrator has to stage the reference's layout again and rt_build_tlas has to move the ray entry back above the
    instances, or every ray would read TLAS leaves as triangles. Frames before and after against the oracle."""
    from test_gpu_parity import compare_frames
    path = instanced_scene_file(str(tmp_path / "s"), count=2)     # floor + two emitters + 2 blobs of one mesh: all five stand still, all are flattened
    grt.config_reset(); grt.config_set(num_bounces=4)
    scene = grt.Scene(path); grt.config_set(num_bounces=4)
    pt = grt.Pathtracer(scene, 160, 100, device=0); pt.update()
    assert pt.static_geometry_whole_scene and pt.static_geometry_members == scene.mesh_count
    compare_frames(grt, oracle, pt, 2, 160, 100)
    grt.config_set(device_tlas=1)
    pt.invalidate("scene"); pt.update()
    assert pt.static_geometry_members == 0 and not pt.static_geometry_whole_scene
    compare_frames(grt, oracle, pt, 2, 160, 100)
    position, rotation, scale = scene.mesh_transform(4)
    scene.set_mesh_transform(4, (position[0] + 1.0, position[1], position[2]), rotation, scale)
    pt.invalidate("scene"); pt.update()
    compare_frames(grt, oracle, pt, 2, 160, 100)
    pt.close(); scene.close(); grt.config_reset()
