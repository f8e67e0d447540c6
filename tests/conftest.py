import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")
    config.addinivalue_line("markers", "reference_layout: the scene is staged exactly as the reference stages it -- one BLAS per mesh under the TLAS "
                                       "(config merge_static = 0) -- because the test feeds the reference's own kernels, a golden vector they produced, or asserts that layout")


@pytest.fixture(scope="session")
def grt():
    """The product package. Building is the driver's job (build()); do it here if it has not run."""
    import __graft_entry__ as entry
    import gpu_raytracer_amd as g
    if not (os.path.exists(g.DEVICE_LIB_PATH) and os.path.exists(g.HOST_LIB_PATH) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        entry.build()
    return g


@pytest.fixture(autouse=True)
def _reference_layout(request, grt, monkeypatch):
    """Tests marked `reference_layout`: every config_reset() of the test leaves merge_static = 0 (the default, 1, flattens the
    static instances into one extra tree, which the reference's kernels and golden vectors know nothing of)."""
    if request.node.get_closest_marker("reference_layout"):
        plain_reset = grt.config_reset

        def reset():
            plain_reset(); grt.config_set(merge_static=0)
        monkeypatch.setattr(grt, "config_reset", reset)
        reset()
    yield
    if request.node.get_closest_marker("reference_layout"):
        monkeypatch.undo(); grt.config_reset()


@pytest.fixture(scope="session")
def oracle(grt):
    """CPU restatement of the reference (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture()
def fresh_config(grt):
    grt.config_reset()
    yield grt
    grt.config_reset()


def make_pathtracer(grt, scene_name, width, height, device, **config):
    grt.config_reset()
    if config:
        grt.config_set(**config)   # before the load: switches the loaders read (textures, BVH type, caches)
    scene = grt.Scene(grt.scene_path(scene_name))
    if config:
        grt.config_set(**config)   # and after it: the scene file's own film size / maxDepth must not win
    pt = grt.Pathtracer(scene, width, height, device=device)
    pt.update()
    return scene, pt


def unpack_hits(hits):
    """uint32[N,4] -> mesh_id, triangle_id (int32), t (float32), u, v (uint16)"""
    return hits[:, 0].view(np.int32), hits[:, 1].view(np.int32), hits[:, 2].view(np.float32), hits[:, 3] & 0xffff, hits[:, 3] >> 16
