import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def grt():
    """The product package. Building is the driver's job (build()); do it here if it has not run."""
    import __graft_entry__ as entry
    import gpu_raytracer_amd as g
    if not (os.path.exists(g.DEVICE_LIB_PATH) and os.path.exists(g.HOST_LIB_PATH) and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        entry.build()
    return g


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
