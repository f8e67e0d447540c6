"""The product's own frame exchange over RCCL (rt_comm_*, rt_all_gather_framebuffer; include/gpu_raytracer_amd.h) EXECUTED:
RCCL accepts a communicator of one rank, so rt_comm_unique_id -> rt_comm_init_rank(rank 0 of 1) -> ncclAllGather runs on
the one MI355X a test box has. (The multi-context tests share that GPU -- which RCCL refuses -- and take the peer-copy
transport; this test is the only one in which librccl.so is bound, a communicator exists and the collective is enqueued
on the context's stream.) The gathered frame must be the frame a plain context renders, bit for bit, also for a second
frame (buffers re-used) and after the communicator has been destroyed and made again."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, make_pathtracer

pytestmark = pytest.mark.gpu

W, H = 320, 192


def _render(grt, with_comm, frames=2):
    lib = grt.device_lib()
    lib.rt_comm_unique_id.argtypes = [ctypes.c_void_p]
    lib.rt_comm_init_rank.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_comm_destroy.argtypes = [ctypes.c_void_p]
    lib.rt_all_gather_framebuffer.argtypes = [ctypes.c_void_p]
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    scene, pt = make_pathtracer(grt, "cornellbox", W, H, 0, num_bounces=5)
    ctx = pt.ctx
    images = []
    if with_comm:
        assert lib.rt_set_pixel_tiles(ctx, W * 8, 0, 1) == 0, lib.rt_last_error(ctx)
    for round_ in range(2 if with_comm else 1):
        if with_comm:
            uid = ctypes.create_string_buffer(128)
            assert lib.rt_comm_unique_id(uid) == 0, "librccl.so could not be bound"
            assert lib.rt_comm_init_rank(ctx, uid, 0, 1) == 0, lib.rt_last_error(ctx)
        for f in range(frames):
            if f or round_:
                pt.update()
            pt.render()
            if with_comm:
                assert lib.rt_all_gather_framebuffer(ctx) == 0, lib.rt_last_error(ctx)
            images.append(pt.read_framebuffer()[:, :W].copy())
        if with_comm:
            assert lib.rt_comm_destroy(ctx) == 0
    pt.close(); scene.close()
    return images


def _child():
    sys.path.insert(0, ROOT)
    import gpu_raytracer_amd as grt
    plain = _render(grt, False, frames=4)
    gathered = _render(grt, True, frames=2)
    assert len(gathered) == 4
    for a, b in zip(plain, gathered):
        assert np.array_equal(a, b) and a[..., :3].max() > 0.0
    print("rccl all-gather ok")


def test_nccl_all_gather_of_a_one_rank_communicator_rebuilds_the_frame(grt):
    # in a child process under a timeout: a collective that never completes must fail this test, not hang the box
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    proc = subprocess.run([sys.executable, "-c", "import test_gpu_rccl as t; t._child()"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert proc.returncode == 0 and "rccl all-gather ok" in proc.stdout, proc.stdout[-3000:]
