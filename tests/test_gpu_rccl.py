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


# ---- two ranks through the native exchange on ONE GPU (missing 3 / next-round item 4) -------------------------------------------------
# RCCL refuses a device twice in one communicator, so on a one-GPU box the collective above never moves another rank's tiles. tests/support/libloopback_ccl.so
# (GRT_COLLECTIVE_LIBRARY) is a stand-in for librccl.so whose ncclAllGather stages every rank's chunk through /dev/shm between PROCESSES: the product code that
# runs is exactly the multi-GPU path -- rt_comm_unique_id on rank 0, the 128 bytes handed over through a side channel (a file), rt_comm_init_rank(rank, 2),
# rt_set_pixel_tiles(tile, rank, 2), render, rt_all_gather_framebuffer (pack kernel -> ncclAllGather on the context's stream -> unpack kernel).
LOOPBACK = os.path.join(ROOT, "tests", "support", "libloopback_ccl.so")
SPLIT_W, SPLIT_H, SPLIT_TILE_ROWS, SPLIT_FRAMES = 352, 200, 8, 3


def _rank_child(rank, world, id_file, out_file):
    import time
    sys.path.insert(0, ROOT)
    import gpu_raytracer_amd as grt
    lib = grt.device_lib()
    lib.rt_comm_unique_id.argtypes = [ctypes.c_void_p]
    lib.rt_comm_init_rank.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.rt_comm_destroy.argtypes = [ctypes.c_void_p]
    lib.rt_all_gather_framebuffer.argtypes = [ctypes.c_void_p]
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 3
    uid = ctypes.create_string_buffer(128)
    if rank == 0:
        assert lib.rt_comm_unique_id(uid) == 0, "the collective library could not be bound"
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.replace(id_file + ".tmp", id_file)
    else:
        deadline = time.time() + 120
        while not os.path.exists(id_file):
            assert time.time() < deadline, "rank 0 never published the communicator id"
            time.sleep(0.05)
        uid.raw = open(id_file, "rb").read()
    scene, pt = make_pathtracer(grt, "sponza", SPLIT_W, SPLIT_H, 0, num_bounces=4)
    ctx = pt.ctx
    assert lib.rt_set_pixel_tiles(ctx, pt.pitch * SPLIT_TILE_ROWS, rank, world) == 0, lib.rt_last_error(ctx)
    assert lib.rt_comm_init_rank(ctx, uid, rank, world) == 0, lib.rt_last_error(ctx)
    frames = []
    for f in range(SPLIT_FRAMES):
        if f:
            pt.update()
        pt.render()
        assert lib.rt_all_gather_framebuffer(ctx) == 0, lib.rt_last_error(ctx)
        frames.append(pt.read_framebuffer()[:, :SPLIT_W].copy())
    assert lib.rt_comm_destroy(ctx) == 0
    pt.close(); scene.close()
    np.save(out_file, np.stack(frames))
    print("rank %d of %d: exchanged %d frames" % (rank, world, len(frames)))


def test_two_ranks_exchange_their_tiles_through_the_native_all_gather(grt, tmp_path):
    """Rank 1's tiles arrive in rank 0's frame (and rank 0's in rank 1's) through rt_all_gather_framebuffer: both ranks end with the frame ONE context renders,
    bit for bit, frame after frame (the send and receive buffers are re-used). Two processes, one GPU, the loopback stand-in as the collective library."""
    assert os.path.exists(LOOPBACK), "tests/support/libloopback_ccl.so is not built (python __graft_entry__.py)"
    world = 2
    id_file = str(tmp_path / "communicator_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GRT_COLLECTIVE_LIBRARY=LOOPBACK, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = []
    for rank in range(world):
        code = "import test_gpu_rccl as t; t._rank_child(%d, %d, %r, %r)" % (rank, world, id_file, str(tmp_path / ("rank%d.npy" % rank)))
        procs.append(subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outputs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outputs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outputs)):
        assert p.returncode == 0 and "exchanged %d frames" % SPLIT_FRAMES in out, "rank %d:\n%s" % (rank, out[-3000:])
    # what one context renders
    scene, pt = make_pathtracer(grt, "sponza", SPLIT_W, SPLIT_H, 0, num_bounces=4)
    whole = []
    for f in range(SPLIT_FRAMES):
        if f:
            pt.update()
        pt.render()
        whole.append(pt.read_framebuffer()[:, :SPLIT_W].copy())
    pt.close(); scene.close()
    whole = np.stack(whole)
    assert whole[..., :3].max() > 0.0
    for rank in range(world):
        got = np.load(str(tmp_path / ("rank%d.npy" % rank)))
        assert np.array_equal(got, whole), "rank %d" % rank
    grt.config_reset()


def test_bench_with_two_ranks_takes_the_native_exchange(grt):
    """`bench.py --gpus 2 --exchange native` with both ranks on GPU 0 (BENCH_SHARE_GPU; torch.distributed over gloo carries the id and the timing reductions):
    the run's per-frame exchange is the library's own (rt_comm_init_rank + rt_all_gather_framebuffer over the loopback stand-in), not the torch fallback."""
    import json
    assert os.path.exists(LOOPBACK)
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_SHARE_GPU="1", GRT_COLLECTIVE_LIBRARY=LOOPBACK, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--exchange", "native"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["value"] > 0
    assert "rt_all_gather_framebuffer" in out["config"]["parallelism"], out["config"]["parallelism"]
