"""N>1 path on CPU: two gloo ranks each render their tiles (with the CPU oracle, which honours
pixel ranges exactly like the device layer) and one all_gather rebuilds the frame, which must
equal the single-process render bit for bit."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
import gpu_raytracer_amd as grt
from oracle import binding as oracle
parallel = __import__("importlib").import_module("gpu_raytracer_amd.parallel")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H = 40, 27
grt.config_reset()
scene = grt.Scene(grt.scene_path("cornellbox"))
grt.config_set(num_bounces=3)
pt = grt.Pathtracer(scene, W, H, device=-1)
pt.update()
view = oracle.SceneView(pt)
split = parallel.TileSplit(rank, world, W, H, tile_rows=4)
frame = oracle.Frame(view)
for offset, count in split.ranges:
    frame.render_sample(0, offset, count)
packed = split.pack(frame.final)
full = split.unpack(split.all_gather(packed)).numpy()
if rank == 0:
    np.save(os.environ["OUT_PATH"], full)
dist.barrier()
dist.destroy_process_group()
'''


def test_tile_split_partition_is_exact():
    import importlib
    sys.path.insert(0, ROOT)
    import gpu_raytracer_amd  # noqa: F401
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")
    for world in (1, 2, 4, 8):
        for (w, h, rows) in ((1920, 1080, 8), (40, 27, 4), (33, 5, 8)):
            seen = np.zeros(w * h, int)
            for r in range(world):
                for off, cnt in parallel.rank_tiles(r, world, w, h, rows):
                    seen[off:off + cnt] += 1
            assert (seen == 1).all()
            order = parallel.gather_order(world, w, h, rows)
            assert np.unique(order).size == w * h
    # 1080p, 8 ranks: every rank owns 17 tiles of 8 rows (135 tiles padded to 136)
    s = parallel.TileSplit(3, 8, 1920, 1080)
    assert s.tiles_per_rank == 17 and s.local_pixels == 17 * 8 * 1920


def test_two_rank_gloo_render_matches_single_process(tmp_path, grt, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    out = tmp_path / "full.npy"
    env = dict(os.environ, OUT_PATH=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    got = np.load(out)

    grt.config_reset()
    scene = grt.Scene(grt.scene_path("cornellbox"))
    grt.config_set(num_bounces=3)
    pt = grt.Pathtracer(scene, 40, 27, device=-1)
    pt.update()
    frame = oracle.Frame(oracle.SceneView(pt))
    frame.render_sample(0)
    want = frame.final[:, :40, :]
    assert np.array_equal(got, want)
    pt.close(); scene.close()


SVGF_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
import gpu_raytracer_amd as grt
from oracle import binding as oracle
parallel = __import__("importlib").import_module("gpu_raytracer_amd.parallel")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H = 48, 36
grt.config_reset()
scene = grt.Scene(grt.scene_path("cornellbox"))
grt.config_set(num_bounces=3, enable_svgf=1, enable_taa=1)
pt = grt.Pathtracer(scene, W, H, device=-1)
pt.update()
view = oracle.SceneView(pt)
split = parallel.TileSplit(rank, world, W, H, tile_rows=4)
frame = oracle.Frame(view)
pitch = view.scene.screen_pitch
frames = []
for f in range(3):
    if f:
        scene.set_camera((0.03 * f, 1.0 + 0.01 * f, 6.8), (0.0, 0.004 * f, 0.0, 1.0)); pt.update()
        view.scene.camera = oracle.SceneView(pt).scene.camera
    vp = pt.view_projection()
    for i in range(16):
        view.scene.view_projection[i] = vp[0][i]; view.scene.view_projection_prev[i] = vp[1][i]
    # this rank's tiles only, no filter
    for offset, count in split.ranges:
        frame.render_sample_unfiltered(pt.sample_index, offset, count)
    # one exchange: the per-frame AOVs and g-buffers of every rank's tiles (SURVEY.md 8e)
    for name, a in frame.svgf_inputs().items():
        rows = a.reshape(H, pitch, -1)
        full = split.unpack(split.all_gather(split.pack(torch.from_numpy(rows)))).numpy()
        rows[:, :W, :] = full
    frame.filter_frame(pt.sample_index)        # every rank filters the whole frame
    frames.append(frame.final[:, :W, :].copy())
if rank == 1:   # the rank that did NOT render the first tile
    np.save(os.environ["OUT_PATH"], np.stack(frames))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_svgf_frames_match_single_process(tmp_path, grt, oracle):
    """BASELINE config 3 under the tile split: each rank path-traces its tiles, ONE all-gather per frame moves the filter's
    per-frame inputs (DIRECT / INDIRECT / ALBEDO + the three g-buffers), every rank filters the whole frame. Three frames
    with a moving camera (reprojection, history lengths, TAA) equal the single-process frames bit for bit."""
    script = tmp_path / "worker_svgf.py"
    script.write_text(SVGF_WORKER.format(root=ROOT))
    out = tmp_path / "frames.npy"
    env = dict(os.environ, OUT_PATH=str(out), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29619", str(script)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    got = np.load(out)

    grt.config_reset()
    scene = grt.Scene(grt.scene_path("cornellbox"))
    grt.config_set(num_bounces=3, enable_svgf=1, enable_taa=1)
    pt = grt.Pathtracer(scene, 48, 36, device=-1)
    pt.update()
    view = oracle.SceneView(pt)
    frame = oracle.Frame(view)
    for f in range(3):
        if f:
            scene.set_camera((0.03 * f, 1.0 + 0.01 * f, 6.8), (0.0, 0.004 * f, 0.0, 1.0)); pt.update()
            view.scene.camera = oracle.SceneView(pt).scene.camera
        vp = pt.view_projection()
        for i in range(16):
            view.scene.view_projection[i] = vp[0][i]; view.scene.view_projection_prev[i] = vp[1][i]
        frame.render_sample(pt.sample_index)
        assert np.array_equal(got[f], frame.final[:, :48, :]), f
    assert np.abs(got[2] - got[0]).max() > 0
    pt.close(); scene.close(); grt.config_reset()
