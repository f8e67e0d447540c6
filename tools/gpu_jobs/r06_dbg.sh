cd $GRAFT_REPO_ROOT
timeout 300 python - <<'PY' 2>&1 | grep -v WARNING | head -60
import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import gpu_raytracer_amd as grt
from conftest import make_pathtracer
grt.config_reset(); grt.config_set(device_blas=1, static_slot_learning_rays=0)
scene = grt.Scene(grt.scene_path("sponza")); grt.config_set(device_blas=1, static_slot_learning_rays=0)
pt = grt.Pathtracer(scene, 320, 180, device=0); pt.update()
nodes = pt.array("bvh8_nodes").view(np.uint8).reshape(-1, 80); words = nodes.view(np.uint32).reshape(-1, 20)
root = pt.static_geometry_top_levels[0]
tri = pt.array("triangles").reshape(-1, 24); alias = pt.array("alias_mesh_ids")
first_copy = int(np.nonzero(alias >= 0)[0].min()); print("nodes", len(nodes), "root", root, "triangles", len(tri), "first copy", first_copy, "copies", int((alias >= 0).sum()), "all copies behind:", bool((alias[first_copy:] >= 0).all()))
# walk the flattened tree: ranges
sys.setrecursionlimit(10000)
bad = [0]; maxdepth = [0]
def walk(k, depth):
    maxdepth[0] = max(maxdepth[0], depth)
    imask = int(words[k, 3] >> 24); base_c = int(words[k, 4]); base_t = int(words[k, 5]); meta = nodes[k, 24:32]
    lo, hi = None, None
    for s in range(8):
        m = int(meta[s])
        if m and not (imask >> s) & 1:
            a = base_t + (m & 31); b = a + bin(m >> 5).count("1")
            lo = a if lo is None else min(lo, a); hi = b if hi is None else max(hi, b)
    rank = 0
    for s in range(8):
        if (imask >> s) & 1:
            c = base_c + rank; rank += 1
            if c >= len(nodes) or c <= k: bad[0] += 1; continue
            clo, chi = walk(c, depth + 1)
            if clo is not None:
                lo = clo if lo is None else min(lo, clo); hi = chi if hi is None else max(hi, chi)
    own_first = base_t
    if lo is not None and own_first > lo and any(int(meta[s]) and not (imask >> s) & 1 for s in range(8)): bad[0] += 0
    return lo, hi
lo, hi = walk(root, 0)
print("tree covers triangles", lo, hi, "depth", maxdepth[0], "bad child indices", bad[0])
PY
