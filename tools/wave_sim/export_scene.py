#!/usr/bin/env python3
"""Dump what the traversal kernel reads (CWBVH nodes, triangle positions, per-instance roots and inverse
transforms, the camera) of a scene to one binary file for tools/wave_sim/wave_sim.cpp. Host only (no GPU)."""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpu_raytracer_amd as grt

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
    out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/wave_sim_%s.bin" % name
    w, h = 1920, 1080
    grt.config_reset()
    scene = grt.Scene(grt.scene_path(name))
    pt = grt.Pathtracer(scene, w, h, device=-1); pt.update()
    nodes = np.ascontiguousarray(pt.array("bvh8_nodes")).view(np.uint8).reshape(-1, 80).copy()
    tlas = np.ascontiguousarray(pt.array("tlas_nodes")).view(np.uint8).reshape(-1, 80)
    nodes[: tlas.shape[0]] = tlas            # unified node array, as stream_sync_tlas does
    tris = np.ascontiguousarray(pt.array("triangles")).view(np.float32).reshape(-1, 24)[:, :9].copy()
    roots = np.ascontiguousarray(pt.array("mesh_bvh_root_indices")).view(np.int32)
    xinv = np.ascontiguousarray(pt.array("mesh_transforms_inv")).view(np.float32).reshape(-1, 12)
    cam = np.frombuffer(bytes(pt.array("camera")), dtype=np.float32)[:15].copy()
    with open(out, "wb") as f:
        f.write(struct.pack("6i", nodes.shape[0], tris.shape[0], roots.shape[0], tlas.shape[0], w, h))
        f.write(nodes.tobytes()); f.write(tris.tobytes()); f.write(roots.tobytes()); f.write(xinv.tobytes()); f.write(cam.tobytes())
    print("wrote %s: %d nodes (%d TLAS), %d triangles, %d instances, identity %d" % (out, nodes.shape[0], tlas.shape[0], tris.shape[0], roots.shape[0], int((roots < 0).sum())))

if __name__ == "__main__":
    main()
