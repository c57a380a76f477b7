#!/usr/bin/env python3
"""Ray files for tools/probe/trav_policy_sim.c (CPU only).

  python tools/probe/trav_policy_sim.py fill   /tmp/rays_fill.bin     # bench.py's pixel_rays_fill scene, every 8th image row
  python tools/probe/trav_policy_sim.py tunnel /tmp/rays_tunnel.bin   # camera -> node rays of the bench's tunnel model (primary pass)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from upsp_processing_amd import synthetic as syn  # noqa: E402


def write(path, soup, org, dirs):
    with open(path, "wb") as f:
        np.asarray([soup.shape[0], dirs.shape[0]], np.int64).tofile(f)
        np.asarray(org, np.float32).tofile(f)
        np.ascontiguousarray(soup, np.float32).tofile(f)
        np.ascontiguousarray(dirs, np.float32).tofile(f)
    print(path, soup.shape[0], "triangles,", dirs.shape[0], "rays")


def pixel_rays(cam, size):
    K, R, t = [np.asarray(cam[k], np.float64) for k in ("K", "R", "t")]
    c = -R.T @ t
    v, u = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    pc = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], -1).reshape(-1, 3)
    d = pc @ R
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return c.astype(np.float32), d.astype(np.float32)


def main():
    kind, path = sys.argv[1], sys.argv[2]
    size = 1024
    if kind == "fill":
        v, t = syn.cube_sphere(289, 6.0)
        s9, _ = syn.soup(v, t)
        cam = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.95)
        org, dirs = pixel_rays(cam, size)
        rows = np.arange(0, size, 8)
        dirs = dirs.reshape(size, size, 3)[rows].reshape(-1, 3)
        write(path, s9.reshape(-1, 9), org, dirs)
    else:
        # bench.py's default model and camera; the rays of the primary pass: camera centre -> every node whose normal makes more
        # than 110 degrees with that direction (psp_process.cpp:298-306), in node order (= the order of the dense list)
        verts, tris = syn.tunnel_model_quad()
        s9, _ = syn.soup(verts, tris)
        nrm = syn.node_normals(verts, tris)
        cam = syn.pinhole_camera(size, size, center=(0, 0, 20), half_extent=6.0, fill=0.7)
        R, t = np.asarray(cam["R"], np.float64), np.asarray(cam["t"], np.float64)
        c = -R.T @ t
        d = verts.astype(np.float64) - c
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        cosang = (d * nrm.astype(np.float64)).sum(1)
        keep = cosang < np.cos(np.radians(110.0))
        write(path, s9.reshape(-1, 9), c.astype(np.float32), d[keep].astype(np.float32))


if __name__ == "__main__":
    main()
