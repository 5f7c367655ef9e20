#!/usr/bin/env python3
"""GPU box: the pruned closest-hit walk against the unpruned one (rt_set_option "prune" 0, which visits every cell of the
root interval like the reference) on random scenes at sizes the CPU oracle cannot cover: 1280x720, depth 6, the mesh
under a random transform, trees built on the device; then the same scene with the vertices moved under the old tree
(rt_update_vertices), where faces hang far out of the cells that list them. Every float colour, byte, hit id and ray
count must be equal. Usage: tools/prune_soak.py FIRST_SEED END_SEED"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401
import test_gpu_parity as T  # noqa: E402
from ray_tracer_amd import abi  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
W, H, D = 1280, 720, 6
bad, rays = [], 0
ctx = {}
for p in (1, 0):
    ctx[p] = abi.RayTracerHIP(abi.default_config(W, H, depth=D))
    ctx[p].setOption("prune", p)


def frames():
    out = []
    for p in (1, 0):
        ctx[p].trace()
        out.append((ctx[p].readFloat(), ctx[p].readRGBA8(), ctx[p].readHitIds(), list(ctx[p].stats().rays_per_level)))
    return out


def equal(a, b):
    return (((a[0] == b[0]) | (np.isnan(a[0]) & np.isnan(b[0]))).all() and np.array_equal(a[1], b[1]) and
            np.array_equal(a[2], b[2]) and a[3] == b[3])


for seed in range(lo, hi):
    sc, cam = T._random_scene(seed, True)
    flat = abi.FlatScene(sc, abi.default_config(W, H, depth=D), cam)
    rng = np.random.default_rng(seed)
    for p in (1, 0):
        ctx[p].setScene(flat, kd_on_device=bool(seed & 1))
    a, b = frames()
    ok = equal(a, b)
    rays += sum(a[3])
    v = np.asarray(flat.verts, np.float64) + rng.uniform(-15, 15, 3) + rng.uniform(-0.5, 0.5, (len(flat.verts), 3))
    for p in (1, 0):
        ctx[p].updateVertices(v)
    a, b = frames()
    ok = ok and equal(a, b)
    rays += sum(a[3])
    if not ok:
        bad.append(seed)
        print("seed", seed, "DIFFERS", flush=True)
    if seed % 20 == 19:
        print("up to seed", seed, "rays compared %.2e" % rays, "failures so far:", bad, flush=True)
print("seeds %d..%d: %.3e rays compared, %d failures %s" % (lo, hi - 1, rays, len(bad), bad))
