#!/usr/bin/env python3
"""GPU box: many device rebuilds of a moving mesh, each compared with the host builder's tree of the same vertices, while a
second context keeps tracing frames on another stream. Usage: tools/kd_build_soak.py [rebuilds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_loader  # noqa: E402

rt_loader.load()
from ray_tracer_amd import abi, scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
teapot = (os.path.join(ROOT, "assets", "models", "teapot.obj"), 60.0, (50.0, 75.0, 10.0))
sc = scene.default_scene(os.path.join(ROOT, "assets", "models", "bun_zipper.ply"), extra_mesh=teapot)
cfg = abi.default_config(1920, 1080, depth=3)
flat = abi.FlatScene(sc, cfg, scene.camera_rays(1920, 1080))
dev = abi.RayTracerHIP(cfg)
dev.setScene(flat, build_kd=False)
busy = abi.RayTracerHIP(cfg)          # keeps the GPU occupied with walks while the other context builds
busy.setScene(flat)
host = abi.RayTracerHIP(abi.default_config(device=-1))
rng = np.random.default_rng(1)
bad = 0
for i in range(n):
    v = flat.verts * rng.uniform(0.7, 1.3, size=3) + rng.uniform(-20, 20, size=3)   # stretched and moved: another tree every time
    os.environ["RT_KD_CHAIN_FLAT"] = "0" if i % 3 == 2 else "2048"
    busy.traceBegin()
    dev.setMesh(v, flat.faces, flat.face_mat)
    dev.buildKdTree(on_device=True)
    pairs, leaves, box = dev.exportKdTree()
    busy.traceEnd()
    host.setMesh(v, flat.faces, flat.face_mat)
    host.buildKdTree()
    hp, hl, hb = host.exportKdTree()
    ok = np.array_equal(hp, pairs) and np.array_equal(hl, leaves) and np.array_equal(hb, box)
    bad += 0 if ok else 1
    if not ok or i % 50 == 49:
        print("rebuild %d: %s (%d nodes, %d leaves)" % (i + 1, "equal" if ok else "DIFFERENT", len(pairs), len(leaves)), flush=True)
print("%d rebuilds, %d different from the host builder's tree" % (n, bad))
sys.exit(1 if bad else 0)
