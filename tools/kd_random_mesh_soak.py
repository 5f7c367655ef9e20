#!/usr/bin/env python3
"""GPU box: device k-d builder against the oracle builder on random meshes of random sizes (spread out, clustered inside a
huge root box, degenerate faces among them). Usage: tools/kd_random_mesh_soak.py [meshes] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rt_loader  # noqa: E402

rt_loader.load()
from oracle import oracle  # noqa: E402
from ray_tracer_amd import abi  # noqa: E402

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
rt = abi.RayTracerHIP(abi.default_config(64, 48))
bad = 0
t0 = time.time()
for i in range(n_meshes):
    kind = i % 4
    # (a flat mesh has a root box without extent on one axis: a third of the levels separate nothing and the tree grows
    # to millions of nodes from a thousand faces - the oracle needs minutes for that, so these stay small)
    n_faces = int(rng.integers(5, 40)) if kind == 3 else int(rng.integers(5, 4000))
    v = rng.uniform(-50, 50, size=(n_faces * 2, 3))
    f = rng.integers(0, len(v), size=(n_faces, 3)).astype(np.uint32)
    size = float(rng.choice([0.2, 2.0, 10.0]))
    v[f[:, 1]] = v[f[:, 0]] + rng.normal(0, size, size=(n_faces, 3))
    v[f[:, 2]] = v[f[:, 0]] + rng.normal(0, size, size=(n_faces, 3))
    if kind == 1:      # nearly everything in one small cluster inside a huge root box
        v *= 0.02
        v[:20] = rng.uniform(-500, 500, size=(20, 3))
    elif kind == 2:    # some faces without area, some repeated
        f[: n_faces // 10, 2] = f[: n_faces // 10, 1]
        f[n_faces // 10: n_faces // 5] = f[0]
    elif kind == 3:    # flat: every vertex in one plane
        v[:, int(rng.integers(0, 3))] = 1.5
    rt.setMesh(v, f, np.zeros((n_faces, 2), np.uint32))
    op, ol, ob, _ = oracle.kd_build(v, f)
    try:
        rt.buildKdTree(on_device=True)
    except abi.RTError as e:
        print("mesh %d (%d faces, kind %d, size %.1f): %s; the oracle's tree has %d nodes, %d leaves" % (i + 1, n_faces, kind, size, e, len(op), len(ol)), flush=True)
        bad += 1
        continue
    pairs, leaves, box = rt.exportKdTree()
    ok = np.array_equal(ob, box) and np.array_equal(ol, leaves) and np.array_equal(op, pairs)
    bad += 0 if ok else 1
    if not ok or i % 10 == 9:
        print("mesh %d (%d faces, kind %d): %s, %d nodes; %.0f s" % (i + 1, n_faces, kind, "equal" if ok else "DIFFERENT", len(pairs), time.time() - t0), flush=True)
print("%d meshes, %d different from the oracle's tree" % (n_meshes, bad))
sys.exit(1 if bad else 0)
