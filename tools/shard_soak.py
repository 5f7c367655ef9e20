#!/usr/bin/env python3
"""GPU box: frames of random sizes cut into random numbers of shards (1 .. 20, more shards than tiles among them): the
assembled render target and the canvas presented by shards must equal the unsharded frame's, byte for byte
(sharded_render_check of tests/test_gpu_parity.py). Usage: tools/shard_soak.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401
import test_gpu_parity as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for i in range(n):
    w, h = int(rng.choice([1, 5, 8, 9, 17, 64, 100, 203, 320])), int(rng.choice([1, 3, 8, 15, 64, 117, 200]))
    shards, depth = int(rng.integers(1, 21)), int(rng.integers(1, 5))
    try:
        T.sharded_render_check(shards, w, h, depth)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("case %d: %dx%d, %d shards, depth %d: %s: %s" % (i, w, h, shards, depth, type(e).__name__, str(e)[:300]), flush=True)
    if i % 20 == 19:
        print("%d cases, %d failures" % (i + 1, bad), flush=True)
print("%d cases, %d failures" % (n, bad))
sys.exit(1 if bad else 0)
