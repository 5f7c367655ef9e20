#!/usr/bin/env python3
"""GPU box: random scenes (the generator of tests/test_gpu_parity.py) under random configurations the test suite does not
sweep - frame sizes from 1x1 up with any aspect ratio, recursion depth 1 .. 8, supersampling 1 .. 3, k-d trees with a
depth limit of 2 .. 24, the brute-force triangle loop, triangles switched off - GPU against oracle on everything
check_against_oracle compares. Usage: tools/random_config_soak.py FIRST_SEED END_SEED"""
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
bad = []
for seed in range(lo, hi):
    rng = np.random.default_rng(1000003 * seed + 17)
    w = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 161]))
    h = int(rng.choice([1, 2, 5, 8, 31, 64, 90, 120]))
    ss = int(rng.choice([1, 1, 2, 3]))
    depth = int(rng.integers(1, 9))
    with_mesh = bool(rng.integers(0, 2))
    kw = dict(depth=depth, supersampling=ss, tr_method=int(rng.integers(0, 2)), max_kd_depth=int(rng.choice([2, 3, 7, 12, 24])),
              use_kd_tree=int(rng.integers(0, 4) != 0), render_triangles=int(rng.integers(0, 6) != 0))
    if not kw["use_kd_tree"] and with_mesh:
        w, h = min(w, 33), min(h, 31)          # every face for every ray: keep the oracle quick
    sc, cam = T._random_scene(seed, with_mesh)
    try:
        from ray_tracer_amd import scene as S
        flat = abi.FlatScene(sc, abi.default_config(w, h, **kw), S.camera_rays(w, h) if cam is None else cam)
        rt, o, (_, got_8, _), cn = T.check_against_oracle(flat)
        Lbar, sf = rt.tone()
        oL, osf = o.tone(got_8)
        assert abs(Lbar - oL) <= T.TONE_RTOL * abs(oL) and abs(sf - osf) <= 1e-8 * abs(osf)
        assert np.array_equal(rt.present(), o.present(got_8, oL, osf))
        rt.close()
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed %d (%dx%d ss %d depth %d mesh %s %s): %s: %s" % (seed, w, h, ss, depth, with_mesh, kw, type(e).__name__, str(e)[:300]), flush=True)
    if seed % 50 == 49:
        print("up to seed", seed, "failures so far:", bad, flush=True)
print("seeds %d..%d: %d failures %s" % (lo, hi - 1, len(bad), bad))
sys.exit(1 if bad else 0)
