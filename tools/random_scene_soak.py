#!/usr/bin/env python3
"""GPU box: the random-scene parity test of tests/test_gpu_parity.py (GPU == oracle on everything it compares) over a
range of seeds far beyond the 24 the test suite runs. Usage: tools/random_scene_soak.py FIRST_SEED END_SEED"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import conftest  # noqa
import test_gpu_parity as T
bad = []
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    try:
        T.test_random_scenes_match_the_oracle(seed)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "FAILED", str(e)[:200], flush=True)
    if seed % 50 == 49: print("up to seed", seed, "failures so far:", bad, flush=True)
print("seeds %d..%d: %d failures %s" % (lo, hi - 1, len(bad), bad))
