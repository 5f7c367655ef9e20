#!/usr/bin/env python3
"""Counts, with the CPU oracle, how many closest-hit node visits of a frame an exact pruning rule would drop
(see prune_probe.c). Usage: python tools/prune_probe/run.py [--width 3840 --height 2160 --step 2 --depth 5]"""
import argparse
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

SO = os.path.join(HERE, "libprune_probe.so")
subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared",
                       "-o", SO, os.path.join(HERE, "prune_probe.c"), os.path.join(ROOT, "oracle", "kd_oracle.c"), "-lm"])
import rt_loader  # noqa: E402
rt_loader.load()
from oracle import oracle  # noqa: E402
oracle.LIB_PATH = SO          # the probe library contains the whole oracle
from ray_tracer_amd import abi, scene  # noqa: E402
import bench  # noqa: E402


class Mode(C.Structure):
    _fields_ = [("prune", C.c_int), ("abs_slack", C.c_float), ("rel_slack", C.c_float), ("per_node", C.c_int)]


MAXM = 8
A8 = C.c_uint64 * 8


class Counts(C.Structure):
    _fields_ = [("rays", A8), ("skipped_whole", A8 * MAXM), ("visits", A8 * MAXM), ("pruned_pops", A8 * MAXM),
                ("tri_tests", A8 * MAXM), ("mismatches", C.c_uint64 * MAXM), ("ref_mismatch", C.c_uint64),
                ("min_margin", C.c_double * MAXM)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--step", type=int, default=2)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--model", default="bun_zipper.ply")
    ap.add_argument("--teapot", type=int, default=1)
    a = ap.parse_args()
    sc = bench.build_workload(a.model, bool(a.teapot))
    cfg = abi.default_config(a.width, a.height, depth=a.depth)
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(a.width, a.height))
    o = oracle.OracleRenderer(flat)
    L = oracle.lib()
    L.probe_sizeof_counts.restype = C.c_size_t
    L.probe_setup.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.probe_counts_get.argtypes = [C.c_void_p]
    assert L.probe_sizeof_counts() == C.sizeof(Counts)
    modes = [(0, 0, 0, 0), (1, 0, 0, 0), (1, 0.05, 2.0 ** -12, 0), (1, 0.25, 2.0 ** -10, 0), (1, 1.0, 2.0 ** -8, 0),
             (1, 0, 0, 1), (1, 0.25, 2.0 ** -10, 1), (1, 0, 0, 2)]
    names = ["none", "tree D, no slack", "tree D +0.05 +2^-12", "tree D +0.25 +2^-10", "tree D +1.0 +2^-8",
             "subtree D, no slack", "subtree D +0.25 +2^-10", "D = 0 (inexact limit)"]
    arr = (Mode * len(modes))(*[Mode(*m) for m in modes])
    ov = (C.c_float * 6)()
    hist = (C.c_uint64 * 64)()
    L.probe_setup(C.addressof(o.s), arr, len(modes), ov, hist)
    print("overhang of listed faces beyond their leaf's cell: below x y z = %.3f %.3f %.3f, above = %.3f %.3f %.3f" % tuple(ov))
    box = [float(v) for v in list(o.s.kd_box_min) + list(o.s.kd_box_max)]
    print("root box", box)
    h = np.array(list(hist), dtype=np.float64)
    cum = np.cumsum(h) / h.sum()
    print("listed (leaf, face) pairs: %d; overhang quantiles: " % h.sum() + ", ".join("%.0f%% <= %.2f" % (100 * q, (np.searchsorted(cum, q) + 1) / 4) for q in (0.5, 0.9, 0.99, 0.999)))
    o.trace(step=a.step, threads=os.cpu_count(), want_float=False, want_ids=False)
    c = Counts()
    L.probe_counts_get(C.byref(c))
    rays = list(c.rays)
    print("closest-hit rays in the box per level:", rays[:a.depth])
    base = [c.visits[0][l] for l in range(8)]
    print("%-26s %14s %8s %10s %10s %9s" % ("mode", "node visits", "share", "pruned pops", "tri tests", "mismatch"))
    for m, name in enumerate(names):
        v = sum(c.visits[m][l] for l in range(8))
        pp = sum(c.pruned_pops[m][l] for l in range(8))
        tt = sum(c.tri_tests[m][l] for l in range(8))
        print("%-26s %14d %7.1f%% %10d %10d %9d" % (name, v, 100.0 * v / max(1, sum(base)), pp, tt, c.mismatches[m]))
        print("    per level: " + "  ".join("%.1f%%" % (100.0 * c.visits[m][l] / max(1, base[l])) for l in range(a.depth)))
    print("stack walk vs the oracle's restart walk (must be 0):", c.ref_mismatch)


if __name__ == "__main__":
    main()
