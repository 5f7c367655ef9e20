#!/usr/bin/env python3
"""GPU box: how long the host needs to queue one frame (rt_trace_begin) against how long the frame takes, for three
frame sizes of the default scene with bunny and teapot."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import rt_loader; rt_loader.load()
import numpy as np
from ray_tracer_amd import abi, scene
ROOT = os.getcwd()
teapot = (os.path.join(ROOT, "assets", "models", "teapot.obj"), 60.0, (50.0, 75.0, 10.0))
for (w, h, depth, label) in ((3840, 2160, 5, "4K depth 5"), (1920, 1080, 1, "1080p depth 1"), (640, 480, 5, "640x480 depth 5")):
    sc = scene.default_scene(os.path.join(ROOT, "assets", "models", "bun_zipper.ply"), extra_mesh=teapot)
    cfg = abi.default_config(w, h, depth=depth)
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(w, h))
    rt = abi.RayTracerHIP(cfg)
    rt.setScene(flat)
    rt.setOption("stage_timing", 0)
    for _ in range(3):
        rt.trace()
    tb = te = 0.0
    n = 20
    for _ in range(n):
        t0 = time.perf_counter(); rt.traceBegin(); t1 = time.perf_counter(); rt.traceEnd(); t2 = time.perf_counter()
        tb += t1 - t0; te += t2 - t0
    print("%-16s: host queues a frame in %.0f us, the frame takes %.0f us" % (label, tb / n * 1e6, te / n * 1e6))
    rt.close()
