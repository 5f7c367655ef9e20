#!/usr/bin/env python3
"""GPU box: time of a k-d rebuild of the bench mesh (bunny + teapot), host builder against device builder."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rt_loader  # noqa: E402

rt_loader.load()
from ray_tracer_amd import abi, scene  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "bun_zipper.ply"
teapot = (os.path.join(ROOT, "assets", "models", "teapot.obj"), 60.0, (50.0, 75.0, 10.0))
sc = scene.default_scene(os.path.join(ROOT, "assets", "models", model), extra_mesh=teapot)
cfg = abi.default_config(640, 480)
flat = abi.FlatScene(sc, cfg, scene.camera_rays(640, 480))
rt = abi.RayTracerHIP(cfg)
rt.setScene(flat, build_kd=False)
for dev in (False, True):
    best = 1e9
    for i in range(6):
        rt.setMesh(flat.verts, flat.faces, flat.face_mat)
        t0 = time.perf_counter()
        rt.buildKdTree(on_device=dev)
        t1 = time.perf_counter()
        rt.trace()          # includes the upload of a host-built tree
        t2 = time.perf_counter()
        st = rt.stats()
        if i:
            best = min(best, (t1 - t0) * 1e3)
        print("%s build %.2f ms (library: %.2f ms build, %.2f ms upload), first trace after it %.2f ms"
              % ("device" if dev else "host  ", (t1 - t0) * 1e3, st.kd_build_ms, st.kd_upload_ms, (t2 - t1) * 1e3))
    print("%s best of 5: %.2f ms for %d faces" % ("device" if dev else "host", best, len(flat.faces)))
