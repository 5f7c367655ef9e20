import numpy as np, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import rt_loader; rt_loader.load()
from helpers import make_flat
from ray_tracer_amd import abi
flat = make_flat("bun_zipper.ply", 640, 480, depth=2)
for dev in (False, True):
    rt = abi.RayTracerHIP(flat.cfg); rt.setScene(flat, kd_on_device=dev); rt.trace()
    v = np.asarray(flat.verts, np.float64)
    import time
    for k in range(6):
        t0 = time.perf_counter(); rt.updateVertices(v + k * 0.5); dt = time.perf_counter() - t0
    print("device tree" if dev else "host tree", "rt_update_vertices call %.3f ms" % (dt * 1e3), rt.kdOverhang())
    rt.close()
