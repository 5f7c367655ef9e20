"""MI355X-native hot path of pxk5958/ray-tracer: the code behind the reference's `RayTracer` class
(k-d build, Whitted ray loop, tone reproduction) as gfx950 HIP kernels behind a C ABI.

  csrc/        HIP kernels, host-side k-d builder, the C ABI (include/rt_hip.h)
  js/          the product host: Node.js scene API + RayTracer class over an N-API addon
  abi.py       ctypes binding used by the Python driver (tests, bench)
  scene.py, three_f32.py   Python twin of the scene API for that driver

The directory name contains a hyphen, so it is imported through `rt_loader.load()` under the module
name `ray_tracer_amd`.
"""
from . import abi, multi_gpu, scene, three_f32  # noqa: F401
