"""Shared test helpers: golden manifests, reference-shaped scenes."""
import hashlib
import json
import os

import numpy as np

from ray_tracer_amd import abi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "assets", "models")


def manifest(name):
    with open(os.path.join(GOLDEN, name, "manifest.json")) as f:
        return json.load(f)


def uniforms(m):
    return {u["name"]: u["value"] for u in m["uniforms"]}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def model(name):
    return os.path.join(MODELS, name)


def lcg_bytes(n):
    """The byte stream the fixture generator hands to readPixels (32-bit LCG, top byte)."""
    out = np.empty(n, np.uint8)
    s = 12345
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s >> 24
    return out


def lcg_bytes_fast(n):
    # closed form per index is awkward; generate in Python chunks with numpy uint64 arithmetic
    out = np.empty(n, np.uint8)
    s = np.uint64(12345)
    a, c, mask = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    # advance in blocks using the affine map composition: s_{k+1} = a s_k + c
    block = 1 << 16
    # precompute powers for a vectorised block: s_{i} = A_i * s_0 + C_i
    A = np.empty(block, np.uint64)
    Cc = np.empty(block, np.uint64)
    ai, ci = np.uint64(1), np.uint64(0)
    for i in range(block):
        ai = (ai * a) & mask
        ci = (ci * a + c) & mask
        A[i], Cc[i] = ai, ci
    pos = 0
    while pos < n:
        m = min(block, n - pos)
        vals = (A[:m] * s + Cc[:m]) & mask
        out[pos:pos + m] = (vals >> np.uint64(24)).astype(np.uint8)
        s = vals[m - 1]
        pos += m
    return out


def make_flat(model_name=None, width=640, height=480, depth=1, ss=1, time=0.0, mesh_material=None,
              second_light=True, mesh_color=(0.1, 0.75, 0.1), extra_mesh=None, **cfg_kw):
    sc = scene.default_scene(model(model_name) if model_name else None, mesh_material=mesh_material,
                             second_light=second_light, mesh_color=mesh_color, extra_mesh=extra_mesh)
    if time is not None:
        scene.animate_light(sc, time)
    cfg = abi.default_config(width, height, depth=depth, supersampling=ss, **cfg_kw)
    return abi.FlatScene(sc, cfg, scene.camera_rays(width, height))


def device_canvas(rt):
    """The presented canvas as it sits in device memory (no present pass is run), [H, W, 4] uint8."""
    import ctypes as C
    import torch
    from ray_tracer_amd import multi_gpu
    p, n = C.c_void_p(), C.c_size_t()
    rt._ck(rt.L.rt_device_canvas(rt.ctx, C.byref(p), C.byref(n)), "rt_device_canvas")
    torch.cuda.synchronize()
    return torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:%d" % rt.cfg.device).cpu().numpy().reshape(rt.cfg.height, rt.cfg.width, 4)


def dragged_frames(model_name, frames, width, height, depth, mesh_material=None, **cfg_kw):
    """benchmark configuration 5's motion: light 0 on the reference's orbit (UI.animate, js/8-kd_tree.js:2711-2716) and the mesh
    dragged along a small circle with Mesh.translate (js/8-kd_tree.js:2055-2064: every vertex through the inverse of the old
    float32 model matrix and through the new one), one translate per listed frame. Returns {frame: FlatScene}."""
    import math
    from ray_tracer_amd.three_f32 import Vector3
    sc = scene.default_scene(model(model_name), mesh_material=mesh_material)
    mesh = sc.objects[-1]
    pos = lambda k: (12.0 * math.sin(2.0 * math.pi * k / 240.0), 0.0, 12.0 * (math.cos(2.0 * math.pi * k / 240.0) - 1.0))
    out, prev = {}, 0
    for k in frames:
        p, q = pos(k), pos(prev)
        if k != prev:
            mesh.translate(Vector3(p[0] - q[0], p[1] - q[1], p[2] - q[2]))
        prev = k
        scene.animate_light(sc, k / 60.0)
        out[k] = abi.FlatScene(sc, abi.default_config(width, height, depth=depth, **cfg_kw), scene.camera_rays(width, height))
    return out
