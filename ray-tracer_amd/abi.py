"""ctypes binding of the C ABI declared in include/rt_hip.h, plus FlatScene: the scene reduced to the
arrays the reference hands to WebGL (uniforms + data textures, js/8-kd_tree.js:2433-2547).

The HIP library is mandatory: loading fails loudly when csrc/librt_hip.so is missing. There is no
CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(HERE, "csrc", "librt_hip.so")   # RT_HIP_LIB: tuning builds


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("supersampling", C.c_int32),
                ("max_depth", C.c_int32), ("tr_method", C.c_int32), ("render_triangles", C.c_int32),
                ("use_kd_tree", C.c_int32), ("max_kd_depth", C.c_int32),
                ("L_max", C.c_float), ("L_dmax", C.c_float), ("A", C.c_float), ("infinity", C.c_float),
                ("epsilon", C.c_float), ("sphere_epsilon", C.c_float), ("plane_rect_epsilon", C.c_float),
                ("triangle_epsilon", C.c_float),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("device", C.c_int32),
                ("parts", C.c_int32)]


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mat_type", C.c_int32), ("mat_id", C.c_int32), ("reserved", C.c_int32),
                ("model", C.c_float * 16), ("inv_model", C.c_float * 16)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("id", C.c_int32),
                ("ka", C.c_float), ("kd", C.c_float), ("ks", C.c_float), ("ke", C.c_float),
                ("kReflect", C.c_float), ("kRefract", C.c_float), ("ior", C.c_float),
                ("Co", C.c_float * 3), ("Co2", C.c_float * 3), ("Cs", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3)]


class Stats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("rays_per_level", C.c_uint64 * 8), ("shadow_rays", C.c_uint64),
                ("pixels", C.c_uint64), ("trace_ms", C.c_double), ("level_kernel_ms", C.c_double * 8),
                ("level_stage_ms", (C.c_double * 7) * 8), ("stage_ms", C.c_double * 7), ("tri_records", C.c_uint64 * 2),
                ("kd_closest_rays", C.c_uint64), ("kd_shadow_rays", C.c_uint64),
                ("composite_ms", C.c_double), ("tone_ms", C.c_double), ("kd_build_ms", C.c_double),
                ("kd_upload_ms", C.c_double), ("kd_device_fallbacks", C.c_uint64)]


def default_config(width=640, height=480, depth=1, supersampling=1, **kw):
    """The constants of js/8-kd_tree.js:25-46."""
    c = Config(width=width, height=height, supersampling=supersampling, max_depth=depth, tr_method=0,
               render_triangles=1, use_kd_tree=1, max_kd_depth=24, L_max=1000.0, L_dmax=500.0, A=0.18,
               infinity=100000.0, epsilon=0.0001, sphere_epsilon=0.095, plane_rect_epsilon=0.095,
               triangle_epsilon=0.0001, shard_index=0, shard_count=1, device=0, parts=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class FlatScene:
    """Scene + camera + config as flat arrays (the form both the C ABI and the oracle take)."""

    def __init__(self, scene, cfg, cam):
        self.cfg = cfg
        self.prims = []
        for o in scene.prim_table():
            self.prims.append((o.kind, o.material.type, o.material.id, o.modelTransform.toArray(),
                               o.invModelTransform.toArray()))
        self.mats = [(m.type, m.id, m.ka, m.kd, m.ks, m.ke, m.kReflect, m.kRefract, m.ior,
                      m.Co.toArray(), m.Co2.toArray(), m.Cs.toArray()) for m in scene.materials]
        self.lights = [(l.position.toArray(), l.color.toArray()) for l in scene.lights]
        self.n_lights_loop = max(scene.n_lights_declared, len(self.lights))
        self.ambient = np.array(scene.ambientLight.toArray(), np.float32)
        self.bg = np.array(scene.bgColor.toArray(), np.float32)
        self.cam = cam["cameraPos"]
        self.ray00, self.ray01, self.ray10, self.ray11 = cam["ray00"], cam["ray01"], cam["ray10"], cam["ray11"]
        self.verts, self.faces, self.face_mat = scene.mesh_arrays()
        if not len(self.faces):
            self.cfg.render_triangles = 0

    def prim_array(self):
        arr = (Prim * len(self.prims))()
        for i, (k, mt, mi, M, IM) in enumerate(self.prims):
            arr[i].kind, arr[i].mat_type, arr[i].mat_id = k, mt, mi
            arr[i].model = (C.c_float * 16)(*M)
            arr[i].inv_model = (C.c_float * 16)(*IM)
        return arr

    def mat_array(self):
        arr = (Material * len(self.mats))()
        for i, (t, mid, ka, kd, ks, ke, kr, kt, ior, Co, Co2, Cs) in enumerate(self.mats):
            a = arr[i]
            a.type, a.id, a.ka, a.kd, a.ks, a.ke, a.kReflect, a.kRefract, a.ior = t, mid, ka, kd, ks, ke, kr, kt, ior
            a.Co, a.Co2, a.Cs = (C.c_float * 3)(*Co), (C.c_float * 3)(*Co2), (C.c_float * 3)(*Cs)
        return arr

    def light_array(self):
        arr = (Light * self.n_lights_loop)()
        for i, (p, c) in enumerate(self.lights):
            arr[i].position, arr[i].color = (C.c_float * 3)(*p), (C.c_float * 3)(*c)
        return arr


_lib = None


def lib():
    global _lib
    if _lib is None:
        # PyTorch ships its own copy of the HIP runtime. If the library pulls in the system's copy first and torch is
        # imported later, the process holds two runtimes and torch finds no GPU: let torch load its copy first.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP library missing: %s (run __graft_entry__.build() / make -C ray-tracer_amd/csrc); "
                               "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
        L.rt_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.rt_destroy.argtypes = [vp]
        L.rt_destroy.restype = None
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = C.c_char_p
        L.rt_default_config.argtypes = [C.POINTER(Config)]
        L.rt_default_config.restype = None
        L.rt_set_primitives.argtypes = [vp, vp, i32]
        L.rt_set_materials.argtypes = [vp, vp, i32]
        L.rt_set_lights.argtypes = [vp, vp, i32, i32]
        L.rt_set_globals.argtypes = [vp, vp, vp]
        L.rt_set_camera.argtypes = [vp, vp, vp, vp, vp, vp]
        L.rt_set_mesh.argtypes = [vp, vp, i32, vp, vp, i32]
        L.rt_update_vertices.argtypes = [vp, vp, i32]
        L.rt_build_kdtree.argtypes = [vp]
        L.rt_build_kdtree_device.argtypes = [vp]
        L.rt_kdtree_counts.argtypes = [vp, i64p, i64p]
        L.rt_export_kdtree_reference.argtypes = [vp, vp, vp, vp]
        L.rt_kdtree_overhang.argtypes = [vp, vp]
        L.rt_trace.argtypes = [vp]
        L.rt_trace_begin.argtypes = [vp]
        L.rt_trace_end.argtypes = [vp]
        L.rt_read_float.argtypes = [vp, vp]
        L.rt_read_rgba8.argtypes = [vp, vp]
        L.rt_read_hit_ids.argtypes = [vp, vp]
        L.rt_pick.argtypes = [vp, i32, i32, C.POINTER(C.c_int32)]
        L.rt_tone_partial.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.rt_tone_finish.argtypes = [vp, C.c_double, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_tone.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_present.argtypes = [vp, vp]
        L.rt_device_rgba8.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.rt_device_float.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.rt_device_canvas.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.rt_device_shard_rgba8.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.rt_assemble_shards.argtypes = [vp, vp, C.c_size_t, i32]
        L.rt_assemble_shards_canvas.argtypes = [vp, vp, C.c_size_t, i32]
        L.rt_present_shard.argtypes = [vp]
        L.rt_device_shard_canvas.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.rt_tone_partial_device.argtypes = [vp, C.POINTER(vp)]
        L.rt_tone_finish_device.argtypes = [vp, vp]
        L.rt_tone_get.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_stream.argtypes = [vp, C.POINTER(vp)]
        L.rt_set_option.argtypes = [vp, C.c_char_p, C.c_double]
        L.rt_luminance_sums_noalloc.argtypes = [vp, C.c_uint64, C.c_float, vp, i32, vp]
        L.rt_noalloc_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class RTError(RuntimeError):
    pass


class RayTracerHIP:
    """Thin object wrapper: one rt_ctx. Method names follow the reference class
    (setScene / update / render, js/8-kd_tree.js:2347, :2433, :2575) where they correspond."""

    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.ctx = C.c_void_p()
        rc = self.L.rt_create(C.byref(cfg), C.byref(self.ctx))
        if rc != 0:
            raise RTError("rt_create: " + (self.L.rt_last_error(None) or b"?").decode())
        self.W = cfg.width * cfg.supersampling
        self.H = cfg.height * cfg.supersampling

    def _ck(self, rc, what):
        if rc != 0:
            raise RTError("%s: %s" % (what, (self.L.rt_last_error(self.ctx) or b"?").decode()))

    def close(self):
        if self.ctx:
            self.L.rt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setScene(self, flat, build_kd=True, kd_on_device=False):
        prims, mats, lights = flat.prim_array(), flat.mat_array(), flat.light_array()
        self._ck(self.L.rt_set_primitives(self.ctx, C.addressof(prims) if len(prims) else None, len(prims)), "rt_set_primitives")
        self._ck(self.L.rt_set_materials(self.ctx, C.addressof(mats) if len(mats) else None, len(mats)), "rt_set_materials")
        self._ck(self.L.rt_set_lights(self.ctx, C.addressof(lights) if len(lights) else None, len(flat.lights),
                                      flat.n_lights_loop), "rt_set_lights")
        amb = np.ascontiguousarray(flat.ambient, np.float32)
        bg = np.ascontiguousarray(flat.bg, np.float32)
        self._ck(self.L.rt_set_globals(self.ctx, amb.ctypes.data, bg.ctypes.data), "rt_set_globals")
        self.setCamera(flat.cam, flat.ray00, flat.ray01, flat.ray10, flat.ray11)
        if flat.cfg.render_triangles and len(flat.faces):
            self.setMesh(flat.verts, flat.faces, flat.face_mat)
            if build_kd and flat.cfg.use_kd_tree:
                self.buildKdTree(on_device=kd_on_device)

    def setCamera(self, pos, r00, r01, r10, r11):
        a = [np.ascontiguousarray(x, np.float32) for x in (pos, r00, r01, r10, r11)]
        self._ck(self.L.rt_set_camera(self.ctx, *[x.ctypes.data for x in a]), "rt_set_camera")

    def setLights(self, lights, n_loop):
        arr = (Light * max(len(lights), 1))()
        for i, (p, c) in enumerate(lights):
            arr[i].position, arr[i].color = (C.c_float * 3)(*p), (C.c_float * 3)(*c)
        self._ck(self.L.rt_set_lights(self.ctx, C.addressof(arr), len(lights), n_loop), "rt_set_lights")

    def setMesh(self, verts, faces, face_mat):
        v = np.ascontiguousarray(verts, np.float64)
        f = np.ascontiguousarray(faces, np.uint32)
        m = np.ascontiguousarray(face_mat, np.uint32)
        self._ck(self.L.rt_set_mesh(self.ctx, v.ctypes.data, len(v), f.ctypes.data, m.ctypes.data, len(f)), "rt_set_mesh")

    def updateVertices(self, verts):
        """Moved vertices under the tree as it stands (what the reference renders between two k-d rebuilds)."""
        v = np.ascontiguousarray(verts, np.float64)
        self._ck(self.L.rt_update_vertices(self.ctx, v.ctypes.data, len(v)), "rt_update_vertices")

    def buildKdTree(self, on_device=False):
        if on_device:
            self._ck(self.L.rt_build_kdtree_device(self.ctx), "rt_build_kdtree_device")
        else:
            self._ck(self.L.rt_build_kdtree(self.ctx), "rt_build_kdtree")

    def exportKdTree(self):
        nn, nl = C.c_int64(), C.c_int64()
        self._ck(self.L.rt_kdtree_counts(self.ctx, C.byref(nn), C.byref(nl)), "rt_kdtree_counts")
        pairs = np.zeros((nn.value, 2), np.uint32)
        leaves = np.zeros((nl.value, 4), np.int32)
        box = np.zeros(6, np.float64)
        self._ck(self.L.rt_export_kdtree_reference(self.ctx, pairs.ctypes.data, leaves.ctypes.data, box.ctypes.data),
                 "rt_export_kdtree_reference")
        return pairs, leaves, box

    def kdOverhang(self):
        """How far listed faces reach below (x, y, z) / above (x, y, z) the cells of the leaves that list them, with the
        safety margin the walk prunes with."""
        out = np.zeros(6, np.float32)
        self._ck(self.L.rt_kdtree_overhang(self.ctx, out.ctypes.data), "rt_kdtree_overhang")
        return out

    def trace(self):
        self._ck(self.L.rt_trace(self.ctx), "rt_trace")

    def traceBegin(self):
        self._ck(self.L.rt_trace_begin(self.ctx), "rt_trace_begin")

    def traceEnd(self):
        """True when the frame had to be traced again with larger buffers (work queued behind the first attempt
        must be redone)."""
        rc = self.L.rt_trace_end(self.ctx)
        if rc < 0:
            self._ck(rc, "rt_trace_end")
        return rc == 1

    def readFloat(self):
        out = np.zeros((self.H, self.W, 3), np.float32)
        self._ck(self.L.rt_read_float(self.ctx, out.ctypes.data), "rt_read_float")
        return out

    def readRGBA8(self, out=None):
        """The canvas as (H, W, 4) bytes; `out`: a caller's buffer of that many bytes to fill instead of a new array
        (anything with the array interface, e.g. a pinned torch tensor's .numpy())."""
        if out is None:
            out = np.zeros((self.H, self.W, 4), np.uint8)
        elif out.nbytes != self.H * self.W * 4 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("readRGBA8: out must be a contiguous buffer of H*W*4 bytes")
        self._ck(self.L.rt_read_rgba8(self.ctx, out.ctypes.data), "rt_read_rgba8")
        return out

    def readHitIds(self):
        out = np.full((self.H, self.W), -2, np.int32)
        self._ck(self.L.rt_read_hit_ids(self.ctx, out.ctypes.data), "rt_read_hit_ids")
        return out

    def pick(self, x, y):
        """Primary hit id of pixel (x, y), y = 0 at the bottom: -1 miss, analytic primitive index, or n_prims + face."""
        v = C.c_int32()
        self._ck(self.L.rt_pick(self.ctx, x, y, C.byref(v)), "rt_pick")
        return v.value

    def tonePartial(self):
        s, n = C.c_double(), C.c_uint64()
        self._ck(self.L.rt_tone_partial(self.ctx, C.byref(s), C.byref(n)), "rt_tone_partial")
        return s.value, n.value

    def toneFinish(self, sum_log, count):
        Lbar, sf = C.c_double(), C.c_double()
        self._ck(self.L.rt_tone_finish(self.ctx, sum_log, count, C.byref(Lbar), C.byref(sf)), "rt_tone_finish")
        return Lbar.value, sf.value

    def tone(self):
        Lbar, sf = C.c_double(), C.c_double()
        self._ck(self.L.rt_tone(self.ctx, C.byref(Lbar), C.byref(sf)), "rt_tone")
        return Lbar.value, sf.value

    def tonePartialDevice(self):
        """Queues the reduction; returns the device address of {sum of log luminance, pixel count} (two doubles)."""
        p = C.c_void_p()
        self._ck(self.L.rt_tone_partial_device(self.ctx, C.byref(p)), "rt_tone_partial_device")
        return p.value

    def toneFinishDevice(self, dptr):
        self._ck(self.L.rt_tone_finish_device(self.ctx, C.c_void_p(dptr)), "rt_tone_finish_device")

    def toneGet(self):
        Lbar, sf = C.c_double(), C.c_double()
        self._ck(self.L.rt_tone_get(self.ctx, C.byref(Lbar), C.byref(sf)), "rt_tone_get")
        return Lbar.value, sf.value

    def presentShard(self):
        """The present pass over this shard's own tiles (supersampling 1), queued on the context's stream."""
        self._ck(self.L.rt_present_shard(self.ctx), "rt_present_shard")

    def presentQueued(self):
        self._ck(self.L.rt_present(self.ctx, None), "rt_present")

    def stream(self):
        p = C.c_void_p()
        self._ck(self.L.rt_stream(self.ctx, C.byref(p)), "rt_stream")
        return p.value

    def setOption(self, name, value):
        self._ck(self.L.rt_set_option(self.ctx, name.encode(), float(value)), "rt_set_option")

    def present(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.uint8)
        self._ck(self.L.rt_present(self.ctx, out.ctypes.data), "rt_present")
        return out

    def deviceRGBA8(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.L.rt_device_rgba8(self.ctx, C.byref(p), C.byref(n)), "rt_device_rgba8")
        return p.value, n.value

    def stats(self):
        s = Stats()
        self._ck(self.L.rt_get_stats(self.ctx, C.byref(s)), "rt_get_stats")
        return s
