"""ctypes access to the CPU oracle (oracle/liboracle.so). TEST INFRASTRUCTURE ONLY: imported by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg - never by the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.kd_oracle_build.restype = C.c_void_p
        _lib.kd_oracle_build.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int, C.c_double]
        _lib.kd_oracle_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        _lib.kd_oracle_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.kd_oracle_free.argtypes = [C.c_void_p]
        _lib.oracle_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib.oracle_tone.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.oracle_present.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        _lib.oracle_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_pow.restype = C.c_float
        _lib.oracle_pow.argtypes = [C.c_float, C.c_float]
        _lib.oracle_sizeof_scene.restype = C.c_size_t
        _lib.oracle_sizeof_counters.restype = C.c_size_t
    return _lib


def kd_build(verts_f64, faces_u32, max_depth=24, infinity=100000.0):
    """Reference-layout k-d tree: (pairs[n,2] u32 ascending heap index, leaves[n,4] i32, box[6] f64, depth)."""
    L = lib()
    v = np.ascontiguousarray(verts_f64, dtype=np.float64)
    f = np.ascontiguousarray(faces_u32, dtype=np.uint32)
    h = L.kd_oracle_build(v.ctypes.data, len(v), f.ctypes.data, len(f), max_depth, infinity)
    nn, nl, dep = C.c_int64(), C.c_int64(), C.c_int32()
    L.kd_oracle_counts(h, C.byref(nn), C.byref(nl), C.byref(dep))
    pairs = np.zeros((nn.value, 2), np.uint32)
    leaves = np.zeros((nl.value, 4), np.int32)
    box = np.zeros(6, np.float64)
    L.kd_oracle_export(h, pairs.ctypes.data, leaves.ctypes.data, box.ctypes.data)
    L.kd_oracle_free(h)
    return pairs, leaves, box, dep.value


class Counters(C.Structure):
    A8 = C.c_uint64 * 8
    _fields_ = [("rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("rays_level", A8), ("rays_in_box", A8),
                ("kd_steps", A8), ("kd_leaf_visits", A8), ("kd_tri_tests", A8),
                ("shadow_level", A8), ("shadow_in_box", A8),
                ("sh_steps", A8), ("sh_leaf_visits", A8), ("sh_tri_tests", A8),
                ("prim_tests", A8), ("sh_prim_tests", A8), ("sh_mesh_hits", C.c_uint64), ("sh_hit_after_leaves", A8)]

    def algorithmic(self, level=None):
        """Bytes and flops the REFERENCE algorithm moves/does for the counted rays: per ray
        bytes = 8*steps + 16*leafVisits + 48*triTests + 128*primTests, flops = 14*steps + 45*triTests + 60*primTests
        (closest-hit and shadow rays alike). level=None sums all levels."""
        lv = range(8) if level is None else [level]
        st = sum(self.kd_steps[l] + self.sh_steps[l] for l in lv)
        lf = sum(self.kd_leaf_visits[l] + self.sh_leaf_visits[l] for l in lv)
        tr = sum(self.kd_tri_tests[l] + self.sh_tri_tests[l] for l in lv)
        pr = sum(self.prim_tests[l] + self.sh_prim_tests[l] for l in lv)
        return {"bytes": 8 * st + 16 * lf + 48 * tr + 128 * pr, "flops": 14 * st + 45 * tr + 60 * pr,
                "steps": st, "leaf_visits": lf, "tri_tests": tr, "prim_tests": pr}

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d


def _scene_struct(abi):
    class OracleScene(C.Structure):
        _fields_ = [("cfg", abi.Config), ("n_prims", C.c_int32), ("prims", C.c_void_p),
                    ("n_mats", C.c_int32), ("mats", C.c_void_p),
                    ("n_lights_loop", C.c_int32), ("lights", C.c_void_p),
                    ("ambient", C.c_float * 3), ("bg", C.c_float * 3),
                    ("cam", C.c_float * 3), ("ray00", C.c_float * 3), ("ray01", C.c_float * 3),
                    ("ray10", C.c_float * 3), ("ray11", C.c_float * 3),
                    ("verts", C.c_void_p), ("faces", C.c_void_p), ("face_mat", C.c_void_p),
                    ("n_verts", C.c_int32), ("n_faces", C.c_int32),
                    ("kd_dense", C.c_void_p), ("kd_dense_len", C.c_int64), ("kd_leaves", C.c_void_p),
                    ("kd_box_min", C.c_float * 3), ("kd_box_max", C.c_float * 3)]
    return OracleScene


class OracleRenderer:
    """Holds one flattened scene (see ray_tracer_amd.abi.FlatScene) and renders it on the CPU."""

    def __init__(self, flat, kd=None):
        from ray_tracer_amd import abi
        self.abi = abi
        self.flat = flat
        self._keep = []
        S = _scene_struct(abi)
        assert C.sizeof(S) == lib().oracle_sizeof_scene(), (C.sizeof(S), lib().oracle_sizeof_scene())
        assert C.sizeof(Counters) == lib().oracle_sizeof_counters()
        s = S()
        s.cfg = flat.cfg
        prims = flat.prim_array()
        mats = flat.mat_array()
        lights = flat.light_array()
        self._keep += [prims, mats, lights]
        s.n_prims, s.prims = len(prims), C.addressof(prims) if len(prims) else None
        s.n_mats, s.mats = len(mats), C.addressof(mats) if len(mats) else None
        s.n_lights_loop, s.lights = flat.n_lights_loop, C.addressof(lights) if len(lights) else None
        for name, src in (("ambient", flat.ambient), ("bg", flat.bg), ("cam", flat.cam), ("ray00", flat.ray00),
                          ("ray01", flat.ray01), ("ray10", flat.ray10), ("ray11", flat.ray11)):
            setattr(s, name, (C.c_float * 3)(*[float(x) for x in src]))
        self.v32 = np.ascontiguousarray(flat.verts, dtype=np.float64).astype(np.float32)
        self.faces = np.ascontiguousarray(flat.faces, dtype=np.uint32)
        self.face_mat = np.ascontiguousarray(flat.face_mat, dtype=np.uint32)
        s.verts, s.faces, s.face_mat = self.v32.ctypes.data, self.faces.ctypes.data, self.face_mat.ctypes.data
        s.n_verts, s.n_faces = len(self.v32), len(self.faces)
        self.kd = None
        if flat.cfg.render_triangles and flat.cfg.use_kd_tree and len(self.faces):
            if kd is None:
                kd = kd_build(flat.verts, self.faces, flat.cfg.max_kd_depth, float(np.float32(flat.cfg.infinity)))
            pairs, leaves, box, _ = kd
            self.kd = kd
            n = int(pairs[:, 0].max()) + 1 if len(pairs) else 1
            self.dense = np.zeros(n, np.uint32)
            self.dense[pairs[:, 0]] = pairs[:, 1]
            self.leaves = np.ascontiguousarray(leaves, dtype=np.int32)
            s.kd_dense, s.kd_dense_len, s.kd_leaves = self.dense.ctypes.data, n, self.leaves.ctypes.data
            b32 = box.astype(np.float32)      # uniform3fv narrowing, js/8-kd_tree.js:1114
            s.kd_box_min = (C.c_float * 3)(*b32[:3])
            s.kd_box_max = (C.c_float * 3)(*b32[3:])
        self.s = s
        self.W = flat.cfg.width * flat.cfg.supersampling
        self.H = flat.cfg.height * flat.cfg.supersampling

    def trace(self, rect=None, step=1, threads=None, want_float=True, want_ids=True):
        x0, y0, x1, y1 = rect or (0, 0, self.W, self.H)
        rgb = np.zeros((self.H, self.W, 3), np.float32) if want_float else None
        rgba = np.zeros((self.H, self.W, 4), np.uint8)
        ids = np.full((self.H, self.W), -2, np.int32) if want_ids else None
        cn = Counters()
        threads = threads or min(os.cpu_count() or 1, 16)
        lib().oracle_trace(C.addressof(self.s), x0, y0, x1, y1, step,
                           rgb.ctypes.data if rgb is not None else None, rgba.ctypes.data,
                           ids.ctypes.data if ids is not None else None, C.addressof(cn), threads)
        return rgb, rgba, ids, cn

    def probe(self, ro, rd):
        """One ray-tree node for an arbitrary ray (known-answer tests): dict of the closest hit, the local colour and the
        two child rays exactly as the pixel loop forms them."""
        o = np.ascontiguousarray(ro, np.float32)
        d = np.ascontiguousarray(rd, np.float32)
        out = np.zeros(31, np.float32)
        lib().oracle_probe(C.addressof(self.s), o.ctypes.data, d.ctypes.data, out.ctypes.data)
        return {"hit": bool(out[0]), "id": int(out[1]) if out[0] else -1, "t": float(out[2]), "point": out[3:6].astype(np.float64),
                "normal": out[6:9].astype(np.float64), "local": out[9:12].astype(np.float64), "colour": out[12:15].astype(np.float64),
                "kReflect": float(out[15]), "kRefract": float(out[16]), "ior": float(out[17]),
                "refr_o": out[18:21].astype(np.float64), "refr_d": out[21:24].astype(np.float64),
                "refl_o": out[24:27].astype(np.float64), "refl_d": out[27:30].astype(np.float64), "lights_blocked": int(out[30])}

    def tone(self, rgba):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        Lbar, sf = C.c_double(), C.c_double()
        lib().oracle_tone(C.addressof(self.s), a.ctypes.data, a.size // 4, C.byref(Lbar), C.byref(sf))
        return Lbar.value, sf.value

    def present(self, rgba, Lbar, sf):
        a = np.ascontiguousarray(rgba, dtype=np.uint8)
        out = np.zeros((self.flat.cfg.height, self.flat.cfg.width, 4), np.uint8)
        lib().oracle_present(C.addressof(self.s), a.ctypes.data, float(np.float32(Lbar)), float(np.float32(sf)),
                             out.ctypes.data)
        return out
