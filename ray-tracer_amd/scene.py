"""Python twin of the reference's scene-description API (js/8-kd_tree.js:1174-1537 primitives,
:1977-2117 Triangle/Mesh, :2123-2291 materials and lights, :2851-2921 loaders, :2927-3024
generateScene), reduced to what feeds the device: it produces exactly the numbers the reference
hands to WebGL as uniforms and data textures. Driver-side helper for tests and bench; the product
host is the JavaScript twin under js/.
"""
import math
import os
import re

import numpy as np

from .three_f32 import Euler, Matrix4, PerspectiveCamera, Vector3, get_primary_ray

PLANE, RECT, SPHERE, CUBE, CYLINDER = 0, 1, 2, 3, 4
PHONG, PHONG_BLINN, PHONG_CHECKERED = 0, 1, 2

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


class Ids:
    """The reference's global id counters (js/8-kd_tree.js:56-58)."""

    def __init__(self):
        self.object = self.phong = self.blinn = self.checkered = self.light = 0


class Primitive:
    kind = None

    def __init__(self, id):
        if type(self) is Primitive:
            raise TypeError("Primitive: cannot instantiate abstract class.")
        self.id = id
        self.modelTransform = Matrix4()
        self.invModelTransform = Matrix4()
        self.material = None

    def setMaterial(self, m):
        self.material = m

    def translate(self, t):
        self.modelTransform = Matrix4().makeTranslation(t.x, t.y, t.z).multiply(self.modelTransform)
        self.invModelTransform = Matrix4().getInverse(self.modelTransform, True)

    def rotate(self, eu):
        self.modelTransform = Matrix4().makeRotationFromEuler(eu).multiply(self.modelTransform)
        self.invModelTransform = Matrix4().getInverse(self.modelTransform, True)

    def scale(self, s):
        self.modelTransform = Matrix4().makeScale(s.x, s.y, s.z).multiply(self.modelTransform)
        self.invModelTransform = Matrix4().getInverse(self.modelTransform, True)


class Plane(Primitive):
    kind = PLANE

    def scale(self, s):
        raise TypeError("Plane: cannot call abstract method 'scale'")


class Rectangle(Primitive):
    kind = RECT


class Sphere(Primitive):
    kind = SPHERE


class Cube(Primitive):
    kind = CUBE


class Cylinder(Primitive):
    kind = CYLINDER


class Material:
    def __init__(self, type, id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Co2, Cs):
        self.type, self.id = type, id
        self.ka, self.kd, self.ks, self.ke = ka, kd, ks, ke
        self.kReflect, self.kRefract, self.ior = kReflect, kRefract, ior
        self.Co, self.Co2, self.Cs = Co, Co2, Cs


def Phong(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs):
    return Material(PHONG, id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Vector3(), Cs)


def PhongBlinn(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Cs):
    return Material(PHONG_BLINN, id, ka, kd, ks, ke, kReflect, kRefract, ior, Co, Vector3(), Cs)


def PhongCheckered(id, ka, kd, ks, ke, kReflect, kRefract, ior, Co1, Co2, Cs):
    return Material(PHONG_CHECKERED, id, ka, kd, ks, ke, kReflect, kRefract, ior, Co1, Co2, Cs)


class PointLight:
    def __init__(self, id, position, color):
        self.id, self.position, self.color = id, position, color


def _apply_matrix4(verts, m):
    """Vector3.applyMatrix4 over an (n,3) float64 array, same operation order (r84)."""
    a = m.e
    b, c, d = verts[:, 0].copy(), verts[:, 1].copy(), verts[:, 2].copy()
    x = a[0] * b + a[4] * c + a[8] * d + a[12]
    y = a[1] * b + a[5] * c + a[9] * d + a[13]
    z = a[2] * b + a[6] * c + a[10] * d + a[14]
    w = a[3] * b + a[7] * c + a[11] * d + a[15]
    s = 1.0 / w
    ok = np.isfinite(s)
    out = np.stack([x * s, y * s, z * s], axis=1)
    out[~ok] = 0.0
    return out


class Mesh(Primitive):
    """js/8-kd_tree.js:1996-2117. Vertices are float64 world-space positions that every transform
    call round-trips through the (f32) inverse of the old model matrix."""

    kind = None

    def __init__(self, id, verts, faces, world):
        super().__init__(id)
        self.verts = np.array(verts, dtype=np.float64).reshape(-1, 3)
        self.faces = np.array(faces, dtype=np.uint32).reshape(-1, 3)
        self.face_material = None
        self.world = world
        # pushToGlobal: js/8-kd_tree.js:2108-2116 appends faces WITHOUT rebasing their vertex indices
        self.startVert = sum(len(m.verts) for m in world)
        world.append(self)

    def setMaterial(self, m):
        self.face_material = m

    def _retransform(self, new_model):
        self.verts = _apply_matrix4(self.verts, self.invModelTransform)
        self.modelTransform = new_model
        self.invModelTransform = Matrix4().getInverse(self.modelTransform, True)
        self.verts = _apply_matrix4(self.verts, self.modelTransform)

    def translate(self, t):
        self._retransform(Matrix4().makeTranslation(t.x, t.y, t.z).multiply(self.modelTransform))

    def rotate(self, eu):
        self._retransform(Matrix4().makeRotationFromEuler(eu).multiply(self.modelTransform))

    def scale(self, s):
        self._retransform(Matrix4().makeScale(s.x, s.y, s.z).multiply(self.modelTransform))


# ---- loaders -------------------------------------------------------------------------------

_PLY_TYPES = {"int8": "b", "char": "b", "uint8": "B", "uchar": "B", "int16": "h", "short": "h", "uint16": "H", "ushort": "H",
              "int32": "i", "int": "i", "uint32": "I", "uint": "I", "float32": "f", "float": "f", "float64": "d", "double": "d"}


def load_ply(path):
    """PLY (ASCII, binary_little_endian or binary_big_endian) -> (verts f64 holding f32 values, faces). Mirrors
    THREE.PLYLoader (js/loaders/PLYLoader.js:252-388 parseASCII, :392-486 parseBinary: positions pass through a
    Float32BufferAttribute; quads split 0-1-3 / 1-2-3; the face list is `vertex_indices` or `vertex_index`) followed by
    Geometry.fromBufferGeometry (vertex and face order kept)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    m = re.search(rb"ply([\s\S]*?)end_header\s", data)
    if not m:
        raise ValueError("not a PLY file: no header")
    fmt = "ascii"
    elements = []
    for line in m.group(1).decode("latin-1").split("\n"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            elements[-1][2].append(tok[1:])
    verts, faces = [], []

    def add_face(idx):
        if len(idx) == 3:
            faces.append((idx[0], idx[1], idx[2]))
        elif len(idx) == 4:
            faces.append((idx[0], idx[1], idx[3]))
            faces.append((idx[1], idx[2], idx[3]))

    if fmt == "ascii":
        lines = [ln for ln in (l.strip() for l in data[m.end():].decode("latin-1").split("\n")) if ln]
        pos = 0
        for name, count, props in elements:
            rows = lines[pos:pos + count]
            pos += count
            if name == "vertex":
                names = [p[-1] for p in props]
                ix, iy, iz = names.index("x"), names.index("y"), names.index("z")
                for r in rows:
                    t = r.split()
                    verts.append((float(t[ix]), float(t[iy]), float(t[iz])))
            elif name == "face":
                for r in rows:
                    t = [int(v) for v in r.split()]
                    add_face(t[1:1 + t[0]])
    elif fmt in ("binary_little_endian", "binary_big_endian"):
        e = "<" if fmt == "binary_little_endian" else ">"
        at = m.end()
        for name, count, props in elements:
            for _ in range(count):
                rec = {}
                for p in props:
                    if p[0] == "list":
                        n, = struct.unpack_from(e + _PLY_TYPES[p[1]], data, at)
                        at += struct.calcsize(_PLY_TYPES[p[1]])
                        item = e + str(n) + _PLY_TYPES[p[2]]
                        rec[p[3]] = list(struct.unpack_from(item, data, at))
                        at += struct.calcsize(item)
                    else:
                        rec[p[1]], = struct.unpack_from(e + _PLY_TYPES[p[0]], data, at)
                        at += struct.calcsize(_PLY_TYPES[p[0]])
                if name == "vertex":
                    verts.append((float(rec["x"]), float(rec["y"]), float(rec["z"])))
                elif name == "face":
                    add_face(rec.get("vertex_indices") or rec.get("vertex_index"))
    else:
        raise ValueError("PLY format %r" % fmt)
    v = np.array(verts, dtype=np.float64).reshape(-1, 3).astype(np.float32).astype(np.float64)
    return v, np.array(faces, dtype=np.uint32).reshape(-1, 3)


def load_obj(path):
    """OBJ -> de-indexed f32 positions, faces (3k, 3k+1, 3k+2): THREE.OBJLoader builds a non-indexed
    BufferGeometry (js/loaders/OBJLoader.js:653-655) and Geometry.fromBufferGeometry numbers its
    faces consecutively. Polygons are fanned (v0, v_i, v_i+1) like the loader's addFace calls."""
    pts, out = [], []
    with open(path, "r") as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                pts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                idx = []
                for tok in t[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(pts) + i)
                if len(idx) == 3:
                    out.extend([pts[idx[0]], pts[idx[1]], pts[idx[2]]])
                elif len(idx) == 4:
                    out.extend([pts[idx[0]], pts[idx[1]], pts[idx[3]]])
                    out.extend([pts[idx[1]], pts[idx[2]], pts[idx[3]]])
                else:
                    raise ValueError("OBJ faces with more than 4 vertices are not handled by the r84 loader")
    v = np.array(out, dtype=np.float64).astype(np.float32).astype(np.float64)
    faces = np.arange(len(out), dtype=np.uint32).reshape(-1, 3)
    return v, faces


# ---- scene -> flat arrays --------------------------------------------------------------------

class Scene:
    """What generateScene returns (js/8-kd_tree.js:3017-3023) plus the global mesh arrays."""

    def __init__(self):
        self.ids = Ids()
        self.objects, self.materials, self.lights = [], [], []
        self.meshes = []
        self.ambientLight = Vector3(0.75, 0.75, 0.75)
        self.bgColor = Vector3(0.1, 0.5, 0.9)
        self.n_lights_declared = 0

    # -- what RayTracer.update marshals (js/8-kd_tree.js:2435-2474) --
    def prim_table(self):
        return [o for o in self.objects if not isinstance(o, Mesh)]

    def mesh_arrays(self, rebase=True):
        """g_verts (f64), g_faces, face materials. rebase=True offsets every mesh's indices by its
        first vertex (needed as soon as there are two meshes; the reference never rebases,
        js/8-kd_tree.js:2114-2115, and only ever loads one mesh)."""
        if not self.meshes:
            return (np.zeros((0, 3)), np.zeros((0, 3), np.uint32), np.zeros((0, 2), np.uint32))
        vs, fs, ms = [], [], []
        base = 0
        for m in self.meshes:
            vs.append(m.verts)
            fs.append(m.faces + np.uint32(base if rebase else 0))
            mat = m.face_material
            ms.append(np.tile(np.array([[mat.type, mat.id]], dtype=np.uint32), (len(m.faces), 1)))
            base += len(m.verts)
        return np.concatenate(vs), np.concatenate(fs), np.concatenate(ms)


def animate_light(scene, time):
    """UI.animate, js/8-kd_tree.js:2711-2716 (both coordinates use cos)."""
    light = scene.lights[0]
    light.position.x = 500 * math.cos((time % 60) / 60 * math.pi * 2)
    light.position.z = 500 * math.cos((time % 60) / 60 * math.pi * 2) - 280


def default_scene(model_path=None, mesh_material=None, second_light=True, mesh_color=(0.1, 0.75, 0.1),
                  extra_mesh=None):
    """generateScene of js/8-kd_tree.js:2927-3024 (and of 7-tone_reproduction.js / project.js, which
    are the same text). model_path=None leaves the mesh out (RENDER_TRIANGLES=false still constructs
    it in the reference, but nothing of it reaches the shader).
    second_light=False gives checkpoint 6's variant (js/6-transmission.js:1933-1947): the second
    light is constructed (so the shader loops over two) but never pushed, so its uniforms stay zero.
    mesh_material = (ka,kd,ks,ke,kReflect,kRefract,ior) overrides the mesh's Phong constants.
    extra_mesh = (path, scale, translate) appends a second mesh (OBJ) with rebased indices."""
    sc = Scene()
    ids = sc.ids
    V = Vector3

    def nid(name):
        v = getattr(ids, name)
        setattr(ids, name, v + 1)
        return v

    l1 = PointLight(nid("light"), V(10, 300, -10), V(1, 1, 1))
    l2 = PointLight(nid("light"), V(-300, 50, 200), V(0.1, 1, 0.1))
    sc.lights.append(l1)
    if second_light:
        sc.lights.append(l2)
    sc.n_lights_declared = ids.light

    sphere1M = Phong(nid("phong"), 0.075, 0.075, 0.2, 20, 0.01, 0.8, 0.95, V(1, 1, 1), V(1, 1, 1))
    sphere2M = Phong(nid("phong"), 0.15, 0.25, 1.0, 20, 0.75, 0, 0, V(0.7, 0.7, 0.7), V(1, 1, 1))
    cubeM = Phong(nid("phong"), 0.5, 0.9, 0.2, 24, 0, 0, 0, V(0.8, 0.1, 0.7), V(0.7, 0.7, 0.7))
    cylM = Phong(nid("phong"), 0.5, 0.3, 0.9, 20, 0, 0, 0, V(0.85, 0.7, 0.1), V(0.7, 0.7, 0.7))
    floorM = PhongCheckered(nid("checkered"), 0.5, 0.9, 0.2, 12, 0.4, 0, 0, V(1, 0, 0), V(1, 1, 0), V(0.7, 0.7, 0.7))
    mm = mesh_material or (0.7, 0.6, 0.7, 16, 0, 0, 0)
    meshM = Phong(nid("phong"), *mm, V(*mesh_color), V(0.7, 0.7, 0.7))
    sc.materials += [sphere1M, sphere2M, cubeM, cylM, floorM, meshM]

    sphere1 = Sphere(nid("object"))
    sphere1.scale(V(50, 50, 50))
    sphere1.translate(V(2, 15, -240))
    sphere1.setMaterial(sphere1M)
    sphere2 = Sphere(nid("object"))
    sphere2.scale(V(40, 40, 40))
    sphere2.translate(V(-75, -20, -325))
    sphere2.setMaterial(sphere2M)
    floor = Rectangle(nid("object"))
    floor.scale(V(450, 1, 1600))
    floor.translate(V(-105, -80, -300))
    floor.setMaterial(floorM)
    cube = Cube(nid("object"))
    cube.scale(V(90, 60, 60))
    cube.rotate(Euler(-math.pi / 4, math.pi / 4, 0, "XYZ"))
    cube.translate(V(-70, 40, -400))
    cube.setMaterial(cubeM)
    cyl = Cylinder(nid("object"))
    cyl.scale(V(30, 60, 30))
    cyl.rotate(Euler(-math.pi / 3, 0, -math.pi / 6, "XYZ"))
    cyl.translate(V(50, 75, -300))
    cyl.setMaterial(cylM)
    sc.objects += [sphere1, sphere2, floor, cube, cyl]

    if model_path is not None:
        v, f = load_ply(model_path)
        mesh = Mesh(nid("object"), v, f, sc.meshes)
        mesh.scale(V(1000, 1000, 1000))
        mesh.translate(V(15, 27, -245))
        mesh.setMaterial(meshM)
        sc.objects.append(mesh)
    if extra_mesh is not None:
        path, s, t = extra_mesh
        v, f = load_obj(path) if path.endswith(".obj") else load_ply(path)
        m2 = Mesh(nid("object"), v, f, sc.meshes)
        m2.scale(V(s, s, s))
        m2.translate(V(*t))
        m2.setMaterial(meshM)
        sc.objects.append(m2)
    return sc


def camera_rays(width, height, fov=45.0, near=0.1, far=10000.0):
    """cameraPos + ray00..ray11 as RayTracer.update computes them (js/8-kd_tree.js:2448-2452) for the
    reference's resting camera pose, narrowed to f32 like setUniforms does (js/8-kd_tree.js:1114)."""
    cam = PerspectiveCamera(fov, width / height, near, far)
    out = {"cameraPos": np.zeros(3, np.float32)}
    for name, (x, y) in {"ray00": (-1, -1), "ray01": (-1, 1), "ray10": (1, -1), "ray11": (1, 1)}.items():
        out[name] = np.array(get_primary_ray(cam, x, y).toArray(), dtype=np.float64).astype(np.float32)
    return out
