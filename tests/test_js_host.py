"""The JavaScript host (ray-tracer_amd/js): the reference's scene API + class RayTracer over the N-API addon.

CPU: the checker script runs the scene through a host-only context and compares every uniform, the mesh arrays
and the k-d export with the fixtures captured from the reference - first with the example scene description,
then (when the reference checkout is present) with the reference's own, unmodified generateScene() text.
GPU: one frame rendered from JavaScript must equal, byte for byte, the frame rendered through the Python binding.
"""
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, make_flat

JS = os.path.join(ROOT, "ray-tracer_amd", "js")
ADDON = os.path.join(JS, "addon", "rt_addon.node")
NODE = shutil.which("node")
REF_SCRIPT = "/root/reference/js/8-kd_tree.js"

needs_node = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is missing")


def run_node(args, timeout=300):
    p = subprocess.run([NODE] + args, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, "node %s failed:\n%s\n%s" % (" ".join(args), p.stdout[-2000:], p.stderr[-2000:])
    return json.loads(p.stdout.strip().splitlines()[-1])


@needs_node
@pytest.mark.parametrize("golden", ["k8_res4_640x480", "k8_res3_640x480", "c2_res4_1920x1080", "k8_full_640x480",
                                    "k8_res4_teapot_640x480", "c3_bunny_teapot", "k8_res4_drag_640x480"])
def test_js_scene_matches_reference_capture(golden):
    """The two *teapot fixtures add a second mesh through loadOBJ with REBASE_MESH_INDICES (BASELINE config 3 as benchmarked).
    The drag fixture was made by driving the reference's UI.mouseDown / mouseMove / mouseUp: class UI (js/ui.js) must
    grab the mesh at the same depth, follow the mouse through the same positions and leave the same transforms,
    vertices and tree."""
    out = run_node([os.path.join(JS, "test", "check_golden.js"), os.path.join(GOLDEN, golden)])
    assert out["failures"] == [] and out["uniforms_checked"] >= 80


@needs_node
@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="reference checkout not present")
@pytest.mark.parametrize("golden", ["k8_res4_640x480", "c2_res4_1920x1080", "k8_res4_teapot_640x480", "k8_res4_drag_640x480"])
def test_unchanged_reference_generate_scene_runs(golden):
    """generateScene() is read from the reference's page script at test time and evaluated as it is."""
    out = run_node([os.path.join(JS, "test", "check_golden.js"), os.path.join(GOLDEN, golden), REF_SCRIPT])
    assert out["failures"] == [] and out["scene"] == "page script"


@needs_node
def test_js_abstract_classes_and_errors():
    code = r"""
const rt = require(%r);
const G = rt.createContext({DEVICE: -1});
const msgs = [];
for (const f of [() => new G.Primitive(0), () => new G.Material(0, 0), () => new G.Light(0), () => new G.Plane(0).scale(new G.THREE.Vector3(1,1,1)),
                 () => new G.Sphere(1).intersect(null), () => new G.Phong(0,1,1,1,1,0,0,0,null,null).constructor.prototype.__proto__.setUniforms.call({}, {})]) {
  try { f(); msgs.push(null); } catch (e) { msgs.push(e.message); }
}
let traceErr = null;
try { const t = new G.RayTracer(); t.setScene({objects: [], materials: [], lights: [], ambientLight: new G.THREE.Vector3(), bgColor: new G.THREE.Vector3()}); t.update(0); }
catch (e) { traceErr = e.message; }
console.log(JSON.stringify({msgs, traceErr}));
""" % JS
    out = run_node(["-e", code])
    assert out["msgs"] == ["Primitive: cannot instantiate abstract class.", "Material: cannot instantiate abstract class.",
                           "Light: cannot instantiate abstract class.", "Plane: cannot call abstract method 'scale'",
                           "Primitive: cannot call abstract method 'intersect'", "Material: cannot call abstract method 'setUniforms'"]
    assert out["traceErr"] and "host-only" in out["traceErr"]   # no CPU fallback behind the JavaScript class either


@needs_node
@pytest.mark.gpu
def test_js_render_equals_python_binding():
    from ray_tracer_amd import abi
    mm = (0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9)
    args = [os.path.join(JS, "examples", "render.js"), "--width", "640", "--height", "480", "--depth", "5",
            "--mesh-material", ",".join(str(v) for v in mm)]
    out = run_node(args)
    # the same frame with the tree built on the device and update() split into its two halves
    out_dev = run_node(args + ["--kd-builder", "device", "--split", "1", "--frames", "2", "--time", str(-1.0 / 60)])
    flat = make_flat("bun_zipper_res4.ply", 640, 480, depth=5, mesh_material=mm)
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    rt.trace()
    rgba = rt.readRGBA8()
    Lbar, sf = rt.tone()
    canvas = rt.present()
    assert out["rgba8_sha256"] == hashlib.sha256(np.ascontiguousarray(rgba).tobytes()).hexdigest()
    assert out["canvas_sha256"] == hashlib.sha256(np.ascontiguousarray(canvas).tobytes()).hexdigest()
    assert out["Lbar"] == Lbar and out["sf"] == sf
    assert out["rays"] == rt.stats().rays_total
    for k in ("rgba8_sha256", "canvas_sha256", "Lbar", "sf", "rays"):
        assert out_dev[k] == out[k], k
    # picking (the reference intersects the pixel's primary ray on the CPU): centre of the canvas = the glass sphere,
    # a corner of the sky = nothing, a pixel of the bunny = the mesh; and rt_pick agrees with the hit-id buffer
    for xy, want in (("320,240", "Sphere"), ("2,2", None), ("330,40", "Mesh")):
        assert run_node(args + ["--pick", xy])["picked"] == want, xy
    ids = rt.readHitIds()
    for x, y in ((320, 239), (2, 477), (330, 439), (100, 50)):
        assert rt.pick(x, y) == ids[y, x]
    rt.close()


@needs_node
@pytest.mark.gpu
def test_js_animation_same_frames_with_either_builder():
    """examples/animate.js: the mesh dragged with Mesh.translate, the tree rebuilt every frame, two RayTracer objects
    alternating frames (updateBegin / updateEnd). The last canvas must not depend on where the tree was built."""
    args = [os.path.join(JS, "examples", "animate.js"), "--width", "320", "--height", "240", "--depth", "3",
            "--model", "models/bun_zipper_res4.ply"]
    dev = run_node(args + ["--kd-builder", "device", "--frames", "6"])
    host = run_node(args + ["--kd-builder", "host", "--frames", "6"])
    assert dev["last_canvas_sha256"] == host["last_canvas_sha256"]
    one = run_node(args + ["--kd-builder", "device", "--tracers", "1", "--frames", "7"])   # same last frame index
    assert one["last_canvas_sha256"] == dev["last_canvas_sha256"]


def _dragged_flat(m, width=640, height=480, depth=3):
    """The scene of the drag fixture after the drag, built with the Python scene twin: the mesh translated by the
    differences of the recorded positions (UI.mouseMove / mouseUp translate by newPosition - oldPosition,
    js/8-kd_tree.js:2771, :2785), the light at the following tick's time."""
    from ray_tracer_amd import abi, scene
    from ray_tracer_amd.three_f32 import Vector3
    from helpers import model
    mm = tuple(float(v) for v in m["constant_edits"]["meshMaterial"].split(","))
    sc = scene.default_scene(model("bun_zipper_res4.ply"), mesh_material=mm)
    mesh = sc.objects[m["drag"]["selected_index"]]
    old = m["drag"]["old_position"]
    for pos in m["drag"]["positions"]:
        mesh.translate(Vector3(pos[0] - old[0], pos[1] - old[1], pos[2] - old[2]))
        old = pos
    scene.animate_light(sc, m["time"] + 0.5)
    return abi.FlatScene(sc, abi.default_config(width, height, depth=depth), scene.camera_rays(width, height)), mesh


def test_python_scene_twin_reproduces_the_dragged_mesh():
    from helpers import manifest, sha
    m = manifest("k8_res4_drag_640x480")
    flat, mesh = _dragged_flat(m)
    assert [float(np.float32(v)) for v in mesh.modelTransform.toArray()] == m["drag"]["model_after"]
    assert sha(flat.verts) == m["sha256"]["verts_f64"] and sha(flat.faces) == m["sha256"]["faces_u32"]


@needs_node
@pytest.mark.gpu
def test_js_drag_the_mesh_and_render_equals_oracle():
    """The bunny clicked (picked from the hit-id buffer), grabbed and dragged with class UI from JavaScript, then one
    more frame: the mesh must have moved exactly as the reference's UI moved it (fixture), and the frame after the drag
    must equal the oracle's render of the moved scene byte for byte."""
    from helpers import manifest
    from oracle import oracle
    from ray_tracer_amd import abi
    m = manifest("k8_res4_drag_640x480")
    d = m["drag"]
    out = run_node([os.path.join(JS, "examples", "render.js"), "--width", "640", "--height", "480", "--depth", "3",
                    "--mesh-material", m["constant_edits"]["meshMaterial"], "--time", str(m["time"]),
                    "--drag", ",".join(str(v) for v in d["mouse"])])
    assert out["drag_selected"] == "Mesh" and out["drag_grabbed"] is True
    assert out["drag_model"] == d["model_after"]
    flat, _ = _dragged_flat(m)
    want_f, want_8, want_id, _ = oracle.OracleRenderer(flat).trace(threads=16)
    assert out["drag_rgba8_sha256"] == hashlib.sha256(np.ascontiguousarray(want_8).tobytes()).hexdigest()
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat, kd_on_device=True)
    rt.trace()
    assert np.array_equal(rt.readRGBA8(), want_8) and np.array_equal(rt.readHitIds(), want_id)
    assert out["drag_rgba8_sha256"] != out["rgba8_sha256"]      # the frame did change
    rt.close()
