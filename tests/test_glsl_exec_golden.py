"""The CPU oracle against arrays made by EXECUTING the reference's own generated shader text (tests/golden/glsl_exec/*,
written by oracle/gen_golden/glsl_exec/make_fixtures.py in the build container: the text of the tracer fragment shader,
js/8-kd_tree.js:327-1057, and of the present shader, :83-131, as the page scripts generate it for each configuration, is
compiled as C++ over a header that gives GLSL's types and built-ins their meaning).

What this is worth: the header is a stand-in for a GLSL implementation and makes the same implementation-defined choices
the oracle does (float32 round-to-nearest without contraction, the same pow, NEAREST texel fetches), so it cannot pin
those choices; it does rule out that the oracle restates the shader's CONTROL FLOW and ARITHMETIC ORDER wrongly: every
branch, loop bound, epsilon and operand order the arrays depend on came from the reference's text, not from a reading of
it. Only the arrays are stored; the text never enters this repository."""
import json
import os

import numpy as np
import pytest

from helpers import make_flat, model
from oracle import oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.join(REPO, "tests", "golden", "glsl_exec")
NAMES = sorted(d for d in os.listdir(ROOT) if os.path.exists(os.path.join(ROOT, d, "meta.json")))


def _scene(meta):
    if "random_scene" in meta:
        # a random scene (oracle/gen_golden/glsl_exec/random_scenes.py): the description realised on the Python twin of the scene
        # classes, with the camera uniforms the reference computed for it
        import importlib.util
        spec = importlib.util.spec_from_file_location("glsl_exec_random_scenes", os.path.join(REPO, "oracle", "gen_golden", "glsl_exec", "random_scenes.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        from ray_tracer_amd import abi
        d = meta["random_scene"]
        cam = {k: np.array(v, np.float32) for k, v in meta["camera"].items()}
        if meta.get("checkpoint6"):   # through js/6-transmission.js: no mesh, INFINITY 100000.1, light 0 where the scene puts it
            return abi.FlatScene(mod.py_scene(d, with_mesh=False, animate=False),
                                 abi.default_config(meta["traced_width"], meta["traced_height"], depth=d["depth"], infinity=100000.1), cam)
        return abi.FlatScene(mod.py_scene(d), abi.default_config(meta["traced_width"], meta["traced_height"], depth=d["depth"]), cam)
    if "drag" in meta:
        return _dragged(meta)[0]
    kw = dict(meta["make_flat"])
    for k in ("mesh_material",):
        if kw.get(k) is not None:
            kw[k] = tuple(kw[k])
    if kw.get("extra_mesh") is not None:
        n, s, t = kw["extra_mesh"]
        kw["extra_mesh"] = (model(n), s, tuple(t))
    flat = make_flat(**kw)
    if "camera" in meta:   # a camera the reference's controls moved: its uniforms as the reference handed them to WebGL
        from ray_tracer_amd import abi, scene
        cam = {k: np.array(v, np.float32) for k, v in meta["camera"].items()}
        assert not np.array_equal(cam["cameraPos"], np.zeros(3, np.float32))
        sc = scene.default_scene(model(kw["model_name"]), mesh_material=kw.get("mesh_material"))
        scene.animate_light(sc, kw["time"])
        flat = abi.FlatScene(sc, flat.cfg, cam)
    return flat


def _dragged(meta):
    """(the scene after the reference's UI dragged the mesh, the tree of the frame before the drag): the mesh translated by the
    differences of the recorded positions (UI.mouseMove / mouseUp, js/8-kd_tree.js:2771, :2785) on the Python twin."""
    from ray_tracer_amd import abi, scene
    from ray_tracer_amd.three_f32 import Vector3
    kw = meta["make_flat"]
    sc = scene.default_scene(model(kw["model_name"]), mesh_material=tuple(kw["mesh_material"]))
    cfg = abi.default_config(kw["width"], kw["height"], depth=kw["depth"])
    before = abi.FlatScene(sc, cfg, scene.camera_rays(kw["width"], kw["height"]))
    kd_old = oracle.kd_build(before.verts, np.asarray(before.faces, np.uint32))
    verts_old = np.array(before.verts, np.float64)
    mesh = sc.objects[meta["drag"]["selected_index"]]
    old = meta["drag"]["old_position"]
    for pos in meta["drag"]["positions"]:
        mesh.translate(Vector3(pos[0] - old[0], pos[1] - old[1], pos[2] - old[2]))
        old = pos
    scene.animate_light(sc, kw["time"])
    flat = abi.FlatScene(sc, abi.default_config(kw["width"], kw["height"], depth=kw["depth"]), scene.camera_rays(kw["width"], kw["height"]))
    return flat, kd_old, before, verts_old


def test_fixture_set_is_complete():
    assert len(NAMES) >= 19
    kinds = [json.load(open(os.path.join(ROOT, n, "meta.json"))) for n in NAMES]
    assert sum("random_scene" in m for m in kinds) >= 4                             # scenes nobody tuned, through the reference's own classes
    assert sum("drag" in m for m in kinds) >= 1                                      # moved vertices under a stale tree
    kinds = [m for m in kinds if "make_flat" in m]
    assert sum("canvas" in m for m in kinds) >= 3                                  # Ward, Reinhard and the odd-N present loop
    assert {m["make_flat"].get("ss", 1) for m in kinds} >= {1, 2, 3}
    assert max(m["make_flat"]["depth"] for m in kinds) >= 5


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_executed_shader_text(name):
    d = os.path.join(ROOT, name)
    meta = json.load(open(os.path.join(d, "meta.json")))
    step = meta["step"]
    if "drag" in meta:
        flat, kd_old, before, verts_old = _dragged(meta)
        assert np.abs(np.array(flat.verts, np.float64) - verts_old).max() > 5.0        # the mesh did move, by more than a leaf cell
        o = oracle.OracleRenderer(flat, kd=kd_old)                                       # moved vertices through the OLD tree and root box
    else:
        o = oracle.OracleRenderer(_scene(meta))
    assert (o.W, o.H) == (meta["traced_width"], meta["traced_height"])
    rgb, rgba, ids, cn = o.trace(step=step)
    want = np.fromfile(os.path.join(d, "tracer.f32"), np.float32).reshape(meta["samples"] + [3])
    got = rgb[::step, ::step]
    assert got.shape == want.shape
    # bit-exact: the colours as float32 BITS (so a NaN or a signed zero would count too)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "%d of %d samples differ" % (int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum()), want.shape[0] * want.shape[1])
    if "drag" in meta:   # and the stale tree shows: the same vertices through a tree of their own are another image
        fresh = oracle.OracleRenderer(flat).trace(step=step)[0][::step, ::step]
        assert (fresh.view(np.uint32) != want.view(np.uint32)).any(-1).sum() > 1000
    assert len(np.unique(ids[::step, ::step])) >= 4                                # the lattice sees several objects, floor and sky
    if "canvas" in meta:
        canvas = np.fromfile(os.path.join(d, "canvas.rgba8"), np.uint8).reshape(meta["canvas"] + [4])
        assert np.array_equal(o.present(rgba, meta["Lbar_f32"], meta["sf_f32"]), canvas)
        # the generator's Lbar / sf (numpy) and the oracle's (C) narrow to the same float32 uniforms
        Lbar, sf = o.tone(rgba)
        assert np.float32(Lbar) == np.float32(meta["Lbar_f32"]) and np.float32(sf) == np.float32(meta["sf_f32"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_equals_executed_shader_text(name):
    """The product path against the same arrays, with no oracle in between: float colours on the lattice, and the presented
    canvas where the configuration has one (the scale factor comes from the device's own luminance reduction)."""
    from ray_tracer_amd import abi
    d = os.path.join(ROOT, name)
    meta = json.load(open(os.path.join(d, "meta.json")))
    step = meta["step"]
    if "drag" in meta:
        flat, _, before, _ = _dragged(meta)
        rt = abi.RayTracerHIP(before.cfg)
        rt.setScene(before)                       # the tree of the frame before the drag
        rt.trace()
        rt.setLights(flat.lights, flat.n_lights_loop)
        rt.updateVertices(flat.verts)             # rt_update_vertices: the moved vertices through that tree
        rt.trace()
    else:
        flat = _scene(meta)
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rt.trace()
    want = np.fromfile(os.path.join(d, "tracer.f32"), np.float32).reshape(meta["samples"] + [3])
    got = np.ascontiguousarray(rt.readFloat()[::step, ::step])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if "canvas" in meta:
        Lbar, sf = rt.tone()
        assert np.float32(Lbar) == np.float32(meta["Lbar_f32"]) and np.float32(sf) == np.float32(meta["sf_f32"])
        canvas = np.fromfile(os.path.join(d, "canvas.rgba8"), np.uint8).reshape(meta["canvas"] + [4])
        assert np.array_equal(rt.present(), canvas)
    rt.close()
