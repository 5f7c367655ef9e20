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
        return abi.FlatScene(mod.py_scene(d), abi.default_config(meta["traced_width"], meta["traced_height"], depth=d["depth"]), cam)
    kw = dict(meta["make_flat"])
    for k in ("mesh_material",):
        if kw.get(k) is not None:
            kw[k] = tuple(kw[k])
    if kw.get("extra_mesh") is not None:
        n, s, t = kw["extra_mesh"]
        kw["extra_mesh"] = (model(n), s, tuple(t))
    return make_flat(**kw)


def test_fixture_set_is_complete():
    assert len(NAMES) >= 13
    kinds = [json.load(open(os.path.join(ROOT, n, "meta.json"))) for n in NAMES]
    assert sum("random_scene" in m for m in kinds) >= 4                             # scenes nobody tuned, through the reference's own classes
    kinds = [m for m in kinds if "make_flat" in m]
    assert sum("canvas" in m for m in kinds) >= 3                                  # Ward, Reinhard and the odd-N present loop
    assert {m["make_flat"].get("ss", 1) for m in kinds} >= {1, 2, 3}
    assert max(m["make_flat"]["depth"] for m in kinds) >= 5


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_executed_shader_text(name):
    d = os.path.join(ROOT, name)
    meta = json.load(open(os.path.join(d, "meta.json")))
    step = meta["step"]
    o = oracle.OracleRenderer(_scene(meta))
    assert (o.W, o.H) == (meta["traced_width"], meta["traced_height"])
    rgb, rgba, ids, cn = o.trace(step=step)
    want = np.fromfile(os.path.join(d, "tracer.f32"), np.float32).reshape(meta["samples"] + [3])
    got = rgb[::step, ::step]
    assert got.shape == want.shape
    # bit-exact: the colours as float32 BITS (so a NaN or a signed zero would count too)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        "%d of %d samples differ" % (int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum()), want.shape[0] * want.shape[1])
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
