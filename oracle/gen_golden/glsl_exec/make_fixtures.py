#!/usr/bin/env python3
"""Build container only: executes the reference's own generated shader text for a few configurations and stores the
resulting pixel arrays under tests/golden/glsl_exec/. Per configuration:
  1. oracle/gen_golden/ref_capture.js runs the reference's page script under Node and dumps the shader sources, uniform
     values and texture payloads of one frame to a scratch directory OUTSIDE the repository;
  2. glsl2cpp.py compiles the tracer fragment shader's text as C++ (glsl_semantics.h) and runs it over a lattice of pixels
     -> tracer.f32 (float32 colours);
  3. where the configuration asks for it, the whole frame is quantised to RGBA8 and the present shader's text is run over
     it with Lbar / sf computed as js/8-kd_tree.js:2559-2569 does -> canvas.rgba8.
tests/test_glsl_exec_golden.py compares the CPU oracle with these arrays. The arrays pin nothing by the rules of parity
(glsl_semantics.h is a stand-in for a GLSL implementation); they are the first EXECUTION of the reference's shader text
the oracle was checked against. Usage: python oracle/gen_golden/glsl_exec/make_fixtures.py [name ...]
"""
import json
import math
import os
import shutil
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
OUT = os.path.join(REPO, "tests", "golden", "glsl_exec")
SCRATCH = "/tmp/glsl_exec_scratch"
MM = (0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9)
TEAPOT = ("teapot.obj", 60.0, (50.0, 75.0, 10.0))

# name -> (ref_capture arguments, lattice step, present pass?, arguments of tests/helpers.make_flat that build the same scene)
CONFIGS = {
    # checkpoint 6 (GLSL ES 1.00, no mesh, one set light of two, INFINITY 100000.1): the reference's own CPU-runnable case
    "c1_checkpoint6_depth5_640x480": (["--script", "6-transmission.js"], 4, False,
                                      dict(model_name=None, width=640, height=480, depth=5, time=None, second_light=False, infinity=100000.1)),
    "k8_res4_depth3_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--depth", "3",
                                "--mesh-material", ",".join(str(v) for v in MM)], 4, False,
                               dict(model_name="bun_zipper_res4.ply", width=640, height=480, depth=3, time=0.0, mesh_material=MM)),
    "k8_res3_teapot_depth5_t7.5_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res3.ply", "--depth", "5", "--time", "7.5",
                                            "--mesh-material", ",".join(str(v) for v in MM), "--extra-obj", "models/teapot.obj",
                                            "--extra-scale", "60", "--extra-translate", "50,75,10"], 4, False,
                                           dict(model_name="bun_zipper_res3.ply", width=640, height=480, depth=5, time=7.5, mesh_material=MM,
                                                extra_mesh=TEAPOT)),
    "c2_res4_depth1_1920x1080": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--width", "1920", "--height", "1080",
                                  "--time", "7.5"], 12, False,
                                 dict(model_name="bun_zipper_res4.ply", width=1920, height=1080, depth=1, time=7.5)),
    # the benchmarked frame (bench.py, benchmark configuration 3): full bunny + teapot, reflective and transmissive, depth 5, 4K
    "c3_bunny_teapot_depth5_3840x2160": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper.ply", "--width", "3840", "--height", "2160",
                                          "--depth", "5", "--mesh-material", ",".join(str(v) for v in MM), "--extra-obj", "models/teapot.obj",
                                          "--extra-scale", "60", "--extra-translate", "50,75,10"], 12, False,
                                         dict(model_name="bun_zipper.ply", width=3840, height=2160, depth=5, time=0.0, mesh_material=MM,
                                              extra_mesh=TEAPOT)),
    # benchmark configuration 5's scene at one instant: project.js with its triangles switched on, full bunny, depth 4, 4K
    "c5_project_triangles_depth4_3840x2160": (["--script", "project.js", "--model", "models/bun_zipper.ply", "--set", "RENDER_TRIANGLES=true",
                                               "--ss", "1", "--width", "3840", "--height", "2160", "--time", "1.5"], 12, False,
                                              dict(model_name="bun_zipper.ply", width=3840, height=2160, depth=4, time=1.5)),
    # the mesh dragged with the reference's own UI between two ticks 0.1 s apart: the second frame reads the moved vertices
    # through the tree and root box of the first (js/8-kd_tree.js:2459-2533: vertices every frame, tree at most every 0.30 s)
    "k8_res4_dragged_stale_tree_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--depth", "3", "--time", "1.0",
                                            "--mesh-material", ",".join(str(v) for v in MM), "--drag", "330,40,335,120,330,200", "--drag-dt", "0.1"], 2, False,
                                           dict(model_name="bun_zipper_res4.ply", width=640, height=480, depth=3, time=1.1, mesh_material=MM)),
    # the camera after a pan, a rotation and a wheel zoom with the reference's own TrackballControls (the steps of the fixture
    # k8_res4_trackball_640x480): corner rays no longer symmetric, camera off the origin; the camera uniforms travel in meta.json
    "k8_res4_trackball_depth3_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--depth", "3",
                                          "--mesh-material", ",".join(str(v) for v in MM),
                                          "--trackball", "down:2:20:20,move:24:22,tick,up:24:22,tick,tick,down:0:20:20,move:22:21,tick,move:23:23,tick,up:23:23,tick,tick,wheel:-100,tick,tick"],
                                         4, False, dict(model_name="bun_zipper_res4.ply", width=640, height=480, depth=3, time="trackball", mesh_material=MM)),
    # the shader's brute-force loop over all faces (USE_KD_TREE false)
    "k8_res4_no_kdtree_depth2_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--depth", "2", "--set", "USE_KD_TREE=false",
                                          "--mesh-material", ",".join(str(v) for v in MM)], 8, False,
                                         dict(model_name="bun_zipper_res4.ply", width=640, height=480, depth=2, time=0.0, mesh_material=MM, use_kd_tree=0)),
    # a shallow tree: most leaves hold more than four faces and keep only the first four (js/8-kd_tree.js:1728-1730)
    "k8_res3_kd_depth8_depth3_640x480": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res3.ply", "--depth", "3", "--set", "MAX_KD_TREE_DEPTH=8",
                                          "--mesh-material", ",".join(str(v) for v in MM)], 4, False,
                                         dict(model_name="bun_zipper_res3.ply", width=640, height=480, depth=3, time=0.0, mesh_material=MM, max_kd_depth=8)),
    "project_ss2_ward_160x120": (["--script", "project.js", "--width", "160", "--height", "120", "--time", "2.25"], 1, True,
                                 dict(model_name="bun_zipper_res3.ply", width=160, height=120, depth=4, ss=2, time=2.25, render_triangles=0)),
    "tone7_ss2_reinhard_160x120": (["--script", "7-tone_reproduction.js", "--width", "160", "--height", "120",
                                    "--set", "TONE_REPRODUCTION_METHOD=1"], 1, True,
                                   dict(model_name="bun_zipper_res3.ply", width=160, height=120, depth=4, ss=2, time=None, render_triangles=0,
                                        tr_method=1)),
    "k8_res4_ss3_depth2_96x72": (["--script", "8-kd_tree.js", "--model", "models/bun_zipper_res4.ply", "--width", "96", "--height", "72",
                                  "--depth", "2", "--ss", "3", "--time", "11.0", "--mesh-material", ",".join(str(v) for v in MM)], 1, True,
                                 dict(model_name="bun_zipper_res4.ply", width=96, height=72, depth=2, ss=3, time=11.0, mesh_material=MM)),
}


def f32(v):
    return struct.unpack("<f", struct.pack("<f", float(v)))[0]


def main():
    names = sys.argv[1:] or list(CONFIGS)
    for name in names:
        args, step, do_present, scene_kw = CONFIGS[name]
        dump = os.path.join(SCRATCH, name)
        shutil.rmtree(dump, ignore_errors=True)
        subprocess.check_call(["node", os.path.join(REPO, "oracle", "gen_golden", "ref_capture.js")] + args +
                              ["--out", os.path.join(dump, "manifest"), "--dump-glsl", dump], stdout=subprocess.DEVNULL)
        out = os.path.join(OUT, name)
        os.makedirs(out, exist_ok=True)
        tracer = os.path.join(out, "tracer.f32")
        subprocess.check_call([sys.executable, os.path.join(HERE, "glsl2cpp.py"), dump, tracer, "--step", str(step)])
        S = scene_kw.get("ss", 1)
        W, H = scene_kw["width"] * S, scene_kw["height"] * S
        meta = {"generator": "oracle/gen_golden/glsl_exec/make_fixtures.py", "ref_capture_args": args, "step": step,
                "traced_width": W, "traced_height": H, "samples": [(H + step - 1) // step, (W + step - 1) // step],
                "make_flat": scene_kw}
        if "--trackball" in args:
            with open(os.path.join(dump, "dump.json")) as f:
                last = {}
                for un in json.load(f)["uniforms"]:
                    last[un["name"]] = un["value"]
            meta["camera"] = {k: last[k] for k in ("cameraPos", "ray00", "ray01", "ray10", "ray11")}
            with open(os.path.join(dump, "manifest", "manifest.json")) as f:
                meta["make_flat"]["time"] = json.load(f)["trackball"]["last_time"]   # the light follows the last tick's time
        if "--drag" in args:
            with open(os.path.join(dump, "manifest", "manifest.json")) as f:
                dr = json.load(f)["drag"]
            assert dr["moving"] and dr["selected_index"] >= 0, "the drag did not grab the mesh"
            meta["drag"] = {k: dr[k] for k in ("selected_index", "old_position", "positions", "dt", "mouse")}
            meta["tree"] = "of the first tick (vertices before the drag)"
        if do_present:
            rgb = np.fromfile(tracer, np.float32).reshape(H, W, 3)
            c = np.where(np.isnan(rgb), 0.0, np.clip(rgb, 0.0, 1.0)).astype(np.float32)
            q = (c * np.float32(255.0) + np.float32(0.5)).astype(np.int32).astype(np.uint8)
            frame = np.concatenate([q, np.full((H, W, 1), 255, np.uint8)], axis=-1)
            fpath = os.path.join(dump, "frame.rgba8")
            frame.tofile(fpath)
            # js/8-kd_tree.js:2559-2569, term by term in pixel order
            acc = 0.0
            flat = frame.reshape(-1, 4).astype(np.float64)
            L = 0.27 * (flat[:, 0] * 1000 / 256.0) + 0.67 * (flat[:, 1] * 1000 / 256.0) + 0.07 * (flat[:, 2] * 1000 / 256.0)
            for v in np.log(0.0001 + L) / len(L):
                acc += v
            Lbar = math.exp(acc)
            sf = ((1.219 + (500 * 0.5) ** 0.4) / (1.219 + Lbar ** 0.4)) ** 2.5
            # ... and by the reference's own loop: the page script once more, its readPixels handed this very frame
            again = os.path.join(dump, "manifest_tone")
            subprocess.check_call(["node", os.path.join(REPO, "oracle", "gen_golden", "ref_capture.js")] + args +
                                  ["--pixels", "file:" + fpath, "--out", again], stdout=subprocess.DEVNULL)
            with open(os.path.join(again, "manifest.json")) as f:
                ref_tone = json.load(f)
            assert f32(ref_tone["Lbar"]) == f32(Lbar) and f32(ref_tone["sf"]) == f32(sf), (ref_tone["Lbar"], Lbar, ref_tone["sf"], sf)
            Lbar, sf = ref_tone["Lbar"], ref_tone["sf"]
            meta["tone_from"] = "the reference's own loop (js/8-kd_tree.js:2559-2569) over the executed frame"
            canvas = os.path.join(out, "canvas.rgba8")
            subprocess.check_call([sys.executable, os.path.join(HERE, "glsl2cpp.py"), dump, canvas, "--present", fpath, repr(f32(Lbar)), repr(f32(sf))])
            meta.update(Lbar_f32=f32(Lbar), sf_f32=f32(sf), canvas=[scene_kw["height"], scene_kw["width"]])
        with open(os.path.join(out, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
        shutil.rmtree(dump, ignore_errors=True)   # the scratch copy of the shader text goes


if __name__ == "__main__":
    main()
