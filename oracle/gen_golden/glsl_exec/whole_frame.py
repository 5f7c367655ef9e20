#!/usr/bin/env python3
"""Build container only: ONE whole frame of a configuration of make_fixtures.py through the reference's executed shader
text (every pixel, several processes) against the oracle's frame, float32 bit for bit. Nothing is stored but the outcome.
Usage: python oracle/gen_golden/glsl_exec/whole_frame.py NAME [--jobs 8]     (NAME: a key of make_fixtures.CONFIGS)"""
import os
import shutil
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import rt_loader  # noqa: E402

rt_loader.load()
import make_fixtures  # noqa: E402


def main():
    name = sys.argv[1]
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 8
    args, _, _, kw = make_fixtures.CONFIGS[name]
    dump = os.path.join(make_fixtures.SCRATCH, name + "_whole")
    shutil.rmtree(dump, ignore_errors=True)
    t0 = time.time()
    subprocess.check_call(["node", os.path.join(REPO, "oracle", "gen_golden", "ref_capture.js")] + args +
                          ["--out", os.path.join(dump, "manifest"), "--dump-glsl", dump], stdout=subprocess.DEVNULL)
    out = os.path.join(dump, "frame.f32")
    subprocess.check_call([sys.executable, os.path.join(HERE, "glsl2cpp.py"), dump, out, "--step", "1", "--jobs", str(jobs)])
    t1 = time.time()
    from helpers import make_flat, model
    from oracle import oracle
    kw = dict(kw)
    if kw.get("extra_mesh") is not None:
        n, s, t = kw["extra_mesh"]
        kw["extra_mesh"] = (model(n), s, tuple(t))
    o = oracle.OracleRenderer(make_flat(**kw))
    rgb, rgba, ids, cn = o.trace()
    want = np.fromfile(out, np.float32).reshape(o.H, o.W, 3)
    diff = (rgb.view(np.uint32) != want.view(np.uint32)).any(-1)
    print("%s: %d x %d pixels, %d rays in the oracle's frame; executed shader text %.0f s; pixels that differ: %d"
          % (name, o.W, o.H, cn.rays, t1 - t0, int(diff.sum())))
    shutil.rmtree(dump, ignore_errors=True)
    return 1 if diff.any() else 0


if __name__ == "__main__":
    sys.exit(main())
