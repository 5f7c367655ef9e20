#!/usr/bin/env python3
"""Animated frames with a k-d rebuild per frame (benchmark configuration 5): the project scene with the mesh switched on,
t = k/60 for k = 0..frames-1, light 0 on the reference's orbit (js/8-kd_tree.js:2711-2716), the mesh dragged along a
small circle (what the reference's mouse drag does through Mesh.translate, js/8-kd_tree.js:2055-2064), so the tree is
rebuilt for every frame (the reference rebuilds at most every 0.3 s, js/8-kd_tree.js:2494-2495).

Frames are independent given the time, so they are sharded over the ranks (frame k on rank k mod N) with no exchange
at all; every rank keeps several frames in flight on alternating contexts. One JSON line: frames/s and Mrays/s of the
whole job.

Launch: python bench_animation.py   |   python -m torch.distributed.run --nproc-per-node N ... bench_animation.py --gpus N
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import rt_loader  # noqa: E402

rt_loader.load()
from ray_tracer_amd import abi, multi_gpu, scene  # noqa: E402


def light_orbit(flat, t):
    x = 500.0 * math.cos(2.0 * math.pi * (t % 60.0) / 60.0)
    return [((x, 300.0, x - 280.0), flat.lights[0][1])] + list(flat.lights[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--ss", type=int, default=1, help="SUPERSAMPLING (project.js has 2: 7680x4320 traced for a 4K canvas)")
    ap.add_argument("--model", default="bun_zipper.ply")
    ap.add_argument("--in-flight", type=int, default=3)
    ap.add_argument("--kd-builder", default="device", choices=["host", "device"])
    ap.add_argument("--static", action="store_true", help="no mesh motion and no rebuild (what the light animation alone costs)")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch one process per GPU with torch.distributed.run" % (world, args.gpus))
    import torch
    torch.cuda.set_device(local)
    sync = multi_gpu.Comm(rank, world, local, backend=args.backend, device="cuda:%d" % local) if world > 1 else multi_gpu.Comm(0, 1, local, backend=None, device="cuda:%d" % local)
    alone = multi_gpu.Comm(0, 1, local, backend=None, device="cuda:%d" % local)   # frames need no exchange

    sc = scene.default_scene(os.path.join(ROOT, "assets", "models", args.model))
    cfg = abi.default_config(args.width, args.height, depth=args.depth, supersampling=args.ss, device=local)
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(args.width, args.height))
    base = np.ascontiguousarray(flat.verts, np.float64)
    on_device = args.kd_builder == "device"
    rts = []
    for _ in range(max(1, args.in_flight)):
        r = abi.RayTracerHIP(cfg)
        r.setScene(flat, kd_on_device=on_device)
        rts.append(r)
    job = multi_gpu.PipelinedFrameJob(rts, alone)

    def frame(k):
        t = k / 60.0
        rt = job.next_context()
        rt.setLights(light_orbit(flat, t), flat.n_lights_loop)
        if not args.static:
            a = 2.0 * math.pi * k / 240.0
            rt.setMesh(base + np.array([12.0 * math.sin(a), 0.0, 12.0 * (math.cos(a) - 1.0)]), flat.faces, flat.face_mat)
            rt.buildKdTree(on_device=on_device)
        job.step()

    for k in range(args.warmup):
        frame(k)
    job.flush()
    job.rays = job.shadow_rays = 0
    sync.barrier()
    torch.cuda.synchronize()
    build_ms = 0.0
    t0 = time.perf_counter()
    mine = range(rank, args.frames, world)
    for k in mine:
        frame(k)
    job.flush()
    sync.barrier()
    torch.cuda.synchronize()
    dt = sync.max_float(time.perf_counter() - t0)
    rays = sync.sum_int(job.rays)
    shadow = sync.sum_int(job.shadow_rays)
    build_ms = rts[0].stats().kd_build_ms
    upload_ms = rts[0].stats().kd_upload_ms
    if rank == 0:
        print(json.dumps({
            "metric": "frames/s, animated project scene with a k-d rebuild per frame", "value": args.frames / dt, "unit": "frames/s",
            "n_gpus": world, "frames": args.frames, "ms_per_frame": dt / args.frames * 1e3, "mrays_per_s": rays / dt / 1e6,
            "mshadow_rays_per_s": shadow / dt / 1e6, "higher_is_better": True, "scaling": "strong", "dtype": "f32",
            "config": {"workload": "BASELINE.json config 5: default scene + %s, %dx%d, supersampling %d, depth %d, t = k/60, mesh dragged along a circle, %s"
                                   % (args.model, args.width, args.height, args.ss, args.depth,
                                      "no rebuild" if args.static else "tree rebuilt every frame by the %s builder" % args.kd_builder),
                       "frames_in_flight": len(rts), "sharding": "frame k on rank k mod N, no exchange"},
            "kd_build_ms_last": build_ms, "kd_upload_ms_last": upload_ms, "frames_retraced": job.redone}))
    sync.close()


if __name__ == "__main__":
    main()
