#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (all bounces) on the 69k-triangle bunny k-d tree at 3840x2160, depth 5.

One step = one whole frame of the hot path with the scene already resident in HBM:
  rt_trace (every ray-tree level + bottom-up composite + RGBA8 resolve) and the log-luminance reduction;
  with N > 1 ranks each rank traces its interleaved 8x8 tiles, the two luminance partial sums are
  all-reduced over RCCL and the RGBA8 tiles are gathered to rank 0, which assembles and presents the frame.
value = rays traced by all ranks (valid ray-tree nodes: primary + reflected + refracted; shadow rays are
reported separately) / max-over-ranks wall time of the K timed steps.

Launch: python bench.py --gpus 1   |   python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import rt_loader  # noqa: E402

rt_loader.load()
from ray_tracer_amd import abi, multi_gpu, scene  # noqa: E402

MESH_MATERIAL = (0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9)   # ka kd ks ke kReflect kRefract ior: the mesh reflects and transmits
HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3


def build_workload(args):
    teapot = (os.path.join(ROOT, "assets", "models", "teapot.obj"), 60.0, (50.0, 75.0, 10.0)) if args.teapot else None
    sc = scene.default_scene(os.path.join(ROOT, "assets", "models", args.model), mesh_material=MESH_MATERIAL,
                             extra_mesh=teapot)
    scene.animate_light(sc, args.time)
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--model", default="bun_zipper.ply")
    ap.add_argument("--teapot", type=int, default=1)
    ap.add_argument("--time", type=float, default=0.0)
    ap.add_argument("--cpu-step", type=int, default=3, help="CPU baseline traces every n-th pixel in x and y")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    torch.cuda.set_device(local)
    comm = multi_gpu.Comm(rank, world, local, backend="nccl") if world > 1 else multi_gpu.Comm(0, 1, local, backend=None)

    sc = build_workload(args)
    cfg = abi.default_config(args.width, args.height, depth=args.depth, shard_index=rank, shard_count=world, device=local)
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(args.width, args.height))
    rt = abi.RayTracerHIP(cfg)
    t0 = time.perf_counter()
    rt.setScene(flat)
    setup_s = time.perf_counter() - t0
    frame = multi_gpu.FrameJob(rt, comm)

    for _ in range(args.warmup):
        frame.step()
    comm.barrier()
    torch.cuda.synchronize()
    lvl_ms = np.zeros(8)
    stage_ms = np.zeros((8, 7))
    trace_ms = comp_ms = tone_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame.step()
        st = rt.stats()
        lvl_ms += np.array(list(st.level_kernel_ms))
        stage_ms += np.array([list(r) for r in st.level_stage_ms])
        trace_ms += st.trace_ms
        comp_ms += st.composite_ms
        tone_ms += st.tone_ms
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = comm.max_float(dt)
    st = rt.stats()
    rays = comm.sum_int(int(st.rays_total))
    shadow = comm.sum_int(int(st.shadow_rays))
    per_level = [comm.sum_int(int(x)) for x in st.rays_per_level]
    K = args.steps
    lvl_ms /= K
    stage_ms /= K

    if rank != 0:
        comm.close()
        return

    ms_per_step = dt / K * 1e3
    out = {
        "metric": "Mrays/s (all bounces), 69k-tri bunny k-d tree @4K, depth 5",
        "value": rays / (dt / K) / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic scene: the reference's generateScene() objects + Stanford bunny" + (" + teapot" if args.teapot else ""),
        "config": {
            "workload": "BASELINE config 3: %s (%d faces)%s, default 8-kd_tree.js scene, reflective+transmissive mesh %s, %dx%d, depth %d, 2 lights, k-d tree, tiles sharded over %d GPU(s)"
                        % (args.model, len(flat.faces), " + teapot.obj" if args.teapot else "", str(MESH_MATERIAL), args.width, args.height, args.depth, world),
            "rays_per_frame": rays, "shadow_rays_per_frame": shadow, "rays_per_level": per_level[:args.depth],
            "step": "rt_trace + luminance reduction" + (" + RCCL all-reduce + tile gather + assemble + present" if world > 1 else " + present"),
        },
        "mshadow_rays_per_s": shadow / (dt / K) / 1e6,
        "trace_only_ms": trace_ms / K, "composite_ms": comp_ms / K, "tone_ms": tone_ms / K,
        "level_kernel_ms": [round(float(x), 4) for x in lvl_ms[:args.depth]],
        "stage_names": ["k_gen_prims", "k_kd_walk<closest>", "k_tri_test<closest>", "k_shade_a", "k_kd_walk<shadow>", "k_tri_test<shadow>", "k_shade_b"],
        "tri_records": [int(x) for x in st.tri_records],
        "level_stage_ms": [[round(float(x), 4) for x in row] for row in stage_ms[:args.depth]],
        "stage_ms": [round(float(x), 4) for x in stage_ms.sum(axis=0)],
        "kd_closest_rays": int(st.kd_closest_rays), "kd_shadow_rays": int(st.kd_shadow_rays),
        "kd_build_ms": st.kd_build_ms, "kd_upload_ms": st.kd_upload_ms, "scene_setup_s": setup_s,
    }

    # ---- CPU baseline + algorithmic work counters (oracle = CPU restatement of the reference shader) ----
    roof = None
    if not args.no_cpu:
        from oracle import oracle
        flat.cfg.shard_index, flat.cfg.shard_count = 0, 1
        ora = oracle.OracleRenderer(flat, kd=None)
        threads = min(os.cpu_count() or 1, 16)
        t0 = time.perf_counter()
        _, _, _, cn = ora.trace(step=args.cpu_step, threads=threads, want_float=False, want_ids=False)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": cn.rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "same frame, every %d-th pixel in x and y (%d rays, %.1f s), oracle/trace_oracle.c with OpenMP"
                      % (args.cpu_step, cn.rays, cpu_s),
        }
        scale = float(args.cpu_step ** 2)
        alg0 = cn.algorithmic(0)
        alg_all = cn.algorithmic(None)
        # dominant kernel = the level-0 launch of k_trace_level (primary rays + their shadow rays); with
        # N ranks each launch handles 1/N of the tiles
        k_ms = float(lvl_ms[0])
        bytes0 = alg0["bytes"] * scale / world
        flops0 = alg0["flops"] * scale / world
        pmc = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_level0.json")
        if os.path.exists(pmc_path) and world == 1:
            with open(pmc_path) as f:
                pmc = json.load(f).get("hbm_bytes_per_launch")
        roof = {
            "bound": "hbm", "kernel": "k_trace_level (level 0)",
            "achieved": bytes0 / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": bytes0 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc,
            "algorithmic_bytes_per_launch": bytes0, "kernel_ms": k_ms,
            "valu_tflops": flops0 / (k_ms * 1e-3) / 1e12, "valu_frac": flops0 / (k_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
            "note": "algorithmic bytes/flops are the REFERENCE restart-traversal counts (8 B/step, 16 B/leaf visit, 48 B/triangle test, 128 B/primitive test; 14/45/60 flop) measured by the instrumented oracle on the sample and scaled by the sampling factor; the scene (about 7 MB) is cache-resident, so neither HBM nor MFMA bounds this kernel",
        }
        out["roofline"] = roof
        out["algorithmic_frame"] = {k: v * scale for k, v in alg_all.items()}
    print(json.dumps(out))
    comm.close()


if __name__ == "__main__":
    main()
