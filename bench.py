#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (all bounces) on the 69k-triangle bunny k-d tree at 3840x2160, depth 5.

One step = one whole frame of the hot path with the scene already resident in HBM:
  rt_trace (every ray-tree level + bottom-up composite + RGBA8 resolve) and the log-luminance reduction;
  with N > 1 ranks each rank traces its interleaved 8x8 tiles, the two luminance partial sums are
  all-reduced over RCCL and the RGBA8 tiles are gathered to rank 0, which assembles and presents the frame.
value = rays traced by all ranks (valid ray-tree nodes: primary + reflected + refracted; shadow rays are
reported separately) / max-over-ranks wall time of the K timed steps.

Launch: python bench.py --gpus N (starts its N ranks itself)   |   python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Every frame in flight owns a HIP stream. The runtime maps streams onto 4 hardware queues by default, and two streams
# that share a queue run their kernels one after the other: with 4 frames in flight a 1/8 shard took 2.23 ms per frame
# against 1.99 ms with 3. Sixteen queues let a shard keep up to twelve frames in flight (1.75 ms). Read by the HIP
# runtime when it starts, so it is set before anything touches the GPU (RT_HW_QUEUES overrides the number; the
# environment's own GPU_MAX_HW_QUEUES usually just restates the default of 4).
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_HW_QUEUES", "16")

import numpy as np  # noqa: E402

import rt_loader  # noqa: E402

rt_loader.load()
from ray_tracer_amd import abi, multi_gpu, scene  # noqa: E402

MESH_MATERIAL = (0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9)   # ka kd ks ke kReflect kRefract ior: the mesh reflects and transmits
HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3
VALU_ISSUE_PEAK = 2.0        # wave64 vector instructions per cycle per CU: four SIMD-32 units, two cycles each
L2_PEAK_BS = 34.5e12         # aggregate L2 bandwidth
L2_LINE_BYTES = 128.0
VALU_MIX_CEILING = 1.0       # what a CU issues of the walk's own instruction mix (compares, selects, IEEE division, three-operand
                             # forms: 0.9 per cycle each; plain two-operand arithmetic 1.5 - 1.65): profiles/r02_valu_issue_rates.txt
PMC_DIGEST = "r03_pmc_digest.json"   # written by tools/prof.sh + tools/prof_summary.py for the committed kernels


def identity(args):
    """What a PMC digest belongs to: the library binary and the workload. Counts per launch from a profile of another
    binary or another workload say nothing about this run."""
    import hashlib
    with open(abi.LIB_PATH, "rb") as f:
        lib = hashlib.sha256(f.read()).hexdigest()
    return {"lib_sha256": lib, "workload": {"model": args.model, "teapot": int(args.teapot), "width": args.width, "height": args.height,
                                            "depth": args.depth, "time": args.time, "shards": max(args.emulate_shards, 1),
                                            "kd_builder": args.kd_builder, "gpus": args.gpus}}


TEAPOT = (os.path.join(ROOT, "assets", "models", "teapot.obj"), 60.0, (50.0, 75.0, 10.0))   # file, scale, translation (checkpoint 6 places it so, js/6-transmission.js:2010-2014)


def build_workload(model="bun_zipper.ply", teapot=True, time=0.0):
    """benchmark configuration 3: the default 8-kd_tree.js scene, full bunny + teapot.obj as a second mesh (indices rebased), the
    mesh material reflective and transmissive. tests/test_gpu_parity.py renders exactly this scene against the oracle."""
    sc = scene.default_scene(os.path.join(ROOT, "assets", "models", model), mesh_material=MESH_MATERIAL,
                             extra_mesh=TEAPOT if teapot else None)
    scene.animate_light(sc, time)
    return sc


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: starts the N ranks as child processes of this one (which never
    touches the GPU), forwards rank 0's JSON line and returns the first non-zero exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    status = 0
    alive = set(range(n))
    out0 = b""
    while alive:
        for r in sorted(alive):
            try:
                if r == 0:
                    o, _ = procs[0].communicate(timeout=0.5)
                    out0 += o or b""
                else:
                    procs[r].wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.discard(r)
            if procs[r].returncode != 0 and status == 0:
                status = procs[r].returncode
                for q in alive:            # the others would wait for the lost rank in their next collective
                    procs[q].terminate()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return status


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
    ap.add_argument("--cpu-step", type=int, default=1, help="CPU baseline traces every n-th pixel in x and y")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--emulate-shards", type=int, default=0, help="single rank only: trace shard 0 of N (what one GPU of an N-GPU job does, without the exchange)")
    ap.add_argument("--emulate-index", type=int, default=0, help="with --emulate-shards: which shard to trace")
    ap.add_argument("--emulate-role", default="root", choices=["root", "worker", "whole"],
                    help="with --emulate-shards: the device work of rank 0 of such a job (own tiles + scatter of the gathered canvas), of any other rank "
                         "(own tiles only), or the shard's trace followed by a present pass over the whole frame")
    ap.add_argument("--parts", type=int, default=0, help="concurrent slices per GPU (0 = library default)")
    ap.add_argument("--in-flight", type=int, default=0, help="frames in flight per GPU (contexts that alternate frames); 1 = finish every frame before the next; "
                                                             "0 = 3 for a whole frame per GPU, 6 / 12 / 12 for a half / a quarter / an eighth of it (a small shard leaves the GPU room for more)")
    ap.add_argument("--walk-waves", type=int, default=0, help="persistent walk waves per CU and context (0 = 24, 20 when tiles are exchanged between GPUs)")
    ap.add_argument("--sync-job", action="store_true", help="the host-synchronised frame job (every stage of the exchange waits on the host)")
    ap.add_argument("--kd-builder", default="host", choices=["host", "device"], help="where the k-d tree is built (same tree; the node blocks are laid out in a different order)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "frames"],
                    help="tiles: every frame is split over the GPUs (all-reduce + gather per frame); frames: step k is traced whole by rank k mod N, no exchange")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--device", type=int, default=None, help="HIP device for every rank (rehearsal on a one-GPU box); default LOCAL_RANK")
    ap.add_argument("--opt", action="append", default=[], help="name=value: rt_set_option on every context (experiments)")
    ap.add_argument("--identity", action="store_true", help="print the identity (library hash + workload) a PMC digest of this command line belongs to, and exit")
    ap.add_argument("--spawn-check", action="store_true", help="only start the ranks and add up their numbers over a gloo group (no GPU needed): checks the single-command launch")
    args = ap.parse_args()

    if args.identity:
        print(json.dumps(identity(args)))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))     # one process per GPU, started from here
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    # stdout carries the JSON line and nothing else: whatever native libraries print on descriptor 1 (gloo announces
    # its connections there) goes to stderr from here on
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.spawn_check:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            json_out.write(json.dumps({"n_gpus": world, "rank_sum": int(t[0])}) + "\n")
            json_out.flush()
        dist.destroy_process_group()
        return

    import torch
    if args.device is not None:
        local = args.device
    torch.cuda.set_device(local)
    comm = (multi_gpu.Comm(rank, world, local, backend=args.backend, device="cuda:%d" % local) if world > 1
            else multi_gpu.Comm(0, 1, local, backend=None, device="cuda:%d" % local))

    sc = build_workload(args.model, bool(args.teapot), args.time)
    by_frames = args.shard == "frames" and world > 1
    job_comm = multi_gpu.Comm(0, 1, local, backend=None, device="cuda:%d" % local) if by_frames else comm
    shard_count = args.emulate_shards if (world == 1 and args.emulate_shards > 1) else (1 if by_frames else world)
    cfg = abi.default_config(args.width, args.height, depth=args.depth, shard_index=0 if by_frames else (args.emulate_index if (world == 1 and args.emulate_shards > 1) else rank), shard_count=shard_count,
                             device=local, parts=args.parts)
    flat = abi.FlatScene(sc, cfg, scene.camera_rays(args.width, args.height))
    pipelined = not args.sync_job and (world == 1 or by_frames or args.backend == "nccl")
    in_flight = args.in_flight if args.in_flight > 0 else (3 if shard_count <= 1 else 6 if shard_count == 2 else 12)
    n_ctx = in_flight if pipelined else 1
    t0 = time.perf_counter()
    rts = []
    for _ in range(n_ctx):
        r = abi.RayTracerHIP(cfg)
        r.setScene(flat, kd_on_device=(args.kd_builder == "device"))
        # with an exchange in flight, the persistent walk leaves a fifth of a CU's wave slots and LDS to the RCCL
        # kernels of the other frames (at partial load the walk does not notice: profiles/r01n_frames_in_flight_sweep.txt)
        waves = args.walk_waves or (20 if (world > 1 and not by_frames) else 24)
        r.setOption("walk_waves_per_cu", waves)
        r.setOption("stage_timing", 0)     # per-kernel events only on the detail frames after the timed region
        for kv in args.opt:
            r.setOption(kv.split("=")[0], float(kv.split("=")[1]))
        rts.append(r)
    rt = rts[0]
    setup_s = (time.perf_counter() - t0) / n_ctx
    frame = None
    if pipelined:
        # one complete frame through the stream-ordered job first; should it raise on any rank, every rank goes on with
        # the host-synchronised job (the ranks agree over the host-side group)
        failed = None
        try:
            if os.environ.get("RT_BENCH_FAIL_PIPELINE"):   # test hook for the fall-back below
                raise RuntimeError("RT_BENCH_FAIL_PIPELINE is set")
            rehearse = (args.emulate_role, shard_count) if (world == 1 and shard_count > 1 and args.emulate_role != "whole") else None
            frame = multi_gpu.PipelinedFrameJob(rts, job_comm, rehearse=rehearse)
            frame.step()
            frame.flush()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            failed = e
        if comm.any_flag(failed is not None):
            sys.stderr.write("rank %d: pipelined frame job unavailable (%r), using the host-synchronised job\n" % (rank, failed))
            pipelined, frame, n_ctx = False, None, 1
            rt.setOption("walk_waves_per_cu", args.walk_waves or 24)
    if frame is None:
        frame = multi_gpu.FrameJob(rt, job_comm)
    flush = frame.flush if pipelined else (lambda: None)

    n_warm = max(args.warmup, n_ctx)       # the first frame of every context sizes its buffers
    for _ in range(n_warm):
        frame.step()
    flush()
    comm.barrier()
    torch.cuda.synchronize()
    lvl_ms = np.zeros(8)
    stage_ms = np.zeros((8, 7))
    trace_ms = comp_ms = tone_ms = 0.0
    my_steps = len(range(rank, args.steps, world)) if by_frames else args.steps
    t0 = time.perf_counter()
    for _ in range(my_steps):
        frame.step()
    flush()
    comm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = comm.max_float(dt)
    if hasattr(frame, "close"):
        frame.close()   # (events and views of the contexts' streams go before the contexts do)
    # per-kernel times: device events of the library around every launch, taken from frames traced one at a time after
    # the timed region (with several frames in flight the kernels of different frames share the GPU)
    n_detail = min(args.steps, 5)
    rt.setOption("walk_waves_per_cu", args.walk_waves or 24)
    rt.setOption("stage_timing", 1)
    for _ in range(n_detail):
        rt.trace()
        rt.tone()
        st = rt.stats()
        lvl_ms += np.array(list(st.level_kernel_ms))
        stage_ms += np.array([list(r) for r in st.level_stage_ms])
        trace_ms += st.trace_ms
        comp_ms += st.composite_ms
        tone_ms += st.tone_ms
    st = rt.stats()
    if by_frames:   # every rank traces whole frames: the per-frame counts are those of any one of them
        rays, shadow, per_level = int(st.rays_total), int(st.shadow_rays), [int(x) for x in st.rays_per_level]
    else:
        rays = comm.sum_int(int(st.rays_total))
        shadow = comm.sum_int(int(st.shadow_rays))
        per_level = [comm.sum_int(int(x)) for x in st.rays_per_level]
    K = args.steps
    lvl_ms /= n_detail
    stage_ms /= n_detail
    trace_ms /= n_detail
    comp_ms /= n_detail
    tone_ms /= n_detail

    # the same frame with the canvas copied to a host buffer (what a caller without device memory pays);
    # reported beside the metric, never as the metric
    pcie_ms = pcie_pinned_ms = None
    if world == 1:
        n_pcie = min(K, 5)
        sync_frame = multi_gpu.FrameJob(rt, comm)
        pageable = np.zeros((args.height, args.width, 4), np.uint8)
        pinned = torch.empty((args.height, args.width, 4), dtype=torch.uint8).pin_memory().numpy()
        for dst in (pageable, pinned):
            sync_frame.step()
            rt.readRGBA8(dst)            # (touches the pages once)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_pcie):
                sync_frame.step()
                rt.readRGBA8(dst)
            ms = (time.perf_counter() - t1) / n_pcie * 1e3
            if dst is pageable:
                pcie_ms = ms
            else:
                pcie_pinned_ms = ms

    if rank != 0:
        comm.close()
        return

    ms_per_step = dt / K * 1e3
    out = {
        "metric": "Mrays/s (all bounces), 69k-tri bunny k-d tree @4K, depth 5",
        "value": rays / (dt / K) / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world, "steps": K, "warmup": n_warm, "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic scene: the reference's generateScene() objects + Stanford bunny" + (" + teapot" if args.teapot else ""),
        "config": {
            "workload": "BASELINE.json config 3: %s (%d faces)%s, default 8-kd_tree.js scene, reflective+transmissive mesh %s, %dx%d, depth %d, 2 lights, k-d tree, %s sharded over %d GPU(s)"
                        % (args.model, len(flat.faces), " + teapot.obj" if args.teapot else "", str(MESH_MATERIAL), args.width, args.height, args.depth, "frames" if by_frames else "tiles", world),
            "rays_per_frame": rays, "shadow_rays_per_frame": shadow, "rays_per_level": per_level[:args.depth],
            "step": "rt_trace + luminance reduction" + (" + RCCL all-reduce + tile gather + assemble + present" if (world > 1 and not by_frames) else " + present"),
        },
        "mshadow_rays_per_s": shadow / (dt / K) / 1e6,
        "pcie_inclusive": None if pcie_ms is None else {"ms_per_step": pcie_ms, "value": rays / (pcie_ms * 1e-3) / 1e6,
                                                        "ms_per_step_pinned": pcie_pinned_ms, "value_pinned": rays / (pcie_pinned_ms * 1e-3) / 1e6,
                                                        "note": "one frame at a time + rt_read_rgba8 of the %dx%d RGBA8 target into the caller's host buffer: pageable memory, and (_pinned) page-locked memory" % (args.width, args.height)},
        "trace_only_ms": trace_ms, "composite_ms": comp_ms, "tone_ms": tone_ms,
        "frames_in_flight": n_ctx, "frame_job": "pipelined (stream-ordered, device-resident exchange)" if pipelined else "host-synchronised",
        "frames_retraced": getattr(frame, "redone", 0),
        # what the collectives saw: the backend, how many ranks the process group of the first device collective held, and
        # the mean host time per frame of completing a frame (rt_trace_end, which waits for the device, + reading the
        # all-reduced overflow flag: finish_flag_ms is that second part alone)
        "backend": (args.backend if world > 1 else None),
        "world_size_joined": (getattr(frame, "joined", None) or (comm.dist.get_world_size() if comm.dist else 1)),
        "finish_host_ms": (frame.finish_s / frame.finished * 1e3) if getattr(frame, "finished", 0) else None,   # mostly the wait for the device (frames in flight)
        "finish_flag_ms": (frame.flag_s / frame.finished * 1e3) if getattr(frame, "finished", 0) else None,
        "step_kernel_ms": [round(float(x), 4) for x in lvl_ms[:args.depth + 1]],
        "stage_names": ["k_gen_prims", "k_kd_walk", "k_shade_b", "k_shade_a"],
        "pipeline_step": "step k = k_gen_prims(level k), k_kd_walk (closest-hit rays of level k with the shadow rays of level k-1, triangle tests inside the walk), k_shade_b(level k-1), k_shade_a(level k)",
        "tri_records": [int(x) for x in st.tri_records],
        "step_stage_ms": [[round(float(x), 4) for x in row[:4]] for row in stage_ms[:args.depth + 1]],
        "stage_ms": [round(float(x), 4) for x in stage_ms.sum(axis=0)[:4]],
        "kd_closest_rays": int(st.kd_closest_rays), "kd_shadow_rays": int(st.kd_shadow_rays),
        "kd_build_ms": st.kd_build_ms, "kd_upload_ms": st.kd_upload_ms, "scene_setup_s": setup_s,
    }

    # ---- CPU baseline + algorithmic work counters (oracle = CPU restatement of the reference shader) ----
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
        # ---- roofline of the dominant kernel, k_kd_walk (one launch per pipeline step: the closest-hit rays of a
        # ray-tree level and the shadow rays of the level before it, triangle tests included).
        # The scene is cache-resident and nothing contracts, so neither HBM nor MFMA bounds this kernel; the roof it
        # sits under is VECTOR INSTRUCTION ISSUE: a CU has four SIMD-32 units, a wave64 vector instruction takes one of
        # them for 2 cycles, so a CU issues at most 2 per cycle. achieved = vector instructions of one launch (PMC
        # SQ_INSTS_VALU of the committed profile of this binary and workload; the count does not depend on timing)
        # / (launch duration measured here with HIP events on the library's stream x shader clock of that profile
        # (GRBM_GUI_ACTIVE / duration) x number of CUs). Beside it: the L2 and HBM fractions from the same profile's
        # request counts over the live duration, and the reference-algorithm bytes and flops of the same rays (what the
        # reference's restart traversal would fetch: 8 B per descent step, 16 B per leaf visit, 48 B per triangle test;
        # counted by the instrumented oracle over the same frame) - a workload figure, not a bandwidth the kernel moves.
        tile_world = 1 if by_frames else world
        scale = float(args.cpu_step ** 2) / tile_world
        D = args.depth
        n_steps = D + 1
        names = out["stage_names"]
        per_stage_bytes = np.zeros(4)
        per_stage_flops = np.zeros(4)
        for lv in range(D):
            per_stage_bytes += np.array([128.0 * cn.prim_tests[lv],
                                         8.0 * (cn.kd_steps[lv] + cn.sh_steps[lv]) + 16.0 * (cn.kd_leaf_visits[lv] + cn.sh_leaf_visits[lv])
                                         + 48.0 * (cn.kd_tri_tests[lv] + cn.sh_tri_tests[lv]), 0.0, 128.0 * cn.sh_prim_tests[lv]])
            per_stage_flops += np.array([60.0 * cn.prim_tests[lv], 14.0 * (cn.kd_steps[lv] + cn.sh_steps[lv])
                                         + 45.0 * (cn.kd_tri_tests[lv] + cn.sh_tri_tests[lv]), 0.0, 60.0 * cn.sh_prim_tests[lv]])
        per_stage_bytes *= scale
        per_stage_flops *= scale
        tot_ms = stage_ms[:n_steps, :4].sum(axis=0)
        launches = np.array([D, n_steps, D, D], dtype=float)   # per frame: a level has a closest-hit half (steps 0..D-1) and a shadow half (steps 1..D)
        digest = {}
        dpath = os.path.join(ROOT, "profiles", PMC_DIGEST)
        digest_note = "no committed PMC digest"
        if os.path.exists(dpath):
            with open(dpath) as f:
                digest = json.load(f)
            # per-launch instruction and request counts only mean something for the binary and workload they were taken on
            if digest.get("__identity") == identity(args):
                digest_note = "PMC digest matches this binary and workload"
            else:
                digest_note = "the committed PMC digest belongs to another binary or workload: counter-based figures left out"
                digest = {}
        n_cus = torch.cuda.get_device_properties(local).multi_processor_count
        table = []
        for k in range(4):
            dur_s = tot_ms[k] / launches[k] * 1e-3
            g = digest.get(names[k], {})
            row = {"kernel": names[k], "launches_per_frame": int(launches[k]), "avg_launch_ms": tot_ms[k] / launches[k],
                   "reference_algorithm_bytes_per_launch": per_stage_bytes[k] / launches[k],
                   "reference_algorithm_tflops": per_stage_flops[k] / launches[k] / dur_s / 1e12,
                   "valu_issue": None, "l2_frac": None, "hbm_frac": None, "traffic": None,
                   "valu_issue_frac_profiled": None, "lanes_per_valu_inst_profiled": g.get("valu_lanes_per_inst"),
                   "l2_hit_profiled": g.get("l2_hit"), "avg_launch_ms_profiled": (g.get("avg_us") or 0.0) / 1e3 or None}
            if world == 1 and g.get("valu_insts_per_launch") and g.get("clock_mhz_profiled"):
                cyc = dur_s * g["clock_mhz_profiled"] * 1e6
                row["valu_issue"] = g["valu_insts_per_launch"] / (cyc * n_cus)
                row["valu_issue_frac_profiled"] = g["valu_issue_frac"] / VALU_ISSUE_PEAK
                row["traffic"] = g.get("hbm_bytes_per_launch")
                if g.get("tcc_req_per_launch"):
                    row["l2_frac"] = g["tcc_req_per_launch"] * L2_LINE_BYTES / dur_s / L2_PEAK_BS
                if row["traffic"] is not None:
                    row["hbm_frac"] = row["traffic"] / dur_s / (HBM_PEAK_GBS * 1e9)
            table.append(row)
        d = table[int(np.argmax(tot_ms))]
        out["roofline"] = {
            "bound": "valu_issue", "kernel": d["kernel"],
            "achieved": d["valu_issue"], "peak": VALU_ISSUE_PEAK, "mix_ceiling": VALU_MIX_CEILING, "unit": "wave64 vector instructions per cycle per CU",
            "frac": None if d["valu_issue"] is None else d["valu_issue"] / VALU_ISSUE_PEAK,
            "traffic": d["traffic"],
            "l2_frac": d["l2_frac"], "hbm_frac": d["hbm_frac"],
            "valu_flop_frac": d["reference_algorithm_tflops"] / FP32_PEAK_TFLOPS,
            "reference_algorithm_bytes_per_launch": d["reference_algorithm_bytes_per_launch"],
            "reference_algorithm_tflops": d["reference_algorithm_tflops"],
            "avg_launch_ms": d["avg_launch_ms"], "launches_per_frame": d["launches_per_frame"],
            "valu_issue_frac_profiled": d["valu_issue_frac_profiled"], "lanes_per_valu_inst_profiled": d["lanes_per_valu_inst_profiled"],
            "l2_hit_profiled": d["l2_hit_profiled"], "pmc_profile": "profiles/" + PMC_DIGEST, "pmc_profile_status": digest_note,
            "note": "frac = vector instructions issued per cycle and CU over the CU's peak of 2 (four SIMD-32, two cycles per wave64 instruction); "
                    "mix_ceiling = what a CU can issue of this kernel's instruction mix (measured per opcode class, profiles/r02_valu_issue_rates.txt); "
                    "instruction and request counts per launch come from the committed rocprofv3 PMC passes of this binary on this workload, the "
                    "duration from HIP events in this run; l2_frac = TCC_REQ x 128 B / duration / 34.5 TB/s (an upper bound: a request may "
                    "carry 64 B); traffic = (2 x FETCH_SIZE + WRITE_SIZE) per launch = bytes that left L2 for the fabric, Infinity-Cache hits "
                    "included, so hbm_frac is an upper bound (the leaf-ordered triangle table of 8.9 MB, the nodes and the tags do not fit the 4 MB L2 of an XCD "
                    "and is served by the Infinity Cache); neither HBM nor MFMA bounds this path",
        }
        out["kernels"] = table
        out["algorithmic_frame"] = {k: v * scale * tile_world for k, v in cn.algorithmic(None).items()}
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()
    comm.close()


if __name__ == "__main__":
    main()
