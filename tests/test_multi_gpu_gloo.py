"""The N > 1 frame path (FrameJob + Comm) over gloo with two CPU ranks.

Each rank runs the same FrameJob the GPU ranks run, with a stand-in shard that renders its interleaved 8x8 tiles
with the oracle and lays them out exactly like rt_device_shard_rgba8 does ([tile][64 pixels][RGBA], alpha 0 for
lanes outside the image). Rank 0 must end up with the full frame, and every rank with the full-image Lbar / sf.
"""
import math
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

# spawned ranks import this module without conftest.py: make the package importable first
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_ROOT, os.path.join(_ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import rt_loader  # noqa: E402

rt_loader.load()
from helpers import make_flat  # noqa: E402

W, H = 100, 60   # ragged on purpose: 13 x 8 tiles, the last column and row are partial


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleShard:
    def __init__(self, flat, rank, world):
        from oracle import oracle
        self.ora = oracle.OracleRenderer(flat)
        self.rank, self.world = rank, world
        # the library's tile order (rt_kernels.h: tile_coords): 8x8-pixel tiles numbered in blocks of 16 x 16 tiles, Z-order
        # inside a block, the blocks row by row; the numbering runs over whole blocks, so tiles beyond the image edge exist
        # in it and hold no pixels. Shard r owns tiles r, r + world, ... of that numbering.
        self.tx, self.ty = (W + 7) // 8, (H + 7) // 8
        self.bx, self.by = (self.tx + 15) // 16, (self.ty + 15) // 16
        self.n_tiles = self.bx * self.by * 256
        self.tiles = list(range(rank, self.n_tiles, world))
        self.stride = ((self.n_tiles + world - 1) // world) * 256
        self.full = None
        self.frame = None

    def trace(self):
        _, self.full, _, _ = self.ora.trace(threads=2, want_float=False, want_ids=False)

    def _tile_pixels(self, t):
        even = lambda w: sum(((w >> (2 * k)) & 1) << k for k in range(4))   # bits 0, 2, 4, 6 -> a 4-bit number
        block, w = t >> 8, t & 255
        x0, y0 = ((block % self.bx) * 16 + even(w)) * 8, ((block // self.bx) * 16 + even(w >> 1)) * 8
        for ly in range(8):
            for lx in range(8):
                yield ly * 8 + lx, x0 + lx, y0 + ly

    def tone_partial(self):
        s, n = 0.0, 0
        for t in self.tiles:
            for _, x, y in self._tile_pixels(t):
                if x < W and y < H:
                    r, g, b = (float(v) * 1000 / 256.0 for v in self.full[y, x, :3])
                    s += math.log(0.0001 + (0.27 * r + 0.67 * g + 0.07 * b))
                    n += 1
        return s, n

    def shard_bytes(self):
        buf = np.zeros(self.stride, np.uint8)
        for k, t in enumerate(self.tiles):
            for lane, x, y in self._tile_pixels(t):
                if x < W and y < H:
                    buf[k * 256 + lane * 4:k * 256 + lane * 4 + 4] = self.full[y, x]
        return torch.from_numpy(buf), self.stride

    def assemble(self, gathered, stride, world):
        g = gathered.numpy()
        self.frame = np.zeros((H, W, 4), np.uint8)
        for r in range(world):
            for k, t in enumerate(range(r, self.n_tiles, world)):
                for lane, x, y in self._tile_pixels(t):
                    px = g[r * stride + k * 256 + lane * 4:r * stride + k * 256 + lane * 4 + 4]
                    if x < W and y < H:
                        assert px[3] == 255
                        self.frame[y, x] = px
                    else:
                        assert px[3] == 0

    def tone_finish(self, s, n):
        Lbar = math.exp(s / n)
        return Lbar, ((1.219 + (500 * 0.5) ** 0.4) / (1.219 + Lbar ** 0.4)) ** 2.5

    def present(self):
        pass


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ray_tracer_amd import multi_gpu
    flat = make_flat("bun_zipper_res4.ply", W, H, depth=2, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.0, 0.0))
    comm = multi_gpu.Comm(rank, world, 0, backend="gloo")
    shard = OracleShard(flat, rank, world)
    job = multi_gpu.FrameJob(shard, comm)
    Lbar, sf = job.step()
    total_px = comm.sum_int(shard.tone_partial()[1])
    tmax = comm.max_float(float(rank))
    # a flag raised on one rank is seen by all (the pipelined job repeats a frame's exchange on every rank this way)
    assert comm.any_flag(rank == 1) is True and comm.any_flag(False) is False
    ok_frame = True
    if rank == 0:
        ok_frame = bool(np.array_equal(shard.frame, shard.full))
    wl, ws = shard.ora.tone(shard.full)
    out.put((rank, Lbar, sf, wl, ws, total_px, tmax, ok_frame))
    comm.close()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 3])
def test_frame_job_ranks_gloo(world):
    """world = 3 does not divide the 104 tiles: the ranks hold 35 / 35 / 34 tiles and the gather pads the last buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, Lbar, sf, wl, ws, total_px, tmax, ok_frame in results:
        assert total_px == W * H            # every pixel belongs to exactly one rank
        assert tmax == float(world - 1)
        assert ok_frame                      # rank 0 reassembled the whole frame from the gathered tiles
        assert abs(Lbar - wl) <= 1e-12 * wl and abs(sf - ws) <= 1e-12 * ws   # all-reduced sums = whole-image statistic
    assert all(r[1] == results[0][1] for r in results)    # every rank holds the same Lbar


# ---- the same job with the HIP shards: two ranks share GPU 0, tensors staged through gloo ------------------
import pytest  # noqa: E402


def hip_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import hashlib
    from ray_tracer_amd import abi, multi_gpu
    torch.cuda.set_device(0)
    flat = make_flat("bun_zipper_res4.ply", 324, 243, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9),
                     shard_index=rank, shard_count=world)
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    comm = multi_gpu.Comm(rank, world, 0, backend="gloo", device="cuda:0")
    def canvas_hash():
        p, n = abi.C.c_void_p(), abi.C.c_size_t()
        rt._ck(rt.L.rt_device_canvas(rt.ctx, abi.C.byref(p), abi.C.byref(n)), "rt_device_canvas")
        torch.cuda.synchronize()
        canvas = torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:0").cpu().numpy()
        return hashlib.sha256(canvas.tobytes()).hexdigest()

    # every rank presents its own tiles, the root assembles the canvas (what "auto" picks at supersampling 1) ...
    job = multi_gpu.FrameJob(rt, comm)
    assert job.by_shards
    for _ in range(2):
        Lbar, sf = job.step()
    res = {"rank": rank, "Lbar": Lbar, "sf": sf, "rays": int(rt.stats().rays_total)}
    if rank == 0:
        res["canvas_by_shards"] = canvas_hash()
    # ... and the root assembling the render target and presenting it
    job = multi_gpu.FrameJob(rt, comm, present="root")
    Lbar2, sf2 = job.step()
    assert (Lbar2, sf2) == (Lbar, sf)
    if rank == 0:
        res["rgba"] = hashlib.sha256(np.ascontiguousarray(rt.readRGBA8()).tobytes()).hexdigest()
        res["canvas"] = canvas_hash()
    out.put(res)
    comm.close()
    rt.close()


@pytest.mark.gpu
def test_cpu_stand_in_lays_its_shards_out_like_the_library():
    """The CPU ranks of the gloo tests (OracleShard) number the tiles themselves; the job logic they exercise means
    something only if that numbering is the library's: every rank's buffer must be, byte for byte, what
    rt_device_shard_rgba8 hands out for the same shard of the same frame (ragged frame, three shards)."""
    import ctypes as C
    from ray_tracer_amd import abi
    flat = make_flat("bun_zipper_res4.ply", W, H, depth=2, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.0, 0.0))
    hip = C.CDLL("libamdhip64.so")
    world = 3
    for r in range(world):
        cpu = OracleShard(flat, r, world)
        cpu.trace()
        want, stride = cpu.shard_bytes()
        flat.cfg.shard_index, flat.cfg.shard_count = r, world
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        rt.trace()
        p, nb = C.c_void_p(), C.c_size_t()
        assert rt.L.rt_device_shard_rgba8(rt.ctx, C.byref(p), C.byref(nb)) == 0
        got = np.zeros(nb.value, np.uint8)
        assert hip.hipMemcpy(C.c_void_p(got.ctypes.data), p, C.c_size_t(nb.value), 2) == 0
        assert nb.value == 256 * len(cpu.tiles) and nb.value <= stride
        assert np.array_equal(got, want.numpy()[:nb.value]), r
        s_cpu, n_cpu = cpu.tone_partial()
        s_gpu, n_gpu = rt.tonePartial()
        assert n_gpu == n_cpu and abs(s_gpu - s_cpu) <= 1e-10 * abs(s_cpu)
        rt.close()
    flat.cfg.shard_index, flat.cfg.shard_count = 0, 1


@pytest.mark.gpu
def test_two_hip_shards_equal_one_context():
    import hashlib
    from ray_tracer_amd import abi
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=hip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flat = make_flat("bun_zipper_res4.ply", 324, 243, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    rt.trace()
    Lbar, sf = rt.tone()
    canvas = rt.present()
    assert results[0]["rgba"] == hashlib.sha256(np.ascontiguousarray(rt.readRGBA8()).tobytes()).hexdigest()
    assert results[0]["canvas"] == hashlib.sha256(np.ascontiguousarray(canvas).tobytes()).hexdigest()
    assert results[0]["canvas_by_shards"] == results[0]["canvas"]
    assert results[0]["rays"] + results[1]["rays"] == rt.stats().rays_total
    for r in results.values():
        assert abs(r["Lbar"] - Lbar) <= 1e-10 * Lbar and np.float32(r["sf"]) == np.float32(sf)
    rt.close()


def pipelined_worker(rank, world, port, out):
    """The stream-ordered job over gloo (the exchange staged through the host), all ranks on GPU 0. The buffers of the
    LAST rank only are far too small for its first frames: its trace raises the overflow flag on the device, the flag is
    all-reduced with the luminance sums, and every rank - reading the same number from its own copy - repeats the
    exchange of exactly those frames."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import faulthandler
    import hashlib
    faulthandler.enable()
    from ray_tracer_amd import abi, multi_gpu
    torch.cuda.set_device(0)
    flat = make_flat("bun_zipper_res4.ply", 324, 243, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9),
                     shard_index=rank, shard_count=world)
    rts = []
    for _ in range(2):
        rt = abi.RayTracerHIP(flat.cfg)
        rt.setScene(flat)
        if rank == world - 1:
            rt.setOption("first_level_rays", 256)
        rts.append(rt)
    comm = multi_gpu.Comm(rank, world, 0, backend="gloo", device="cuda:0")
    job = multi_gpu.PipelinedFrameJob(rts, comm)
    assert job.by_shards and job.staged
    for _ in range(5):
        job.step()
    job.flush()
    torch.cuda.synchronize()
    res = {"rank": rank, "redone": job.redone, "joined": job.joined, "finished": job.finished, "tone": [rt.toneGet() for rt in rts]}
    if rank == 0:
        res["canvas"] = []
        for rt in rts:
            p, n = abi.C.c_void_p(), abi.C.c_size_t()
            rt._ck(rt.L.rt_device_canvas(rt.ctx, abi.C.byref(p), abi.C.byref(n)), "rt_device_canvas")
            canvas = torch.as_tensor(multi_gpu._DevView(p.value, n.value), device="cuda:0").cpu().numpy()
            res["canvas"].append(hashlib.sha256(canvas.tobytes()).hexdigest())
    out.put(res)
    job.close()
    comm.close()
    for rt in rts:
        rt.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_job_over_gloo_repeats_a_frame_that_one_rank_retraced(world):
    import hashlib
    from ray_tracer_amd import abi
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=pipelined_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r["rank"]: r for r in (q.get(timeout=240) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flat = make_flat("bun_zipper_res4.ply", 324, 243, depth=3, mesh_material=(0.7, 0.6, 0.7, 16, 0.3, 0.5, 0.9))
    rt = abi.RayTracerHIP(flat.cfg)
    rt.setScene(flat)
    rt.trace()
    Lbar, sf = rt.tone()
    want = hashlib.sha256(np.ascontiguousarray(rt.present()).tobytes()).hexdigest()
    assert results[0]["canvas"] == [want, want]
    for r in results.values():
        # both contexts of the last rank were too small on their first frame: every rank repeated those two exchanges
        assert r["redone"] == 2 and r["joined"] == world and r["finished"] == 5
        for gl, gs in r["tone"]:
            assert abs(gl - Lbar) <= 1e-10 * Lbar and np.float32(gs) == np.float32(sf)
    rt.close()


def _run_bench(args, timeout=600):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(_ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()   # the JSON line and nothing else (gloo's connection notices belong on stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_from_one_command():
    """`python bench.py --gpus N` with no launcher: the process starts N child ranks itself (it never touches the GPU),
    forwards rank 0's single JSON line and its exit status. --spawn-check stops after a gloo all-reduce of the rank
    numbers, so this runs without a GPU."""
    out = _run_bench(["--gpus", "3", "--spawn-check"], timeout=300)
    assert out == {"n_gpus": 3, "rank_sum": 6}


def test_bench_spawn_reports_a_failing_rank():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(_ROOT, "bench.py"), "--gpus", "2", "--spawn-check", "--no-such-flag"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_from_one_command():
    """The whole N = 2 bench path (tiles sharded, luminance all-reduce, tile gather, assemble, present) started by
    `python bench.py --gpus 2` alone, rehearsed with both ranks on device 0 over gloo."""
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--device", "0", "--no-cpu", "--model", "bun_zipper_res4.ply", "--teapot", "0",
                      "--width", "320", "--height", "240", "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    one = _run_bench(["--gpus", "1", "--no-cpu", "--model", "bun_zipper_res4.ply", "--teapot", "0",
                      "--width", "320", "--height", "240", "--steps", "3", "--warmup", "1"])
    assert one["config"]["rays_per_frame"] == out["config"]["rays_per_frame"]
    assert one["config"]["shadow_rays_per_frame"] == out["config"]["shadow_rays_per_frame"]
