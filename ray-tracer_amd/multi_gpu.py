"""One process per GPU: tile-sharded frames over torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards by image tiles with the scene replicated (every pixel is independent: the reference's
fragment shader reads only uniforms and textures, js/8-kd_tree.js:1060-1069). Two exchange steps exist:
  1. the log-average luminance is a whole-image statistic (js/8-kd_tree.js:2559-2568): each rank reduces
     its own tiles to (sum log(EPS+L), pixel count) and the two numbers are all-reduced (SUM);
  2. the finished RGBA8 tiles are gathered to rank 0, which scatters them into the full frame and runs the
     present pass - or, with SUPERSAMPLING 1 (where the present shader reads only its own pixel), every rank runs
     the present pass over its own tiles first and rank 0 only scatters finished canvas tiles (present="shards").
The same FrameJob drives a CPU stand-in over gloo in the tests.
"""
import ctypes as C

import numpy as np


class Comm:
    def __init__(self, rank, world, local_rank=0, backend="nccl", device=None, single_rank_group=False):
        """backend "nccl": collectives run on GPU tensors over RCCL. backend "gloo" with a cuda `device` is the
        rehearsal mode (several ranks sharing one GPU, which RCCL refuses): tensors are staged through the host."""
        self.rank, self.world, self.local = rank, world, local_rank
        self.dist = None
        self.side = None
        self.device = device or "cpu"
        self.backend = backend
        if backend == "nccl":
            self.device = device or "cuda:%d" % local_rank
        if world > 1 or single_rank_group:   # a one-rank group runs every collective call (tests)
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=rank, world_size=world)
            self.dist = dist
            # host-side flags travel over their own gloo group, so that they never queue behind device collectives
            self.side = dist.new_group(backend="gloo") if backend == "nccl" else None

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def _reduce(self, value, dtype, op):
        if not self.dist:
            return value
        import torch
        t = torch.tensor([value], dtype=dtype, device=self._wire())
        self.dist.all_reduce(t, op=op)
        return t.item()

    def _wire(self):
        return self.device if self.backend == "nccl" else "cpu"

    def max_float(self, v):
        import torch.distributed as dist
        import torch
        return self._reduce(float(v), torch.float64, dist.ReduceOp.MAX) if self.dist else v

    def sum_int(self, v):
        import torch.distributed as dist
        import torch
        return int(self._reduce(int(v), torch.int64, dist.ReduceOp.SUM)) if self.dist else v

    def any_flag(self, flag):
        """True on every rank when `flag` is true on any rank (host-side, no device work)."""
        if not self.dist:
            return bool(flag)
        import torch
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if self.side is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.side)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(t.item())

    def allreduce_tone(self, sum_log, count):
        """(sum of log luminance, pixel count) summed over ranks; float64, so the count is exact to 2^53."""
        if not self.dist:
            return sum_log, count
        import torch
        t = torch.tensor([sum_log, float(count)], dtype=torch.float64, device=self._wire())
        self.dist.all_reduce(t)
        s, n = t.tolist()
        return s, int(n)

    def gather_bytes(self, shard, stride):
        """Gathers every rank's `stride`-byte uint8 tensor to rank 0; returns one [world*stride] tensor there."""
        import torch
        if not self.dist:
            return shard
        out = None
        wire = shard if self.backend == "nccl" else shard.cpu()
        if self.rank == 0:
            out = torch.empty(self.world * stride, dtype=torch.uint8, device=wire.device)
            parts = list(out.split(stride))
        else:
            parts = None
        self.dist.gather(wire, gather_list=parts, dst=0)
        if out is not None and out.device != shard.device:
            out = out.to(shard.device)
        return out

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()


class _DevView:
    """Zero-copy torch view of device memory owned by the HIP library."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class HipShard:
    """Adapter: one RayTracerHIP context = one shard of the frame."""

    def __init__(self, rt, device):
        import torch
        from . import abi
        self.rt, self.torch, self.abi = rt, torch, abi
        self.device = device
        p, n = C.c_void_p(), C.c_size_t()
        rt._ck(rt.L.rt_device_shard_rgba8(rt.ctx, C.byref(p), C.byref(n)), "rt_device_shard_rgba8")
        # a shard without tiles (more ranks than tiles) has nothing to copy: no view over zero bytes
        self.compact = torch.as_tensor(_DevView(p.value, n.value), device=device) if n.value else None
        # the library numbers the 8x8 tiles over whole blocks of 16 x 16 tiles (rt_kernels.h: tile_coords)
        g_tiles = (((rt.W + 7) // 8 + 15) // 16) * (((rt.H + 7) // 8 + 15) // 16) * 256
        sc = max(rt.cfg.shard_count, 1)
        self.stride = ((g_tiles + sc - 1) // sc) * 256
        self.staging = torch.zeros(self.stride, dtype=torch.uint8, device=device)
        # the same tiles after the present pass (rt_present_shard), supersampling 1 only
        self.can_present_shard = rt.cfg.supersampling == 1
        rt._ck(rt.L.rt_device_shard_canvas(rt.ctx, C.byref(p), C.byref(n)), "rt_device_shard_canvas")
        self.compact_canvas = torch.as_tensor(_DevView(p.value, n.value), device=device) if n.value else None

    def trace(self):
        self.rt.trace()

    def tone_partial(self):
        return self.rt.tonePartial()

    def shard_bytes(self):
        if self.compact is not None:
            self.staging[:self.compact.numel()].copy_(self.compact)
        return self.staging, self.stride

    def assemble(self, gathered, stride, world):
        self.torch.cuda.synchronize()
        self.rt._ck(self.rt.L.rt_assemble_shards(self.rt.ctx, C.c_void_p(gathered.data_ptr()), C.c_size_t(stride), world),
                    "rt_assemble_shards")

    def present_shard(self):
        self.rt.presentShard()

    def shard_canvas_bytes(self):
        if self.compact_canvas is not None:
            self.staging[:self.compact_canvas.numel()].copy_(self.compact_canvas)
        return self.staging, self.stride

    def assemble_canvas(self, gathered, stride, world):
        self.torch.cuda.synchronize()
        self.rt._ck(self.rt.L.rt_assemble_shards_canvas(self.rt.ctx, C.c_void_p(gathered.data_ptr()), C.c_size_t(stride), world),
                    "rt_assemble_shards_canvas")

    def tone_finish(self, s, n):
        return self.rt.toneFinish(s, n)

    def present(self):
        self.rt._ck(self.rt.L.rt_present(self.rt.ctx, None), "rt_present")


class FrameJob:
    """One frame across all ranks: trace -> luminance all-reduce -> tile gather -> assemble -> present."""

    def __init__(self, rt_or_shard, comm, present="auto"):
        """present: "root" = rank 0 assembles the render target and presents it; "shards" = every rank presents its own
        tiles and rank 0 assembles the canvas (rank 0 then holds the finished canvas, but only its own tiles of the
        render target); "auto" = "shards" where the shard can (supersampling 1), else "root"."""
        self.comm = comm
        self.shard = rt_or_shard if hasattr(rt_or_shard, "tone_partial") else HipShard(rt_or_shard, comm.device)
        able = bool(getattr(self.shard, "can_present_shard", False))
        assert present in ("auto", "root", "shards") and (present != "shards" or able)
        self.by_shards = able if present == "auto" else present == "shards"
        self.Lbar = self.sf = None

    def step(self):
        sh, comm = self.shard, self.comm
        sh.trace()
        s, n = sh.tone_partial()
        if comm.world > 1 and self.by_shards:
            s, n = comm.allreduce_tone(s, n)
            self.Lbar, self.sf = sh.tone_finish(s, n)     # every rank: the tone values of the whole image
            sh.present_shard()
            buf, stride = sh.shard_canvas_bytes()
            gathered = comm.gather_bytes(buf, stride)
            if comm.rank == 0:
                sh.assemble_canvas(gathered, stride, comm.world)
            return self.Lbar, self.sf
        if comm.world > 1:
            s, n = comm.allreduce_tone(s, n)
            buf, stride = sh.shard_bytes()
            gathered = comm.gather_bytes(buf, stride)
            if comm.rank == 0:
                sh.assemble(gathered, stride, comm.world)
        self.Lbar, self.sf = sh.tone_finish(s, n)
        if comm.rank == 0:
            sh.present()
        return self.Lbar, self.sf


class PipelinedFrameJob:
    """The same frame as FrameJob, queued without host synchronisation and with several frames in flight.

    Every context (one per frame in flight) owns a HIP stream; the trace, the luminance reduction, the RCCL all-reduce
    of three doubles (in place, in the library's device memory: the two luminance sums and a flag the trace raises on
    the device when it ran out of buffer space), the tile gather, the assemble, the tone finish and the present pass of a
    frame are all queued on that stream, so the device orders them and the host only waits - when it reuses the context
    some frames later - for the trace's buffer check (rt_trace_end) and for its copy of the reduced flag: every rank
    reads the same number there, so all ranks repeat the exchange of a frame that any rank had to trace again, and no
    host-side collective is on the per-frame path. While the exchange of frame i runs, the kernels of frame i+1
    (another context, another stream) fill the GPU. backend "nccl": collectives on device memory over RCCL; backend
    "gloo": rehearsal, the three doubles and the tiles are staged through the host (several ranks may share a GPU)."""

    def __init__(self, rts, comm, present="auto", rehearse=None):
        """rehearse = (role, n_shards), single rank only: the contexts are shards of an n_shards-way split and every
        frame does the device work of one rank of such a job without the collectives - "worker": trace, sums, present
        of the own tiles; "root": also the scatter of n_shards gathered canvas buffers (whatever they hold)."""
        import torch
        self.torch, self.comm = torch, comm
        self.staged = comm.dist is not None and comm.backend != "nccl"
        able = all(rt.cfg.supersampling == 1 for rt in rts)
        assert present in ("auto", "root", "shards") and (present != "shards" or able)
        self.by_shards = (able if present == "auto" else present == "shards") and comm.dist is not None   # see FrameJob
        self.rehearse = rehearse if (rehearse and comm.dist is None and able) else None
        self.slots = []
        for rt in rts:
            sh = HipShard(rt, comm.device)
            st = torch.cuda.ExternalStream(rt.stream(), device=comm.device)
            gathered = (torch.empty(comm.world * sh.stride, dtype=torch.uint8, device=comm.device)
                        if comm.rank == 0 and comm.dist is not None else None)
            if self.rehearse and self.rehearse[0] == "root":
                gathered = torch.zeros(self.rehearse[1] * sh.stride, dtype=torch.uint8, device=comm.device)
            self.slots.append({"rt": rt, "shard": sh, "stream": st, "gathered": gathered, "busy": False, "pair": None,
                               "pair_host": torch.zeros(3, dtype=torch.float64).pin_memory(), "ev": torch.cuda.Event()})
        self.i = 0
        self.redone = 0
        self.finish_s = 0.0    # host time spent completing frames (waiting in rt_trace_end for the device + the reduced flag), and how many
        self.flag_s = 0.0      # of that: waiting for and reading the all-reduced overflow flag (after rt_trace_end returned)
        self.finished = 0
        self.joined = None     # ranks that took part in the first device collective
        self.rays = 0          # rays of all completed frames (this rank's shard)
        self.shadow_rays = 0

    def _post(self, slot):
        """Everything behind the trace of one frame, queued on the slot's stream."""
        rt, sh, comm, torch = slot["rt"], slot["shard"], self.comm, self.torch
        dptr = rt.tonePartialDevice()
        if slot["pair"] is None:
            slot["pair"] = torch.as_tensor(_DevView64(dptr, 3), device=comm.device)
        if self.rehearse:
            rt.toneFinishDevice(dptr)
            rt.presentShard()
            buf, stride = sh.shard_canvas_bytes()
            if self.rehearse[0] == "root":
                rt._ck(rt.L.rt_assemble_shards_canvas(rt.ctx, C.c_void_p(slot["gathered"].data_ptr()), C.c_size_t(stride), self.rehearse[1]),
                       "rt_assemble_shards_canvas")
            return
        if self.by_shards:
            self._allreduce(slot)
            rt.toneFinishDevice(dptr)
            rt.presentShard()
            buf, stride = sh.shard_canvas_bytes()
            self._gather(slot, buf, stride)
            if comm.rank == 0:
                rt._ck(rt.L.rt_assemble_shards_canvas(rt.ctx, C.c_void_p(slot["gathered"].data_ptr()), C.c_size_t(stride), comm.world),
                       "rt_assemble_shards_canvas")
            return
        if comm.dist is not None:
            self._allreduce(slot)
            buf, stride = sh.shard_bytes()
            self._gather(slot, buf, stride)
            if comm.rank == 0:
                rt._ck(rt.L.rt_assemble_shards(rt.ctx, C.c_void_p(slot["gathered"].data_ptr()), C.c_size_t(stride), comm.world),
                       "rt_assemble_shards")
        rt.toneFinishDevice(dptr)
        if comm.rank == 0:
            rt.presentQueued()

    def _allreduce(self, slot):
        """{sum of log luminance, pixel count, overflow flag} summed over the ranks in place, on the slot's stream; a
        copy of the result goes to page-locked host memory behind it (read by _finish when the context is reused)."""
        comm, pair = self.comm, slot["pair"]
        if self.staged:
            t = pair.cpu()                      # (waits for the slot's stream)
            comm.dist.all_reduce(t)
            pair.copy_(t)
        else:
            comm.dist.all_reduce(pair)
        if self.joined is None:
            self.joined = comm.dist.get_world_size()
        slot["pair_host"].copy_(pair, non_blocking=True)
        slot["ev"].record()

    def _gather(self, slot, buf, stride):
        comm = self.comm
        if self.staged:
            g = comm.gather_bytes(buf, stride)
            if comm.rank == 0:
                slot["gathered"].copy_(g)
            return
        parts = list(slot["gathered"].split(stride)) if comm.rank == 0 else None
        comm.dist.gather(buf, gather_list=parts, dst=0)

    def _finish(self, slot):
        if not slot["busy"]:
            return
        slot["busy"] = False
        import time
        t0 = time.perf_counter()
        # every rank must take the same decision: a rank whose buffers overflowed traced again, and the sums and
        # tiles it sent were void, so all ranks repeat the exchange of that frame. The decision comes out of the
        # exchange itself: the third all-reduced number counts the ranks whose trace overflowed
        redo = slot["rt"].traceEnd()
        if self.comm.dist is not None:
            t1 = time.perf_counter()
            slot["ev"].synchronize()
            anyone = float(slot["pair_host"][2]) > 0.0
            assert anyone or not redo, "a trace that was repeated must have raised its flag in the exchange"
            redo = anyone
            self.flag_s += time.perf_counter() - t1
        self.finish_s += time.perf_counter() - t0
        self.finished += 1
        if redo:
            self.redone += 1
            with self.torch.cuda.stream(slot["stream"]):
                self._post(slot)
        st = slot["rt"].stats()
        self.rays += int(st.rays_total)
        self.shadow_rays += int(st.shadow_rays)

    def next_context(self):
        """The context the next step() will use, with its previous frame completed: the caller may now change its
        scene (lights, mesh, tree) for the coming frame."""
        slot = self.slots[self.i % len(self.slots)]
        self._finish(slot)
        return slot["rt"]

    def step(self):
        """Queues one frame on the next context, then completes the frame queued before it on the other one."""
        slot = self.slots[self.i % len(self.slots)]
        self._finish(slot)              # the context is reused: its previous frame must be checked first
        with self.torch.cuda.stream(slot["stream"]):
            slot["rt"].traceBegin()
            self._post(slot)
        slot["busy"] = True
        self.i += 1
        if len(self.slots) == 1:
            self._finish(slot)

    def flush(self):
        for k in range(len(self.slots)):
            self._finish(self.slots[(self.i + k) % len(self.slots)])

    def close(self):
        """Completes what is queued and lets go of the torch objects that refer to the contexts' streams and memory
        (events, views); call it before the contexts are destroyed."""
        self.flush()
        self.torch.cuda.synchronize()
        self._release()

    def _release(self):
        # events first, contexts (whose streams the events were recorded on) last
        for slot in self.slots:
            for k in ("ev", "pair_host", "pair", "gathered", "stream", "shard"):
                slot[k] = None
        self.slots = []

    def __del__(self):
        try:
            self.torch.cuda.synchronize()
            self._release()
        except Exception:   # noqa: BLE001 (interpreter shutdown)
            pass

    def tone(self, k=-1):
        """Lbar, sf of the frame last completed on context k (waits for its stream)."""
        return self.slots[k]["rt"].toneGet()


class _DevView64:
    """Zero-copy torch view of `n` doubles of device memory owned by the HIP library."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
