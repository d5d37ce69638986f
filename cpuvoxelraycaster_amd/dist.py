"""Multi-GPU frame sharding: one process per GPU, the
read-only LSVO replicated, the frame sharded by interleaved ROW_BLOCK-row
blocks (block b belongs to rank b % world), RNG keyed by the global pixel index
so the image does not depend on the GPU count.  The only exchange step is one
collective over the resolved RGBA8 shards per frame (RCCL over xGMI on the GPU
box; the same code runs over gloo on CPU tensors in the tests): an all-gather
(every rank ends up with the frame) or a gather to rank 0, the presenting rank
(north_star / SURVEY 8e; 1/N of the bytes on the wire, one direct hop per peer).

Shard buffer layout (what vrc_pack_shard writes and vrc_unpack_shards reads):
    gathered[rank][slot][row_in_block][x]  RGBA8, slot k <-> row block rank + k*world,
    slots = ceil(ceil(H / ROW_BLOCK) / world), missing rows zero-padded.
"""
import numpy as np

ROW_BLOCK = 8


def shard_slots(height, row_block, world):
    nblocks = (height + row_block - 1) // row_block
    return (nblocks + world - 1) // world


def shard_bytes(width, height, row_block, world):
    return shard_slots(height, row_block, world) * row_block * width * 4


def owned_rows(height, row_block, rank, world):
    return [y for y in range(height) if (y // row_block) % world == rank]


def pack_shard_np(image, row_block, rank, world):
    """numpy restatement of k_pack_shard (CPU tests): image (H, W, 4) u8 -> flat shard."""
    H, W, _ = image.shape
    slots = shard_slots(H, row_block, world)
    out = np.zeros((slots, row_block, W, 4), np.uint8)
    for k in range(slots):
        y0 = (k * world + rank) * row_block
        if y0 < H:
            n = min(row_block, H - y0)
            out[k, :n] = image[y0:y0 + n]
    return out.reshape(-1)


def unpack_shards_np(gathered, width, height, row_block, world):
    """numpy restatement of k_unpack_shards: flat [rank][slot][row][x] -> (H, W, 4)."""
    slots = shard_slots(height, row_block, world)
    g = np.asarray(gathered, np.uint8).reshape(world, slots, row_block, width, 4)
    img = np.zeros((height, width, 4), np.uint8)
    for y in range(height):
        b = y // row_block
        img[y] = g[b % world, b // world, y % row_block]
    return img


def all_gather_frame(mine, world):
    """mine: this rank's packed shard (torch uint8 tensor, CPU or GPU).  Returns the
    concatenation over ranks ([rank][slot]...) on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return mine
    gathered = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    return gathered


def exchange_shards(mine, gathered, world, rank, mode="all", async_op=False):
    """The frame's one exchange step.  mine: this rank's packed shard; gathered: world * mine.numel() bytes laid out
    [rank][slot]...  mode "all": all-gather into every rank's `gathered`; mode "root": gather to rank 0 -- only rank
    0's `gathered` is written (the other ranks may pass None).  Returns the work handle when async_op."""
    import torch.distributed as dist
    if mode == "root":
        views = list(gathered.view(world, mine.numel()).unbind(0)) if rank == 0 else None
        return dist.gather(mine, views, dst=0, async_op=async_op)
    if mode != "all":
        raise ValueError(f"unknown exchange mode {mode!r}")
    return dist.all_gather_into_tensor(gathered, mine, async_op=async_op)


class PeerFrames:
    """The third exchange form (SURVEY 8e "direct peer writes"; bench.py --gather peer): rank 0, the presenting rank, owns the
    framebuffers -- those of its `renderers` --, exports them (vrc_ipc_export_image), and every other rank's frame kernel
    writes its rows of frame i straight into rank 0's framebuffer i % F (full-frame layout: no pack, no collective, no unpack).

    Ordering is by frame flags in a shared-memory segment (include/vrc.h: vrc_ipc_flags_open), written and waited for in
    stream order: written[slot][k] = 1 + the last frame whose rows rank k has put into framebuffer `slot` (rank 0's stream
    waits for it before the frame is read), consumed[slot] = 1 + the last frame of that framebuffer rank 0 has consumed (the
    writers' streams wait for it before the next frame goes in).  A wait names a value, so the hosts exchange nothing per
    frame; the image handles travel once, at set-up, through all_gather_object on `group` (any backend).

    A stream-ordered wait has no timeout: a stream that waits for a rank that died waits for ever, and so does every
    synchronize on it.  So the hosts never synchronise these streams blindly: `wait(streams)` is the watchdog
    (include/vrc.h: vrc_ipc_stream_wait) -- it polls the streams, the peers' process ids and a deadline, and when a peer is gone
    it releases every wait on the flags in every process, lets the streams drain and raises."""

    def __init__(self, renderers, device, rank, world, group, name):
        import ctypes as C
        import os
        import torch.distributed as dist
        from . import capi
        L = capi.load()
        self.rank, self.world, self.device = rank, world, device
        self.renderers = renderers
        F = self.F = len(renderers)
        self._flags = C.c_void_p()
        n = F * world + F                                      # written[slot][rank] then consumed[slot]
        # the name carries the creator's pid: two runs on one node never share a segment (the creator tells the others)
        name = "%s_%d" % (name, os.getpid()) if rank == 0 else None
        if rank == 0:
            capi.check(L.vrc_ipc_flags_open(name.encode(), n, device, 1, C.byref(self._flags)))
        everyone = [None] * world
        dist.all_gather_object(everyone, ([r.exportImage() for r in renderers], name, os.getpid()) if rank == 0 else (None, None, os.getpid()), group=group)
        name = everyone[0][1]
        self.pids = [e[2] for k, e in enumerate(everyone) if k != rank]      # the peers this rank's streams may wait for
        everyone = [e[0] for e in everyone]
        self.targets = []
        # A rank that cannot map the segment or a framebuffer (IPC mode mismatch, handle refused) must not leave the others in a
        # barrier until the process-group timeout: every rank reports how its set-up went, and every rank raises the first
        # failure by name (round-5 advice).  The name is unlinked only when all of them have the segment mapped.
        problem = None
        if rank != 0:
            try:
                capi.check(L.vrc_ipc_flags_open(name.encode(), n, device, 0, C.byref(self._flags)))
                for s, r in enumerate(renderers):
                    buf = (C.c_ubyte * 64).from_buffer_copy(everyone[0][s])
                    tgt = C.c_void_p()
                    capi.check(L.vrc_ipc_open_image(device, C.byref(buf), C.byref(tgt)))
                    self.targets.append(tgt)
                    r.setImageTarget(tgt)
            except Exception as e:                              # noqa: BLE001  (reported below, on every rank)
                problem = f"rank {rank}: {e}"
        outcome = [None] * world
        dist.all_gather_object(outcome, problem, group=group)
        # every rank has the segment mapped now (or has said why not): its name can go (the mappings stay valid), so that a run
        # that is killed later -- launch_ranks ends stragglers by SIGKILL -- leaves nothing behind in /dev/shm
        if rank == 0:
            capi.check(L.vrc_ipc_flags_unlink(self._flags))
        failed = [o for o in outcome if o]
        if failed:
            self.close()
            raise capi.VrcError("direct peer writes could not be set up: " + "; ".join(failed))

    def _written(self, slot, k):
        return slot * self.world + k

    def _consumed(self, slot):
        return self.F * self.world + slot

    def before_render(self, frame, stream):
        """a writer must not touch framebuffer frame % F before rank 0 has consumed frame - F, which it holds"""
        from . import capi
        if self.rank != 0 and frame >= self.F:
            capi.check(capi.load().vrc_stream_wait_flag(self._flags, self._consumed(frame % self.F), frame - self.F + 1, capi.ptr(stream)))

    def after_render(self, frame, stream):
        """writer: my rows are in.  Rank 0: wait for every writer's rows -- the frame is complete on `stream` after this call"""
        from . import capi
        L, slot = capi.load(), frame % self.F
        if self.rank != 0:
            capi.check(L.vrc_stream_write_flag(self._flags, self._written(slot, self.rank), frame + 1, capi.ptr(stream)))
        else:
            for k in range(1, self.world):
                capi.check(L.vrc_stream_wait_flag(self._flags, self._written(slot, k), frame + 1, capi.ptr(stream)))

    def consumed_on(self, frame, stream):
        """rank 0: everything enqueued on `stream` so far has read the frame; writers may reuse its framebuffer after that"""
        from . import capi
        if self.rank == 0:
            capi.check(capi.load().vrc_stream_write_flag(self._flags, self._consumed(frame % self.F), frame + 1, capi.ptr(stream)))

    def wait(self, streams, timeout_s=120.0):
        """Instead of synchronising `streams` (raw stream pointers): returns when everything enqueued on them has completed;
        raises VrcError (VRC_ERR_PEER) when a peer process is gone or no flag of the segment has moved for timeout_s (an
        inactivity limit: a long queue that keeps completing frames is never declared dead) -- every wait on the flags is then
        released in every process, so nobody hangs; the frames in flight are lost."""
        import ctypes as C
        from . import capi
        pids = (C.c_int32 * len(self.pids))(*self.pids)
        L, failed = capi.load(), None
        for s in streams:
            # after a failure the remaining streams are still drained (each call holds the release while its stream empties):
            # a process must not leave with a wait pending on the device
            rc = L.vrc_ipc_stream_wait(self._flags, capi.ptr(s), pids, len(self.pids), int(timeout_s * 1000))
            if rc != 0 and failed is None:
                failed = capi.VrcError(f"vrc error {rc}: {L.vrc_last_error().decode()}")
        if failed is not None:
            raise failed

    def close(self):
        """after every rank has synchronised its streams (the caller's barrier)"""
        from . import capi
        for r in self.renderers:
            if self.rank != 0:
                r.setImageTarget(None)
        for t in self.targets:
            capi.load().vrc_ipc_close_image(self.device, t)
        self.targets = []
        if self._flags:
            capi.load().vrc_ipc_flags_close(self._flags)
            self._flags = None


def reduce_stats(values, device):
    """Sum per-rank counters (rays, loop iterations, pixels) over ranks."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
