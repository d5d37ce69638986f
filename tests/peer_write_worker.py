"""One rank of the direct-peer-write test (tests/test_gpu_peer_write.py): WORLD_SIZE processes on GPU 0, gloo for the host
messages.  Every rank renders its row blocks of frames 0..N-1 with IN_FLIGHT renderers / streams taking turns (whole-spp work
units, fused resolve, exactly like bench.py); ranks > 0 write their rows straight into rank 0's framebuffers
(cpuvoxelraycaster_amd.dist.PeerFrames).  Rank 0 reads every completed frame back, compares it with the unsharded render of
the same samples and prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def dbg(*a):
    if os.environ.get("PW_DEBUG"):
        print(f"[rank {os.environ.get('RANK')}]", *a, file=sys.stderr, flush=True)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import cpuvoxelraycaster_amd as vrc
    import cpuvoxelraycaster_amd.dist as vd
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    depth, W, H, spp = int(os.environ.get("PW_DEPTH", "8")), int(os.environ.get("PW_W", "1280")), int(os.environ.get("PW_H", "720")), 4
    frames, in_flight = int(os.environ.get("PW_FRAMES", "9")), 3
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    textures = vrc.load_textures()
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, device=0, textures=textures)
    cam = vrc.reference_camera(depth, pitch=-0.5)

    def make():
        r = vrc.RayCaster(svo, (W, H))
        r.setLightPosition(vrc.reference_light(depth))
        r.use_gi, r.use_samples, r.shadow_samples = True, True, 1
        r.setSampleChunk(spp)
        return r

    rcs = [make() for _ in range(in_flight)]
    streams = [torch.cuda.Stream() for _ in range(in_flight)]
    dbg("renderers made")
    if rank == int(os.environ.get("PW_BAD_SETUP_RANK", "-1")):
        # the set-up failure test: this rank cannot open rank 0's framebuffer (what an IPC mode mismatch looks like)
        vrc.capi.load().vrc_ipc_open_image = lambda *a: vrc.capi.load().vrc_ipc_close_image(-1, 1)   # an error code + vrc_last_error
    try:
        peers = vd.PeerFrames(rcs, 0, rank, world, dist.group.WORLD, "/vrc_peer_test_%s" % os.environ["MASTER_PORT"])
    except vrc.capi.VrcError as e:
        print(json.dumps({"rank": rank, "error": str(e)}), flush=True)
        os._exit(8)
    dbg("peer frames set up")

    class _Alias:                       # a renderer's framebuffer as a torch tensor (no copy)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    L = vrc.capi.load()
    fb = [torch.as_tensor(_Alias(L.vrc_image_device_ptr(r._h), H * W * 4), device="cuda") for r in rcs]
    keep = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(frames)] if rank == 0 else None
    dist.barrier()
    t0 = time.perf_counter()
    try:
        for i in range(frames):
            rc, ts = rcs[i % in_flight], streams[i % in_flight]
            with torch.cuda.stream(ts):
                dbg("frame", i, "before_render")
                if rank == int(os.environ.get("PW_DIE_RANK", "-1")) and i == int(os.environ.get("PW_DIE_AT_FRAME", "-1")):
                    os._exit(5)             # the failure test: this rank vanishes with frames in flight and flags unwritten
                peers.before_render(i, ts.cuda_stream)
                rc.frame_index = i * spp
                rc.renderFrameResolved(cam, spp=spp, stream=ts.cuda_stream, row_block=vd.ROW_BLOCK, shard_index=rank, shard_count=world)
                dbg("frame", i, "enqueued")
                peers.after_render(i, ts.cuda_stream)
                dbg("frame", i, "after_render done")
                if rank == 0:
                    # "presenting" the frame = copying it out on the frame's stream, behind the waits for the writers' rows
                    keep[i].copy_(fb[i % in_flight], non_blocking=True)
                peers.consumed_on(i, ts.cuda_stream)
        dbg("all frames enqueued")
        peers.wait([ts.cuda_stream for ts in streams], timeout_s=float(os.environ.get("PW_TIMEOUT_S", "60")))   # never a blind synchronize
    except vrc.capi.VrcError as e:
        # a peer is gone (noticed by this rank's watchdog, or by another rank's while this one was still issuing frames):
        # drain what is queued with the release held, then leave at once, non-zero -- no collective clean-up with a dead peer
        try:
            peers.wait([ts.cuda_stream for ts in streams], timeout_s=10.0)
        except vrc.capi.VrcError:
            pass
        print(json.dumps({"rank": rank, "error": str(e)}), flush=True)
        os._exit(7)
    torch.cuda.synchronize()
    dbg("synchronized")
    dt = time.perf_counter() - t0
    dist.barrier()
    dbg("barrier 1 passed")
    out = {"rank": rank, "world": world, "frames": frames, "frames_in_flight": in_flight, "ms_per_frame": round(dt / frames * 1e3, 3),
           "size": [W, H], "depth": depth, "spp": spp}
    if rank == 0:
        ref = make()
        ref.setSampleChunk(0)
        equal = []
        for i in range(frames):
            ref.frame_index = i * spp
            ref.renderFrameResolved(cam, spp=spp)
            equal.append(bool(np.array_equal(keep[i].cpu().numpy().reshape(H, W, 4), ref.readImage())))
        out["composed_frame_equals_unsharded"] = equal
        out["kernel"] = rcs[0].lastKernel()
        dbg("verified", equal)
    dist.barrier()
    peers.close()
    dbg("peers closed")
    dist.barrier()
    dbg("barrier 2 passed")
    dist.destroy_process_group()
    dbg("group destroyed")
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
