"""Fixed per-frame cost of the multi-GPU step on a tiny frame (64x32): host enqueue time and GPU launch latencies."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cpuvoxelraycaster_amd as vrc
depth = 7
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
rc = vrc.RayCaster(svo, (64, 32))
rc.setLightPosition(vrc.reference_light(depth))
rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
cam = vrc.reference_camera(depth)
stream = torch.cuda.current_stream().cuda_stream
L = vrc.capi.load()
buf = torch.empty(L.vrc_shard_bytes(64, 32, 8, 2), dtype=torch.uint8, device="cuda")
full = torch.empty(64*32*4, dtype=torch.uint8, device="cuda")
g = torch.empty(2*buf.numel(), dtype=torch.uint8, device="cuda")
def step_separate():
    rc.resetSamples(stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc.renderFrame(cam, spp=4, stream=stream, row_block=8, shard_index=0, shard_count=2)
    e1.record()
    rc.samples_to_image(stream)
    rc.packShard(8, 0, 2, buf.data_ptr(), stream)
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(g.data_ptr()), 64, 32, 8, 2, vrc.capi.ptr(full.data_ptr()), vrc.capi.ptr(stream)))


def step_fused():
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc.renderFrame(cam, spp=4, stream=stream, row_block=8, shard_index=0, shard_count=2)
    e1.record()
    rc.resolveShard(8, 0, 2, buf.data_ptr(), reset=True, stream=stream)
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(g.data_ptr()), 64, 32, 8, 2, vrc.capi.ptr(full.data_ptr()), vrc.capi.ptr(stream)))


for name, step in (("separate calls (reset, render, resolve, pack, unpack)", step_separate), ("fused (render, resolve_shard, unpack)", step_fused)):
    for _ in range(50): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue per step {1e6*(t1-t0)/2000:.1f} us; incl. drain {1e6*(t2-t0)/2000:.1f} us")


def step_fused_noev():
    rc.renderFrame(cam, spp=4, stream=stream, row_block=8, shard_index=0, shard_count=2)
    rc.resolveShard(8, 0, 2, buf.data_ptr(), reset=True, stream=stream)
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(g.data_ptr()), 64, 32, 8, 2, vrc.capi.ptr(full.data_ptr()), vrc.capi.ptr(stream)))


def step_render_only():
    rc.renderFrame(cam, spp=4, stream=stream, row_block=8, shard_index=0, shard_count=2)


for name, step in (("fused, no events", step_fused_noev), ("render only", step_render_only)):
    for _ in range(50): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue per step {1e6*(t1-t0)/2000:.1f} us; incl. drain {1e6*(t2-t0)/2000:.1f} us")
