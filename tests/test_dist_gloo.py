"""N > 1 path on CPU: two gloo processes each render their row-block shard of the
frame (with the oracle standing in for the GPU renderer), pack it in the
documented shard layout, all-gather, unpack -- the result must equal the
unsharded frame, and summed counters must equal the unsharded counters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    import cpuvoxelraycaster_amd.dist as vd
    import cpuvoxelraycaster_amd.scenes as scenes
    depth = 7
    nodes = O.compile_terrain(depth, O.load_terrain_heights())
    top, side = O.load_textures()
    light = scenes.reference_light(depth)
    cam = O.make_camera(scenes.reference_camera_position(depth), O.make_rotation(0.0, -0.5))
    acc = None
    rays = steps = 0
    for s in range(2):
        p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=s,
                          row_block=vd.ROW_BLOCK, shard_index=rank, shard_count=world)
        _, acc, _, st = O.render_frame(nodes, depth, top, side, cam, p, accum=acc, threads=2)
        rays += st.rays
        steps += st.sum_complexity
    img = O.samples_to_image(acc)
    mine = torch.from_numpy(vd.pack_shard_np(img, vd.ROW_BLOCK, rank, world))
    assert mine.numel() == vd.shard_bytes(W, H, vd.ROW_BLOCK, world)
    gathered = vd.all_gather_frame(mine, world)
    full = vd.unpack_shards_np(gathered.numpy(), W, H, vd.ROW_BLOCK, world)
    # the two exchange forms bench.py offers (--gather all / root), synchronous and asynchronous
    for mode in ("all", "root"):
        for async_op in (False, True):
            buf = torch.zeros(world * mine.numel(), dtype=torch.uint8)
            work = vd.exchange_shards(mine, buf, world, rank, mode, async_op=async_op)
            if async_op:
                work.wait()
            if mode == "all" or rank == 0:
                assert torch.equal(buf, gathered), (mode, async_op)
            else:
                assert not buf.any()                      # a gather writes on the presenting rank only
    tot = vd.reduce_stats([rays, steps], "cpu")
    if rank == 0:
        q.put((full, tot))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(2, 200, 100),      # 13 row blocks: uneven split over 2 and over 3 ranks, last block partial
                                       (3, 200, 100),
                                       (4, 48, 1080),      # ... on 4 ranks: 135 row blocks = 34 + 34 + 34 + 33 (33.75 each: the uneven case between 3 and 8)
                                       (8, 48, 1080)])     # BASELINE configs 3/4's height on 8 ranks: 135 row blocks = seven ranks own 17, rank 7 owns 16
def test_sharded_frame_over_gloo_ranks_matches_unsharded(built, heights, textures, world, W, H):
    import oracle_lib as O
    import cpuvoxelraycaster_amd as vrc
    import cpuvoxelraycaster_amd.dist as vd
    if world == 4:
        assert [len(vd.owned_rows(H, vd.ROW_BLOCK, r, world)) // vd.ROW_BLOCK for r in range(4)] == [34, 34, 34, 33]
    if world == 8:
        assert [len(vd.owned_rows(H, vd.ROW_BLOCK, r, world)) // vd.ROW_BLOCK for r in range(8)] == [17] * 7 + [16]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, tot = q.get(timeout=480)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    depth = 7
    nodes = O.compile_terrain(depth, heights)
    top, side = textures
    cam = O.make_camera(vrc.reference_camera_position(depth), O.make_rotation(0.0, -0.5))
    acc = None
    rays = steps = 0
    for s in range(2):
        p = O.make_params(W, H, vrc.reference_light(depth), use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, st = O.render_frame(nodes, depth, top, side, cam, p, accum=acc, threads=4)
        rays += st.rays
        steps += st.sum_complexity
    assert np.array_equal(full, O.samples_to_image(acc))
    assert tot == [rays, steps]


def test_shard_layout_roundtrip():
    import cpuvoxelraycaster_amd.dist as vd
    rng = np.random.default_rng(0)
    for (W, H, world) in [(17, 100, 3), (64, 64, 8), (5, 7, 2), (33, 1080, 8)]:
        img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        shards = [vd.pack_shard_np(img, vd.ROW_BLOCK, r, world) for r in range(world)]
        assert all(s.size == vd.shard_bytes(W, H, vd.ROW_BLOCK, world) for s in shards)
        back = vd.unpack_shards_np(np.concatenate(shards), W, H, vd.ROW_BLOCK, world)
        assert np.array_equal(back, img)
        rows = sorted(sum((vd.owned_rows(H, vd.ROW_BLOCK, r, world) for r in range(world)), []))
        assert rows == list(range(H))
