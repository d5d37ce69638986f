#!/usr/bin/env python3
"""Diagnostics: build libvrc_hip with -DVRC_PROBE into gpurun_out/ and print when the waves of one k_render_sync
launch start and end (tail study).  Usage (GPU box): python tools/probe_waves.py [--config c3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--lpt", type=int, default=0, help="extra frames scheduled longest-unit-first")
    args = ap.parse_args()
    import torch
    import bench
    import cpuvoxelraycaster_amd as vrc
    cfg = bench.CONFIGS[args.config]
    depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = bool(cfg["use_gi"]), True, 1
    cam = vrc.reference_camera(depth, pitch=-0.5, aperture=cfg["aperture"])
    stream = torch.cuda.current_stream().cuda_stream
    L = vrc.capi.load()
    rc.setSampleChunk(args.chunk)
    probe = torch.zeros(W * H * 12, dtype=torch.int32, device="cuda")
    rc.setPrimaryCapture(probe.data_ptr())
    n_units = ((W + 7) // 8) * ((H + 7) // 8) * (spp // args.chunk if args.chunk else 1)
    pi = probe.view(torch.int32)
    for it in range(3 + args.lpt):
        if it >= 3:   # longest-processing-time-first order from the previous frame's measured unit durations
            dur = pi[(1 << 18):(1 << 18) + n_units].clone()
            order = torch.argsort(dur, descending=True).to(torch.int32)
            if it == 3:
                d = dur.cpu().numpy() / 100.0
                print("unit duration us: pct 0/10/50/90/99/100:", np.percentile(d, [0, 10, 50, 90, 99, 100]).round(1), "sum/5120 =", (d.sum() / 5120).round(1))
            if it > 3:
                print("corr(prev-frame duration, this order's duration):", float(torch.corrcoef(torch.stack([prev_dur.float(), dur.float()]))[0, 1]))
                rank_dur = dur[order_prev.long()].cpu().numpy() / 100.0
                print("duration by LPT rank (deciles of the hand-out order), mean us:", [round(float(x.mean()), 1) for x in np.array_split(rank_dur, 10)])
            prev_dur, order_prev = dur, order
            probe.zero_()
            pi[(1 << 19):(1 << 19) + n_units] = order
            pi[(1 << 19) - 1] = 1
        else:
            if it > 0:
                d_now = pi[(1 << 18):(1 << 18) + n_units].clone()
                if it > 1:
                    print("natural order: corr between consecutive frames:", float(torch.corrcoef(torch.stack([d_prev.float(), d_now.float()]))[0, 1]),
                          "by position deciles:", [round(float(x.mean()) / 100.0, 1) for x in np.array_split(d_now.cpu().numpy(), 10)])
                d_prev = d_now
            probe.zero_()
        rc.resetSamples(stream)
        rc.frame_index = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc.renderFrame(cam, spp=spp, stream=stream)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    p = probe.cpu().numpy().view(np.int64).reshape(-1, 8)[:16384]
    p = p[p[:, 1] != 0]
    t0 = p[:, 0].min()
    start = (p[:, 0] - t0) / 100.0      # us (100 MHz)
    end = (p[:, 1] - t0) / 100.0
    units = p[:, 2] & 0xffffffff
    last_unit = p[:, 2] >> 32
    last_dur = (p[:, 1] - p[:, 3]) / 100.0
    longest = p[:, 4] / 100.0
    print("last-unit duration us: pct 0/50/90/99/100:", np.percentile(last_dur, [0, 50, 90, 99, 100]).round(1))
    print("longest earlier unit per wave us: pct 0/50/90/99/100:", np.percentile(longest, [0, 50, 90, 99, 100]).round(1))
    du = pi[(1 << 18):(1 << 18) + n_units].cpu().numpy() / 100.0
    iters = pi[(3 << 18):(3 << 18) + n_units].cpu().numpy().astype(np.float64)
    print("wave-iterations per unit: pct 0/10/50/90/99/100:", np.percentile(iters, [0, 10, 50, 90, 99, 100]), "total", iters.sum())
    print("corr(duration, wave-iterations):", np.corrcoef(du, iters)[0, 1].round(3), " us per wave-iteration: pct 1/10/50/90/99:",
          np.percentile(du / np.maximum(iters, 1), [1, 10, 50, 90, 99]).round(3))
    print("us per wave-iteration by position 20-quantiles:", [round(float(a.sum() / b.sum()), 3) for a, b in zip(np.array_split(du, 20), np.array_split(iters, 20))])
    print("wave-iterations by position deciles:", [round(float(x.mean()), 1) for x in np.array_split(iters, 10)])
    late = np.argsort(end)[-10:]
    for i in late:
        print(f"   late wave: end {end[i]:.1f} last unit {last_unit[i]} started {(p[i, 3] - t0) / 100.0:.1f} dur {last_dur[i]:.1f}; units {units[i]}; iters {iters[last_unit[i]]:.0f}")
    print("last unit ids: pct 0/10/50/90/100", np.percentile(last_unit, [0, 10, 50, 90, 100]))
    print(f"event time {ms:.3f} ms; waves {len(p)}; span {end.max():.1f} us")
    print("wave start us: pct 0/50/90/99/100:", np.percentile(start, [0, 50, 90, 99, 100]).round(1))
    print("wave end   us: pct 0/1/10/25/50/75/90/100:", np.percentile(end, [0, 1, 10, 25, 50, 75, 90, 100]).round(1))
    print("units per wave: min/mean/max", units.min(), units.mean().round(2), units.max(), "hist", np.bincount(units.astype(int)))
    hw = p[:, 6]
    wave_id, simd_id, cu_id, se_id = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7
    xcc = p[:, 7] & 15
    print("units per wave by hardware wave slot:", {int(w): (int((wave_id == w).sum()), round(float(units[wave_id == w].mean()), 2), round(float(end[wave_id == w].mean()), 1)) for w in np.unique(wave_id)})
    print("units per wave by simd:", {int(w): round(float(units[simd_id == w].mean()), 2) for w in np.unique(simd_id)})
    print("units per wave by xcc:", {int(w): (int((xcc == w).sum()), round(float(units[xcc == w].mean()), 2), round(float(end[xcc == w].max()), 1)) for w in np.unique(xcc)})
    key = (xcc * 8 + se_id) * 16 + cu_id
    per_cu = np.bincount(key.astype(int))
    print("waves per CU: ", np.bincount(per_cu[per_cu > 0]))
    busy = (end - start).sum()
    print(f"wave-busy fraction of span: {busy / (len(p) * end.max()):.3f}")
    # how many waves still run at time t
    for t in np.linspace(end.max() * 0.7, end.max(), 13):
        print(f"  t={t:7.1f} us running {(end > t).sum():5d}")


if __name__ == "__main__":
    main()
