#!/usr/bin/env python3
"""One table from the bench lines of a 1 / 2 / 4 / 8-GPU run (round-5 verdict, Next 2): where a scaling curve goes.

    python tools/scale_report.py n1.json n2.json n4.json n8.json        (files whose LAST line is bench.py's JSON line;
                                                                         a driver record with the line under "parsed" works too)

Per N: ms per frame, Grays/s, speed-up and efficiency against the N = 1 line, and efficiency against the COMPUTE-ONLY ceiling --
the time one GPU needs for one rank's shard of the frame with frames in flight and no exchange at all
(tools/shard_inflight.py, profiles/r05/shard_inflight_final.txt: 907 / 472 / 249 / 128 us at N = 1 / 2 / 4 / 8; sharding by
8-row blocks is not perfectly balanced and a short launch has a longer tail, so the ceiling itself is below N x).  What the
measured line loses against that ceiling is the exchange and its waits; what the ceiling loses against N x is compute balance.
"render-only" is the same ceiling measured IN the run (the slowest rank's config.per_rank[].kernel_ms_render_only: its shard with the
timed renderers and no exchange, right after the timed region) -- on a real node that is the figure to trust; the table is one box's.
The line's own per-rank record (config.per_rank) says which rank and which part; the last block lists the exchange forms the
same run measured (extra.exchange_forms: gather to rank 0, all-gather, direct peer writes), each with its parity bit.

The frame is one fork/join of main.cpp:139-154 spread over N GPUs; nothing here measures -- it only reads lines."""
import json
import sys

# us per frame of ONE rank's shard rendered alone on one MI355X, 3 frames in flight, whole-spp units, fused resolve (C3, pitch -0.5)
COMPUTE_ONLY_US = {"c3": {1: 906.6, 2: 472.4, 4: 249.1, 8: 128.4},        # profiles/r05/shard_inflight_final.txt (again in round 6: 906.2 / 472.4 / 249.1 / 127.9, profiles/r06/shard_inflight_c3.txt)
                   "c4": {1: 1211.2, 2: 628.6, 4: 327.3, 8: 165.8}}       # BASELINE config 4 (lens), the frame north_star shards: profiles/r06/shard_inflight_c4.txt


def load_line(path):
    with open(path) as f:
        text = f.read().strip()
    try:
        d = json.loads(text)                                    # a driver record (BENCH_rNN.json / SCALE_rNN.json entry) or a bare line
        if isinstance(d, dict) and "parsed" in d:
            d = d["parsed"]
        if isinstance(d, dict) and "metric" in d:
            return d
    except ValueError:
        pass
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{"):
            return json.loads(ln)
    raise ValueError(f"{path}: no JSON line")


def workload_key(d):
    w = d.get("config", {}).get("workload", "")
    return "c3" if w.startswith("512^3 LSVO terrain (FastNoise") else "c4" if "GI + DOF" in w else None


def report(lines):
    lines = sorted(lines, key=lambda d: d["n_gpus"])
    base = next((d for d in lines if d["n_gpus"] == 1), None)
    out = []
    wk = workload_key(lines[0])
    ceil = COMPUTE_ONLY_US.get(wk, {})
    out.append(f"workload: {lines[0]['config'].get('workload')}")
    out.append(f"{'N':>2s} {'ms/frame':>9s} {'Grays/s':>8s} {'speed-up':>9s} {'eff vs N=1':>11s} {'compute-only us':>16s} {'eff vs ceiling':>15s} "
               f"{'lost to':>22s} {'render-only':>12s} {'eff vs it':>9s}  parity  exchange")
    rows = []
    for d in lines:
        n, ms = d["n_gpus"], d["ms_per_step"]
        sp = base["ms_per_step"] / ms if base else None
        eff = sp / n if sp is not None else None
        c_us = ceil.get(n)
        eff_c = (c_us / 1e3) / ms if c_us else None
        # compute balance: what the ceiling itself loses against N x (of the N = 1 ceiling); exchange: the rest
        lost = ""
        if c_us and ceil.get(1):
            balance = 1.0 - (ceil[1] / n) / c_us
            exchange = 1.0 - eff_c
            lost = f"balance {balance * 100:4.1f}% exch {exchange * 100:5.1f}%"
        cfg = d.get("config", {})
        # the ceiling measured IN the run: every rank renders its shard with the timed renderers and no exchange right after the
        # timed region (config.per_rank[].kernel_ms_render_only); the slowest rank's figure is what the frame period cannot beat
        live = [p.get("kernel_ms_render_only") for p in (cfg.get("per_rank") or []) if p.get("kernel_ms_render_only")]
        live_ms = max(live) if live else (d.get("roofline", {}).get("kernel_ms") if n == 1 else None)
        eff_live = live_ms / ms if live_ms else None
        parity = cfg.get("frame_equals_oracle")
        ex = cfg.get("sharding", "").split(", ", 1)[-1][:60] if n > 1 else "-"
        rows.append({"n": n, "ms": ms, "grays": d["value"] / 1e3, "speedup": sp, "eff": eff, "ceiling_us": c_us, "eff_ceiling": eff_c,
                     "render_only_ms": live_ms, "eff_render_only": eff_live})
        f = lambda v, fmt: (fmt % v) if v is not None else "-"
        out.append(f"{n:2d} {ms:9.4f} {d['value'] / 1e3:8.2f} {f(sp, '%9.2f'):>9s} {f(eff, '%11.3f'):>11s} {f(c_us, '%16.1f'):>16s} "
                   f"{f(eff_c, '%15.3f'):>15s} {lost:>22s} {f(live_ms, '%12.4f'):>12s} {f(eff_live, '%9.3f'):>9s}  {str(parity):6s}  {ex}")
    for d in lines:
        pr = d.get("config", {}).get("per_rank")
        if pr:
            out.append(f"N = {d['n_gpus']}: per rank  " + "  ".join(
                f"[{p['rank']}] render {p.get('kernel_ms_render_only')} + exch {p.get('exchange_ms_per_step')} ms, drain {p.get('drain_wait_ms_per_block')} ms" for p in pr))
    for d in lines:
        forms = d.get("extra", {}).get("exchange_forms")
        if forms:
            out.append(f"N = {d['n_gpus']}: exchange forms measured in this run")
            for name in ("root", "all", "peer"):
                r = forms.get(name)
                if not r:
                    continue
                if "error" in r:
                    out.append(f"    {name:5s} FAILED: {r['error'][:120]}")
                    continue
                waits = r.get("per_rank_exchange_wait_ms") or []
                out.append(f"    {name:5s} {r['ms_per_step']:9.4f} ms/frame  {r.get('value', 0) / 1e3:7.2f} Grays/s  oracle {str(r.get('frame_equals_oracle')):5s}  "
                           f"max rank wait {max(waits) if waits else float('nan'):7.3f} ms" + ("   <- headline (`value`)" if r.get("headline") else ""))
            if forms.get("deadline"):
                out.append(f"    {forms['deadline']}")
    return "\n".join(out) + "\n", rows


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    text, _ = report([load_line(p) for p in sys.argv[1:]])
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
