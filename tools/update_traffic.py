#!/usr/bin/env python3
"""Turn one tools/pmc_roofline.sh result into the stamped record bench.py reads (profiles/traffic.json) and keep
the summaries under profiles/.

    python tools/update_traffic.py gpurun_out/roofline_<tag> <key e.g. c3_n1> <profiles/rNN/prefix>

The record carries the hash of the kernel sources it was measured on (written on the GPU box next to the counters);
bench.py reports roofline.traffic = null with the reason when the tree's kernel sources have a different hash.
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB per launch; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src, key, dst_prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    vals = {}
    kernel_symbol = None
    for line in open(os.path.join(src, "summary.txt")):
        f = line.split()
        if len(f) == 3 and f[0] == "==" and f[1] == "kernel":
            if kernel_symbol is not None:
                break                                             # only the first block: the kernel with the most launches
            kernel_symbol = f[2]
        if len(f) >= 5 and f[1] == "launches":
            vals[f[0]] = float(f[10]) if len(f) >= 11 and f[9] == "median" else float(f[4])
    # the settings of the profiled bench command (its JSON line, kept by tools/pmc_roofline.sh)
    settings = {}
    log = os.path.join(src, "stats.log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            settings = {"frames_in_flight": d["config"].get("frames_in_flight"), "fused": d["config"].get("resolve_fused_into_frame_kernel"),
                        "whole_spp_work_units": d["config"].get("whole_spp_work_units"), "bench_kernel": d["roofline"].get("kernel")}
    source_hash = open(os.path.join(src, "source_hash.txt")).read().strip()
    import bench
    commit = None
    if bench.kernel_source_hash() == source_hash:
        try:
            commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
            if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "cpuvoxelraycaster_amd/csrc"], text=True).strip():
                commit += "+uncommitted"
        except Exception:   # noqa: BLE001
            commit = None
    kernel_ms = None
    ks = os.path.join(src, "kernel_stats.csv")
    if os.path.exists(ks):
        for r in csv.DictReader(open(ks)):
            if r.get("Name", "") == kernel_symbol or (kernel_symbol is None and r.get("Name", "").startswith("k_render")):
                kernel_ms = float(r["AverageNs"]) / 1e6
    fetch = 2.0 * vals["FETCH_SIZE"] * 1024.0
    write = vals["WRITE_SIZE"] * 1024.0
    rec = {
        "traffic_bytes": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
        "valu_wave_instr": int(vals["SQ_INSTS_VALU"]),
        "lane_utilisation": round(vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"]), 4),
        "simd_cycles_per_valu_instr": round(vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / vals["SQ_INSTS_VALU"], 3),
        "wait_inst_any_frac_of_wave_cycles": round(vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"], 4),
        "profiled_kernel_ms": round(kernel_ms, 4) if kernel_ms else None,
        "source_hash": source_hash, "commit": commit, "profile": dst_prefix + "_pmc.txt",
        "kernel": kernel_symbol, "settings": settings,
    }
    for extra in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH", "SQ_WAVES"):
        if extra in vals:
            rec[extra.lower()] = int(vals[extra])
    os.makedirs(os.path.dirname(os.path.join(ROOT, dst_prefix)), exist_ok=True)
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(ROOT, dst_prefix + "_pmc.txt"))
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(ROOT, dst_prefix + "_kernel_stats.csv"))
    log = os.path.join(src, "stats.log")
    if os.path.exists(log):
        with open(os.path.join(ROOT, dst_prefix + "_stats_bench.json.log"), "w") as f:
            lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]      # the bench line, not rocprofv3's chatter
            f.write((lines[-1] if lines else "") + "\n")
    path = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    if "entries" not in doc:
        doc = {"_comment": doc.get("_comment", ""), "entries": {}}
    doc["_comment"] = ("Per-launch rocprofv3 PMC records of k_render_sync (tools/pmc_roofline.sh, separate --pmc passes; "
                       "traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE in bytes, MI355X_MICROARCH.md HBM section), each stamped with the "
                       "hash of the kernel sources it was measured on; bench.py uses a record only for that hash.")
    doc["entries"][key] = rec
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
