"""The profile tooling behind bench.py's roofline record, on synthetic rocprofv3 output (CPU): tools/pmc_summary.py sums a
dispatch's rows, reports the median (a launch that carries a neighbour's counts must not move the record), and
tools/update_traffic.py's unit conversions are the ones MI355X_MICROARCH.md prescribes for gfx950."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_pass(d, name, rows):
    os.makedirs(os.path.join(d, name, "x"), exist_ok=True)
    with open(os.path.join(d, name, "x", "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerows(rows)


def test_pmc_summary_median_and_traffic_units(tmp_path):
    d = str(tmp_path / "roofline_t")
    k = "k_render_sync_pinhole_resolved_w8"
    rows = []
    for disp, scale in ((1, 1.0), (2, 1.0), (3, 2.0), (4, 1.0), (5, 1.0)):     # dispatch 3 carries a neighbour's counts
        rows += [(disp, k, "FETCH_SIZE", 500.0 * scale), (disp, k, "FETCH_SIZE", 500.0 * scale),   # two rows per dispatch: summed
                 (disp, "k_resolve_shard", "FETCH_SIZE", 7.0)]
    write_pass(d, "fetch", rows)
    write_pass(d, "write", [(i, k, "WRITE_SIZE", 3000.0) for i in range(1, 6)])
    write_pass(d, "sq1", [(i, k, c, v) for i in range(1, 6) for c, v in (("SQ_INSTS_VALU", 1e9), ("SQ_ACTIVE_INST_VALU", 1e9),
                                                                       ("SQ_THREAD_CYCLES_VALU", 3.2e10), ("SQ_WAVE_CYCLES", 4e9),
                                                                       ("SQ_WAIT_INST_ANY", 2e9))])
    write_pass(d, "grbm", [(i, k, "GRBM_GUI_ACTIVE", 8 * 3.0e6) for i in range(1, 6)])
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d, "k_render"], text=True)
    line = [ln.split() for ln in out.splitlines() if ln.startswith("FETCH_SIZE")][0]
    assert line[1:3] == ["launches", "5"]
    assert float(line[line.index("median") + 1]) == 1000.0 and float(line[line.index("max") + 1]) == 2000.0
    assert float(line[line.index("mean") + 1]) == 1200.0
    # what update_traffic.py makes of it (the arithmetic only; it is not run here because it writes into profiles/)
    vals = {ln.split()[0]: float(ln.split()[ln.split().index("median") + 1]) for ln in out.splitlines() if "median" in ln}
    fetch_bytes = 2.0 * vals["FETCH_SIZE"] * 1024.0            # gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, unit KiB
    write_bytes = vals["WRITE_SIZE"] * 1024.0
    assert (fetch_bytes, write_bytes) == (2048000.0, 3072000.0)
    assert vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"]) == 0.5
    assert vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / vals["SQ_INSTS_VALU"] == 3.072


def test_traffic_records_are_stamped_and_used_only_for_their_sources(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert doc["entries"]
    for key, rec in doc["entries"].items():
        assert len(rec["source_hash"]) == 16 and rec["commit"], key
        assert abs(rec["traffic_bytes"] - (rec["fetch_bytes"] + rec["write_bytes"])) <= 2, key
        assert os.path.exists(os.path.join(ROOT, rec["profile"])), key
    key, rec = next(iter(doc["entries"].items()))
    # a record is used when the tree's kernel sources + compiler flags hash to its stamp ...
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: rec["source_hash"])
    got, why = bench.pmc_record(key)
    assert got is not None and got["traffic_bytes"] == rec["traffic_bytes"]
    # ... and refused, with the reason, when they do not, or when there is none for the workload
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "0" * 16)
    got, why = bench.pmc_record(key)
    assert got is None and why
    got, why = bench.pmc_record("no_such_workload")
    assert got is None and why


def test_issue_model_reads_the_walk_loop_and_is_current():
    """tools/issue_model.py parses the asm text of csrc/vrc_walk.h into per-block instruction classes; profiles/issue_model.json
    (what bench.py prices roofline.issue.floor_ms with) must be the model of THIS tree's loop -- it is regenerated on the CPU
    (python tools/issue_model.py), so a kernel change that forgets it fails here rather than silently dropping the floor."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    import bench
    import issue_model
    nolod, lod = issue_model.block_counts(0, False), issue_model.block_counts(1, False)
    valu = lambda b: sum(b.get(c, 0) for c in "ABSQP")
    assert [valu(nolod[k]) for k in ("header", "descend", "advance", "pop")] == [14, 24, 16, 21]
    assert valu(lod["header"]) == 16 and lod["header"]["Q"] == 6          # + the LOD product and its compare
    assert valu(lod["descend"]) == 24                                     # no t_max update: the stack entry is the parent alone
    assert valu(nolod["start"]) == 11 and nolod["start"]["lds"] == 1 and nolod["start"]["vmem"] == 1   # entry of a ray that starts below the root
    assert all(b.get("P", 0) == 0 for b in nolod.values())               # no packed / lane-access instruction in the loop
    assert nolod["descend"]["lds"] == 1 and nolod["pop"]["lds"] == 1 and nolod["descend"]["vmem"] == 1 and nolod["pop"]["vmem"] == 1
    # the header is read in the SHIPPED variant (the product defines none of the A/B switches): the wait for the node comes
    # before the exits' compare class, and the text with VRC_WALK_LATE_WAIT defined is the other one (round-5 advice)
    assert list(nolod["header"]).index("wait") < list(nolod["header"]).index("Q")
    src = open(issue_model.WALK).read()
    shipped = issue_model.macro_body(issue_model.preprocess(src), "VRC_WALK_HEADER_REST")
    late = issue_model.macro_body(issue_model.preprocess(src, defined=("VRC_WALK_LATE_WAIT",)), "VRC_WALK_HEADER_REST")
    first = lambda body: body.lstrip("\\ \n\t")
    assert first(shipped).startswith('"s_waitcnt') and first(late).startswith('"v_sub_f32') and "s_waitcnt" in late
    assert "buffer_load" in issue_model.macro_body(issue_model.preprocess(src), "VRC_WALK_LOAD_NODE")
    lane = issue_model.block_counts(0, True)                             # per-lane complexity: one VALU more, two scalar fewer
    assert valu(lane["header"]) == 15 and lane["header"]["salu"] == nolod["header"]["salu"] - 2
    doc = json.load(open(os.path.join(ROOT, "profiles", "issue_model.json")))
    assert doc["blocks_no_lod"] == nolod and doc["blocks_lod"] == lod
    assert doc["source_hash"] == bench.kernel_source_hash()
    freq = json.load(open(os.path.join(ROOT, doc["frequencies"])))
    pts = issue_model.per_tile_sample(freq, nolod, lod)
    assert abs(sum(pts.get(c, 0) for c in "ABSQP") - sum(doc["walks_per_tile_sample"].get(c, 0) for c in "ABSQP")) < 0.1
    f = issue_model.floor_cycles(1000.0, 100.0, 0.0, 400.0)
    assert f["valu_arch"] == 2000.0 and f["slow_unit"] == 430.0 and abs(f["scalar_unit"] - 440.0) < 1e-9


def test_path_schedule_policies_on_a_synthetic_walk():
    """tests/tools/path_schedule.py's wave simulations (the shipped loop and round 5's work list of pending blocks) on a walk
    whose answer is known: lanes that all take the same path issue each block once per round whatever the policy, and a
    divergent wave never issues fewer blocks than its fullest lane needs."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import path_schedule as ps
    n, L = 64, 12
    # op codes per iteration: 0 descend, 1 advance, 2 advance + pop; the same for every lane
    row = np.array([0, 0, 1, 2, 0, 1, 2, 2, 1, 0, 1, 2], np.uint8)
    ops = np.tile(row, (n, 1))
    length = np.full(n, L, np.int64)
    cost, rounds = ps.simulate(ops, length, {}, lod=False)
    want = L * ps.H_ + (row == 0).sum() * ps.D_ + (row != 0).sum() * ps.A_ + (row == 2).sum() * ps.P_
    assert (cost, rounds) == (want, L)
    c2, r2 = ps.simulate_work_list(ops, length, 1, lod=False)
    assert c2 == want                      # one path per round: nothing waits for anything (the pops run a round late)
    assert r2 >= rounds
    # a divergent wave: every policy executes every lane's iterations, and costs at least the uniform wave's blocks
    rng = np.random.default_rng(1)
    ops = rng.integers(0, 3, (n, L)).astype(np.uint8)
    length = rng.integers(1, L + 1, n).astype(np.int64)
    base, _ = ps.simulate(ops, length, {}, lod=True)
    for T in (1, 8, 32):
        c, r = ps.simulate_work_list(ops, length, T, lod=True)
        assert c > 0 and r >= int(length.max())
    assert base >= int(length.max()) * (ps.H_ + 2.0)
    # the mask dump records one entry per issued block, in issue order, with the lanes that executed it
    ps.MASKS = []
    try:
        ps.simulate(ops, length, {}, lod=True)
        blocks = np.array([b for b, _ in ps.MASKS])
        lanes = np.array([k for _, k in ps.MASKS])
        assert (blocks == 0).sum() == int(length.max()) and lanes.min() >= 1 and lanes.max() <= n
        assert lanes[blocks == 0].sum() == int(length.sum())          # every lane-iteration went through one header
    finally:
        ps.MASKS = None


def test_scale_report_attributes_a_curve_to_balance_and_exchange(tmp_path):
    """tools/scale_report.py (round-5 verdict, Next 2): four bench lines in, one table out -- efficiency against N = 1 and
    against the compute-only ceiling of one rank's shard (profiles/r05/shard_inflight_final.txt), and the exchange forms a
    line carries.  Synthetic lines: N = 2 exactly at its ceiling (all of its loss is balance), N = 8 at twice its ceiling
    (half of its time is exchange)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_report as sr
    wl = "512^3 LSVO terrain (FastNoise seed 1337, generated and built on the GPU), 1920x1080, 4 spp GI (1 indirect bounce), shadow_samples=1"
    rays = 32222738

    def line(n, ms, forms=None):
        d = {"metric": "m", "n_gpus": n, "ms_per_step": ms, "value": rays / (ms * 1e-3) / 1e6,
             "config": {"workload": wl, "frame_equals_oracle": True, "sharding": f"8-row blocks round-robin over {n} GPU(s), RCCL gather to rank 0 of RGBA8 shards"}}
        if forms:
            d["extra"] = {"exchange_forms": forms}
        return d
    forms8 = {"root": {"headline": True, "ms_per_step": 0.2568, "value": 1.0e5, "frame_equals_oracle": True, "per_rank_exchange_wait_ms": [0.1] * 8},
              "all": {"headline": False, "ms_per_step": 0.30, "value": 9.0e4, "frame_equals_oracle": True, "per_rank_exchange_wait_ms": [0.2] * 8},
              "peer": {"headline": False, "error": "VrcError('hipIpcOpenMemHandle refused')"}}
    lines = [line(1, 0.9066), line(2, 0.4724), line(4, 0.2491 * 1.25), line(8, 0.2568, forms8)]
    paths = []
    for d in lines:
        p = tmp_path / f"n{d['n_gpus']}.json"
        p.write_text("RCCL banner line\n" + json.dumps(d) + "\n")            # the JSON line is the LAST line of a log
        paths.append(str(p))
    (tmp_path / "rec.json").write_text(json.dumps({"rc": 0, "parsed": lines[0]}))    # a driver record holds it under "parsed"
    assert sr.load_line(str(tmp_path / "rec.json"))["n_gpus"] == 1
    text, rows = sr.report([sr.load_line(p) for p in reversed(paths)])
    by_n = {r["n"]: r for r in rows}
    assert abs(by_n[2]["eff_ceiling"] - 1.0) < 1e-6 and abs(by_n[2]["eff"] - 906.6 / 472.4 / 2) < 1e-3
    assert abs(by_n[4]["eff_ceiling"] - 0.8) < 1e-6
    assert abs(by_n[8]["eff_ceiling"] - 0.5) < 1e-3 and abs(by_n[8]["speedup"] - 0.9066 / 0.2568) < 1e-6
    assert "balance 11.7% exch  50.0%" in text
    assert "peer  FAILED: VrcError" in text and "<- headline" in text and "all      0.3000 ms/frame" in text
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "scale_report.py")] + paths, text=True)
    assert out == text
