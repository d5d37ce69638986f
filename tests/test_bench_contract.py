"""bench.py's output contract, checked on CPU against the bench lines kept under profiles/ (the lines a GPU run of this
tree printed): one JSON object with the driver's keys, the roofline and cpu_baseline objects with theirs, no baseline
ratio claimed, a workload-named config, and the consistency relations a reader would check by hand."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "profiles", "r02")

DRIVER_KEYS = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
               "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def load(name):
    with open(os.path.join(R, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["final_bench.json.log", "final_bench_c2.json.log", "final_bench_c4.json.log", "final_bench_c5.json.log"])
def test_bench_line_contract(name):
    d = load(name)
    for k, t in DRIVER_KEYS.items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None                      # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["unit"] == "Mrays/s" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # achieved = algorithmic bytes per launch / kernel time; the kernel cannot take longer than the step it is part of
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert r["achieved"] < r["peak"]
    # value = rays per frame / frame period
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    # measured traffic is far below the algorithmic bytes (the tree is cache resident) and carries its provenance
    assert r["traffic"] is not None and r["traffic"] < r["algorithmic_bytes_per_launch"]
    assert os.path.exists(os.path.join(ROOT, r["traffic_detail"]["profile"]))
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1
    kinds = {(v["decomp"].split(" ")[0], v["parity"]) for v in c["variants"]}
    assert ("4x4x16", True) in kinds and ("omp", True) in kinds and ("omp", False) in kinds


def test_rocprof_stats_agree_with_the_bench_lines():
    """profiles/r02/final_c3*_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same commands) against the bench
    lines: one frame in flight, events = rocprof's average; three in flight, rocprof's average = the bracketed figure."""
    import csv

    def avg_ms(csv_name):
        with open(os.path.join(R, csv_name)) as f:
            for row in csv.DictReader(f):
                if row["Name"].startswith("k_render_sync"):
                    return row["Name"], float(row["AverageNs"]) / 1e6
        raise AssertionError(csv_name)

    d = load("final_bench.json.log")
    name, ms = avg_ms("final_c3_kernel_stats.csv")
    assert name == d["roofline"]["kernel"]
    assert abs(ms - d["roofline"]["kernel_ms_bracketed"]) < 0.05 * ms
    name1, ms1 = avg_ms("final_c3_inflight1_kernel_stats.csv")
    one = d["extra"]["one_frame_in_flight"]
    assert abs(ms1 - one["kernel_ms"]) < 0.02 * ms1
    assert one["kernel_ms"] <= one["ms_per_step"]


def test_dist_path_logs_verified_the_gathered_frame():
    for name in ("force_dist_bench.json.log", "force_dist_root_bench.json.log", "gloo2_bench.json.log"):
        d = load(name)
        assert d["config"]["gathered_frame_equals_unsharded"] is True, name
        assert d["config"]["ranks_in_process_group"] == d["n_gpus"] or name.startswith("force_dist")


# ---- round 3: the same contract on this round's kept lines, plus what the round added (spread over timed blocks, issue floor)
R3 = os.path.join(ROOT, "profiles", "r03")


def load3(name):
    with open(os.path.join(R3, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["final_bench.json.log", "final_bench_steps20.json.log", "final_bench_inflight1.json.log",
                                  "final_bench_c2.json.log", "final_bench_c4.json.log", "final_bench_c5.json.log"])
def test_round3_bench_line_contract(name):
    d = load3(name)
    for k, t in DRIVER_KEYS.items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # the nominal HBM form stays ...
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001 and r["achieved"] < r["peak"]
    assert r["traffic"] is not None and r["traffic"] < r["algorithmic_bytes_per_launch"]
    assert os.path.exists(os.path.join(ROOT, r["traffic_detail"]["profile"]))
    # ... and the bound that binds is named, with its floor: no faster than the floor, and the floor is the largest resource
    assert r["bound"] == "valu_issue"
    i = r["issue"]
    assert 0.3 < i["issue_frac"] <= 1.0 and abs(i["issue_frac"] - i["floor_ms"] / r["kernel_ms"]) < 2e-3
    assert i["floor_ms"] == max(i["floor_ms_by_resource"].values()) and i["binding_resource"] in i["floor_ms_by_resource"]
    assert abs(i["floor_ms"] - 2.0 * i["valu_wave_instr_used"] / 1024 / 2.4e9 * 1e3) < 2e-3 or i["binding_resource"] != "valu_issue"
    # the median block is the one reported
    b = d["extra"]["ms_per_step_blocks"]
    lo, med, hi = b["min_median_max"]
    assert b["blocks"] >= 5 and b["steps_per_block"] == d["steps"] and lo <= med <= hi and abs(med - d["ms_per_step"]) < 1e-3
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]


def test_round3_headline_line():
    d = load3("final_bench.json.log")
    assert d["steps"] == 100 and d["config"]["frames_in_flight"] == 3 and d["roofline"]["kernel"] == "k_render_sync_pinhole_resolved"
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1
    one = d["extra"]["one_frame_in_flight"]
    assert one["kernel_ms"] <= one["ms_per_step"] and one["blocks"] >= 3
    lo, med, hi = one["ms_per_step_blocks"]
    assert lo <= med <= hi and med == one["ms_per_step"]
    # rocprofv3 --stats of the same commands (tools/pmc_roofline.sh): alone on the chip rocprof's average = the event time
    import csv

    def avg_ms(csv_name):
        with open(os.path.join(R3, csv_name)) as f:
            for row in csv.DictReader(f):
                if row["Name"].startswith("k_render"):
                    return row["Name"], float(row["AverageNs"]) / 1e6
        raise AssertionError(csv_name)
    name1, ms1 = avg_ms("final_c3_inflight1_kernel_stats.csv")
    assert name1 == one["kernel"] and abs(ms1 - one["kernel_ms"]) < 0.02 * ms1
    name3, ms3 = avg_ms("final_c3_kernel_stats.csv")
    assert name3 == d["roofline"]["kernel"] and abs(ms3 - d["roofline"]["kernel_ms_bracketed"]) < 0.05 * ms3


def test_round3_dist_path_logs_verified_the_gathered_frame():
    for name in ("force_dist_bench.json.log", "force_dist_root_bench.json.log", "force_dist_peer_bench.json.log",
                 "gloo2_bench.json.log", "gloo2_peer_bench.json.log"):
        d = load3(name)
        assert d["config"]["gathered_frame_equals_unsharded"] is True, name
    assert "direct peer writes" in load3("gloo2_peer_bench.json.log")["config"]["sharding"]
    for name in ("peer_write_world2.json.log", "peer_write_world3.json.log"):
        assert load3(name)["composed_frame_equals_unsharded"] == [True] * 9
    for name in ("cpp_rccl_all_world1.json.log", "cpp_rccl_root_world1.json.log"):
        assert load3(name)["gathered_frames_equal_unsharded"] is True
    for name in ("cpp_peer_write_world2.json.log", "cpp_peer_write_world3.json.log"):
        assert load3(name)["composed_frames_equal_unsharded"] is True


# ---- round 4: the parity bit, the CPU leg's stated host, the self-launched N > 1 lines, the world-8 rehearsals
R4 = os.path.join(ROOT, "profiles", "r04")


def load4(name):
    with open(os.path.join(R4, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["final_bench.json.log", "final_bench_steps20.json.log", "final_bench_c2.json.log", "final_bench_c4.json.log",
                                  "final_bench_c5.json.log"])
def test_round4_bench_line_contract(name):
    d = load4(name)
    for k, t in DRIVER_KEYS.items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    # the parity bit of the run itself: the timed renderer's frame 0 against the oracle's
    assert d["config"]["frame_equals_oracle"] is True and "oracle/vrc_oracle.c" in d["config"]["frame_equals_oracle_is"]
    assert d["config"]["library"].endswith("(the product)") and d["config"]["launched_by"] == "plain"
    r = d["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001 and r["achieved"] < r["peak"]
    b = d["extra"]["ms_per_step_blocks"]
    lo, med, hi = b["min_median_max"]
    assert b["blocks"] >= 5 and lo <= med <= hi and abs(med - d["ms_per_step"]) < 1e-3
    # the CPU leg says what the host gave it: the rate peaks at the container's quota because a team beyond it only gets throttled
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s"
    h = c["host"]
    assert h["cgroup_cpu_max"] and h["usable"] >= c["cores"]
    if h["cgroup_quota_cores"]:
        omp = [v for v in c["variants"] if v["decomp"] == "omp" and v["parity"]]
        assert all(v["cpu_seconds_per_second"] <= h["cgroup_quota_cores"] * 1.1 for v in omp)
        assert c["cores"] <= 2 * h["cgroup_quota_cores"]


def test_round4_headline_line_carries_the_stamped_pmc_record():
    d = load4("final_bench.json.log")
    r = d["roofline"]
    assert d["steps"] == 100 and d["config"]["frames_in_flight"] == 3 and r["kernel"] == "k_render_sync_pinhole_resolved"
    assert r["traffic"] is not None and r["traffic"] < r["algorithmic_bytes_per_launch"] and r["bound"] == "valu_issue"
    assert r["traffic_detail"]["profile"].startswith("profiles/r04/") and os.path.exists(os.path.join(ROOT, r["traffic_detail"]["profile"]))
    # the timed region is one stretch of GPU work several seconds long (round-3 verdict item 5)
    assert d["extra"]["ms_per_step_blocks"]["blocks"] * d["steps"] * d["ms_per_step"] * 1e-3 > 6.0
    # a launch alone on the chip runs the samples-abreast kernel (the library's choice), and faster than round 3's 1.116 ms
    one = d["extra"]["one_frame_in_flight"]
    assert one["kernel"] == "k_render_sync_pinhole_s4" and one["ms_per_step"] < 1.11
    import csv
    with open(os.path.join(R4, "final_c3_kernel_stats.csv")) as f:
        row = [x for x in csv.DictReader(f) if x["Name"].startswith("k_render")][0]
    assert row["Name"] == r["kernel"] and abs(float(row["AverageNs"]) / 1e6 - r["kernel_ms_bracketed"]) < 0.05 * r["kernel_ms_bracketed"]


def test_round4_self_launched_and_world8_lines():
    for name, n in (("self_launch_gloo2_full.json.log", 2), ("self_launch_gloo2_root.json.log", 2), ("self_launch_gloo8_peer.json.log", 8)):
        d = load4(name)
        assert d["n_gpus"] == n and d["config"]["ranks_in_process_group"] == n and d["config"]["gathered_frame_equals_unsharded"] is True, name
        assert d["config"]["launched_by"].startswith("bench.py itself") and d["config"]["distinct_gpus"] == 1
    assert "direct peer writes" in load4("self_launch_gloo8_peer.json.log")["config"]["sharding"]
    for w in (2, 3, 8):
        assert load4(f"peer_write_world{w}.json.log")["composed_frame_equals_unsharded"] == [True] * 9
        assert load4(f"cpp_peer_write_world{w}.json.log")["composed_frames_equal_unsharded"] is True
    dead = load4("peer_write_dead_writer.json.log")
    assert dead["exit_codes"] == [7, 5, 7] and dead["seconds"] < 80


# ---- round 5: the timed symbol alone (verdict Next 1), period vs latency (Next 5), the self-validating N > 1 line (Next 2)
R5 = os.path.join(ROOT, "profiles", "r05")


def load5(name):
    with open(os.path.join(R5, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["final_bench.json.log", "final_bench_steps20.json.log", "final_bench_c2.json.log", "final_bench_c4.json.log",
                                  "final_bench_c5.json.log"])
def test_round5_bench_line_contract(name):
    d = load5(name)
    for k, t in DRIVER_KEYS.items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    assert d["config"]["frame_equals_oracle"] is True
    # the line says which of its figures is a period and where the latency is
    assert "PERIOD" in d["config"]["ms_per_step_is"] and "extra.one_frame_in_flight" in d["config"]["ms_per_step_is"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel_ms"] <= d["ms_per_step"] * 1.001
    # one launch of the timed symbol alone on the chip: slower than the period, and priced on the same algorithmic bytes
    assert r["kernel_ms_alone"] > r["kernel_ms"]
    assert abs(r["frac_alone"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_alone"] * 1e-3) / 1e9 / r["peak"]) < 1e-3
    assert d["extra"]["timed_symbol_alone"]["kernel"] == r["kernel"]
    assert d["extra"]["one_frame_in_flight"]["ms_per_step"] > d["ms_per_step"]
    assert d["cpu_baseline"]["host"]["cgroup_quota_from"] in (None, "cgroup v2 cpu.max", "cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us")


def test_round5_headline_quotes_a_clean_rocprof_record_of_the_timed_symbol():
    """round-4 verdict, Next 1: 5.150 GB / that file's average / 8 TB/s reproduces roofline.frac_alone within 3 %"""
    import csv
    d = load5("final_bench.json.log")
    r = d["roofline"]
    rp = r["kernel_ms_alone_rocprof"]
    assert rp["profile"] == "profiles/r05/final_c3_timed_symbol_alone_kernel_stats.csv" and rp["agrees_with_kernel_ms_alone_within_3pct"] is True
    with open(os.path.join(ROOT, rp["profile"])) as f:
        row = [x for x in csv.DictReader(f) if x["Name"].startswith("k_render")][0]
    assert row["Name"] == r["kernel"] == "k_render_sync_pinhole_resolved_q" and int(row["Calls"]) >= 100      # (quadrant walks: the default since round 5)
    avg_ms = float(row["AverageNs"]) / 1e6
    assert abs(avg_ms - rp["average_ms"]) < 1e-3
    frac_from_file = r["algorithmic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / 8000.0
    assert abs(frac_from_file / r["frac_alone"] - 1.0) < 0.03
    # the bench line of the profiled command itself: one launch at a time, whole-spp units, the same symbol
    alone = load5("final_bench_timed_symbol_alone.json.log")
    assert alone["config"]["frames_in_flight"] == 1 and alone["config"]["whole_spp_work_units"] is True
    assert alone["roofline"]["kernel"] == r["kernel"] and abs(alone["roofline"]["kernel_ms"] / r["kernel_ms_alone"] - 1.0) < 0.03
    # (round 6 took the records again for its kernel-source hash: profiles/traffic.json now points at profiles/r06/, and
    # test_round6_headline_... holds the same relations on that round's line)


def test_round5_self_launched_lines_validate_themselves():
    """round-4 verdict, Next 2: at N > 1 the GATHERED frame 0 against the oracle, and what the exchange ran on, in the line"""
    for name, n, gather in (("self_launch_gloo4_root.json.log", 4, "root"), ("self_launch_gloo4_peer.json.log", 4, "peer"),
                            ("self_launch_gloo2_root.json.log", 2, "root"), ("self_launch_gloo8_peer.json.log", 8, "peer")):
        d = load5(name)
        c = d["config"]
        assert d["n_gpus"] == n and c["ranks_in_process_group"] == n, name
        assert c["gathered_frame_equals_unsharded"] is True and c["frame_equals_oracle"] is True, name
        assert f"{n} ranks" in c["frame_equals_oracle_is"] and "summed over ranks" in c["frame_equals_oracle_is"]
        assert [p["rank"] for p in c["per_rank"]] == list(range(n)) and all(p["kernel_ms_render_only"] > 0 for p in c["per_rank"])
        # the four ranks' rows make up the frame: their rays add up to the unsharded frame's
        assert abs(sum(p["rays_per_frame"] for p in c["per_rank"]) - c["rays_per_frame"]) <= n      # (each is a floor of a mean over the block's frames)
        ex = c["exchange"]
        assert ex["backend"] == "gloo" and ex["world"] == n and ex["rccl_version"] and ex["device_can_access_peer"] == [[1]]
        assert ex["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        if gather == "root":
            assert "gloo" in c["sharding"] and "REHEARSAL" in c["sharding"] and "RCCL gather" not in c["sharding"]
        else:
            assert "direct peer writes" in c["sharding"]


R6 = os.path.join(ROOT, "profiles", "r06")


def load6(name):
    with open(os.path.join(R6, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["final_bench.json.log", "final_bench_steps20.json.log", "final_bench_c2.json.log", "final_bench_c4.json.log",
                                  "final_bench_c5.json.log"])
def test_round6_bench_line_contract(name):
    d = load6(name)
    for k, t in DRIVER_KEYS.items():
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["unit"] == "Mrays/s" and d["n_gpus"] == 1
    assert d["config"]["frame_equals_oracle"] is True
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert r["kernel_ms_alone"] > r["kernel_ms"] and d["extra"]["timed_symbol_alone"]["kernel"] == r["kernel"]
    # the PMC record is this round's (taken again on this tree's kernel sources), and the --stats file it names exists
    assert r["traffic"] is not None and r["traffic"] < 0.2 * r["algorithmic_bytes_per_launch"]        # (C2, a 0.07 ms frame: 6 MB of 0.3 GB; C3: 0.4 %)
    assert r["traffic_detail"]["profile"].startswith("profiles/r06/") and os.path.exists(os.path.join(ROOT, r["traffic_detail"]["profile"]))
    rp = r["kernel_ms_alone_rocprof"]
    assert rp["profile"].startswith("profiles/r06/") and os.path.exists(os.path.join(ROOT, rp["profile"])) and rp["agrees_with_kernel_ms_alone_within_3pct"] is True
    assert r["bound"] == "valu_issue" and 0.4 < r["issue"]["issue_frac"] <= 1.0             # (C2, a 0.07 ms launch that is mostly tail: 0.48; C3 0.78)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_round6_headline_says_which_work_is_distinct():
    """round-5 verdict, Next 3: a reader of `value` and `lane_utilisation` must be able to tell that part of the rays are
    lockstep copies of a pixel's sample-invariant walks (raycaster.hpp:131,147-158)"""
    import csv
    d = load6("final_bench.json.log")
    c, r = d["config"], d["roofline"]
    # (means over a block's frames, whose GI rays differ frame by frame: frame 0 alone has 32 222 738 rays, 12.4 M of them repeats)
    assert abs(c["rays_per_frame"] - 32222738) < 5000 and abs(c["rays_distinct_per_frame"] - 19829529) < 5000
    assert c["rays_distinct_per_frame"] == d["extra"]["invariant_rays_walked_once"]["rays_per_frame_executed"]
    assert c["ms_per_identical_image"] == d["extra"]["invariant_rays_walked_once"]["ms_per_step"] < d["ms_per_step"]
    assert "raycaster.hpp:131,147-158" in c["rays_distinct_note"]
    by_stage = r["issue"]["lane_utilisation_by_stage"]
    assert set(by_stage) == {"primary", "shadow", "gi", "gi_shadow"}
    assert by_stage["primary"] > 0.9 and by_stage["shadow"] > 0.85 and by_stage["gi"] < 0.5 and by_stage["gi_shadow"] < 0.55
    assert by_stage == json.load(open(os.path.join(ROOT, "profiles", "issue_model.json")))["lane_utilisation_by_stage"]
    assert "lockstep copies" in r["issue"]["lane_utilisation_by_stage_note"]
    assert min(by_stage.values()) < r["issue"]["lane_utilisation"] < max(by_stage.values())     # the PMC average lies between
    # the headline's roofline figures, recomputed from the files it names
    rp = r["kernel_ms_alone_rocprof"]
    with open(os.path.join(ROOT, rp["profile"])) as f:
        row = [x for x in csv.DictReader(f) if x["Name"].startswith("k_render")][0]
    assert row["Name"] == r["kernel"] == "k_render_sync_pinhole_resolved_q" and int(row["Calls"]) >= 100
    avg_ms = float(row["AverageNs"]) / 1e6
    assert abs(r["algorithmic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / 8000.0 / r["frac_alone"] - 1.0) < 0.03
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["entries"]["c3_n1_inflight1_wholespp"]
    assert abs(rec["profiled_kernel_ms"] - avg_ms) < 1e-3 and rec["kernel"] == r["kernel"] and rec["settings"]["frames_in_flight"] == 1


@pytest.mark.parametrize("n", [2, 4, 8])
def test_round6_one_invocation_reports_the_three_exchange_forms(n):
    """round-5 verdict, Next 2 (done criterion): `VRC_DIST_BACKEND=gloo python3 bench.py --gpus N --steps 5` on the box -- rc 0,
    three forms in extra.exchange_forms, every parity bit true.  (A rehearsal: the ranks share the box's one GPU and the
    collectives are host-staged gloo; the peer writes are real hipIpc mappings.)"""
    d = load6(f"exchange_forms_gloo{n}.json.log")
    c = d["config"]
    assert d["n_gpus"] == n and c["ranks_in_process_group"] == n and c["frame_equals_oracle"] is True and c["gathered_frame_equals_unsharded"] is True
    forms = d["extra"]["exchange_forms"]
    assert "deadline" not in forms and "close_error" not in forms
    assert forms["root"]["headline"] is True and forms["root"]["ms_per_step"] == d["ms_per_step"] and forms["root"]["value"] == d["value"]
    for name in ("root", "all", "peer"):
        f = forms[name]
        assert "error" not in f and f["frame_equals_oracle"] is True and f["ms_per_step"] > 0 and f["value"] > 0, (name, f)
        assert len(f["per_rank_exchange_wait_ms"]) == n
    assert "REHEARSAL" in forms["note"] and "rehearsal" in c["sharding"].lower()
    assert c["exchange"]["backend"] == "gloo" and c["exchange"]["device_can_access_peer"] == [[1]]


def test_round6_rccl_runs_the_three_forms_with_a_world_of_one():
    """the only RCCL this project can run on a one-GPU box: VRC_FORCE_DIST=1 VRC_FORCE_FORMS=1 takes the N > 1 code path -- process
    group on the nccl backend, shard buffers, the async gather / all-gather handles, the switch between the forms, PeerFrames'
    set-up and flags -- with one rank.  Every form renders the oracle's frame 0; their periods are the N = 1 frame's."""
    d = load6("force_forms_rccl_world1.json.log")
    c = d["config"]
    assert d["n_gpus"] == 1 and c["exchange"]["backend"] == "nccl" and c["exchange"]["rccl_version"][0].isdigit()
    assert "RCCL gather to rank 0" in c["sharding"] and c["frame_equals_oracle"] is True
    forms = d["extra"]["exchange_forms"]
    for name in ("root", "all", "peer"):
        assert "error" not in forms[name] and forms[name]["frame_equals_oracle"] is True, name
        assert abs(forms[name]["ms_per_step"] / d["ms_per_step"] - 1.0) < 0.06, name
    assert "REHEARSAL" not in forms["note"]
