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
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
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
