"""The plain launch line for N > 1 on the GPU box (round-3 verdict, item 1): `python bench.py --gpus N` with no launcher in
the line starts its own ranks.  One GPU here, so the ranks share it and exchange over gloo (VRC_DIST_BACKEND=gloo); the
gathered frame must equal the unsharded one and the JSON line must be the last line of stdout."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,gather", [(2, "root"), (4, "root"), (8, "peer")])
def test_plain_bench_line_launches_its_ranks(built, world, gather):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VRC_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "2", "--blocks", "1",
                          "--blocks-min-seconds", "0", "--gather", gather, "--config", "c4"] + ([] if world == 4 else ["--no-extras"]),
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == world and d["config"]["ranks_in_process_group"] == world
    assert d["config"]["gathered_frame_equals_unsharded"] is True
    # the line validates itself at N > 1 (round-4 verdict, Next 2): the GATHERED frame 0 and the ranks' summed counters against the oracle
    assert d["config"]["frame_equals_oracle"] is True, d["config"].get("frame_equals_oracle_is")
    assert f"{world} ranks" in d["config"]["frame_equals_oracle_is"]
    assert len(d["config"]["per_rank"]) == world and [r["rank"] for r in d["config"]["per_rank"]] == list(range(world))
    assert all(r["kernel_ms_render_only"] > 0 for r in d["config"]["per_rank"])
    assert d["config"]["exchange"]["backend"] == "gloo" and "rehearsal" in d["config"]["sharding"].lower() or gather == "peer"
    assert d["config"]["launched_by"].startswith("bench.py itself") and d["config"]["distinct_gpus"] == 1
    if world == 4:
        # round-5 verdict, Next 2: the one invocation reports all three exchange forms, each with its own parity bit (on this
        # one-GPU box 'root' and 'all' are the same host-staged gloo rehearsal; the peer writes are real hipIpc mappings)
        forms = d["extra"]["exchange_forms"]
        assert forms[gather]["headline"] is True and forms[gather]["ms_per_step"] == d["ms_per_step"]
        for name in ("root", "all", "peer"):
            f = forms[name]
            assert "error" not in f, (name, f)
            assert f["frame_equals_oracle"] is True and f["ms_per_step"] > 0 and len(f["per_rank_exchange_wait_ms"]) == world, (name, f)
        assert "deadline" not in forms and "REHEARSAL" in forms["note"]
    else:
        assert "exchange_forms" not in d.get("extra", {})
    log = os.path.join(ROOT, "gpurun_out", "r06")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, f"self_launch_gloo{world}_{gather}.json.log"), "w") as f:
        f.write(json.dumps(d) + "\n")


def test_the_drivers_launch_line(built):
    """the line the driver uses for N > 1, word for word: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` (here N = 2 ranks sharing the box's one GPU,
    hence VRC_DIST_BACKEND=gloo).  bench.py is then a rank, not a launcher: rc 0, rank 0's JSON line is the last line of
    stdout, it says who launched it, validates itself and carries the three exchange forms."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["VRC_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                          "--blocks", "1", "--blocks-min-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_in_process_group"] == 2 and c["launched_by"] == "torch.distributed.run"
    assert c["frame_equals_oracle"] is True and c["gathered_frame_equals_unsharded"] is True
    forms = d["extra"]["exchange_forms"]
    assert all(forms[f]["frame_equals_oracle"] is True for f in ("root", "all", "peer")) and "deadline" not in forms
    log = os.path.join(ROOT, "gpurun_out", "r06")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "torchrun_gloo2.json.log"), "w") as f:
        f.write(json.dumps(d) + "\n")


def test_the_extras_deadline_keeps_the_headline_line(built):
    """round 6: the block that measures the other exchange forms is untimed and must never cost the run its result.  With a
    deadline of 50 ms it cannot finish: rank 0 prints the headline line it already has -- parity bit and all, the reason under
    extra.exchange_forms.deadline -- as the LAST line of stdout, and every rank leaves by itself with exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VRC_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--blocks", "1",
                          "--blocks-min-seconds", "0", "--config", "c4", "--forms-deadline-s", "0.05"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["frame_equals_oracle"] is True and d["value"] > 0
    forms = d["extra"]["exchange_forms"]
    assert "did not finish within" in forms["deadline"] and forms["root"]["headline"] is True and set(forms) == {"root", "deadline"}
    assert "did not finish within" in out.stderr


def test_nccl_on_too_few_gpus_fails_loudly(built):
    """RCCL needs one GPU per rank: on this one-GPU box the plain line must leave non-zero with the reason, not hang"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "VRC_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "RCCL needs one per rank" in out.stderr
