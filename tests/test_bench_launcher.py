"""CPU: bench.py's launcher contract -- `--gpus N` without a launcher starts the ranks itself (or fails loudly), before any GPU call;
under a launcher the flag must match WORLD_SIZE.  No GPU is touched here: every case exits before the first device call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_gpus_n_without_enough_devices_exits_nonzero_and_prints_no_line():
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2          # (device_count does not initialise the GPU)
    r = run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_flag_must_match_world_size():
    r = run(["--gpus", "4", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE = 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launch_ranks_builds_a_torchrun_command_on_localhost(monkeypatch):
    """The spawned command: one process per GPU through torch.distributed.run, rendezvous on 127.0.0.1, this file and the caller's flags."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"]) == 7         # (the children's exit code is the run's)
    c = seen["cmd"]
    assert c[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in c and c[c.index("--master-addr") + 1] == "127.0.0.1"
    assert c[-5] == os.path.join(ROOT, "bench.py") and c[-4:] == ["--gpus", "4", "--steps", "3"]


def test_self_launch_starts_four_ranks_that_meet_on_localhost():
    """`python bench.py --gpus 4 --rendezvous-only` with no launcher around it: bench.py starts the four ranks itself (torch.distributed.run,
    127.0.0.1), every rank joins the group (gloo here: no GPU), sees all four and prints its OWN line with world_size == 4; exit code 0."""
    import json
    r = run(["--gpus", "4", "--rendezvous-only"])
    import re
    lines = [json.loads(m) for m in re.findall(r'\{"rendezvous".*?\}', r.stdout)]      # (four processes write to one pipe: lines may share a row)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert sorted(d["rank"] for d in lines) == [0, 1, 2, 3]
    for d in lines:
        assert d["world_size"] == 4 and d["ranks_seen"] == [0, 1, 2, 3] and d["master"] == "127.0.0.1"
