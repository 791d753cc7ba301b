"""GPU: bench.py's config-5 leg (the child process bench.py spawns through bench_legs.py for BASELINE.json configs[4]) at a small table size -- the leg's code path
(large-table engine, captured step, JSON contract) without its 154 GB."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_child_prints_one_json_object():
    env = dict(os.environ, RECBENCH_C5_ITEMS="300000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_legs.py"), "--leg", "config5"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    d = json.loads(lines[0])
    assert "skipped" not in d, d
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] == 128 and d["distinct_batches"] == 64
    assert 0.0 < d["final_loss"] < 2.0


@pytest.mark.parametrize("leg,unit", [("config1", "triplets/s"), ("config4", "rows/s")])
def test_small_config_legs_print_one_json_object(leg, unit):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_legs.py"), "--leg", leg], capture_output=True, text=True, timeout=300, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    d = json.loads(lines[0])
    assert "skipped" not in d, d
    assert d["unit"] == unit and d["value"] > 0 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    if leg == "config4":
        # the step's dominant kernel family is the MLP's GEMM (mfma); the field bag (hbm) is reported beside it
        assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["roofline"]["peak"] == 157.3
        assert d["roofline_fm_bag"]["bound"] == "hbm" and 0 < d["roofline_fm_bag"]["frac"] < 1
    if leg == "config1":
        # (26 MB of parameters and moments: the launch runs out of the Infinity Cache and is priced against ITS measured ceiling)
        assert d["roofline"]["bound"] == "infinity_cache" and d["roofline"]["peak"] == 8600.0 and 0 < d["roofline"]["frac"] < 1


def test_gpus_flag_must_match_the_launcher():
    """--gpus 2 under a launcher that started ONE rank is an error (exit 2), not an N = 1 line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode == 2 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], (r.returncode, r.stdout[-300:], r.stderr[-300:])
