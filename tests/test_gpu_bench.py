"""GPU: bench.py's config-5 leg (the child process bench.py spawns for BASELINE.json configs[4]) at a small table size -- the leg's code path
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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config5-child"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    d = json.loads(lines[0])
    assert "skipped" not in d, d
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["steps"] == 100
    assert 0.0 < d["final_loss"] < 2.0
