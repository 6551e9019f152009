"""bench.py's output contract (the driver parses this line): one JSON object on stdout with the agreed keys, from the
launcher -> rank-process path with a live RCCL group, at a reduced batch so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pairs", "8",
                          "--no-latency", *extra], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract(dev):
    d = _run("--no-cpu")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "nccl" in d["config"]["process_group"]
    assert d["value"] > 0 and abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert 0 < r["frac"] <= 1 and r["bound"] in ("valu-issue", "mfma", "hbm", "fp64-valu")
    assert d["cpu_baseline"] is None                       # --no-cpu


def test_bench_shard_mode_line(dev):
    d = _run("--no-cpu", "--mode", "shard")
    assert d["scaling"] == "strong" and d["config"]["mode"] == "shard" and d["n_gpus"] == 1 and d["value"] > 0
