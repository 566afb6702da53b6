"""The driver's contract with bench.py: ONE JSON line on stdout with the agreed keys, run here on the small config C2 with
every side block that needs minutes switched off (the side blocks that remain -- the seam, the launch counts -- are
checked for their shape)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_agreed_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "C2",
           "--no-cpu-baseline", "--no-c5", "--no-rank-proxy", "--no-small", "--no-dist-one-rank"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == "CG solves/s" and d["unit"] == "solves/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] - 1e3) < 1e-6 * 1e3               # value = steps / elapsed, ms_per_step = elapsed / steps
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.05 < r["frac"] < 1.0                                                # a measured rate, below the peak
    assert d["iterations"] > 5 and d["rel_residual"] <= 1.1e-12 and d["preconditioner"]["kind"] == "amg"
    assert d["assembly"]["ms"] > 0 and len(d["assembly"]["ms_each"]) == 10 and d["seam"]["ms_per_solve"] > d["ms_per_step"]
    assert d["preconditioner"]["launches"]["per_setup"] > 50
