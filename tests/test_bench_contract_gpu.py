"""The driver's contract with bench.py, checked where it runs: `python bench.py --gpus 1 --steps K --warmup W` prints exactly ONE line on stdout, a JSON
object with the keys the contract names (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline /
dtype / data / config.workload), the `roofline` object of the dominant kernel, and the verdict of the oracle gate that ran behind the timed region."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_keeps_the_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    assert d["unit"] == "point-residuals/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "8-KF" in d["config"]["workload"] and "2000 active points" in d["config"]["workload"] and "model" not in d["config"]
    # value = units of all ranks / the timed region; ms_per_step = that region / K
    assert abs(d["value"] - 14000 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert 0.02 < d["ms_per_step"] < 0.2
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and 0.0 < ro["frac"] < 1.0
    assert ro["algorithmic_bytes_per_launch"] == 14000 * 468 and ro["launch_samples"] >= 1
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_launch"] / (ro["launch_us"] * 1e-6) / 1e9) <= 1e-6 * ro["achieved"]
    assert "traffic" in ro
    assert d["parity_checked"] is True and d["parity_ok"] is True and d["parity"]["R"] == 14000
    assert "invalid" not in d
