"""The driver's contract with bench.py, checked where it runs: `python bench.py --gpus 1 --steps K --warmup W` prints exactly ONE line on stdout, a
compact JSON object (< 4 KB: the driver keeps ~8 KB of stdout and round 4's 27.5 KB line was unreadable to it) with the keys the contract names
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload), the
`roofline` object of the dominant kernel, `cpu_baseline`, and the verdict of the oracle gate that ran behind the timed region; the full object goes to
bench_detail.json."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_contract(d, steps, warmup):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"]
    assert d["unit"] == "point-residuals/s" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
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
    assert d["parity_checked"] is True and d["parity_ok"] is True
    assert "invalid" not in d


def _run(args, detail):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--detail", detail], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:3]
    return lines[0], r.stdout


def test_bench_line_keeps_the_contract(tmp_path):
    detail = str(tmp_path / "detail.json")
    line, _ = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-extras", "--no-cpu-baseline"], detail)
    assert len(line) < 4096
    d = json.loads(line)
    _check_contract(d, 6, 2)
    full = json.load(open(detail))
    assert full["parity"]["R"] == 14000 and full["value"] == d["value"]


def test_the_exact_driver_command_prints_a_line_the_driver_can_read(tmp_path):
    """`python bench.py --gpus 1 --steps 20 --warmup 5` — the command BENCH_rNN.json records — with every default leg on (configs C / E, tracker,
    sequence shard, CPU baseline): the LAST stdout line is < 4 KB, parses from the tail the driver keeps, and carries roofline + cpu_baseline."""
    detail = str(tmp_path / "detail.json")
    line, stdout = _run(["--gpus", "1", "--steps", "20", "--warmup", "5"], detail)
    assert len(line) < 4096, len(line)
    tail = stdout[-8000:]                                        # what the driver keeps
    d = json.loads([l for l in tail.splitlines() if l.strip()][-1])
    _check_contract(d, 20, 5)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "point-residuals/s" and cb["sample"]
    assert set(d["configs"]) == {"C", "E"} and all(c.get("parity_ok") is True for c in d["configs"].values()), d["configs"]
    assert d["sequence"]["parity_ok"] is True and d["sequence"]["frames_per_s"] > 0 and d["sequence"]["yardstick_used"] <= 1
    assert d["tracker"]["optimize_ms_1"] > 0 and d["solve"]["us"] > 0
    full = json.load(open(detail))
    assert len(json.dumps(full)) > len(line)                    # tables, notes, min/max lists live in the detail file


def test_the_contract_region_carries_no_instrumentation(tmp_path):
    """The timed region runs without profile events on its dispatches (the residual kernel's duration comes from an event-carrying region behind it):
    the driver's `--steps 20 --warmup 5` figure must agree with the default `--steps 200 --warmup 20` one — with events on every 4th step it read 46.1
    against 43.9 us.  What remains between the two is the region's own closing sync + barrier (25-30 us of host wake-up, inside the region by the
    bench contract) amortised over K steps: 1.3-1.5 us per step at K = 20, 0.15 at K = 200 — 3 % by construction; measured 2.7 - 4.7 % between separate
    processes on a shared box.  Bar 6 % (one retry), and the 20-step figure must not be the FASTER one by more than 2 %."""
    worst = None
    for attempt in range(2):
        a, _ = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"], str(tmp_path / "a.json"))
        b, _ = _run(["--gpus", "1", "--steps", "200", "--warmup", "20", "--no-extras", "--no-cpu-baseline"], str(tmp_path / "b.json"))
        ma, mb = json.loads(a)["ms_per_step"], json.loads(b)["ms_per_step"]
        worst = (ma - mb) / mb
        print("ms_per_step at --steps 20: %.5f, at --steps 200: %.5f (%+.1f %%)" % (ma, mb, 100 * worst))
        if -0.02 < worst < 0.06:
            break
    assert -0.02 < worst < 0.06, worst
