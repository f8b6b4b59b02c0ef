#!/usr/bin/env python3
"""Run on the GPU box (through gpurun): rocprofv3 passes over the default bench command.

  pass 1  --kernel-trace --stats                  -> per-kernel table (markdown)
  pass 2  --pmc FETCH_SIZE   (own run, no traces) -> memory-side read bytes per k_ba_linearize launch
  pass 3  --pmc WRITE_SIZE   (own run)            -> memory-side write bytes per launch

Writes gpurun_out/<tag>_kernels.md and gpurun_out/<tag>_pmc.json; copy what should be judged into profiles/.
Counter units/corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB-ish
units of the TCC_EA request counters; on gfx950 a 16 B/lane streaming read is under-counted by 2x (FETCH_SIZE = RDREQ x 64 B
with 128-B requests), so the read figure is doubled; other widths are uncalibrated, which is why both raw and corrected
values are stored."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
bench_args = sys.argv[2:] or ["--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
env = dict(os.environ, TMPDIR="/tmp")


def run(extra, sub):
    d = os.path.join(OUT, "prof_%s_%s" % (tag, sub))
    cmd = ["rocprofv3"] + extra + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=False, text=True)
    run.last_stdout = r.stdout or ""
    return d


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return f[0] if f else None


os.makedirs(OUT, exist_ok=True)
d1 = run(["--kernel-trace", "--stats"], "stats")
rows = list(csv.DictReader(open(find(d1, "kernel_stats.csv"))))
with open(os.path.join(OUT, tag + "_kernels.md"), "w") as f:
    f.write("`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py %s` on 1x MI355X\n\n" % " ".join(bench_args))
    f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in rows[:26]:
        f.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                             float(r["MaxNs"]) / 1e3, r["Percentage"]))
KERNEL = os.environ.get("CML_PROF_KERNEL", "k_ba_lin")          # the residual kernel (k_ba_linearize<..> / k_ba_lin_rs<..>)
lin = [r for r in rows if KERNEL in r["Name"] and "finish" not in r["Name"]]
res = {"bench_args": bench_args, "kernel": lin[0]["Name"][:60] if lin else None, "linearize_avg_us": float(lin[0]["AverageNs"]) / 1e3 if lin else None}
for line in run.last_stdout.splitlines():          # the bench line of the SAME (profiled) command: its roofline.launch_us must agree with the average above
    if line.startswith("{") and "roofline" in line:
        try:
            b = json.loads(line)
            res["bench_line_under_profiler"] = {"launch_us": b["roofline"]["launch_us"], "frac": b["roofline"]["frac"], "ms_per_step": b["ms_per_step"],
                                                "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
        except Exception:
            pass
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = run(["--pmc", ctr], ctr.lower())
    f = find(d, "counter_collection.csv")
    vals = []
    if f:
        for r in csv.DictReader(open(f)):
            if KERNEL in r.get("Kernel_Name", "") and "finish" not in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr:
                vals.append(float(r["Counter_Value"]))
    res[ctr + "_raw_avg"] = sum(vals) / len(vals) if vals else None
    res[ctr + "_launches"] = len(vals)
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes (derived: RDREQ*64/1024 ...)
if res.get("FETCH_SIZE_raw_avg") is not None and res.get("WRITE_SIZE_raw_avg") is not None:
    res["read_bytes_per_launch_raw"] = res["FETCH_SIZE_raw_avg"] * 1024
    res["read_bytes_per_launch_gfx950_x2"] = res["FETCH_SIZE_raw_avg"] * 1024 * 2
    res["write_bytes_per_launch_raw"] = res["WRITE_SIZE_raw_avg"] * 1024
    res["traffic_bytes_per_launch"] = res["read_bytes_per_launch_gfx950_x2"] + res["write_bytes_per_launch_raw"]
json.dump(res, open(os.path.join(OUT, tag + "_pmc.json"), "w"), indent=1)
print(json.dumps(res))
