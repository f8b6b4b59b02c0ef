#!/usr/bin/env python3
"""Run on the GPU box (through gpurun): one `rocprofv3 --pmc <counter>` pass of the bench command per counter (own runs, no
traces), averaged over the launches of one kernel.

    python tools/profile_counters.py <tag> <kernel substring[,substring...]> <counter,counter,...> [bench.py arguments ...]

Writes gpurun_out/<tag>_counters.json."""
import csv, glob, json, os, subprocess, sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
tag, kern, ctrs = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
bench_args = sys.argv[4:] or ["--steps", "100", "--warmup", "10", "--no-cpu-baseline"]
env = dict(os.environ, TMPDIR="/tmp")
os.makedirs(OUT, exist_ok=True)
res = {"bench_args": bench_args}
for ctr in ctrs:
    d = os.path.join(OUT, "prof_%s_%s" % (tag, ctr.lower()))
    cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=600)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    for kn in kern.split(","):
        vals = [float(r["Counter_Value"]) for r in rows if kn in r.get("Kernel_Name", "") and r.get("Counter_Name") == ctr]
        res.setdefault(kn, {})[ctr] = {"avg": sum(vals) / len(vals) if vals else None, "launches": len(vals)}
    subprocess.run(["rm", "-rf", d])
json.dump(res, open(os.path.join(OUT, tag + "_counters.json"), "w"), indent=1)
print(json.dumps(res))
