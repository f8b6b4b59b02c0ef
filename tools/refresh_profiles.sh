#!/bin/bash
# Run on the GPU box (through gpurun): every measurement profiles/ quotes for the current build, under gpurun_out/refresh/.
#   bench lines B / C / E (unprofiled), rocprofv3 kernel tables + PMC traffic of the same commands, in-kernel phases, tracker kernel table
set -u
OUT=gpurun_out/refresh
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python bench.py > $OUT/bench_B.json 2> $OUT/bench_B.err
python bench.py --config C --no-cpu-baseline > $OUT/bench_C.json 2> $OUT/bench_C.err
python bench.py --config E > $OUT/bench_E.json 2> $OUT/bench_E.err
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_B_20steps.json 2>/dev/null
for cfg in B C E; do
  python tools/profile_bench.py refresh/prof_$cfg --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/profile_$cfg.log 2>&1
done
python tools/probe_phases.py B > $OUT/phases_B.txt 2>&1
python tools/probe_phases.py E > $OUT/phases_E.txt 2>&1
# tracker kernels (the bench's tracker object under the kernel tracer)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_tracker -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 )
python - <<'PY'
import csv, glob, os
d = "gpurun_out/refresh/prof_tracker"
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/refresh/kernels_tracker.md", "w") as o:
        o.write("`rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline` (tracker / multi-window objects included) on 1x MI355X\n\n")
        o.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:30]:
            o.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $OUT/prof_tracker gpurun_out/prof_refresh* 2>/dev/null
ls -la $OUT
