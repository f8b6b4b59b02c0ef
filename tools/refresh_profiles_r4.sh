#!/bin/bash
# Run on the GPU box (through gpurun): every measurement profiles/round4_* quotes for the current build, under gpurun_out/refresh4/.
set -u
OUT=gpurun_out/refresh4
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>/dev/null
for cfg in B C E; do
  python tools/profile_bench.py refresh4/prof_$cfg --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/profile_$cfg.log 2>&1
done
BENCH_ARITH_RELAXED=1 python tools/profile_bench.py refresh4/prof_E_relaxed --config E --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/profile_E_relaxed.log 2>&1      # CMLHIP_ARITH_RELAXED (the bench line's bit-exact gate reports its mismatch: expected)
python tools/valu_roof.py B > $OUT/valu_B.log 2>&1
python tools/valu_roof.py E > $OUT/valu_E.log 2>&1
python tools/probe_phases.py B > $OUT/phases_B.txt 2>&1
python tools/probe_phases.py E > $OUT/phases_E.txt 2>&1
bash tools/probe_rs_scaling.sh > $OUT/rs_scaling_E.txt 2>&1
bash tools/probe_rs_stagger.sh > $OUT/rs_stagger_E.txt 2>&1
python tools/probe_run_cost.py 48 > $OUT/run_cost.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_seq -- python $OLDPWD/tools/probe_sequence.py 48 > /dev/null 2>&1 )
python - <<'PY'
import csv, glob, os
d = "gpurun_out/refresh4/prof_seq"
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open("gpurun_out/refresh4/kernels_sequence.md", "w") as o:
        o.write("`rocprofv3 --kernel-trace --stats -- python tools/probe_sequence.py 48` (a 48-frame sequence shard through the host mirror) on 1x MI355X\n\n")
        o.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            o.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $OUT/prof_seq gpurun_out/prof_refresh4* gpurun_out/prof_valu* 2>/dev/null
ls -la $OUT gpurun_out/*.json gpurun_out/*.md 2>/dev/null | tail -40
