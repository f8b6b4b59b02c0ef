#!/bin/bash
# Run on the GPU box (through gpurun): every measurement profiles/round5_* quotes, from ONE build, under gpurun_out/refresh5/.
#   bench lines (default command, the driver's command, --extras), rocprofv3 kernel tables + PMC traffic of the bench command at B / C / E, the
#   VALU-issue inputs, in-kernel phases, run() cost inside the sequence shard, kernel tables of the sequence shard and of the batched launches.
set -u
OUT=gpurun_out/refresh5
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
cat libcml_amd/BUILD_COMMIT > $OUT/BUILD_COMMIT 2>/dev/null
python bench.py --detail $OUT/bench_default_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 --detail $OUT/bench_driver_cmd_detail.json > $OUT/bench_driver_cmd.json 2>/dev/null
python bench.py --extras --detail $OUT/bench_extras_detail.json > $OUT/bench_extras.json 2>/dev/null
for cfg in B C E; do
  python tools/profile_bench.py refresh5/prof_$cfg --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/profile_$cfg.log 2>&1
done
python tools/valu_roof.py B > $OUT/valu_B.log 2>&1
python tools/valu_roof.py E > $OUT/valu_E.log 2>&1
python tools/probe_phases.py B > $OUT/phases_B.txt 2>&1
python tools/probe_phases.py E > $OUT/phases_E.txt 2>&1
python tools/probe_run_cost.py 48 > $OUT/run_cost.txt 2>&1
table () {   # $1 = profile directory, $2 = output markdown, $3 = command line quoted in the header
python - "$1" "$2" "$3" <<'PY'
import csv, glob, os, sys
d, out, cmd = sys.argv[1:4]
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(out, "w") as o:
        o.write("`rocprofv3 --kernel-trace --stats -- %s` on 1x MI355X\n\n" % cmd)
        o.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            o.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
}
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_seq -- python $OLDPWD/tools/probe_sequence.py 48 > /dev/null 2>&1 )
table $OUT/prof_seq $OUT/kernels_sequence.md "python tools/probe_sequence.py 48 (a 48-frame sequence shard through the host mirror)"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_batched -- python $OLDPWD/tools/probe_multi_window.py 8 32 > $OLDPWD/$OUT/batched_values.txt 2>&1 )
table $OUT/prof_batched $OUT/kernels_batched.md "python tools/probe_multi_window.py 8 32 (S = 8 and S = 32 config-B windows per launch, one and two stream groups)"
rm -rf $OUT/prof_seq $OUT/prof_batched gpurun_out/prof_refresh5* gpurun_out/prof_valu* 2>/dev/null
ls -la $OUT gpurun_out/*.json gpurun_out/*.md 2>/dev/null | tail -40
