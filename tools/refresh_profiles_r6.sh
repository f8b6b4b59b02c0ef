#!/bin/bash
# Run on the GPU box (through gpurun): every measurement profiles/round6_* quotes, from ONE build, under gpurun_out/refresh6/.
#   bench lines (default command, the driver's command, --extras), rocprofv3 kernel tables + PMC traffic of the bench command at B / C / E, the
#   VALU-issue inputs, in-kernel phases, run() cost inside the sequence shard, kernel tables of the sequence shard and of the batched launches.
set -u
OUT=gpurun_out/refresh6
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
cat libcml_amd/BUILD_COMMIT > $OUT/BUILD_COMMIT 2>/dev/null
python bench.py --detail $OUT/bench_default_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 --detail $OUT/bench_driver_cmd_detail.json > $OUT/bench_driver_cmd.json 2>/dev/null
python bench.py --extras --detail $OUT/bench_extras_detail.json > $OUT/bench_extras.json 2>/dev/null
for cfg in B C E; do
  python tools/profile_bench.py refresh6/prof_$cfg --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $OUT/profile_$cfg.log 2>&1
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
for S in 8 32; do      # one table per S (a merged table averages the two dispatch sizes in one column: VERDICT round 5, item 9)
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_batched_S$S -- python $OLDPWD/tools/probe_multi_window.py $S > $OLDPWD/$OUT/batched_values_S$S.txt 2>&1 )
  table $OUT/prof_batched_S$S $OUT/kernels_batched_S$S.md "python tools/probe_multi_window.py $S (S = $S config-B windows per launch, one and two stream groups)"
done
# round 6: the gather roof of the config-E residual kernel, what its stores cost, the tracker's in-kernel phases, S sequence shards on the one GPU
python tools/gather_roof.py $OUT/gather_roof_E.json > $OUT/gather_roof_E.log 2>&1
( echo "K1 of the config-E resident loop under the development switches of ba_linearize_rs_body.inc (CMLHIP_RS_DBG): 0 as shipped | 2 no pair tile / reduced-record stores |"; \
  echo "4 every texel tap on one L1-resident line per lane | 6 both | 16 no reduced-record store | 32 no pair-tile store | 48 neither | 64 no per-residual state stores"; \
  bash tools/probe_rs_dbg.sh "0 2 4 6 16 32 48 64"; echo "phase shift of the waves sharing a SIMD (CMLHIP_RS_STAGGER, x 0.43 us per slot; 8 is the default):"; \
  for st in 0 4 6 8 10 12; do echo "stagger=$st: $(CMLHIP_RS_STAGGER=$st bash tools/probe_rs_dbg.sh 0)"; done ) > $OUT/rs_stores_E.txt 2>&1
python tools/probe_shards_per_gpu.py $OUT/shards_per_gpu.json > $OUT/shards_per_gpu.log 2>&1
python tools/probe_keyframe_calls.py > $OUT/keyframe_calls.txt 2>&1
if [ -f ab_tmp/libcmlhip_toprof.so ]; then
  cp libcml_amd/libcmlhip.so /tmp/orig.so; cp ab_tmp/libcmlhip_toprof.so libcml_amd/libcmlhip.so
  ( echo "k_tracker_optimize, one hypothesis, config-B shape (build with -DTO_PROFILE: in-kernel clocks of the single-wave algebra), three runs:"; for i in 1 2 3; do python tools/probe_tracker_algebra.py; done ) > $OUT/tracker_phases.txt 2>&1
  cp /tmp/orig.so libcml_amd/libcmlhip.so
fi
python tools/probe_tracker_opt.py > $OUT/tracker_opt.txt 2>&1
rm -rf $OUT/prof_seq $OUT/prof_batched_S8 $OUT/prof_batched_S32 gpurun_out/prof_refresh6* gpurun_out/prof_valu* 2>/dev/null
ls -la $OUT gpurun_out/*.json gpurun_out/*.md 2>/dev/null | tail -40
