#!/bin/bash
# Run on the GPU box: the kernels of a tracked frame's chain (tracker batch -> trace -> publish) from the 48-frame sequence shard, and the
# bench's sequence object three times (frame_ms is a median over ~47 frames: +-0.005 ms run to run).
set -u
OUT=gpurun_out/frame_chain
mkdir -p $OUT
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -- python $OLDPWD/tools/probe_sequence.py 48 > /dev/null 2>&1 )
python - $OUT/prof <<'PY' | tee $OUT/kernels.txt
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("trace", "tracker", "ticket")):
        print("%-64s %4s  avg %7.2f us  min %7.2f  max %7.2f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
for i in 1 2 3; do
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sequence']; print('frame_ms', s['frame_ms'], 'lib fps', s['library_frames_per_s'], 'run_ms', s['run_ms'], 'fps', s['frames_per_s'])" | tee -a $OUT/frame_ms.txt
done
