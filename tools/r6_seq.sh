#!/bin/bash
# sequence shard: fused frame path — tests, host-clock split of a frame, kernel trace
mkdir -p gpurun_out/r6
(timeout 600 python -m pytest tests/test_frame_fused_gpu.py tests/test_tracer_gpu.py -x -q 2>&1 | tail -5)
(CMLHOST_TIMING=1 timeout 300 python tools/probe_sequence.py 24 2>&1 | grep "\[frame\]" | tail -12)
(timeout 300 python tools/probe_sequence.py 48 2>&1 | tail -16)
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_seq -- python $GRAFT_REPO_ROOT/tools/probe_sequence.py 48 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_seq/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:14]:
        print("%-60s calls %5s avg %8.2f us  %s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd $GRAFT_REPO_ROOT
timeout 400 python tools/probe_shards_per_gpu.py gpurun_out/r6/shards_per_gpu.json 2>&1 | tail -60
