#!/bin/bash
mkdir -p gpurun_out/r6
(timeout 900 python -m pytest tests/test_frame_fused_gpu.py tests/test_tracer_gpu.py tests/test_sequence_gpu.py tests/test_sequence_split_gpu.py -x -q 2>&1 | tail -5)
(CMLHOST_TIMING=1 timeout 300 python tools/probe_sequence.py 24 2>&1 | grep "\[frame\]" | tail -10)
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/r6/b4_bench_detail.json > gpurun_out/r6/b4_bench.json 2> gpurun_out/r6/b4_bench.err); tail -c 900 gpurun_out/r6/b4_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6/b4_bench_detail.json'))
s=d.get('sequence',{})
print({k:s.get(k) for k in ('frames_per_s','library_frames_per_s','frame_ms','traces_redone')})
for k,v in (s.get('library_ms_per_stage') or {}).items(): print("%-28s calls %3d mean/stage %.3f ms" % (k, v['calls'], v['mean_ms_per_stage']))
print(s.get('error'), d.get('ms_per_step'))
PY
