#!/bin/bash
mkdir -p gpurun_out/r6
(timeout 1200 python -m pytest tests/test_frame_fused_gpu.py tests/test_tracer_gpu.py tests/test_tracker_optimize_gpu.py tests/test_sequence_gpu.py tests/test_sequence_split_gpu.py tests/test_threads_gpu.py -x -q 2>&1 | tail -25) > gpurun_out/r6/b3_pytest.txt
cat gpurun_out/r6/b3_pytest.txt
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail gpurun_out/r6/b3_bench_detail.json > gpurun_out/r6/b3_bench.json 2> gpurun_out/r6/b3_bench.err); tail -c 1500 gpurun_out/r6/b3_bench.json; tail -5 gpurun_out/r6/b3_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6/b3_bench_detail.json'))
s=d.get('sequence',{})
print({k:s.get(k) for k in ('frames_per_s','library_frames_per_s','frame_ms','traces_redone','per_frame_ms','per_keyframe_ms')})
for k,v in (s.get('library_ms_per_stage') or {}).items(): print(k, v)
print(s.get('error'))
PY
