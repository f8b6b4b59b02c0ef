#!/bin/bash
# sequence shard: fused frame path — tests, host-clock split of a frame, stage times, shards per GPU
mkdir -p gpurun_out/r6
(timeout 900 python -m pytest tests/test_frame_fused_gpu.py tests/test_tracer_gpu.py tests/test_sequence_gpu.py tests/test_sequence_split_gpu.py -x -q 2>&1 | tail -8)
(CMLHOST_TIMING=1 timeout 300 python tools/probe_sequence.py 24 2>&1 | grep "\[frame\]" | tail -10)
(timeout 300 python tools/probe_sequence.py 48 2>&1 | tail -15)
timeout 500 python tools/probe_shards_per_gpu.py gpurun_out/r6/shards_per_gpu.json 2>&1 | tail -50
