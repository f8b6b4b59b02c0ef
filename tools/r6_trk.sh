#!/bin/bash
mkdir -p gpurun_out/r6
(timeout 900 python -m pytest tests/test_tracker_optimize_gpu.py tests/test_tracker_parity_gpu.py tests/test_frame_fused_gpu.py -x -q 2>&1 | tail -5)
cp libcml_amd/libcmlhip.so /tmp/orig.so
cp ab_tmp/libcmlhip_toprof.so libcml_amd/libcmlhip.so
(for i in 1 2 3; do timeout 300 python tools/probe_tracker_algebra.py; done 2>&1 | tail -3)
cp /tmp/orig.so libcml_amd/libcmlhip.so
timeout 300 python tools/probe_tracker_opt.py 2>&1 | tail -12
