#!/bin/bash
(timeout 900 python -m pytest tests/test_sequence_gpu.py tests/test_sequence_split_gpu.py tests/test_marginalization_gpu.py tests/test_frame_fused_gpu.py -x -q 2>&1 | tail -5)
timeout 300 python tools/probe_keyframe_calls.py 2>&1 | tail -30
