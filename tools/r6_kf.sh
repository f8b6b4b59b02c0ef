#!/bin/bash
(timeout 900 python -m pytest tests/test_frame_fused_gpu.py tests/test_tracer_gpu.py tests/test_sequence_gpu.py tests/test_sequence_split_gpu.py tests/test_marginalization_gpu.py tests/test_pyramid_async_gpu.py tests/test_window_edits_gpu.py -x -q 2>&1 | tail -8)
timeout 300 python tools/probe_keyframe_calls.py 2>&1 | tail -20
