#!/bin/bash
mkdir -p gpurun_out/r6
(timeout 1500 python tests/soak_parity.py 40 2>&1 | tail -30) > gpurun_out/r6/soak.txt
(timeout 1500 python tests/soak_parity.py 40 --tolerance 2>&1 | tail -30) > gpurun_out/r6/soak_tolerance.txt
(timeout 2400 python tests/soak_parity.py 30 --sequence 2>&1 | tail -45) > gpurun_out/r6/soak_sequence.txt
tail -12 gpurun_out/r6/soak.txt; tail -14 gpurun_out/r6/soak_tolerance.txt; tail -14 gpurun_out/r6/soak_sequence.txt
