#!/bin/bash
# round 6, first GPU batch: new parity statement (config E vs fp32 texels), gather roof, tracker phases, event-free contract region
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_config_e_gpu.py tests/test_tracker_optimize_gpu.py -x -q -s 2>&1 | tail -15) > gpurun_out/r6/b1_pytest.txt
(timeout 600 python tools/gather_roof.py gpurun_out/r6/gather_roof_E.json 2>&1 | tail -60) > gpurun_out/r6/b1_gather.txt
cp libcml_amd/libcmlhip.so /tmp/orig.so
cp ab_tmp/libcmlhip_toprof.so libcml_amd/libcmlhip.so
(for i in 1 2 3; do timeout 300 python tools/probe_tracker_algebra.py; done 2>&1 | tail -8) > gpurun_out/r6/b1_tracker_phases.txt
cp /tmp/orig.so libcml_amd/libcmlhip.so
(timeout 900 python -m pytest tests/test_bench_contract_gpu.py -x -q -s -k "no_instrumentation" 2>&1 | tail -8) > gpurun_out/r6/b1_contract.txt
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r6/b1_bench_detail.json > gpurun_out/r6/b1_bench.json 2> gpurun_out/r6/b1_bench.err)
tail -c 3000 gpurun_out/r6/b1_bench.json
cat gpurun_out/r6/b1_pytest.txt gpurun_out/r6/b1_gather.txt gpurun_out/r6/b1_tracker_phases.txt gpurun_out/r6/b1_contract.txt
