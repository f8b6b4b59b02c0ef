#!/bin/bash
OUT=gpurun_out/refresh6; mkdir -p $OUT
python bench.py --detail $OUT/bench_default_detail.json > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 5 --detail $OUT/bench_driver_cmd_detail.json > $OUT/bench_driver_cmd.json 2>/dev/null
python bench.py --extras --detail $OUT/bench_extras_detail.json > $OUT/bench_extras.json 2>/dev/null
python tools/probe_shards_per_gpu.py $OUT/shards_per_gpu.json > $OUT/shards_per_gpu.log 2>&1
tail -c 700 $OUT/bench_default.json; echo; tail -c 700 $OUT/bench_driver_cmd.json
