#!/bin/bash
# A/B of builds of libcmlhip.so on ONE box (run through gpurun): the variants are ab_tmp/libcmlhip_<name>.so (untracked), alternated, three bench runs each.
#   gpurun -- "bash tools/ab_run.sh base variant"     (AB_ARGS="--config E" for another workload)
mkdir -p gpurun_out/ab
cp libcml_amd/libcmlhip.so /tmp/orig.so
for i in 1 2 3; do
  for v in "$@"; do
    cp ab_tmp/libcmlhip_$v.so libcml_amd/libcmlhip.so
    python bench.py --no-extras --no-cpu-baseline ${AB_ARGS:-} > gpurun_out/ab/${v}_$i.json 2>/dev/null
  done
done
cp /tmp/orig.so libcml_amd/libcmlhip.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "ms %.4f rep %.4f K1 %.2f ok %s" % (d['ms_per_step'], d['ms_per_step_repeats']['median'], d['linearize_kernel_us'], d.get('parity_ok')))
    except Exception as e:
        print(f, 'err', e)
PY
