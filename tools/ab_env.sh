#!/bin/bash
# A/B of a development switch (an environment variable of libcmlhip.so) on ONE box: bash tools/ab_env.sh CMLHIP_NO_SPLIT [bench args]
V=$1; shift
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab/on_$i.json 2>/dev/null
  env $V=1 python bench.py --no-extras --no-cpu-baseline "$@" > gpurun_out/ab/off_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab/o*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "ms %.4f rep %.4f K1 %.2f ok %s solve %s" % (d['ms_per_step'], d['ms_per_step_repeats']['median'], d['linearize_kernel_us'], d.get('parity_ok'), d.get('schur_solve_ms')))
    except Exception as e:
        print(f, 'err', e)
PY
