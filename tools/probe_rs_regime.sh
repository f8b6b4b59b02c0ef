#!/bin/bash
# K1 duration of the two resident residual kernels (CMLHIP_RS_TILE=16: k_ba_lin_rs4, 64: k_ba_lin_rs) on one window size
cfg=${1:-M}
for t in 16 64; do
  out=$(CMLHIP_RS_TILE=$t python bench.py --config $cfg --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f us  step %.1f us' % (d['linearize_kernel_us'], 1e3*d['ms_per_step']))")
  echo "config $cfg tile=$t : $out"
done
