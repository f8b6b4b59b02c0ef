#!/bin/bash
# K1 at config E with the waves of a SIMD delayed by slot x st x 0.43 us at their start (development switch, ba_linearize_rs_body.inc)
for st in 0 2 4 8 16 24; do
  out=$(CMLHIP_RS_DBG=$((st*256)) python bench.py --config E --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f us  step %.1f us' % (d['linearize_kernel_us'], 1e3*d['ms_per_step']))")
  echo "stagger=$st : $out"
done
