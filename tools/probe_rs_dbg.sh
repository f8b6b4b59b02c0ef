#!/bin/bash
# K1 at config E under the development switches of ba_linearize_rs.hip: 0 real | 2 no tile / reduced-record stores | 4 one L1-resident line per lane | 6 both
for dbg in ${1:-0 2 4 6}; do
  out=$(CMLHIP_RS_DBG=$dbg python bench.py --config E --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f us  step %.1f us' % (d['roofline']['launch_us'], 1e3*d['ms_per_step']))")
  echo "dbg=$dbg : K1 $out"
done
