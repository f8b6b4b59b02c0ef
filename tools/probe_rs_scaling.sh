#!/bin/bash
# K1 at config E against the number of tiles launched (waves per SIMD) and the texel source (development switches of ba_linearize_rs.hip):
#   dbg 0 real taps | 4 one L1-resident line per lane | 12 one line per lane and tile (L2-resident)
for dbg in 0 4 12; do
  for mt in 512 1024 1536 2048 2660 99999; do
    out=$(CMLHIP_RS_DBG=$dbg CMLHIP_RS_MAXTILES=$mt python bench.py --config E --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f' % d['linearize_kernel_us'])")
    echo "dbg=$dbg maxtiles=$mt : K1 $out us"
  done
done
