#!/bin/bash
# K1 duration of the resident residual kernel at config E under the development switches of ba_linearize_rs.hip
# (CMLHIP_RS_LDM: 1 = nontemporal texel loads; CMLHIP_RS_WPB: waves per workgroup; CMLHIP_RS_MAXTILES: launch only the first n tiles;
#  CMLHIP_RS_DBG: 1 = taps at texel 0, 2 = no stores)
cfg=${1:-E}
VARS=${2:-"0,4 1,4"}; for v in $VARS; do wpe=${v%,*}; wpb=${v#*,}; for mt in ${3:-100000 2048 1024 512}; do for dbg in ${4:-0 1}; do
  out=$(CMLHIP_RS_LDM=$wpe CMLHIP_RS_WPB=$wpb CMLHIP_RS_MAXTILES=$mt CMLHIP_RS_DBG=$dbg python bench.py --config $cfg --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f us  step %.1f us' % (d['linearize_kernel_us'], 1e3*d['ms_per_step']))")
  echo "ldm=$wpe wpb=$wpb maxtiles=$mt dbg=$dbg : $out"
done; done; done
