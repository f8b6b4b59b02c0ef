#!/bin/bash
# A/B of libcmlhip.so variants (ab_tmp/libcmlhip_<name>.so) at config E: K1 and the step, two rounds.   bash tools/ab_e.sh new touch ...
cp libcml_amd/libcmlhip.so /tmp/orig.so
run() { python bench.py --config ${AB_CONFIG:-E} --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('K1 %.2f us  step %.1f us parity %s' % (d['roofline']['launch_us'], 1e3*d['ms_per_step'], d.get('parity_ok')))"; }
for i in 1 2; do
  for v in "$@"; do
    cp ab_tmp/libcmlhip_$v.so libcml_amd/libcmlhip.so
    echo "$v: $(run)"
  done
done
cp /tmp/orig.so libcml_amd/libcmlhip.so
