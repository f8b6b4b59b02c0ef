#!/bin/bash
# round 6, batch 2: the residual kernel's stores — A/B at config E (base = round 5; new = transposed reduced-record store + lean outputs), then the GPU suite
mkdir -p gpurun_out/r6
cp libcml_amd/libcmlhip.so /tmp/orig.so
run() { python bench.py --config E --no-cpu-baseline --no-extras --steps 60 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('K1 %.2f us  step %.1f us parity %s' % (d['roofline']['launch_us'], 1e3*d['ms_per_step'], d.get('parity_ok')))"; }
for i in 1 2; do
  for v in base new notr; do
    cp ab_tmp/libcmlhip_$v.so libcml_amd/libcmlhip.so
    echo "$v: $(run)"
    if [ $v = new ]; then echo "new, CMLHIP_RS_FULL=1: $(CMLHIP_RS_FULL=1 run)"; fi
  done
done > gpurun_out/r6/b2_ab.txt 2>&1
cp /tmp/orig.so libcml_amd/libcmlhip.so
cat gpurun_out/r6/b2_ab.txt
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r6/b2_pytest.txt
cat gpurun_out/r6/b2_pytest.txt
