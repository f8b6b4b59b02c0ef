for st in 0 2 4 6 8 12; do
  out=$(CMLHIP_RS_DBG=$((st*256)) python bench.py --config E --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%.2f us  step %.1f us' % (d['linearize_kernel_us'], 1e3*d['ms_per_step']))")
  echo "stagger=$st : $out"
done
