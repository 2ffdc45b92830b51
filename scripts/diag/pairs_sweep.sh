# pair kernel across embed sizes, rows of weight-0 categories skipped (default) or fetched
for e in 32 64 128 200 256; do for sk in 1 0; do
python bench.py --embed $e --users 500000 --opt skip_masked=$sk --no-cpu-baseline --no-side --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('E=$e skip=$sk', round(d['value']/1e9,3), 'G pairs/s', round(d['ms_per_step'],4), 'ms', r['bound'], r['frac'] and round(r['frac'],3), d['config']['kernel'])"
done; done
