# timing experiment: the retrieval leg with variants of libm2d.so (build/exp/), restored afterwards
cp foodrec_amd/libm2d.so /tmp/libm2d_keep.so
for v in r02 exp1 exp2 exp4 exp7 cur; do
  if [ $v = cur ]; then cp /tmp/libm2d_keep.so foodrec_amd/libm2d.so; else cp build/exp/libm2d_$v.so foodrec_amd/libm2d.so; fi
  python3 scripts/prof_mfma.py topk 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', ' | '.join('%s %.3f ms (median %.3f)' % (k, v['event_avg_ms'], v['event_median_ms']) for k, v in d.items()))"
done
cp /tmp/libm2d_keep.so foodrec_amd/libm2d.so
