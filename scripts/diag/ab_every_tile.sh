# A/B on one box: the every-tile scan (topk_prune = 0) of two builds of libm2d.so: $1 (a path) against the tree's
cp foodrec_amd/libm2d.so /tmp/new.so
for rep in 1 2; do
for lib in $1 /tmp/new.so; do
  cp $lib foodrec_amd/libm2d.so
  echo "== $lib"
  PROBE_PRUNES="0" PROBE_VARIANTS="0" python scripts/diag/prune_probe.py 100000 64 2>&1 | grep 'prune=0\|prune=1'
done; done
cp /tmp/new.so foodrec_amd/libm2d.so
