# A/B on one box: scripts/diag/prune_probe.py under two builds of libm2d.so.  Usage: ab_probe.sh <other.so> <probe args...>   (env passes through)
other=$1; shift
cp foodrec_amd/libm2d.so /tmp/new.so
for rep in 1 2; do
for lib in $other /tmp/new.so; do
  cp $lib foodrec_amd/libm2d.so
  echo "== $lib"
  python scripts/diag/prune_probe.py "$@" 2>&1 | grep 'prune=1\|prune=0'
done; done
cp /tmp/new.so foodrec_amd/libm2d.so
