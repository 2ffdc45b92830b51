# A/B on one box: a command under two builds of libm2d.so -- $1 (a path) and the tree's -- twice each.  Usage: ab_libs.sh <other.so> <command...>
other=$1; shift
cp foodrec_amd/libm2d.so /tmp/new.so
for rep in 1 2; do
for lib in $other /tmp/new.so; do
  cp $lib foodrec_amd/libm2d.so
  echo "== $lib"
  "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step', round(d['ms_per_step'],4), 'value', d['value'])"
done; done
cp /tmp/new.so foodrec_amd/libm2d.so
