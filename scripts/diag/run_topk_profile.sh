mkdir -p gpurun_out/r03/p1; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/p1 -- python3 scripts/prof_mfma.py topk > gpurun_out/r03/p1/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r03/p1/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(d.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    if ("m2d_topk" in k or "rocclr" in k):
        last=[x[1] for x in sorted(v)[-10:]]
        print("%-70s n=%3d last10 avg %.1f us"%(k[:70],len(v),sum(last)/len(last)/1e3))
PY
rm -rf gpurun_out/r03/p1/*/*.db
python3 scripts/prof_mfma.py topk | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items(): print(k, 'event avg %.3f ms median %.3f frac %.3f repaired %s' % (v['event_avg_ms'], v['event_median_ms'], v['frac_of_peak'], v['repaired_users']))"
