mkdir -p gpurun_out/r03/p3; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/p3 -- python3 bench.py --workload train --learner sgd --steps 200 > gpurun_out/r03/p3/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r03/p3/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])))
allk=sorted((s,e,k) for k,v in d.items() for s,e in v if "m2d_train" in k)
for k,v in d.items():
    if "m2d_train" in k:
        dur=[e-s for s,e in v][-100:]
        print("%-60s n=%4d avg %.1f us"%(k[:60],len(v),sum(dur)/len(dur)/1e3))
gaps=[allk[i+1][0]-allk[i][1] for i in range(len(allk)-200,len(allk)-1)]
print("avg gap between consecutive train kernels %.1f us"%(sum(gaps)/len(gaps)/1e3))
PY
rm -rf gpurun_out/r03/p3
for l in sgd adam; do python bench.py --workload train --learner $l --steps 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['roofline']['frac'])"; done
