# rocprofv3 kernel trace of a python command; prints the last-10 average of every m2d kernel.  Usage: trace_cmd.sh <outdir> <script.py> [args...]
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(d.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    if "m2d_" in k:
        last=[x[1] for x in sorted(v)[-10:]]
        print("%-84s n=%3d last10 avg %.1f us"%(k.replace("(anonymous namespace)::","")[:84],len(v),sum(last)/len(last)/1e3))
PY
