mkdir -p gpurun_out/r04/p2; export TMPDIR=/tmp
cat > /tmp/pp.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, foodrec_amd
I, E, n, var = int(sys.argv[1]), int(sys.argv[2]), 65536, int(sys.argv[3])
U, C = 1_000_000, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
RE = torch.randn((I, E), generator=g, device="cuda") * s
CE = torch.randn((C, E), generator=g, device="cuda") * s
pat = torch.randint(1, 16, (I,), generator=g, device="cuda", dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
users = torch.randperm(U, generator=g, device="cuda")[:n].to(torch.int32)
eng.set_option("variant", var)
eng.set_option("topk_prune", int(os.environ.get("PRUNE", "1")))
for _ in range(12):
    eng.topk_users(users, 10)
eng.check()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/p2 -- python3 /tmp/pp.py $1 $2 $3 > gpurun_out/r04/p2/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04/p2/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(d.items(), key=lambda kv:-sum(x[1] for x in kv[1])):
    if ("m2d_" in k or "rocclr" in k):
        last=[x[1] for x in sorted(v)[-10:]]
        print("%-70s n=%3d last10 avg %.1f us"%(k[:70],len(v),sum(last)/len(last)/1e3))
PY
rm -rf gpurun_out/r04/p2
