# Run ON THE GPU BOX: kernel trace of a small retrieval call (users = $1, default 1) at 100 k dishes, E = 64.
mkdir -p gpurun_out/r03/p4; export TMPDIR=/tmp
cat > /tmp/ps.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, foodrec_amd
I, E, n = 100000, 64, int(sys.argv[1])
U, C = 200_000, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
RE = torch.randn((I, E), generator=g, device="cuda") * s
CE = torch.randn((C, E), generator=g, device="cuda") * s
pat = torch.randint(1, 16, (I,), generator=g, device="cuda", dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
users = torch.randperm(U, generator=g, device="cuda")[:n].to(torch.int32)
for _ in range(30):
    eng.topk_users(users, 10)
torch.cuda.synchronize(); eng.check()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/p4 -- python3 /tmp/ps.py ${1:-1} > gpurun_out/r03/p4/log.txt 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03/p4/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
# the last call: from the last user_plan launch on
idx=[i for i,r in enumerate(rows) if "user_plan" in r[2]]
last=rows[idx[-1]:]
t0=last[0][0]
for s,e,k in last:
    print("%8.1f us  +%6.1f us  %s" % ((s-t0)/1e3,(e-s)/1e3,k.replace("(anonymous namespace)::","")[:70]))
print("call: %.1f us from first start to last end" % ((last[-1][1]-t0)/1e3))
PY
rm -rf gpurun_out/r03/p4
