import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, foodrec_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
C, E, U, I = 4, 64, 200_000, 100_000
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device=dev) * s
RE = torch.randn((I, E), generator=g, device=dev) * s
CE = torch.randn((C, E), generator=g, device=dev) * s
pat = torch.randint(1, 16, (I,), generator=g, device=dev, dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device=dev, dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
for kv in sys.argv[1:]:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for n in (256, 1024, 8192):
    users = torch.randperm(U, generator=g, device=dev)[:n].to(torch.int32)
    for _ in range(3): eng.topk_users(users, 10)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); eng.topk_users(users, 10); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print(n, "host call time median %.1f us" % (sorted(ts)[10] * 1e6))
