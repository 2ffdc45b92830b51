import sys, os
sys.path.insert(0, os.getcwd())
import torch, foodrec_amd
I, E, n = int(sys.argv[1]), int(sys.argv[2]), 65536
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
for _ in range(12):
    eng.topk_users(users, 10)
eng.check()
print("refined", eng.get_option("topk_refined"), "sent to the repair", eng.get_option("topk_refine_repaired"), "tie-repaired", eng.get_option("topk_repaired"))
