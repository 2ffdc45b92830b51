"""One round of the sharded path as a single call (users x dishes, E = 64, k = 10), for a kernel trace:
   scripts/diag/trace_cmd.sh <outdir> scripts/diag/round_probe.py [users] [dishes]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, foodrec_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
I = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
E, C = 64, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
s = E ** -0.5
PM = torch.randn((n, C + 1, E), generator=g, device="cuda") * s
RE = torch.randn((I, E), generator=g, device="cuda") * s
CE = torch.randn((C, E), generator=g, device="cuda") * s
pat = torch.randint(1, 16, (I,), generator=g, device="cuda", dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
users = torch.arange(n, dtype=torch.int32, device="cuda")
for _ in range(6):
    eng.topk_users(users, 10)
eng.check()
print("refined", eng.get_option("topk_refined"), "sent to the repair", eng.get_option("topk_refine_repaired"), "tie-repaired", eng.get_option("topk_repaired"),
      "tiles", eng.get_option("topk_tiles_scanned") / eng.get_option("topk_tiles_full"))
