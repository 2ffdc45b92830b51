"""How many users of a retrieval call take the tie-repair path, and does a user's list depend on the batch it is in?
   python scripts/diag/tie_flags_probe.py [dishes] [E]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, foodrec_amd
I = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
U, C = 200000, 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
RE = torch.randn((I, E), generator=g, device="cuda") * s
CE = torch.randn((C, E), generator=g, device="cuda") * s
pat = torch.randint(1, 16, (I,), generator=g, device="cuda", dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
big = torch.randperm(U, generator=g, device="cuda")[:65536].to(torch.int32)
for x3, form in ((1, 0), (1, 1), (0, 0)):
  eng.set_option("topk_bf16x3", x3); eng.set_option("topk_form", form)
  print("--- topk_bf16x3 = %d, topk_form = %d" % (x3, form))
  for n in (65536, 64, 256, 4096):
    s1, i1 = eng.topk_users(big[:n], 10); eng.check()
    if n == 65536:
        S, Iid = s1, i1
    d = (s1 != S[:n]).any(1)
    print("n=%d kernel=%s repaired users %d; rows whose scores differ from the 65536-user call: %d, ids differ: %d, max |ds| %.3e" %
          (n, eng.last_kernel(), eng.get_option("topk_repaired"), int(d.sum()), int((i1 != Iid[:n]).any(1).sum()), float((s1 - S[:n]).abs().max())))
