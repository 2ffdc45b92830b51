"""Pattern pruning of the pipelined retrieval kernel: time and tiles scanned with the option on / off, lists compared.
   python scripts/diag/prune_probe.py [dishes] [E] [users]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, foodrec_amd
I = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
K = int(os.environ.get("PROBE_K", "10"))
U, C = max(1_000_000, n), 4
g = torch.Generator(device="cuda"); g.manual_seed(1)
s = E ** -0.5
PM = torch.randn((U, C + 1, E), generator=g, device="cuda") * s
RE = torch.randn((I, E), generator=g, device="cuda") * s
CE = torch.randn((C, E), generator=g, device="cuda") * s
pat = torch.randint(1, 16, (I,), generator=g, device="cuda", dtype=torch.int32)
cats = ((pat[:, None] >> torch.arange(C, device="cuda", dtype=torch.int32)[None, :]) & 1).float()
eng = foodrec_amd.ScoringEngine(PM, RE, CE); eng.set_dish_categories(cats)
users = torch.randperm(U, generator=g, device="cuda")[:n].to(torch.int32)
if os.environ.get("PROBE_BLOCK"):
    eng.set_option("topk_block", int(os.environ["PROBE_BLOCK"]))    # users per block of the pruned pipelined launch (128 / 256)
if os.environ.get("PROBE_NPROBE"):
    eng.set_option("topk_probes", int(os.environ["PROBE_NPROBE"]))
if os.environ.get("PROBE_REFINE"):
    eng.set_option("topk_refine", int(os.environ["PROBE_REFINE"]))  # near-tied lists finished in the repair's arithmetic (default 1)
if os.environ.get("PROBE_F32"):
    eng.set_option("topk_bf16x3", 0)                      # the exact-f32 kernel
res = {}
import itertools
PRUNES = [int(v) for v in os.environ.get("PROBE_PRUNES", "1").split()]
VARS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "101 116").split()]
for prune, var in [(0, 0), (6, 0), (1, 0)] + [(pr, v) for v in VARS for pr in PRUNES]:      # 6: Cauchy-Schwarz bounds only; 1: + probe rows      # 5: pruned, grid launch order; 1: longest item first
    eng.set_option("topk_prune", prune); eng.set_option("variant", var)
    for _ in range(5):
        s1, i1 = eng.topk_users(users, K)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    for i in range(10):
        ev[i].record(); s1, i1 = eng.topk_users(users, K)
    ev[10].record(); torch.cuda.synchronize(); eng.check()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5]
    res[prune] = (s1, i1)
    print("variant %d " % var, end="")
    print("prune=%d  %.3f ms  %.2f T pairs/s  tiles scanned %d of %d (%.3f)  repaired %d  kernel %s" %
          (prune, ms, n * I / ms / 1e9, eng.get_option("topk_tiles_scanned"), eng.get_option("topk_tiles_full"),
           eng.get_option("topk_tiles_scanned") / max(1, eng.get_option("topk_tiles_full")), eng.get_option("topk_repaired"), eng.last_kernel()))
print("lists identical with and without pruning:", bool(torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[6][1])))
