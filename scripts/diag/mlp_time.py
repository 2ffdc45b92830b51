import torch, time, sys
sys.path.insert(0,'.')
import foodrec_amd
dev=torch.device('cuda',0); g=torch.Generator(device=dev); g.manual_seed(1)
U,I,E,C=200000,100000,128,4
s=E**-0.5
PM=torch.randn((U,C+1,E),generator=g,device=dev)*s; RE=torch.randn((I,E),generator=g,device=dev)*s; CE=torch.randn((C,E),generator=g,device=dev)*s
pat=torch.randint(1,16,(I,),generator=g,device=dev,dtype=torch.int32)
cats=((pat[:,None]>>torch.arange(C,device=dev,dtype=torch.int32)[None,:])&1).float()
eng=foodrec_amd.ScoringEngine(PM,RE,CE); eng.set_dish_categories(cats)
K=(C+1)*E; rn=lambda *sh: torch.randn(sh,generator=g,device=dev)
eng.set_mlp_head(rn(K,256)/K**0.5, rn(256)*0.1, rn(256,64)/16, rn(64)*0.1, rn(64)/8, 0.0)
B=1<<22
users=torch.randint(0,U,(B,),generator=g,device=dev,dtype=torch.int32); items=torch.randint(0,I,(B,),generator=g,device=dev,dtype=torch.int32)
out=torch.empty(B,device=dev)
for _ in range(3): eng.score_pairs_mlp(users,items,out=out)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(10): eng.score_pairs_mlp(users,items,out=out)
torch.cuda.synchronize(); print('ms/step', (time.perf_counter()-t)/10*1e3)
