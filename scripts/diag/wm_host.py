import sys, time; sys.path.insert(0,'.')
import torch, foodrec_amd
dev=torch.device('cuda',0); g=torch.Generator(device=dev); g.manual_seed(3)
U,I,C,E,L,B=64657,4548,4,200,95,256
s=E**-0.5
PM=torch.randn((U,C+1,E),generator=g,device=dev)*s; RE=torch.randn((I,E),generator=g,device=dev)*s; CE=torch.randn((C,E),generator=g,device=dev)*s
GM=torch.randn((L,C+1,E),generator=g,device=dev)*s
eng=foodrec_amd.ScoringEngine(PM,RE,CE)
users=torch.randint(0,U,(B,),generator=g,device=dev,dtype=torch.int32); items=torch.randint(0,I,(B,),generator=g,device=dev,dtype=torch.int32)
cats=torch.ones((B,C),device=dev); sign=torch.ones((B,1),device=dev); labels=(torch.rand((B,L),generator=g,device=dev)<0.05).float()
for kw in (dict(write_pm=False,write_gm=True), dict(write_pm=True,write_gm=False)):
    for _ in range(20): eng.write_memory(users,items,cats,sign,labels,GM,0.5,0.5,0.001,**kw)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(500): eng.write_memory(users,items,cats,sign,labels,GM,0.5,0.5,0.001,**kw)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(kw, 'host us/call', (t1-t)/500*1e6, 'wall us/call', (t2-t)/500*1e6)
