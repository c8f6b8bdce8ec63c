"""Per-entry-point HIP-event profile of one eager mn10 training step: python tools/prof_train.py [batch]  (GPU diagnostic)."""
import sys, io, contextlib, collections; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from efficientat_amd import _lib
dev=torch.device('cuda:0')
mel,model=bench.build_model(dev)
import os
if os.environ.get('MODEL'): model=bench.make_train_model(os.environ['MODEL'], dev)   # e.g. MODEL=mn40_bf16
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
wave=(0.1*torch.randn(B,320000,device=dev)).clamp_(-1,1)
y=(torch.rand(B,527,device=dev)<0.005).float()
opt=torch.optim.Adam(model.parameters(),lr=8e-4)
model.train(); mel.train()
def tstep():
    opt.zero_grad(set_to_none=True)
    logits,_=model(mel(wave).unsqueeze(1)); loss=F.binary_cross_entropy_with_logits(logits,y); loss.backward(); opt.step()
for _ in range(2): tstep()
torch.cuda.synchronize()
real=_lib.call; real_rc=_lib.call_rc; rec=[]
def traced(name,*a):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); real(name,*a); e1.record(); rec.append((name,a,e0,e1))
def traced_rc(name,*a):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); rc=real_rc(name,*a); e1.record()
    if rc==0: rec.append((name,a,e0,e1))
    return rc
_lib.call=traced; _lib.call_rc=traced_rc
import time
t0=time.perf_counter(); tstep(); torch.cuda.synchronize(); wall=time.perf_counter()-t0
_lib.call=real; _lib.call_rc=real_rc
agg=collections.defaultdict(lambda:[0,0.0])
for n,a,e0,e1 in rec:
    agg[n][0]+=1; agg[n][1]+=e0.elapsed_time(e1)
tot=sum(v[1] for v in agg.values())
print(f'wall {wall*1e3:.1f} ms, sum of HIP-lib kernels {tot:.1f} ms, launches {len(rec)}')
for n,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1]): print(f'{n:28s} {c:4d} {t:8.2f} ms')
top=sorted(rec,key=lambda r:-r[2].elapsed_time(r[3]))[:14]
for n,a,e0,e1 in top: print(f'  {n:26s} {e0.elapsed_time(e1)*1e3:8.1f} us', [v for v in a if isinstance(v,int) and abs(v)<10**7])
import os
if os.environ.get("EAT_PROF_ALL"):
    want = os.environ["EAT_PROF_ALL"].split(",")
    for n,a,e0,e1 in rec:
        if any(w in n for w in want):
            print(f'  {n:26s} {e0.elapsed_time(e1)*1e3:8.1f} us', [v for v in a if isinstance(v,int) and abs(v)<10**7])
