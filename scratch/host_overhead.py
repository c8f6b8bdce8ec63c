import sys, time; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
import bench
from efficientat_amd import _lib
dev=torch.device('cuda:0')
mel,model=bench.build_model(dev)
B=128
wave=(0.1*torch.randn(B,320000,device=dev)).clamp_(-1,1)
y=(torch.rand(B,527,device=dev)<0.005).float()
opt=torch.optim.Adam(model.parameters(),lr=8e-4)
model.train(); mel.train()
def tstep():
    opt.zero_grad(set_to_none=True)
    logits,_=model(mel(wave).unsqueeze(1)); loss=F.binary_cross_entropy_with_logits(logits,y); loss.backward(); opt.step()
for _ in range(3): tstep()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): tstep()
torch.cuda.synchronize(); print('real step ms', (time.perf_counter()-t0)*100)
real=_lib.call; n=[0]
def noop(name,*a): n[0]+=1
_lib.call=noop
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): tstep()
torch.cuda.synchronize(); print('host-only step ms (HIP lib calls stubbed)', (time.perf_counter()-t0)*100, 'lib calls/step', n[0]/10)
_lib.call=real
# forward eval eager vs graph
model.eval(); mel.eval()
w2=(0.1*torch.randn(256,320000,device=dev))
def f():
    with torch.no_grad(): model(mel(w2).unsqueeze(1))
for _ in range(3): f()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): f()
torch.cuda.synchronize(); print('eval fwd eager ms', (time.perf_counter()-t0)*50)
