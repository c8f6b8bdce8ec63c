import sys, io, contextlib; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
_empty, _empty_like = torch.empty, torch.empty_like
def pe(*a, **k):
    t=_empty(*a,**k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float('nan'))
    return t
def pel(x, **k):
    t=_empty_like(x,**k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float('nan'))
    return t
torch.empty, torch.empty_like = pe, pel
from efficientat_amd.dymn import get_model
import bench
dev=torch.device('cuda:0'); torch.manual_seed(0)
mel,_=bench.build_model(dev)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0).to(dev)
model.train(); mel.train()
B=32
w=(0.1*torch.randn(B,320000,device=dev)).clamp_(-1,1); y=(torch.rand(B,527,device=dev)<0.005).float()
lo,_=model(mel(w).unsqueeze(1)); l=F.binary_cross_entropy_with_logits(lo,y); l.backward()
bad=[n for n,p in model.named_parameters() if not torch.isfinite(p.grad).all()]
print('poisoned eager: loss',float(l),'bad grads',len(bad),bad[:10])
