import sys, io, contextlib; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from efficientat_amd.dymn import get_model
from efficientat_amd.graphs import GraphedTrainStep
dev=torch.device('cuda:0'); torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0).to(dev)
model.train()
B=8
x=torch.randn(B,1,128,1000,device=dev); y=(torch.rand(B,527,device=dev)<0.01).float()
cap = sys.argv[1]=='cap'
opt=torch.optim.Adam(model.parameters(),lr=8e-4,capturable=cap)
if cap:
    g=GraphedTrainStep(model,opt,F.binary_cross_entropy_with_logits,x,y)
    for i in range(5):
        l=g(x,y); torch.cuda.synchronize()
        badp=[n for n,p in model.named_parameters() if not torch.isfinite(p).all()]
        badg=[n for n,p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        print('replay',i,'loss',float(l),'bad params',len(badp),badp[:3],'bad grads',len(badg),badg[:3])
else:
    for i in range(9):
        opt.zero_grad(set_to_none=True); lo,_=model(x); l=F.binary_cross_entropy_with_logits(lo,y); l.backward(); opt.step()
        print('eager',i,float(l))
