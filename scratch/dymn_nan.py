import sys, io, contextlib; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from efficientat_amd.dymn import get_model
from efficientat_amd.preprocess import AugmentMelSTFT
dev=torch.device('cuda:0'); torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=2.0).to(dev); mel=AugmentMelSTFT(freqm=0,timem=0).to(dev)
B=16
w=(0.1*torch.randn(B,320000,device=dev)).clamp_(-1,1); y=(torch.rand(B,527,device=dev)<0.005).float()
opt=torch.optim.Adam(model.parameters(),lr=8e-4)
model.train(); mel.train()
for it in range(6):
    opt.zero_grad(set_to_none=True)
    logits,_=model(mel(w).unsqueeze(1)); loss=F.binary_cross_entropy_with_logits(logits,y); loss.backward()
    bad=[n for n,p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    gn=max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None and torch.isfinite(p.grad).all())
    print(it,'loss',loss.item(),'logit absmax',float(logits.abs().max()),'nonfinite grads',len(bad),bad[:3],'max|g|',gn)
    opt.step()
