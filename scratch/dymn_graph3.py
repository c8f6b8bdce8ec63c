import sys, io, contextlib; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
import bench
from efficientat_amd.dymn import get_model
from efficientat_amd.graphs import GraphedTrainStep
dev=torch.device('cuda:0'); torch.manual_seed(0)
mel,_=bench.build_model(dev)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0).to(dev)
reinit = sys.argv[2]=='reinit'
if reinit:
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan_in = m.weight.shape[1]*m.weight.shape[2]*m.weight.shape[3]; m.weight.normal_(0,(2.0/fan_in)**0.5)
model.train(); mel.train()
B=int(sys.argv[1])
w=(0.1*torch.randn(B,320000,device=dev)).clamp_(-1,1); y=(torch.rand(B,527,device=dev)<0.005).float()
opt=torch.optim.Adam(model.parameters(),lr=8e-4,capturable=True)
g=GraphedTrainStep(model,opt,F.binary_cross_entropy_with_logits,mel(w).unsqueeze(1),y)
for i in range(8):
    l=g(mel(w).unsqueeze(1),y); torch.cuda.synchronize()
    badp=[n for n,p in model.named_parameters() if not torch.isfinite(p).all()]
    badg=[n for n,p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print('B',B,'reinit',reinit,'replay',i,'loss',float(l),'bad params',len(badp),badp[:2],'bad grads',len(badg),badg[:2])
    if badg:
        n=badg[0]; pg=dict(model.named_parameters())[n].grad
        idx=(~torch.isfinite(pg)).nonzero().flatten()
        print('  bad grad',n,'shape',tuple(pg.shape),'n bad',idx.numel(),'idx',idx[:8].tolist(),'vals',pg[idx[:4]].tolist())
        wn=n.replace('bias','weight'); wg=dict(model.named_parameters())[wn].grad
        print('  weight grad finite:',bool(torch.isfinite(wg).all()),'rows nonfinite', (~torch.isfinite(wg).all(dim=tuple(range(1,wg.dim())))).nonzero().flatten()[:8].tolist())
        break
