import sys, io, contextlib, os; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
from efficientat_amd.dymn import get_model
from efficientat_amd.dymn_train import CtxPool, Linear
DEV=torch.device('cuda:0')
def rel(a,b): a=a.double().reshape(-1).cpu(); b=b.double().reshape(-1); return float((a-b).norm()/max(1e-30,float(b.norm())))
# unit: CtxPool backward
x=torch.randn(2,5,6,7).requires_grad_(True)
ref=torch.cat([x.mean(3),x.mean(2)],2).transpose(1,2); w=torch.randn_like(ref); (ref*w).sum().backward()
xd=x.detach().to(DEV).requires_grad_(True); out=CtxPool.apply(xd); (out*w.to(DEV)).sum().backward()
print('ctxpool fwd',rel(out.detach(),ref.detach()),'bwd',rel(xd.grad,x.grad))
# unit: Linear
a=torch.randn(37,24).requires_grad_(True); W=torch.randn(9,24).requires_grad_(True); b=torch.randn(9).requires_grad_(True)
y=F.linear(a,W,b); dy=torch.randn_like(y); y.backward(dy)
ad,Wd,bd=[t.detach().to(DEV).requires_grad_(True) for t in (a,W,b)]
yd=Linear.apply(ad,Wd,bd); yd.backward(dy.to(DEV)); print('linear',rel(yd.detach(),y.detach()),rel(ad.grad,a.grad),rel(Wd.grad,W.grad),rel(bd.grad,b.grad))
g=np.load('tests/golden/dymn10_ref.npz')
sd=synth.synth_state(synth.dymn_shapes(1.0),seed=0)
for k in g.files:
    if k.startswith('bn/'): sd[k[3:]]=torch.from_numpy(g[k])
temp=float(g['temp_train'])
x=O.mel_forward(synth.parity_clips(320000,seed=1234)).unsqueeze(1)
y=torch.from_numpy(g['train_labels']); keep=torch.from_numpy(g['drop_keep'].astype(np.float32))
skip=('running_mean','running_var','num_batches_tracked','lambdas','init_v')
sdr={k:(v.clone().requires_grad_(True) if not k.endswith(skip) else v.clone()) for k,v in sd.items()}
l,_=O.dymn_forward(sdr,x,temperature=temp,train=True,stats={},drop_mask=keep)
F.binary_cross_entropy_with_logits(l,y).backward()
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0)
model.load_state_dict(sd)
for m in model.modules():
    if hasattr(m,'temperature'): m.temperature=temp
model.to(DEV).train(); model._drop_mask_override=keep
logits,_=model(x.to(DEV)); F.binary_cross_entropy_with_logits(logits,y.to(DEV)).backward()
for i in range(15):
    row=[]
    for nm in ['exp_conv.weight','exp_norm.weight','depth_conv.weight','depth_norm.weight','depth_act.coef_net.0.weight','proj_conv.weight','proj_norm.weight','context_gen.joint_conv.weight','context_gen.conv_f.weight','context_gen.conv_t.weight']:
        n=f'layers.{i}.{nm}'
        if n in sdr: row.append(f'{rel(dict(model.named_parameters())[n].grad,sdr[n].grad):.1e}')
        else: row.append('   -   ')
    print(i,' '.join(row))
