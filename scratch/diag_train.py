import sys, io, contextlib, os; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
from efficientat_amd.mn import get_model
DEV=torch.device('cuda:0')
g=np.load('tests/golden/mn10_ref.npz')
sd=synth.synth_state(synth.mn_shapes(1.0),seed=0)
for k in g.files:
    if k.startswith('bn/'): sd[k[3:]]=torch.from_numpy(g[k])
x=O.mel_forward(synth.parity_clips(320000,seed=1234)).unsqueeze(1)
y=torch.from_numpy(g['train_labels']); keep=torch.from_numpy(g['drop_keep'].astype(np.float32))
def ref(dtype):
    sdr={k:(v.clone().to(dtype).requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(('running_mean','running_var')) else (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone())) for k,v in sd.items()}
    l,_=O.mn_forward(sdr,x.to(dtype),train=True,stats={},drop_mask=keep.to(dtype))
    F.binary_cross_entropy_with_logits(l,y.to(dtype)).backward()
    return sdr
r32=ref(torch.float32); r64=ref(torch.float64)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0)
model.load_state_dict(sd); model.to(DEV).train(); model._drop_mask_override=keep
logits,_=model(x.to(DEV)); F.binary_cross_entropy_with_logits(logits,y.to(DEV)).backward()
def rel(a,b): a=a.double().reshape(-1).cpu(); b=b.double().reshape(-1); return float((a-b).norm()/max(1e-30,float(b.norm())))
rows=[]
for n,p in model.named_parameters():
    if float(r64[n].grad.norm())<1e-9: continue
    rows.append((n, rel(p.grad,r64[n].grad), rel(r32[n].grad,r64[n].grad), rel(p.grad,r32[n].grad)))
print('name  hip_vs_f64  cpu32_vs_f64  hip_vs_cpu32')
for r in sorted(rows,key=lambda r:-r[1])[:14]: print(f'{r[0]:44s} {r[1]:.2e} {r[2]:.2e} {r[3]:.2e}')
a=np.array([[r[1],r[2],r[3]] for r in rows]); print('median',np.median(a,axis=0),'max',a.max(axis=0))
