import sys, io, contextlib; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
from efficientat_amd.dymn import get_model
from efficientat_amd.dymn_train import _block_train
DEV=torch.device('cuda:0')
def rel(a,b): a=a.double().reshape(-1).cpu(); b=b.double().reshape(-1); return float((a-b).norm()/max(1e-30,float(b.norm())))
sd=synth.synth_state(synth.dymn_shapes(1.0),seed=0)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0)
model.load_state_dict(sd); model.to(DEV).train()
blocks,_=O.block_table(1.0)
shapes={0:(64,500),1:(64,500),3:(32,250),6:(16,125),12:(8,63),13:(4,32),5:(16,125)}
for i in [13,12,6,5,3,1,0]:
    c=blocks[i]; H=O.context_dim(c['cexp'],1.0); Fq,T=shapes[i]
    B=3
    x=torch.randn(B,c['cin'],Fq,T,generator=torch.Generator().manual_seed(i))
    for temp in (30.0,):
        skip=('running_mean','running_var','num_batches_tracked','lambdas','init_v')
        sdr={k:(v.clone().requires_grad_(True) if not k.endswith(skip) else v.clone()) for k,v in sd.items() if k.startswith(f'layers.{i}.')}
        xr=x.clone().requires_grad_(True)
        out_ref=O._dy_block(sdr,f'layers.{i}',xr,c,H,True,{},temp)
        dout=torch.randn(out_ref.shape,generator=torch.Generator().manual_seed(99))
        out_ref.backward(dout)
        blk=model.layers[i]
        for m in blk.modules():
            if hasattr(m,'temperature'): m.temperature=temp
        for p in blk.parameters(): p.grad=None
        xd=x.to(DEV).requires_grad_(True)
        out=_block_train(blk,xd); out.backward(dout.to(DEV))
        worst=max(((rel(p.grad,sdr[f'layers.{i}.{n}'].grad),n) for n,p in blk.named_parameters() if float(sdr[f'layers.{i}.{n}'].grad.norm())>1e-7))
        print(f'block {i} s{c["stride"]} k{c["k"]} out {rel(out.detach(),out_ref.detach()):.1e} dx {rel(xd.grad,xr.grad):.1e} worst param {worst[0]:.1e} {worst[1]}')
