import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
def rel(a,b): a=a.double().reshape(-1); b=b.double().reshape(-1); return float((a-b).norm()/max(1e-30,float(b.norm())))
g=np.load('tests/golden/dymn10_ref.npz')
sd=synth.synth_state(synth.dymn_shapes(1.0),seed=0)
for k in g.files:
    if k.startswith('bn/'): sd[k[3:]]=torch.from_numpy(g[k])
temp=float(g['temp_train'])
x=O.mel_forward(synth.parity_clips(320000,seed=1234)).unsqueeze(1)
y=torch.from_numpy(g['train_labels']); keep=torch.from_numpy(g['drop_keep'].astype(np.float32))
skip=('running_mean','running_var','num_batches_tracked','lambdas','init_v')
def run(dt):
    sdr={k:((v.to(dt) if v.dtype.is_floating_point else v).clone().requires_grad_(True) if not k.endswith(skip) else (v.to(dt) if v.dtype.is_floating_point else v).clone()) for k,v in sd.items()}
    l,_=O.dymn_forward(sdr,x.to(dt),temperature=temp,train=True,stats={},drop_mask=keep.to(dt))
    F.binary_cross_entropy_with_logits(l,y.to(dt)).backward(); return sdr
torch.set_num_threads(8)
a=run(torch.float32); b=run(torch.float64)
for n in ['in_c.0.weight','layers.0.depth_conv.weight','layers.2.proj_conv.weight','layers.5.proj_conv.weight','layers.8.proj_conv.weight','layers.11.proj_conv.weight','layers.13.proj_conv.weight','layers.14.proj_conv.weight','out_c.0.weight']:
    print(f'{n:40s} cpu32_vs_f64 {rel(a[n].grad,b[n].grad):.2e}')
