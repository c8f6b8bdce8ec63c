import sys, io, contextlib, os, importlib; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import eat_oracle as O, synth
import efficientat_amd.mn as mn
DEV=torch.device('cuda:0')
g=np.load('tests/golden/mn10_ref.npz')
sd=synth.synth_state(synth.mn_shapes(1.0),seed=0)
for k in g.files:
    if k.startswith('bn/'): sd[k[3:]]=torch.from_numpy(g[k])
x=O.mel_forward(synth.parity_clips(320000,seed=1234)).unsqueeze(1)
xn=O.mel_forward((0.1*torch.randn(8,320000,generator=torch.Generator().manual_seed(5))).clamp(-1,1)).unsqueeze(1)
with torch.no_grad():
    ref,_=O.mn_forward(sd,x); refn,_=O.mn_forward(sd,xn)
    ref64,_=O.mn_forward({k:(v.double() if v.dtype.is_floating_point else v) for k,v in sd.items()},x.double())
print('cpu fp32 vs fp64 logits err', float((ref.double()-ref64).abs().max()), 'logit absmax', float(ref.abs().max()))
for mode in ['fp32','auto','bf16x3','bf16']:
    mn._PW_MODE=mode
    with contextlib.redirect_stdout(io.StringIO()):
        model=mn.get_model(width_mult=1.0)
    model.load_state_dict(sd); model.to(DEV).eval()
    with torch.no_grad():
        l,_=model(x.to(DEV)); ln,_=model(xn.to(DEV))
    print(mode,'parity clips err vs oracle',float((l.cpu()-ref).abs().max()),'vs f64',float((l.cpu().double()-ref64).abs().max()),'| noise clips err',float((ln.cpu()-refn).abs().max()))
