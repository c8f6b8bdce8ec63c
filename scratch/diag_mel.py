import sys, io, contextlib; sys.path.insert(0,'.')
import torch, numpy as np
from oracle import eat_oracle as O, synth
from efficientat_amd.preprocess import AugmentMelSTFT
dev=torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    mel=AugmentMelSTFT(freqm=0,timem=0).to(dev).eval()
for n,seed in [(32000,77),(320000,1234)]:
    w=synth.parity_clips(n,seed=seed)
    got=mel(w.to(dev)).cpu(); ref=O.mel_forward(w)
    err=(got-ref).abs()
    print(n,'max err per clip',err.amax(dim=(1,2)).tolist())
    for c in range(5):
        i=int(err[c].argmax()); m,t=divmod(i,err.shape[2])
        print('  clip',c,'worst at mel',m,'t',t,'got',float(got[c,m,t]),'ref',float(ref[c,m,t]))
    print('  err by t (first/last 4):',err.amax(dim=(0,1))[:4].tolist(),err.amax(dim=(0,1))[-4:].tolist())
z=mel(torch.zeros(2,320000,device=dev)).cpu()
print('silence',z.min().item(),z.max().item(),(np.float32(np.log(np.float32(1e-5)))+4.5)/5.0, float((torch.log(torch.tensor(1e-5))+4.5)/5))
