import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np, io, contextlib
from oracle import eat_oracle as O, synth
from efficientat_amd import mn
from efficientat_amd.mn import get_model
import test_gpu_parity as T
sd, g = T._calibrated_state('tests/golden')
wave = synth.parity_clips(320000, seed=1234)
x_ref = O.mel_forward(wave).unsqueeze(1)
with torch.no_grad():
    ref_logits, ref_fmaps = O.mn_forward(sd, x_ref, return_fmaps=True)
    _, ref_feat = O.mn_forward(sd, x_ref)
for mode in ['fp32','auto','bf16x3']:
    mn._PW_MODE = mode
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model(width_mult=1.0)
    model.load_state_dict(sd, strict=True); model.to('cuda:0').eval()
    with torch.no_grad():
        logits, fmaps = model._forward_impl(x_ref.to('cuda:0'), return_fmaps=True)
        _, feat = model(x_ref.to('cuda:0'))
    fe = [float((a.cpu()-b).abs().max())/max(1.0,float(b.std())) for a,b in zip(fmaps, ref_fmaps)]
    print(mode, 'logit err', float((logits.cpu()-ref_logits).abs().max()), 'vs golden', float(np.abs(logits.cpu().numpy()-g['eval_logits']).max()),
          'feat err', float((feat.cpu()-ref_feat).abs().max()), 'fmap rel max', max(fe), ['%.1e'%e for e in fe])
