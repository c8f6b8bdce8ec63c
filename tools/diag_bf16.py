"""Where does the HIP bf16 train step leave the oracle's emulation of the same arithmetic? (GPU diagnostic)"""
import contextlib, io, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
from efficientat_amd import mn as mn_mod, ops
DEV = torch.device("cuda:0")
def q(fn,*a,**k):
    with contextlib.redirect_stdout(io.StringIO()): return fn(*a,**k)
W = float(os.environ.get("W", "1.0"))
wave = synth.parity_clips(320000, seed=21)
x = O.mel_forward(wave).unsqueeze(1)
fwd = lambda sd, xm, **k: O.mn_forward(sd, xm, width_mult=W, **k)
sd = synth.calibrate(synth.synth_state(synth.mn_shapes(W), seed=0), fwd, x)
# eval-mode per-block comparison: HIP bf16 eval vs emulated oracle eval vs fp32 oracle
with torch.no_grad():
    l32, f32 = fwd(sd, x, return_fmaps=True)
    with O.emulate_bf16_pointwise():
        le, fe = fwd(sd, x, return_fmaps=True)
mn_mod._PW_MODE = "bf16"
model = q(mn_mod.get_model, width_mult=W); model.load_state_dict(sd); model.to(DEV).eval()
with torch.no_grad():
    lh, fh = model._forward_impl(x.to(DEV), return_fmaps=True)
rel = lambda a,b: float((a.cpu().double()-b.double()).norm()/b.double().norm())
print("eval (note: eval folds BN into the weights BEFORE bf16 rounding in HIP; the emulation rounds the raw weights)")
for i,(a,b,c) in enumerate(zip(fh,fe,f32)):
    print(f"  fmap {i:2d}: hip-vs-emu {rel(a,b):.3e}  emu-vs-fp32 {rel(b,c):.3e}  hip-vs-fp32 {rel(a,c):.3e}")
print("  logits: hip-vs-emu", float((lh.cpu()-le).abs().max()), "emu-vs-fp32", float((le-l32).abs().max()))
# train-mode forward: logits + first-layer checks
mn_mod._PW_MODE = "auto"
y = (torch.rand(5, 527, generator=torch.Generator().manual_seed(2)) < 0.01).float()
keep = torch.ones(5, model.classifier[2].out_features)
def gstate():
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and not k.endswith(("running_mean","running_var")) else v.clone()) for k,v in sd.items()}
res = {}
for name, ctx in (("fp32", contextlib.nullcontext()), ("emu", O.emulate_bf16_pointwise())):
    s = gstate()
    with ctx:
        lg,_ = fwd(s, x, train=True, stats={}, drop_mask=keep)
        F.binary_cross_entropy_with_logits(lg, y).backward()
    res[name] = (lg.detach(), {k:v.grad for k,v in s.items() if getattr(v,"grad",None) is not None})
for prec in ("fp32", "bf16"):
    model = q(mn_mod.get_model, width_mult=W); model.load_state_dict(sd); model.to(DEV).train()
    model.train_precision = prec; model._drop_mask_override = keep
    lg,_ = model(x.to(DEV)); F.binary_cross_entropy_with_logits(lg, y.to(DEV)).backward()
    for ref in ("fp32","emu"):
        rl, rg = res[ref]
        gmax = max(float(g.norm()) for g in rg.values())
        names = [n for n,p in model.named_parameters() if float(rg[n].norm()) > 1e-5*gmax]
        r = np.array([rel(dict(model.named_parameters())[n].grad, rg[n]) for n in names])
        print(f"train hip-{prec} vs oracle-{ref}: logits max diff {float((lg.detach().cpu()-rl).abs().max()):.3e} (|l| {float(rl.abs().max()):.2f}); grad rel median {np.median(r):.3e} max {r.max():.3e}")
        if prec=="bf16" and ref=="emu":
            for n,v in list(zip(names,r))[::12]: print(f"      {n:50s} {v:.3e}")
