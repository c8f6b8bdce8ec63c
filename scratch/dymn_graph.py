import sys, io, contextlib; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from efficientat_amd.dymn import get_model
from efficientat_amd.dymn_train import _block_train
dev=torch.device('cuda:0'); torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    model=get_model(width_mult=1.0).to(dev)
model.train()
def run_block(i, shape):
    blk=model.layers[i]
    x=torch.randn(*shape,device=dev).requires_grad_(True)
    def step():
        for p in blk.parameters(): p.grad=None
        x.grad=None
        out=_block_train(blk,x); loss=out.square().mean(); loss.backward(); return loss.detach()
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): l0=step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref={n:p.grad.clone() for n,p in blk.named_parameters()}
    g=torch.cuda.CUDAGraph()
    for p in blk.parameters(): p.grad=None
    x.grad=None
    with torch.cuda.graph(g):
        out=_block_train(blk,x); loss=out.square().mean(); loss.backward()
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        bad=[n for n,p in blk.named_parameters() if not torch.isfinite(p.grad).all()]
        worst=max(float((p.grad-ref[n]).abs().max()/(ref[n].abs().max()+1e-12)) for n,p in blk.named_parameters() if n not in bad)
        print('block',i,'replay',r,'loss',float(loss),'eager loss',float(l0),'nonfinite',bad[:4],'worst rel diff',worst)
run_block(13,(4,160,4,32))
run_block(1,(4,16,64,500))
