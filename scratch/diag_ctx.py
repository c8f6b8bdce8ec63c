import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from oracle import eat_oracle as O, synth
def rel(a,b): a=a.double().reshape(-1); b=b.double().reshape(-1); return float((a-b).norm()/max(1e-30,float(b.norm())))
sd=synth.synth_state(synth.dymn_shapes(1.0),seed=0)
blocks,_=O.block_table(1.0)
i=12; c=blocks[i]; H=O.context_dim(c['cexp'],1.0); Fq,T=8,63; B=3; stride=c['stride']
x=torch.randn(B,c['cin'],Fq,T,generator=torch.Generator().manual_seed(i))
p=f'layers.{i}.context_gen.'
def run(mode, dev='cpu'):
    W={k:sd[p+k].clone().to(dev).requires_grad_(True) for k in ['joint_conv.weight','joint_norm.weight','joint_norm.bias','conv_f.weight','conv_f.bias','conv_t.weight','conv_t.bias']}
    xr=x.clone().to(dev).requires_grad_(True)
    if mode=='oracle':
        cf, ct = xr.mean(dim=3, keepdim=True), xr.mean(dim=2, keepdim=True).permute(0, 1, 3, 2)
        g = F.conv2d(torch.cat([cf, ct], dim=2), W['joint_conv.weight'])
        g = F.hardswish(F.batch_norm(g, None, None, W['joint_norm.weight'], W['joint_norm.bias'], True, 0.01, 1e-3))
        h_cf, h_ct = g[:, :, :Fq], g[:, :, Fq:].permute(0, 1, 3, 2)
        h_c = g.mean(dim=2).reshape(B, H)
        if stride>1:
            h_cf = F.avg_pool2d(h_cf, (3, 1), (stride, 1), (1, 0)); h_ct = F.avg_pool2d(h_ct, (1, 3), (1, stride), (0, 1))
        g_cf = F.conv2d(h_cf, W['conv_f.weight'], W['conv_f.bias']); g_ct = F.conv2d(h_ct, W['conv_t.weight'], W['conv_t.bias'])
        outs=(h_c, g_cf.squeeze(3).permute(0,2,1), g_ct.squeeze(2).permute(0,2,1))
    else:
        L=Fq+T
        if mode=='hip':
            from efficientat_amd.dymn_train import CtxPool, Linear
            lin=lambda a,w,b=None: Linear.apply(a,w,b)
            seq=CtxPool.apply(xr)
        else:
            lin=F.linear
            seq=torch.cat([xr.mean(3),xr.mean(2)],2).transpose(1,2)
        gj=lin(seq.reshape(B*L,-1), W['joint_conv.weight'].flatten(1))
        gj=F.batch_norm(gj,None,None,W['joint_norm.weight'],W['joint_norm.bias'],True,0.01,1e-3)
        g=F.hardswish(gj).view(B,L,H)
        h_c=g.mean(1); h_cf,h_ct=g[:,:Fq],g[:,Fq:]
        if stride>1:
            pool=lambda t: F.avg_pool1d(t.transpose(1,2),3,stride,1).transpose(1,2)
            h_cf,h_ct=pool(h_cf),pool(h_ct)
        Fo,To=h_cf.shape[1],h_ct.shape[1]
        g_cf=lin(h_cf.reshape(B*Fo,H),W['conv_f.weight'].flatten(1),W['conv_f.bias']).view(B,Fo,-1)
        g_ct=lin(h_ct.reshape(B*To,H),W['conv_t.weight'].flatten(1),W['conv_t.bias']).view(B,To,-1)
        outs=(h_c,g_cf,g_ct)
    gen=torch.Generator().manual_seed(5)
    loss=sum((o*torch.randn(o.shape,generator=gen).to(dev)).sum() for o in outs)
    loss.backward()
    return [o.detach().cpu() for o in outs],xr.grad.cpu(),{k:v.grad.cpu() for k,v in W.items()}
o1,dx1,g1=run('oracle'); o2,dx2,g2=run('mine')
print('outs',[rel(a,b) for a,b in zip(o2,o1)],'dx',rel(dx2,dx1))
for k in g1: print(k, rel(g2[k].reshape(g1[k].shape),g1[k]))

o3,dx3,g3=run('hip','cuda:0')
print('HIP outs',[rel(a,b) for a,b in zip(o3,o1)],'dx',rel(dx3,dx1))
for k in g1: print(k, rel(g3[k].reshape(g1[k].shape),g1[k]))
