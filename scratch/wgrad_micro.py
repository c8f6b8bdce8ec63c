import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 128
shapes = [(16,16,32000),(16,64,32000),(64,24,8000),(24,72,8000),(72,24,8000),(72,40,2000),(40,120,2000),(120,40,2000),(40,240,2000),
          (240,80,504),(80,200,504),(200,80,504),(80,184,504),(184,80,504),(80,480,504),(480,112,504),(112,672,504),(672,112,504),
          (672,160,128),(160,960,128),(960,160,128)]
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0
for Ci, Co, S in shapes:
    x = torch.randn(B, Ci, S, 1, device=dev); dz = torch.randn(B, Co, S, 1, device=dev)
    us = timeit(lambda: ops.pw_conv_wgrad(dz, x)); tot += us
    gb = 4 * B * S * (Ci + Co) / 1e9
    ref = torch.einsum('bos,bis->oi', dz[:8, :, :, 0].double(), x[:8, :, :, 0].double())
    got = ops.pw_conv_wgrad(dz[:8].contiguous(), x[:8].contiguous())
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    print(f"{Ci:4d}->{Co:4d} S={S:5d}: {us:7.1f} us  {gb/us*1e3:5.2f} TB/s  rel err {err:.1e}", flush=True)
print('total', tot)
