import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 128
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0
for C, F_, T, k, s in [(16,64,500,3,1),(64,64,500,3,2),(72,32,250,3,1),(72,32,250,5,2),(120,16,125,5,1),(240,16,125,3,2),(200,8,63,3,1),(480,8,63,3,1),(672,8,63,3,1),(672,8,63,5,2),(960,4,32,5,1)]:
    Fo, To = ops.conv_out(F_, k, s), ops.conv_out(T, k, s)
    x = torch.randn(B, C, F_, T, device=dev); dz = torch.randn(B, C, Fo, To, device=dev)
    us = timeit(lambda: ops.dw_conv_wgrad(dz, x, k, s)); tot += us
    got = ops.dw_conv_wgrad(dz[:4].contiguous(), x[:4].contiguous(), k, s)
    xd = torch.nn.functional.unfold(x[:4].double().reshape(4 * C, 1, F_, T), k, padding=(k - 1) // 2, stride=s)   # (4C, k*k, Fo*To)
    ref = (xd * dz[:4].double().reshape(4 * C, 1, Fo * To)).sum(-1).reshape(4, C, k * k).sum(0)
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    gb = 4 * B * C * (F_ * T + Fo * To) / 1e9
    print(f"C{C:4d} {F_}x{T} k{k}s{s}: {us:7.1f} us {gb/us*1e3:5.2f} TB/s  rel err {err:.1e}", flush=True)
print('total', tot)
