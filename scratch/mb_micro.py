"""Microbenchmark of the fused block kernel on the early mn10 blocks (B=256)."""
import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256
# (Cin, Cexp, Cout, F, T, k, s, act, se)
blocks = [(16,64,24,64,500,3,2,1,0),(24,72,24,32,250,3,1,1,0),(24,72,40,32,250,5,2,1,1),(40,120,40,16,125,5,1,1,1),
          (40,240,80,16,125,3,2,2,0)]
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for Ci, Ce, Co, F_, T, k, s, act, se in blocks:
    x = torch.randn(B, Ci, F_, T, device=dev)
    we = ops.pw_prepack(torch.randn(Ce, Ci, device=dev) / Ci ** 0.5, None); be = torch.zeros(Ce, device=dev)
    wd = torch.randn(Ce, k * k, device=dev) / k; bd = torch.zeros(Ce, device=dev)
    wp = ops.pw_prepack(torch.randn(Co, Ce, device=dev) / Ce ** 0.5, None); bp = torch.zeros(Co, device=dev)
    Fo, To = ops.conv_out(F_, k, s), ops.conv_out(T, k, s)
    if se:
        pool = torch.zeros(B, Ce, device=dev)
        f = lambda: ops.fused_expand_dw(x, we, be, wd, bd, Ce, k, s, act, pool)
        actual = 4 * B * (Ci * F_ * T + Ce * Fo * To)
    else:
        f = lambda: ops.mbconv(x, we, be, wd, bd, wp, bp, Ce, Co, k, s, act, res=x if (s == 1 and Ci == Co) else None)
        actual = 4 * B * (Ci * F_ * T * (2 if (s == 1 and Ci == Co) else 1) + Co * Fo * To)
    us = timeit(f)
    print(f"Cin {Ci:3d} Cexp {Ce:3d} Cout {Co:3d} {F_}x{T} k{k} s{s} se{se}: {us:7.1f} us  actual traffic {actual/1e9:5.2f} GB -> {actual/us*1e-6:5.2f} TB/s", flush=True)
