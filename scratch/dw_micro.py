import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256
cases = [(672,8,63,5,2,2,1),(960,4,32,5,1,2,1),(240,16,125,3,2,2,0),(672,8,63,3,1,2,1),(480,8,63,3,1,2,1),(200,8,63,3,1,2,0)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for C, F, T, k, s, act, se in cases:
    x = torch.randn(B, C, F, T, device=dev); w = torch.randn(C, k * k, device=dev); b = torch.zeros(C, device=dev)
    pool = torch.zeros(B, C, device=dev) if se else None
    Fo, To = ops.conv_out(F, k, s), ops.conv_out(T, k, s)
    us = timeit(lambda: ops.dw_conv(x, w, b, k, s, act, pool))
    gb = 4 * B * C * (F * T + Fo * To) / 1e9
    us2 = timeit(lambda: torch.empty_like(x))
    print(f"C {C:4d} {F}x{T} k{k} s{s} se{se}: {us:7.1f} us  {gb:5.2f} GB  {gb/us*1e3:5.2f} TB/s   (alloc {us2:.1f} us)", flush=True)
