"""Microbenchmark of the 1x1 kernels on the late mn10 layer shapes (B=256)."""
import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256
import os
if os.environ.get('EARLY'): shapes_early = True
shapes = [(16,16,32000),(16,64,32000),(64,24,8000),(24,72,8000),(72,24,8000),(72,40,2000),(40,120,2000),(120,40,2000)] if os.environ.get('EARLY') else [(40,240,2000),(240,80,504),(80,200,504),(200,80,504),(80,184,504),(184,80,504),(80,480,504),(480,112,504),
          (112,672,504),(672,112,504),(672,160,128),(160,960,128),(960,160,128)]
modes = sys.argv[1:] or ['fp32', 'bf16x3', 'bf16']
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for Ci, Co, S in shapes:
    x = torch.randn(B, Ci, S, 1, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    bias = torch.zeros(Co, device=dev)
    gb = 4 * B * S * (Ci + Co) / 1e9; gf = 2 * B * S * Ci * Co / 1e9
    ref = None; line = f"{Ci:4d}->{Co:4d} S={S:4d} {gb:5.2f} GB hbm_floor {gb/5.5e3*1e6:6.1f} us |"
    for m in modes:
        if m == 'fp32':
            wp = ops.pw_prepack(w, None); f = lambda: ops.pw_conv(x, wp, bias, Co, 0)
        else:
            wp = ops.pw_prepack_bf16(w, None, split=(m == 'bf16x3')); f = lambda: ops.pw_conv_bf16(x, wp, bias, Co, 0, m == 'bf16x3')
        y = f(); us = timeit(f)
        if ref is None: ref = y
        err = float((y - ref).abs().max())
        line += f" {m} {us:6.1f} us {gb/us*1e3:5.2f} TB/s {gf/us*1e-3:5.1f} TF err {err:.1e} |"
    print(line, flush=True)
