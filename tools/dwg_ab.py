"""A/B of the depthwise weight-gradient kernels at the late-layer geometries: EAT_DWP=0/1 python tools/dwg_ab.py  (GPU diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import _lib, ops
dev = torch.device("cuda:0")
def s(): return torch.cuda.current_stream().cuda_stream
cases = [(256, 16, 64, 500, 3, 1), (256, 64, 64, 500, 3, 2), (256, 72, 32, 250, 3, 1), (256, 72, 32, 250, 5, 2), (128, 128, 64, 500, 3, 2),
         (256, 200, 8, 63, 3, 1), (256, 672, 8, 63, 3, 1), (256, 120, 16, 125, 5, 1), (256, 672, 8, 63, 5, 2), (256, 240, 16, 125, 3, 2),
         (256, 960, 4, 32, 5, 1), (128, 1920, 4, 32, 5, 1), (128, 1344, 8, 63, 5, 2), (128, 240, 16, 125, 5, 1)]
for B, C, F, T, k, st in cases:
    Fo, To = ops.conv_out(F, k, st), ops.conv_out(T, k, st)
    x = torch.randn(B, C, F, T, device=dev); dz = torch.randn(B, C, Fo, To, device=dev)
    for per in (0, 1):
        out = torch.zeros((B if per else 1) * C * k * k, device=dev)
        name = "eat_dw_conv_dyn_wgrad" if per else "eat_dw_conv_wgrad"
        def run():
            if per: _lib.call(name, dz.data_ptr(), x.data_ptr(), out.data_ptr(), B, C, F, T, Fo, To, k, st, s())
            else: _lib.call(name, dz.data_ptr(), x.data_ptr(), out.data_ptr(), B, C, C, F, T, Fo, To, k, st, s())
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        gb = (x.numel() + dz.numel()) * 4 / 1e9
        print(f"B={B} C={C} {F}x{T} k{k}s{st} {'per-plane' if per else 'static   '}: {t*1e3:8.1f} us  {gb/t:5.2f} TB/s", flush=True)
