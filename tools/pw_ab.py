"""A/B of the bf16x3 1x1 kernels on the mn10 layer shapes at B = 256 (tools, not product):
LDS-staged kernel (conv_pw_bf16.hip) vs the barrier-free kernels (conv_pw_stream.hip), HIP events on one stream.
  python tools/pw_ab.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEV = "cuda"
# (Ci, Co, F, T, act, se, res)
LAYERS = [(40, 120, 16, 125, 1, 0, 0), (120, 40, 16, 125, 0, 1, 1), (40, 240, 16, 125, 2, 0, 0),
          (240, 80, 8, 63, 0, 0, 0), (80, 200, 8, 63, 2, 0, 0), (200, 80, 8, 63, 0, 0, 1), (80, 184, 8, 63, 2, 0, 0),
          (184, 80, 8, 63, 0, 0, 1), (80, 480, 8, 63, 2, 0, 0), (480, 112, 8, 63, 0, 1, 0), (112, 672, 8, 63, 2, 0, 0),
          (672, 112, 8, 63, 0, 1, 1), (672, 160, 4, 32, 0, 1, 0), (160, 960, 4, 32, 2, 0, 0), (960, 160, 4, 32, 0, 1, 1)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


MODES = (0, 3, 7)
tot = {m: 0.0 for m in MODES}
for Ci, Co, F, T, act, se, res in LAYERS:
    x = torch.randn(B, Ci, F, T, device=DEV)
    w = torch.randn(Co, Ci, device=DEV) * Ci ** -0.5
    bias = torch.randn(Co, device=DEV) * 0.1
    sc = torch.rand(B, Ci, device=DEV) if se else None
    r = torch.randn(B, Co, F, T, device=DEV) if res else None
    wp = ops.pw_prepack_bf16(w, None, True)
    gb = 4 * B * F * T * (Ci + Co * (2 if res else 1)) / 1e9
    row = f"{Ci:4d}->{Co:4d} @{F}x{T} se={se} res={res}  {gb:6.3f} GB"
    outs = {}
    for mode in MODES:
        ops.pw_stream_mode(mode)
        us = timeit(lambda: ops.pw_conv_bf16(x, wp, bias, Co, act, True, in_scale=sc, res=r))
        outs[mode] = ops.pw_conv_bf16(x, wp, bias, Co, act, True, in_scale=sc, res=r)
        tot[mode] += us
        row += f" | mode {mode}: {us:7.1f} us {gb / us * 1e3:5.2f} TB/s"
    d = max(float((outs[0] - outs[m]).abs().max()) for m in MODES[1:])
    print(row + f" | max diff {d:.1e}", flush=True)
ops.pw_stream_mode(0)
print("total us per mode:", {m: round(v) for m, v in tot.items()})
