"""Run the fused expand + depthwise kernel alone on one mn10 shape (for rocprofv3 passes): python tools/edw_one.py [Ci Ce B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops
Ci, Ce, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (112, 672, 256)))
DEV = "cuda"
x = torch.randn(B, Ci, 8, 63, device=DEV)
wp = ops.pw_prepack_bf16(torch.randn(Ce, Ci, device=DEV) * Ci ** -0.5, None, True)
be, bd, w9 = torch.randn(Ce, device=DEV) * 0.1, torch.randn(Ce, device=DEV) * 0.1, torch.randn(Ce, 9, device=DEV) * 0.3
pool = torch.zeros(B, Ce, device=DEV)
for _ in range(5):
    ops.expand_dw_bf16(x, wp, be, w9, bd, Ce, 3, 1, 2, pool)
torch.cuda.synchronize()
