import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256; Ci, Co, S = [int(v) for v in sys.argv[1:4]]
x = torch.randn(B, Ci, S, 1, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
bias = torch.zeros(Co, device=dev)
wp = ops.pw_prepack_bf16(w, None, split=True)
for _ in range(3): ops.pw_conv_bf16(x, wp, bias, Co, 2, True)
torch.cuda.synchronize()
