import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 256
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for Ci, Co, S in [(112,672,504),(112,672,512),(112,672,496),(112,672,2048),(80,480,504),(80,480,512),(672,112,504),(672,112,512)]:
    x = torch.randn(B, Ci, S, 1, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5; bias = torch.zeros(Co, device=dev)
    wp = ops.pw_prepack_bf16(w, None, split=True)
    us = timeit(lambda: ops.pw_conv_bf16(x, wp, bias, Co, 2, True))
    gb = 4 * B * S * (Ci + Co) / 1e9
    print(f"{Ci}->{Co} S={S}: {us:.1f} us {gb/us*1e3:.2f} TB/s", flush=True)
