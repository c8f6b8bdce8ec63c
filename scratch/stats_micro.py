import sys; sys.path.insert(0, '.')
import torch
from efficientat_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
B = 128
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode, shapes in (("fp32", [(16,64,32000),(24,72,8000)]), ("bf16x3", [(112,672,504),(672,112,504),(160,960,128),(40,240,2000)])):
    for Ci, Co, S in shapes:
        x = torch.randn(B, Ci, S, 1, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5; bias = torch.zeros(Co, device=dev)
        with ops.precision(mode):
            wp = ops.pw_prepack(w)
            t0 = timeit(lambda: ops.pw_conv(x, wp, bias, Co, 0))
            t1 = timeit(lambda: ops.pw_conv_stats(x, wp, bias, Co))
            z = ops.pw_conv(x, wp, bias, Co, 0)
            t2 = timeit(lambda: ops.bn_stats(z))
        print(f"pw {mode} {Ci}->{Co} S={S}: conv {t0:.1f} us, conv+stats {t1:.1f} us, separate bn_stats {t2:.1f} us", flush=True)
for C, F_, T, k, s in [(64,64,500,3,2),(120,16,125,5,1),(672,8,63,3,1),(960,4,32,5,1)]:
    x = torch.randn(B, C, F_, T, device=dev); w = torch.randn(C, k*k, device=dev); bias = torch.zeros(C, device=dev)
    t0 = timeit(lambda: ops.dw_conv(x, w, bias, k, s, 0)); t1 = timeit(lambda: ops.dw_conv_stats(x, w, bias, k, s))
    z = ops.dw_conv(x, w, bias, k, s, 0); t2 = timeit(lambda: ops.bn_stats(z))
    print(f"dw C{C} {F_}x{T} k{k}s{s}: conv {t0:.1f} us, conv+stats {t1:.1f} us, separate bn_stats {t2:.1f} us", flush=True)
