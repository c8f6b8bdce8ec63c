"""Does batch-chunked scheduling (producer chunk -> consumer chunk while the chunk is still in L2 / Infinity Cache) beat
whole-batch passes for a train-mode conv -> stats -> apply -> depthwise chain?   python tools/mall_probe.py   (GPU diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import _lib, ops
dev = torch.device("cuda:0")
B = 256
NONE, RELU = ops.ACT_NONE, ops.ACT_RELU
def s(): return torch.cuda.current_stream().cuda_stream

def run(shape, nb):
    Cin, Cexp, F, T, k, stride = shape
    S = F * T
    x = torch.randn(B, Cin, F, T, device=dev)
    w = torch.randn(Cexp, Cin, device=dev) * 0.2
    wp = ops.pw_prepack(w)
    zb = torch.zeros(Cexp, device=dev)
    wd = torch.randn(Cexp, k * k, device=dev) * 0.2
    a = torch.rand(Cexp, device=dev) + 0.5
    bb = torch.randn(Cexp, device=dev) * 0.1
    z = torch.empty(B, Cexp, F, T, device=dev)
    y = torch.empty_like(z)
    Fo, To = ops.conv_out(F, k, stride), ops.conv_out(T, k, stride)
    zd = torch.empty(B, Cexp, Fo, To, device=dev)
    sums = torch.zeros(2 * Cexp, device=dev, dtype=torch.float64)
    sums2 = torch.zeros(2 * Cexp, device=dev, dtype=torch.float64)
    def phase1(i0, n):
        _lib.call("eat_pw_conv_fwd", x[i0:i0+n].data_ptr(), wp.data_ptr(), zb.data_ptr(), None, None, z[i0:i0+n].data_ptr(), None,
                  n, Cin, Cexp, S, NONE, s())
        _lib.call("eat_bn_stats", z[i0:i0+n].data_ptr(), n, Cexp, S, sums.data_ptr(), s())
    def phase2(i0, n):
        _lib.call("eat_bn_act_fwd", z[i0:i0+n].data_ptr(), a.data_ptr(), bb.data_ptr(), None, y[i0:i0+n].data_ptr(), None, n, Cexp, S, RELU, s())
        _lib.call("eat_dw_conv_fwd", y[i0:i0+n].data_ptr(), wd.data_ptr(), zb.data_ptr(), zd[i0:i0+n].data_ptr(), None, n, Cexp, F, T, Fo, To,
                  k, stride, NONE, s())
        _lib.call("eat_bn_stats", zd[i0:i0+n].data_ptr(), n, Cexp, Fo * To, sums2.data_ptr(), s())
    def step():
        for i0 in range(0, B, nb): phase1(i0, min(nb, B - i0))
        for i0 in range(0, B, nb): phase2(i0, min(nb, B - i0))
    step(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10

for shape in [(16, 64, 64, 500, 3, 2), (24, 72, 32, 250, 3, 1), (40, 120, 16, 125, 5, 1), (112, 672, 8, 63, 3, 1)]:
    base = None
    for nb in (256, 64, 32, 16, 8):
        t = run(shape, nb)
        base = base or t
        print(shape, "chunk", nb, f"{t:.3f} ms  ({t / base:.2f}x)", flush=True)
