"""A/B of the fused expand + depthwise kernel on the mn10 block shapes it covers (tools, not product): separate kernels
(eat_pw_conv_bf16_fwd + eat_dw_conv_fwd) vs eat_expand_dw_bf16_fwd, then the whole forward with and without it.
  python tools/edw_ab.py [B]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
DEV = "cuda"


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


for Ci, Ce, se in [(80, 200, 0), (80, 184, 0), (80, 480, 1), (112, 672, 1)]:
    F, T = 8, 63
    x = torch.randn(B, Ci, F, T, device=DEV)
    we = torch.randn(Ce, Ci, device=DEV) * Ci ** -0.5
    be, bd = torch.randn(Ce, device=DEV) * 0.1, torch.randn(Ce, device=DEV) * 0.1
    w9 = torch.randn(Ce, 9, device=DEV) * 0.3
    wp = ops.pw_prepack_bf16(we, None, True)
    pool = torch.zeros(B, Ce, device=DEV) if se else None
    sep = lambda: ops.dw_conv(ops.pw_conv_bf16(x, wp, be, Ce, 2, True), w9, bd, 3, 1, 2, pool)
    fus = lambda: ops.expand_dw_bf16(x, wp, be, w9, bd, Ce, 3, 1, 2, pool)
    d = float((sep() - fus()).abs().max())
    ts, tf = timeit(sep), timeit(fus)
    gb = 4 * B * F * T * (Ci + Ce) / 1e9
    print(f"{Ci:4d}->{Ce:4d} @8x63: separate {ts:7.1f} us, fused {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s of x + y), max diff {d:.1e}", flush=True)

# whole forward: one graph per setting, replayed alternately
import bench
from efficientat_amd.graphs import GraphedForward
dev = torch.device("cuda:0")
mel, model = bench.build_model(dev)
wave = (0.1 * torch.randn(256, bench.CLIP_SAMPLES, device=dev)).clamp_(-1, 1)
graphs = {}
for on in (False, True):
    ops._FUSE_EXPAND_DW = on
    graphs[on] = GraphedForward(model, mel, wave, streams=2)
with torch.no_grad():
    ops._FUSE_EXPAND_DW = False
    l0 = model(mel(wave[:8]).unsqueeze(1))[0]
    ops._FUSE_EXPAND_DW = True
    l1 = model(mel(wave[:8]).unsqueeze(1))[0]
print("logit max diff fused vs separate:", float((l0 - l1).abs().max()), "on |logits| <=", float(l0.abs().max()))


def timed(run, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for on in graphs:
    for _ in range(5):
        graphs[on].replay()
res = {on: [] for on in graphs}
for rnd in range(7):
    for on in graphs:
        res[on].append(timed(graphs[on].replay))
for on in graphs:
    med = statistics.median(res[on])
    print(f"expand+dw fused {str(on):5s}: median {med:.3f} ms ({256 / med * 1e3:.0f} clips/s, {256 / med * 1e3 * 96.37e6 / 8e12:.3f} of the HBM roofline)", flush=True)
