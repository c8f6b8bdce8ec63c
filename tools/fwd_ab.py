"""Whole-forward A/B of the 1x1 kernel variants in ONE process on one box (tools, not product): one hipGraph per
eat_pw_stream_mode value (the graph freezes the kernels chosen at capture), replayed alternately, median of rounds.
  python tools/fwd_ab.py [modes, e.g. 0,1,3,15] [streams]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from efficientat_amd import ops
from efficientat_amd.graphs import GraphedForward

modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "0,1,3,15").split(",")]
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
mel, model = bench.build_model(dev)
wave = (0.1 * torch.randn(256, bench.CLIP_SAMPLES, device=dev)).clamp_(-1, 1)
graphs = {}
for m in modes:
    ops.pw_stream_mode(m)
    graphs[m] = GraphedForward(model, mel, wave, streams=streams)
ops.pw_stream_mode(0)


def timed(run, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for m in modes:
    for _ in range(5):
        graphs[m].replay()
res = {m: [] for m in modes}
for rnd in range(7):
    for m in modes:
        res[m].append(timed(graphs[m].replay))
for m in modes:
    med = statistics.median(res[m])
    print(f"mode {m:2d}: median {med:.3f} ms  ({256 / med * 1e3:.0f} clips/s, {256 / med * 1e3 * 96.37e6 / 8e12:.3f} of the HBM roofline)"
          f"  rounds {[round(v, 3) for v in res[m]]}", flush=True)
