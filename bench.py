"""bench.py -- hot-path throughput on MI355X.

One "step" = one pass of the hot path (fused log-mel front-end + mn10_as forward, eval, fp32)
over one batch of synthetic 10 s @ 32 kHz clips already resident in HBM (BASELINE.json configs[1]:
"mn10_as forward-only, batch 256 synthetic 10 s clips, 1xMI355X, fp32").  With --gpus N every rank
processes its own batch of the same size (clips are independent: no data-path collective), so
value = N * batch * steps / max-over-ranks time and scaling is "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel's algorithmic HBM bytes per launch / its mean launch duration
                (HIP events on the launch stream), against the 8 TB/s HBM3E peak
  cpu_baseline  the CPU oracle (a port of the reference's torch-CPU path) timed on this host
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                      # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15                 # FLOP/s, dense bf16 MFMA (no sparsity), same guide
MFMA_F32_PEAK = 157.3e12               # FLOP/s, dense fp32-input MFMA (= fp32 vector peak), same guide
CLIP_SAMPLES = 320000                  # 10 s @ 32 kHz
ALG_BYTES_PER_CLIP = 96.37e6           # mn10 fwd 94.50 MB + weights/B + mel 1.79 MB


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def build_model(dev):
    from efficientat_amd.mn import get_model
    from efficientat_amd.preprocess import AugmentMelSTFT
    torch.manual_seed(0)
    mel = quiet(AugmentMelSTFT, freqm=0, timem=0).to(dev).eval()
    model = quiet(get_model, width_mult=1.0)
    # random init that keeps activations O(1) through 46 un-trained BN layers (fan-in scaling),
    # so the kernels see realistic (non-collapsed, non-zero) data
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.normal_(0, (2.0 / fan_in) ** 0.5)
            elif isinstance(m, torch.nn.Linear):
                m.weight.normal_(0, (1.0 / m.weight.shape[1]) ** 0.5)
        model.classifier[5].weight.mul_(0.05)      # keep |logits| O(1) so the parity probe reads like 1e-3 abs
    return mel, model.to(dev).eval()


# ------------------------------------------------------------------ per-kernel event profile
def _alg_bytes(name, a):
    """(kernel symbol as rocprofv3 prints it, algorithmic HBM bytes, flops) of one launch:
    activations in + out once, weights once (SURVEY.md 8d per-layer traffic model)."""
    if name == "eat_pw_conv_fwd":
        x, wp, bias, sc, res, y, pool, B, Ci, Co, S, act = a[:12]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        ns = min(B, 256 // S + 2) if sc else 0
        pipe = "true" if 16 * ns <= 64 else "false"
        nbytes = 4 * B * S * (Ci + (Co if y else 0) + (Co if res else 0)) + 4 * Co * Ci
        return f"pw_conv_kernel<{mtw},{pipe}>", nbytes, 2 * B * S * Ci * Co
    if name == "eat_pw_conv_bf16_fwd":
        x, wp, bias, sc, res, y, pool, B, Ci, Co, S, act, split = a[:13]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + (Co if y else 0) + (Co if res else 0)) + (4 if split else 2) * Co * Ci
        return f"pw_conv_bf16_kernel<{mtw},{3 if split else 1},*>", nbytes, 2 * B * S * Ci * Co
    if name == "eat_dw_conv_fwd":
        x, w, bias, y, pool, B, C, F, T, Fo, To, k, s, act = a[:14]
        return f"dw_conv_kernel<{k},{s},{act}>", 4 * B * C * (F * T + Fo * To) + 4 * C * k * k, 2 * B * C * Fo * To * k * k
    if name == "eat_fused_expand_dw_fwd":
        x, wp, be, wd, bd, y, pool, B, Cin, Cexp, F, T, Fo, To, k, s, act = a[:17]
        nbytes = 4 * B * (Cin * F * T + Cexp * Fo * To) + 4 * Cexp * (Cin + k * k)
        return f"mbconv_kernel<{k},{s},*,{act},false>", nbytes, 2 * B * Cexp * (Cin * F * T + k * k * Fo * To)
    if name == "eat_mbconv_fwd":
        x, wpe, be, wd, bd, wpp, bp, res, y, B, Cin, Cexp, Cout, F, T, Fo, To, k, s, act = a[:20]
        # a fused kernel is priced at its OWN unavoidable traffic (input once [+ residual re-read], output once),
        # not at the traffic of the three layers it replaces
        nbytes = 4 * B * (Cin * F * T * (2 if res else 1) + Cout * Fo * To) + 4 * Cexp * (Cin + k * k + Cout)
        flops = 2 * B * (Cexp * Cin * F * T + Cexp * k * k * Fo * To + Cout * Cexp * Fo * To)
        return f"mbconv_kernel<{k},{s},*,{act},true>", nbytes, flops
    if name == "eat_front_fwd":
        x, ws, bs, wd, bd, wpp, bp, y, B, C, F, T, Fo, To, act = a[:15]
        return f"front_kernel<{act}>", 4 * B * (F * T + C * Fo * To) + 4 * C * (9 + 9 + C), 2 * B * C * Fo * To * (9 + 9 + C)
    if name == "eat_stem_conv_fwd":
        x, w, bias, y, B, C, F, T, Fo, To, act = a[:11]
        return f"stem_conv_kernel<{act}>", 4 * B * (F * T + C * Fo * To), 2 * B * C * Fo * To * 9
    if name == "eat_mel_fwd":
        B, L, n_mels, T = a[1], a[2], a[10], a[13]
        return "mel_fwd_kernel", 4 * B * (L + n_mels * T), B * T * 60000
    if name == "eat_linear_fwd":
        x, w, bias, y, B, K, N = a[:7]
        return "linear_kernel", 4 * (B * K + N * K + B * N), 2 * B * K * N
    return name, 0, 0


def kernel_profile(step, iters=3):
    """Run `step` eagerly with a HIP event pair around every C-ABI launch (same stream)."""
    from efficientat_amd import _lib
    real_call = _lib.call
    rec = []

    def traced(name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *args)
        e1.record()
        rec.append((name, args, e0, e1))

    _lib.call = traced
    try:
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
    finally:
        _lib.call = real_call
    agg = {}
    if os.environ.get("EAT_BENCH_LAUNCHES"):      # debug: one line per launch of the last iteration
        for name, args, e0, e1 in rec[-(len(rec) // iters):]:
            sym, nbytes, _ = _alg_bytes(name, args)
            us = e0.elapsed_time(e1) * 1e3
            ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < 10 ** 7]
            print(f"[launch] {sym:24s} {us:9.1f} us {nbytes / us / 1e3:8.1f} GB/s  {ints}", file=sys.stderr)
    for name, args, e0, e1 in rec:
        sym, nbytes, flops = _alg_bytes(name, args)
        d = agg.setdefault(sym, [0, 0.0, 0, 0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += nbytes
        d[3] += flops
    return {k: dict(launches=v[0] // iters, total_ms=v[1] / iters * 1e3, bytes=v[2] / iters, flops=v[3] / iters,
                    gbps=(v[2] / v[1] / 1e9) if v[1] > 0 else 0.0) for k, v in agg.items()}


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(budget_s=12.0, batch=16):
    """The CPU oracle (port of the reference torch-CPU path: mel + mn10 eval forward) on host cores."""
    from oracle import eat_oracle as O
    from oracle import synth
    sd = synth.synth_state(synth.mn_shapes(1.0), seed=0)
    g = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(batch, CLIP_SAMPLES, generator=g)).clamp_(-1, 1)

    def once():
        with torch.no_grad():
            O.mn_forward(sd, O.mel_forward(x).unsqueeze(1))

    once()
    t0 = time.perf_counter()
    n = 0
    while True:
        once()
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    return {"value": round(batch * n / dt, 2), "unit": "clips/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} iters x batch {batch} of the same workload (mel + mn10 fwd, fp32, torch CPU)"}


def parity_probe(mel, model, dev):
    """logit max-abs-err of the HIP path vs the CPU oracle on 4 synthetic clips (same weights)."""
    from oracle import eat_oracle as O
    g = torch.Generator().manual_seed(7)
    x = (0.1 * torch.randn(4, CLIP_SAMPLES, generator=g)).clamp_(-1, 1)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref, _ = O.mn_forward(sd, O.mel_forward(x).unsqueeze(1))
        got, _ = model(mel(x.to(dev)).unsqueeze(1))
    return float((got.cpu() - ref).abs().max()), float(ref.abs().max())


def train_bench(args, mel, model, wave, dev, dist, world, barrier):
    """Full training step per GPU: log-mel (train mode) -> forward (batch-stat BN) -> BCE-with-logits ->
    hand-written backward -> [bucketed RCCL all-reduce of the 19.5 MB gradient, overlapped] -> Adam.
    Mirrors ex_audioset.py:139-199 without data loading / wandb / KD teacher."""
    import torch.nn.functional as F
    from efficientat_amd.dp import enable_data_parallel
    bt = min(args.train_batch, wave.shape[0])
    w = wave[:bt]
    alg = 285.8e6                                    # SURVEY 8(d) train-step bytes per clip, mn10 fp32
    if args.train_model != "mn10":                   # BASELINE configs 3 / 4 (parity cases, timed on request)
        torch.manual_seed(0)
        if args.train_model.startswith("dymn"):
            from efficientat_amd.dymn import get_model as gm
            model = quiet(gm, width_mult=2.0 if args.train_model == "dymn20" else 1.0).to(dev)
            alg = 324.6e6 if args.train_model == "dymn20" else None
        else:
            from efficientat_amd.mn import get_model as gm
            model = quiet(gm, width_mult=4.0).to(dev)
            model.train_precision = "bf16" if args.train_model.endswith("bf16") else "fp32"
            alg = 581.5e6
        with torch.no_grad():
            for m in model.modules():
                if isinstance(m, torch.nn.Conv2d):
                    fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                    m.weight.normal_(0, (2.0 / fan_in) ** 0.5)
    g = torch.Generator(device=dev).manual_seed(99)
    y = (torch.rand((bt, 527), device=dev, generator=g) < 2.7 / 527).float()
    use_dp = world > 1 or dist is not None
    if use_dp:
        enable_data_parallel(model)
    graphed = not use_dp and not args.no_graph
    # fused=True: one multi-tensor kernel per step (the foreach path spends ~350 tiny launches per step on the
    # per-parameter bias-correction scalars)
    opt = torch.optim.Adam(model.parameters(), lr=8e-4, capturable=graphed, fused=True)
    model.train()
    mel.train()
    launch = "eager"
    if graphed:
        # the step is launch-bound when issued eagerly (~35 ms of host time for ~500 launches):
        # capture fwd + loss + bwd + Adam once, replay per step; the mel front-end stays outside
        try:
            from efficientat_amd.graphs import GraphedTrainStep
            gstep = GraphedTrainStep(model, opt, F.binary_cross_entropy_with_logits, mel(w).unsqueeze(1), y)
            launch = "hipGraph replay (mel eager)"
        except Exception as e:  # pragma: no cover
            print(f"[bench] train-step graph capture failed ({e}); eager", file=sys.stderr)
            graphed = False
            opt = torch.optim.Adam(model.parameters(), lr=8e-4, fused=True)

    def tstep():
        if graphed:
            return gstep(mel(w).unsqueeze(1), y)
        opt.zero_grad(set_to_none=True)
        logits, _ = model(mel(w).unsqueeze(1))
        loss = F.binary_cross_entropy_with_logits(logits, y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(2, args.warmup // 2)):
        tstep()
    steps = max(3, args.steps // 3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tstep()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    cps = world * bt * steps / el
    return {"value": round(cps, 1), "unit": "clips/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
            "batch_per_gpu": bt, "final_loss": round(float(loss), 5), "launch": launch,
            "what": "mel + fwd(train BN) + BCE + bwd (HIP) + " + ("RCCL all-reduce + " if world > 1 else "") + "Adam, fp32",
            "model": args.train_model,
            "roofline_e2e_frac": round(cps / world * alg / HBM_PEAK, 4) if alg else None,
            "alg_bytes_per_clip": alg}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-train", action="store_true", help="skip the train-step measurement")
    ap.add_argument("--train-batch", type=int, default=128, help="clips per GPU per train step")
    ap.add_argument("--train-model", default="mn10", choices=["mn10", "mn40", "mn40_bf16", "dymn10", "dymn20"],
                    help="network of the train-step measurement (mn40_bf16 / dymn20 = BASELINE configs 3 / 4)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--streams", type=int, default=2, help="sub-batches issued on concurrent HIP streams per step")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel event profile to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("EAT_BENCH_FORCE_DIST") == "1"    # exercise the RCCL path with a single rank (debug)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    mel, model = build_model(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    wave = (0.1 * torch.randn(args.batch, CLIP_SAMPLES, device=dev, generator=g)).clamp_(-1, 1)

    out = {}
    n_str = max(1, args.streams)
    streams = [torch.cuda.Stream() for _ in range(n_str)] if n_str > 1 else []
    chunks = wave.chunk(n_str) if n_str > 1 else [wave]

    def step():
        with torch.no_grad():
            if n_str == 1:
                out["logits"], out["feat"] = model(mel(wave).unsqueeze(1))
                return
            # the batch is cut into `streams` sub-batches issued on their own HIP streams (fork / join inside
            # the captured graph): latency-bound kernels of one sub-batch (SE GEMMs, mel, kernel tails) overlap
            # with bandwidth-bound kernels of the other
            cur = torch.cuda.current_stream()
            res = []
            for st, wv in zip(streams, chunks):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    res.append(model(mel(wv).unsqueeze(1)))
            for st in streams:
                cur.wait_stream(st)
            out["logits"] = [r[0] for r in res]
            out["feat"] = [r[1] for r in res]

    step()                               # folds / packs weights, builds mel tables
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        except Exception as e:  # pragma: no cover - report, then measure eagerly
            print(f"[bench] hipGraph capture failed ({e}); timing eager launches", file=sys.stderr)
            graph = None
    run = graph.replay if graph is not None else step

    for _ in range(args.warmup):
        run()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    clips_per_s = world * args.batch * args.steps / elapsed
    result = {
        "metric": "clips/sec (10 s @ 32 kHz) mn10_as", "value": round(clips_per_s, 1), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "mn10_as forward-only (log-mel front-end + MN eval forward), batch 256 synthetic "
                               "10 s @ 32 kHz clips per GPU, fp32 [BASELINE.json configs[1]]",
                   "batch_per_gpu": args.batch,
                   "arithmetic": "fp32 activations and accumulation; 1x1 convs: exact fp32 MFMA for C_in < 40, split-operand "
                                 "bf16x3 MFMA (x = hi + lo, 3 products, ~2^-16 rel. error) for C_in >= 40 [EAT_PW_MODE=fp32 forces "
                                 "exact fp32 everywhere]",
                   "launch": ("hipGraph replay" if graph is not None else "eager") +
                             (f", {n_str} concurrent sub-batch streams" if n_str > 1 else ""),
                   "parallelism": f"dp{world} (independent clips, no collective)"},
        "roofline_e2e": {"bound": "hbm", "achieved": round(clips_per_s / world * ALG_BYTES_PER_CLIP / 1e9, 1),
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(clips_per_s / world * ALG_BYTES_PER_CLIP / HBM_PEAK, 4),
                         "note": "whole forward: clips/s per GPU x 96.37 MB algorithmic bytes per clip (SURVEY 8d)"},
    }

    if not args.no_train:
        result["train_step"] = train_bench(args, mel, model, wave, dev, dist, world, barrier)
        model.eval()
        mel.eval()

    if rank == 0:
        def profile_step():      # one stream: concurrent sub-batch streams would time overlapping kernels
            with torch.no_grad():
                model(mel(wave).unsqueeze(1))
        prof = kernel_profile(profile_step)
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, d = dom
        per_launch_bytes = d["bytes"] / d["launches"]
        per_launch_flops = d["flops"] / d["launches"]
        per_launch_s = d["total_ms"] * 1e-3 / d["launches"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_r1.json")
        if os.path.exists(tpath):
            # `*` in our symbol stands for template arguments chosen inside the library (tile rows, stages)
            import fnmatch
            ks = json.load(open(tpath))["kernels"]
            hit = [v for kk, v in ks.items() if fnmatch.fnmatchcase(kk, name.replace(" ", ""))]
            traffic = int(sum(h["hbm_bytes_per_launch"] * h["launches_sampled"] for h in hit) /
                          sum(h["launches_sampled"] for h in hit)) if hit else None
        # which roof binds this kernel: the larger of its HBM time and its MFMA time.  The MFMA peak is the
        # one of the instruction the kernel issues: fp32 16x16x4 (157 TF), bf16 16x16x32 (2.5 PF dense), and
        # for the bf16x3 split kernel 2.5 PF / 3 because each useful product costs three bf16 MFMAs.
        mfma_peak = MFMA_F32_PEAK
        if name.startswith("pw_conv_bf16_kernel"):
            mfma_peak = MFMA_BF16_PEAK / (3 if ",3," in name else 1)
        mfma_bound = per_launch_flops / mfma_peak > per_launch_bytes / HBM_PEAK
        common = {"kernel": name, "traffic": traffic, "launches_per_step": d["launches"],
                  "avg_launch_us": round(per_launch_s * 1e6, 2), "alg_bytes_per_launch": int(per_launch_bytes),
                  "alg_flops_per_launch": int(per_launch_flops),
                  "hbm_gbps": round(per_launch_bytes / per_launch_s / 1e9, 1),
                  "mfma_tflops": round(per_launch_flops / per_launch_s / 1e12, 2),
                  "traffic_source": "profiles/pmc_traffic_r1.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                  "share_of_step": round(d["total_ms"] / sum(v["total_ms"] for v in prof.values()), 3)}
        if mfma_bound:
            ach = per_launch_flops / per_launch_s
            result["roofline"] = {"bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": round(mfma_peak / 1e12, 1),
                                  "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4), **common}
        else:
            ach = per_launch_bytes / per_launch_s
            result["roofline"] = {"bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9,
                                  "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4), **common}
        if args.kernel_table:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
                print(f"[bench] {k:28s} launches {v['launches']:3d}  {v['total_ms']:8.3f} ms  "
                      f"{v['bytes'] / 1e9:7.3f} GB  {v['gbps']:8.1f} GB/s", file=sys.stderr)
        err, scale = parity_probe(mel, model, dev)
        result["parity"] = {"logit_max_abs_err": err, "logit_abs_max": scale, "vs": "CPU oracle, 4 clips, same weights"}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
