"""bench.py -- hot-path throughput on MI355X.

BASELINE.json's metric is "clips/sec (10 s @ 32 kHz) mn10_as fwd+bwd, 1/2/4/8 MI355X; logit max-abs-err".  One "step" =
one full training step of mn10_as on one batch of 256 synthetic 10 s @ 32 kHz clips per GPU, already resident in HBM:
log-mel front-end (train mode) -> forward with batch-statistics BatchNorm -> BCE-with-logits -> hand-written backward ->
[bucketed RCCL all-reduce of the gradient, overlapped with backward, when N > 1] -> fused Adam (ex_audioset.py:139-199
without data loading / wandb / teacher).  `value` = N * 256 * steps / max-over-ranks time, scaling "weak".
(Rounds 1-2 printed the forward-only figure of BASELINE configs[1] as `value`; it is the `forward` object now, so that
the headline is the metric BASELINE names and the N = 2/4/8 runs exercise the collective.)

`python bench.py --gpus N` started WITHOUT a torchrun environment re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` (one rank per GPU, RCCL); started by torchrun it
reads RANK / LOCAL_RANK / WORLD_SIZE.  It refuses to print a line when fewer than N ranks / GPUs are available.

Rank 0 prints ONE JSON line (contract in the task statement) with these extra objects:
  roofline       the kernel with the largest share of the timed step: algorithmic HBM bytes (or flops) per launch / its
                 mean launch duration (HIP events on the launch stream) against the 8 TB/s HBM3E (or MFMA) peak;
                 `traffic` = measured HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic_*.json)
  roofline_e2e   the whole training step at SURVEY 8(d)'s 285.8 MB algorithmic bytes per clip
  forward        BASELINE configs[1]: log-mel + mn10 eval forward, batch 256, hipGraph replay: clips/s, ms, fraction of the
                 HBM roofline at the 96.37 MB/clip contract bytes AND at the bytes the launch plan actually moves
                 (sum of every kernel's own input + output), `fp32_exact` = every 1x1 conv on the exact fp32 MFMA
  train_step_mn40_bf16 / train_step_dymn20   BASELINE configs[2] / configs[3] (batch 128; N = 1 only)
  forward_mn40 / forward_dymn20   eval forward of the same two models (batch 128, one stream; N = 1 only)
  parity         logit max-abs-err of the HIP path vs the CPU oracle (4 clips, same weights)
  cpu_baseline   the CPU oracle (a port of the reference's torch-CPU path) timed on this host: the same training step
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12                      # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2.5e15                 # FLOP/s, dense bf16 MFMA (no sparsity), same guide
MFMA_F32_PEAK = 157.3e12               # FLOP/s, dense fp32-input MFMA (= fp32 vector peak), same guide
CLIP_SAMPLES = 320000                  # 10 s @ 32 kHz
ALG_BYTES_PER_CLIP = 96.37e6           # mn10 fwd 94.50 MB + weights/B + mel 1.79 MB
ALG_TRAIN = {"mn10": 285.8e6, "mn40": 581.5e6, "mn40_bf16": 581.5e6, "dymn20": 324.6e6, "dymn20_bf16": 324.6e6, "dymn10": None}
# SURVEY 8(d) quotes the dymn20 contract for bf16 activations (3 x 104.70 + 8.7 + 1.79 MB); the same model with fp32 activations
# moves 3 x 209.39 + 8.7 + 1.79 MB per clip by the same per-layer rule - reported beside the fp32-activation leg
ALG_TRAIN_FP32_ACT = {"dymn20": 638.7e6}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _fan_in_init(model, head_scale=None):
    """random init that keeps activations O(1) through ~46 un-trained BN layers (fan-in scaling), so the kernels
    see realistic (non-collapsed, non-zero) data"""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.normal_(0, (2.0 / fan_in) ** 0.5)
            elif isinstance(m, torch.nn.Linear) and head_scale is not None:
                m.weight.normal_(0, (1.0 / m.weight.shape[1]) ** 0.5)
        if head_scale is not None:
            model.classifier[5].weight.mul_(head_scale)   # keep |logits| O(1) so the parity probe reads like 1e-3 abs


def build_model(dev):
    from efficientat_amd.mn import get_model
    from efficientat_amd.preprocess import AugmentMelSTFT
    torch.manual_seed(0)
    mel = quiet(AugmentMelSTFT, freqm=0, timem=0).to(dev).eval()
    model = quiet(get_model, width_mult=1.0)
    _fan_in_init(model, head_scale=0.05)
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
    if name == "eat_pw_conv_gstats_fwd":
        x, wp, wmode, zb, y, gz, ga, gb, gact, part, B, Ci, Co, S = a[:14]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + 2 * Co) + 4 * Co * Ci                          # x in, y out, z_d in (the epilogue)
        return (f"pw_conv_kernel<{mtw},true>" if wmode == 0 else f"pw_conv_bf16_kernel<{mtw},3,*>"), nbytes, 2 * B * S * Ci * Co
    if name == "eat_pw_conv_stats_fwd":
        # train-mode 1x1 conv with the BatchNorm statistics in its epilogue (project / last convs; ops.pw_conv_stats)
        x, wp, wmode, per_sample, tfa, tfb, act, sc, zb, y, part, B, Ci, Co, S = a[:15]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + Co) + 4 * Co * Ci * (B if per_sample else 1)
        tag = ",tf" if tfa else ""
        return ((f"pw_conv_kernel<{mtw},true{tag}>" if wmode == 0 else f"pw_conv_bf16_kernel<{mtw},{3 if wmode == 2 else 1},*{tag}>"),
                nbytes, 2 * B * S * Ci * Co)
    if name == "eat_bn_bwd_sums_from_tiles":
        return "bn_bwd_sums_from_tiles_kernels", 8 * a[1] * a[2] + 64, a[1] * a[2]
    if name == "eat_pw_conv_bf16_fwd":
        x, wp, bias, sc, res, y, pool, B, Ci, Co, S, act, split = a[:13]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + (Co if y else 0) + (Co if res else 0)) + (4 if split else 2) * Co * Ci
        flops, np_ = 2 * B * S * Ci * Co, 3 if split else 1
        # which kernel the library picks (csrc/conv_pw_stream.hip: try_stream)
        from efficientat_amd import ops as _ops
        mode, n_chunks = _ops.pw_stream_mode(), (Ci + 31) // 32
        if S % 4 == 0 and (mode & 1) and not sc and not res and not pool and y and n_chunks <= 4 and Co >= 2 * Ci \
                and 4 * B * Co * S < 2 ** 31 - 1:
            return f"pw_expand_kernel<{n_chunks},{np_},*>", nbytes, flops
        win = Co <= 2 * Ci and Ci >= 160 and (mt + 5) // 6 == (mt + 7) // 8
        if S % 4 == 0 and (((mode & 2) and win) or (mode & 4)):
            mc = (mt + 5) // 6
            return f"pw_kstream_kernel<{(mt + mc - 1) // mc},{np_},*>", nbytes, flops
        return f"pw_conv_bf16_kernel<{mtw},{np_},*>", nbytes, flops
    if name == "eat_expand_dw_bf16_fwd":
        x, wp, be, wd, bd, y, pool, B, Ci, Ce, F, T, k, st, act = a[:15]
        nbytes = 4 * B * F * T * (Ci + Ce) + 4 * Ce * (Ci + k * k)
        return f"expand_dw_kernel<{(Ci + 31) // 32}>", nbytes, 2 * B * Ce * F * T * (Ci + k * k)
    if name == "eat_dw_conv_fwd":
        x, w, bias, y, pool, B, C, F, T, Fo, To, k, s, act = a[:14]
        nbytes, flops = 4 * B * C * (F * T + Fo * To) + 4 * C * k * k, 2 * B * C * Fo * To * k * k
        # the small late-layer planes run on the register-resident kernel (csrc/dw_plane.hip: dw_plane_try's table)
        plane = {(3, 1, 8): (32, 64, 1, 64), (5, 1, 16): (64, 128, 2, 64), (5, 2, 8): (32, 64, 2, 32),
                 (3, 2, 16): (64, 128, 2, 64), (5, 1, 4): (0, 32, 1, 32)}.get((k, s, F))
        if plane and plane[0] < T <= plane[1]:
            return f"dw_plane_kernel<{k},{s},{plane[2]},{plane[3]},{F},*>", nbytes, flops
        return f"dw_conv_kernel<{k},{s},{act}>", nbytes, flops
    if name == "eat_fused_expand_dw_fwd":
        x, wp, be, wd, bd, y, pool, B, Cin, Cexp, F, T, Fo, To, k, s, act = a[:17]
        nbytes = 4 * B * (Cin * F * T + Cexp * Fo * To) + 4 * Cexp * (Cin + k * k)
        return f"irb_kernel<{k},{s},{Cin // 4},*,false,*>", nbytes, 2 * B * Cexp * (Cin * F * T + k * k * Fo * To)
    if name == "eat_mbconv_fwd":
        x, wpe, be, wd, bd, wpp, bp, res, y, B, Cin, Cexp, Cout, F, T, Fo, To, k, s, act = a[:20]
        # a fused kernel is priced at its OWN unavoidable traffic (input once [+ residual re-read], output once),
        # not at the traffic of the three layers it replaces
        nbytes = 4 * B * (Cin * F * T * (2 if res else 1) + Cout * Fo * To) + 4 * Cexp * (Cin + k * k + Cout)
        flops = 2 * B * (Cexp * Cin * F * T + Cexp * k * k * Fo * To + Cout * Cexp * Fo * To)
        return f"irb_kernel<{k},{s},{Cin // 4},*,true,*>", nbytes, flops
    if name == "eat_front_fwd":
        x, ws, bs, wd, bd, wpp, bp, y, B, C, F, T, Fo, To, act = a[:15]
        return "irb_kernel<3,1,3,*,true,*,true>", 4 * B * (F * T + C * Fo * To) + 4 * C * (9 + 9 + C), 2 * B * C * Fo * To * (9 + 9 + C)
    if name == "eat_stem_conv_fwd":
        x, w, bias, y, B, C, F, T, Fo, To, act = a[:11]
        return f"stem_conv_kernel<{act}>", 4 * B * (F * T + C * Fo * To), 2 * B * C * Fo * To * 9
    if name == "eat_mel_fwd":
        B, L, n_mels, T = a[1], a[2], a[11], a[14]
        # the front-end costs ~0.03 GFLOP per 10 s clip (1000 frames: radix FFT-1024 + banded mel)
        return "mel_fwd_kernel", 4 * B * (L + n_mels * T), int(B * T * 30000)
    if name == "eat_linear_fwd":
        x, w, bias, y, B, K, N = a[:7]
        return "linear_kernel", 4 * (B * K + N * K + B * N), 2 * B * K * N
    if name == "eat_se_gate_fwd":
        pool, w1, b1, w2, b2, out, B, C, Q = a[:9]
        return "se_gate_kernel", 4 * (2 * B * C + 2 * C * Q), 4 * B * C * Q
    if name == "eat_head_fwd":
        pool, w1, b1, w2, b2, out, B, C, H, N = a[:10]
        return "head_kernel", 4 * (B * C + C * H + H * N + B * N), 2 * B * H * (C + N)
    # ---- training-step entry points (symbols: the dispatch rules of csrc/train.hip / dw_plane.hip)
    if name in ("eat_pw_conv_wgrad", "eat_pw_conv_wgrad_ws", "eat_pw_conv_wgrad_tf"):
        if name == "eat_pw_conv_wgrad":
            dz, x, xs, dW, B, Co, Ci, S, mode = a[:9]
            tf = None
        elif name == "eat_pw_conv_wgrad_ws":
            dz, x, xs, dW, ws, nsl, B, Co, Ci, S, mode = a[:11]
            tf = None
        else:
            dz, x, tf, tfb, tfact, xs, dW, ws, nsl, B, Co, Ci, S, mode = a[:14]
        same = dz == x and tf is None
        nbytes = 4 * B * S * (Co if same else Co + Ci) + 4 * Co * Ci
        # the library's own dispatch (csrc/train.hip: wgrad_plan), asked through its host helper
        from efficientat_amd import _lib as _l
        kind = int(_l.lib().eat_pw_wgrad_kernel_kind(B, Co, Ci, S, mode, 1 if dz == x else 0, 1 if xs else 0, 1 if tf else 0))
        npr = 1 if mode == 2 else 3
        if kind == 2:
            sym = "pw_wgrad_kernel"
        elif kind == 1:
            sym = f"pw_wgrad_x3_kernel<{npr}>"
        elif kind == 3:
            sym = f"pw_wgrad_wide_kernel<{npr},*>"
        else:
            d = kind // 10
            sym = f"pw_wgrad_x3_narrow_kernel<{d // 1000},{(d % 1000) // 10},{'true' if d % 10 else 'false'}>"
        return sym, nbytes, 2 * B * S * Co * Ci
    if name == "eat_pw_conv_tf_fwd":
        x, ta, tb, tact, wp, wmode, bias, sc, res, y, B, Ci, Co, S, act = a[:15]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + Co + (Co if res else 0)) + (4 if wmode != 1 else 2) * Co * Ci
        sym = f"pw_conv_kernel<{mtw},*,true>" if wmode == 0 else f"pw_conv_bf16_kernel<{mtw},{3 if wmode == 2 else 1},*,true>"
        return sym, nbytes, 2 * B * S * Ci * Co
    if name == "eat_se_bn_bwd_partials":
        B, C, S = a[6:9]
        return "se_bn_bwd_partials_kernel", 8 * B * C * S, 12 * B * C * S
    if name == "eat_col_sum":
        return "col_sum_kernel", 4 * a[2] * a[3], a[2] * a[3]
    if name == "eat_dw_conv_fwd_stats":
        x, ia, ib, iact, w, y, part, cap, hin, B, C, F, T, Fo, To, k, s = a[:17]
        return f"dw_conv_fwd_stats<{k},{s}>", 4 * B * C * (F * T + Fo * To), 2 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_fwd_tf":
        x, ia, ib, iact, w, bias, y, B, C, F, T, Fo, To, k, s = a[:15]
        return f"dw_conv_fwd_tf<{k},{s}>", 4 * B * C * (F * T + Fo * To), 2 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_dgrad_g":
        dz, w, gz, ga, gb, gact, g, gp, cap, hin, B, C, F, T, Fo, To, k, s = a[:18]
        return f"dw_conv_dgrad_g<{k},{s}>", 4 * B * C * (Fo * To + 2 * F * T), 2 * B * C * F * T * k * k // (s * s)
    if name == "eat_dw_conv_dgrad":
        dz, w, res, dx, B, C, F, T, Fo, To, k, s = a[:12]
        return f"dw_conv_dgrad<{k},{s}>", 4 * B * C * (Fo * To + F * T * (2 if res else 1)), 2 * B * C * F * T * k * k // (s * s)
    if name == "eat_dw_conv_wgrad_tf":
        dz, x, ia, ib, iact, dw, B, C, F, T, Fo, To, k, s = a[:14]
        return f"dw_conv_wgrad<{k},{s}>", 4 * B * C * (F * T + Fo * To), 2 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_wgrad":
        dz, x, dw, B, C, XC, F, T, Fo, To, k, s = a[:12]
        return f"dw_conv_wgrad<{k},{s}>", 4 * B * (XC * F * T + C * Fo * To), 2 * B * C * Fo * To * k * k
    if name == "eat_bn_act_fwd":
        z, aa, bb, res, y, pool, B, C, S, act = a[:10]
        return f"bn_act_fwd_kernel<{act}>", 4 * B * C * S * (1 + (1 if y else 0) + (1 if res else 0)), 4 * B * C * S
    if name == "eat_bn_act_bwd_reduce":
        B, C, S, act = a[8:12]
        return f"bn_act_bwd_reduce_kernel<{act}>", 8 * B * C * S, 8 * B * C * S
    if name == "eat_bn_act_bwd_apply":
        B, C, S, act = a[10:14]
        return f"bn_act_bwd_apply_kernel<{act}>", 12 * B * C * S, 10 * B * C * S
    if name in ("eat_bn_stats", "eat_bn_stats_partial"):
        B, C, S = a[1:4]
        return "bn_stats_kernel", 4 * B * C * S, 3 * B * C * S
    if name == "eat_plane_dot":
        B, C, S = a[5:8]
        return "plane_dot_kernel", 8 * B * C * S, 2 * B * C * S
    if name == "eat_act_grad_sum":
        B, C, S = a[7:10]
        return "act_grad_sum_kernel", 12 * B * C * S, 4 * B * C * S
    # ---- round 5: bf16 activation storage (BASELINE configs[2]): the wide tensor of each launch moves 2 bytes per element
    if name == "eat_pw_conv_b16_fwd":
        x, x16, x2, c1, wp, bias, ta, tb, tact, sc, res, y, y16, part, gz, ga, gb, gact, B, Ci, Co, S, act = a[:23]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        cw = (c1 if x2 else Ci) if x16 else 0                                      # bf16 input channels
        nbytes = B * S * (2 * cw + 4 * (Ci - cw) + (2 if y16 else 4) * Co + (4 * Co if res else 0)
                          + (2 * Co if gz else 0)) + 2 * Co * Ci               # (gz: the epilogue reads the bf16 z_d tile)
        return (f"pw_conv_bf16_kernel<{mtw},1,*,{'true' if ta else 'false'},{'bf16' if x16 else 'float'},{'bf16' if y16 else 'float'}>",
                nbytes, 2 * B * S * Ci * Co)
    if name == "eat_dw_conv_fwd_stats_b16":
        x, x16, ia, ib, iact, w, y, part, cap, hin, B, C, F, T, Fo, To, k, s = a[:18]
        return f"dw_conv_fwd_stats<{k},{s},bf16>", B * C * ((2 if x16 else 4) * F * T + 2 * Fo * To), 2 * B * C * Fo * To * k * k
    if name == "eat_bn_act_fwd_b16":
        z, aa, bb, res, y, y16, yc, pool, B, C, S, act = a[:12]
        per = 2 + (4 if res else 0) + ((2 if y16 else 4) if y else 0) + (2 if yc else 0)
        return f"bn_act_fwd_kernel<{act},bf16,{'bf16' if y16 else 'float'}>", per * B * C * S, 4 * B * C * S
    if name == "eat_bn_act_bwd_reduce_b16":
        d16 = a[1]
        B, C, S, act = a[9:13]
        return f"bn_act_bwd_reduce_kernel<{act},bf16,{'bf16' if d16 else 'float'}>", ((2 if d16 else 4) + 2) * B * C * S, 8 * B * C * S
    if name == "eat_bn_act_bwd_apply_b16":
        B, C, S, act = a[11:15]
        return f"bn_act_bwd_apply_kernel<{act},bf16>", (4 + 2 + 4 + (2 if a[10] else 0)) * B * C * S, 10 * B * C * S
    if name == "eat_cast_b16":
        return "cast_b16_kernel", 6 * a[2], 0
    if name == "eat_se_bn_bwd_partials_b16":
        B, C, S = a[6:9]
        return "se_bn_bwd_partials_kernel<bf16>", 4 * B * C * S, 12 * B * C * S
    if name == "eat_dw_conv_bwd_bn_g_b16":
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        xb = 2 if a[12] else 4                                                    # x and g: bf16, or fp32 in the first block
        return (f"dw_bwd_tile_kernel<{k},{s},*,true,*,false,bf16>", B * C * (2 * 2 * Fo * To + 2 * xb * F * T),
                4 * B * C * F * T * k * k // (s * s) + 2 * B * C * Fo * To * k * k)
    if name == "eat_pw_conv_wgrad_b16":
        dz, d16, x, x16, ta, tb, tact, xs, dW, ws, nsl, B, Co, Ci, S = a[:15]
        return f"pw_wgrad_wide_kernel<1,{'true' if x16 else 'false'},*,true,*>", B * S * ((2 if d16 else 4) * Co + (2 if x16 else 4) * Ci) + 4 * Co * Ci, 2 * B * S * Co * Ci
    # ---- round 4: every entry point of the training steps has a byte model (argument order = include/eat_hip.h)
    if name in ("eat_dw_conv_bwd_bn_g", "eat_dw_conv_dyn_bwd_bn_g"):
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        dyn = name.endswith("dyn_bwd_bn_g")
        has_res = dyn and a[14] is not None
        # reads dy + z (conv-output sized) and x (input sized) [+ res], writes g (input sized) [+ per-plane tap gradients]
        nbytes = 4 * B * C * (2 * Fo * To + (3 if has_res else 2) * F * T) + (4 * B * C * k * k if dyn else 0)
        return f"dw_bwd_tile_kernel<{k},{s},*,true,*,{'true' if dyn else 'false'}>", nbytes, 4 * B * C * F * T * k * k // (s * s) + 2 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_bwd_g":
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        return f"dw_bwd_tile_kernel<{k},{s},*,false,*>", 4 * B * C * (Fo * To + 2 * F * T), 4 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_dyn_fwd_stats":
        x, ia, ib, iact, w, y, part, cap, hin, B, C, F, T, Fo, To, k, s = a[:17]
        return f"dw_conv_fwd_stats<{k},{s},dyn>", 4 * B * C * (F * T + Fo * To + k * k), 2 * B * C * Fo * To * k * k
    if name == "eat_pw_conv_cat_fwd":
        x1, C1, x2, C2, wp, wmode, bias, res, y, B, Co, S, act = a[:13]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (C1 + C2 + Co + (Co if res else 0)) + (4 if wmode != 1 else 2) * Co * (C1 + C2)
        sym = f"pw_conv_kernel<{mtw},*,cat>" if wmode == 0 else f"pw_conv_bf16_kernel<{mtw},{3 if wmode == 2 else 1},*,cat>"
        return sym, nbytes, 2 * B * S * (C1 + C2) * Co
    if name == "eat_se_mlp_bwd":
        B, C, Cr = a[13:16]
        return "se_mlp_bwd_kernels", 4 * (4 * B * C + 2 * B * Cr + 4 * C * Cr + C + Cr), 8 * B * C * Cr
    if name == "eat_mlp_head_bwd":
        B, C, H, N = a[13:17]
        return "head_bwd_kernels", 4 * (B * (2 * N + 5 * H + 2 * C) + 2 * (H * C + N * H)), 4 * B * H * (N + C)
    if name == "eat_stem_bwd":
        B, C, F, T = a[10:14]
        Fo, To = (F - 1) // 2 + 1, (T - 1) // 2 + 1
        n_dy = 2 if a[1] is not None else 1
        return "stem_bwd_kernel", 4 * B * (F * T + n_dy * C * Fo * To), 2 * B * C * Fo * To * 9 * 2
    if name == "eat_stem_gram":
        B, C, F, T = a[5:9]
        return "stem_gram_kernel", 4 * B * F * T, 2 * B * ((F - 1) // 2 + 1) * ((T - 1) // 2 + 1) * 54
    if name in ("eat_gram_bn_finalize_g", "eat_gram_bn_finalize"):
        Co, Ci = a[3:5]
        per_row = Ci * Ci if name.endswith("_g") else Ci           # the one-launch form streams G once per output channel (L2)
        return name.replace("eat_", "") + "_kernel", 4 * (Ci * Ci + 3 * Co * Ci + 8 * Co) if name.endswith("_g") else 4 * (2 * Co * Ci + 8 * Co), 2 * Co * (per_row + Ci)
    if name == "eat_gram_centered":
        B, C, S = a[6:9]
        mt = (C + 15) // 16
        from efficientat_amd import _lib as _l
        kind = int(_l.lib().eat_pw_wgrad_kernel_kind(B, C, C, S, a[9], 1, 0, 0))
        sym = ("pw_wgrad_kernel" if kind == 2 else "pw_wgrad_x3_kernel<3>" if kind == 1 else "pw_wgrad_wide_kernel<3,*>" if kind == 3
               else f"pw_wgrad_x3_narrow_kernel<{kind // 10 // 1000},{(kind // 10 % 1000) // 10},true>")
        return sym, 4 * B * S * C + 4 * C * C, 2 * B * S * C * C
    if name == "eat_expand_bwd_wcat":
        W, a_, e2, e1, Co, Ci, kind = a[:7]
        K = Co + Ci
        pack = 4 * K * Ci if kind == 0 else (4 if kind == 2 else 2) * ((K + 31) // 32 * 32) * ((Ci + 15) // 16 * 16)
        return "expand_bwd_wcat_kernel", 4 * Co * Ci + pack + 12 * Co + 4 * Ci, 2 * Co * Ci * Ci + 2 * Co * Ci
    if name == "eat_expand_bwd_coef":
        Co, Ci = a[7:9]
        return "expand_bwd_coef_kernel", 4 * 7 * Co * Ci, 12 * Co * Ci
    if name in ("eat_bn_finalize", "eat_bn_finalize_partials", "eat_se_bn_bwd_combine", "eat_bn_bwd_combine_partials"):
        if name == "eat_bn_finalize":
            n_in = 16 * a[8]
        elif name == "eat_bn_finalize_partials":
            n_in = 8 * a[1] * a[2] * a[3]
        elif name == "eat_se_bn_bwd_combine":
            n_in = 4 * 7 * a[4] * a[5]
        else:
            n_in = 8 * a[3] * a[4] * a[5]
        return name.replace("eat_", "") + "_kernel", n_in + 64, n_in // 2
    if name in ("eat_pw_prepack", "eat_pw_prepack_t", "eat_pw_prepack_bf16", "eat_pw_prepack_bf16_t"):
        Co, Ci = a[3:5]
        return "pw_prepack_kernel", 8 * Co * Ci, Co * Ci
    if name == "eat_pw_prepack_multi":
        return "pw_prepack_multi_kernel", 8 * a[1] * a[2], a[1] * a[2]          # n matrices x max_threads elements: upper bound
    if name in ("eat_pw_conv_dyn_fwd", "eat_pw_conv_dyn_bf16_fwd"):
        x, wp, bias, res, y, B, Ci, Co, S, act = a[:10]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = 4 * B * S * (Ci + Co + (Co if res else 0)) + 4 * B * Co * Ci            # + the sample's own packed weights
        sym = f"pw_conv_kernel<{mtw},*,dyn>" if name == "eat_pw_conv_dyn_fwd" else f"pw_conv_bf16_kernel<{mtw},3,*,dyn>"
        return sym, nbytes, 2 * B * S * Ci * Co
    if name == "eat_pw_conv_kcat_fwd":
        x, wp, bias, att, res, y, B, Ci, nb, Co, S, act = a[:12]
        return "pw_conv_bf16_kernel<kcat>", 4 * B * S * (Ci + Co + (Co if res else 0)) + 4 * nb * Co * Ci, 2 * B * S * nb * Ci * Co
    if name == "eat_pw_conv_dyn_wgrad":
        dz, x, dW, B, Co, Ci, S = a[:7]
        return "pw_wgrad_dyn", 4 * B * (S * (Co + Ci) + Co * Ci), 2 * B * S * Co * Ci
    if name in ("eat_dyn_pw_pack", "eat_dyn_pw_pack_t", "eat_dyn_pw_pack_bf16", "eat_dyn_pw_pack_bf16_t"):
        if "bf16" in name:
            B, K, Co, Ci = a[3:7]
        else:
            B, K, Co, Ci = a[4:8]
        return "dyn_pw_pack_kernel", 4 * Co * Ci * (K + B), 2 * B * K * Co * Ci         # banks once (L2-resident), packs written
    if name == "eat_dyn_aggregate":
        B, K, N = a[4:7]
        return "dyn_aggregate_kernel", 4 * N * (K + B), 2 * B * K * N
    if name == "eat_dyn_bank_grad":
        B, K, N = a[5:8]
        return "dyn_bank_grad_fused_kernel", 4 * N * (B + 2 * K), 4 * B * K * N
    if name in ("eat_dyrelu_ca_fwd2", "eat_dyrelu_ca_fwd"):
        B, C, Fo, To = a[-5:-1]
        return "dyrelu_ca_fwd_kernel", 4 * B * C * (2 * Fo * To + Fo + To + 4), 10 * B * C * Fo * To
    if name in ("eat_dyrelu_ca_bwd2", "eat_dyrelu_ca_bwd"):
        B, C, Fo, To = a[-5:-1]
        return "dyrelu_ca_bwd_kernel", 4 * B * C * (3 * Fo * To + 2 * (Fo + To) + 10), 24 * B * C * Fo * To
    if name in ("eat_ctx_pool", "eat_ctx_pool_cm"):
        B, C, F, T = a[2:6]
        return "ctx_pool_kernel", 4 * B * C * (F * T + F + T), 2 * B * C * F * T
    if name in ("eat_ctx_pool_bwd", "eat_ctx_pool_cm_bwd"):
        B, C, F, T = a[3:7]
        return "ctx_pool_bwd_kernel", 4 * B * C * ((2 if a[1] is not None else 1) * F * T + F + T), 2 * B * C * F * T
    if name in ("eat_ctx_split", "eat_ctx_split_bwd"):
        H, B, F, T = a[4:8]
        return name.replace("eat_", "") + "_kernel", 8 * H * B * (F + T), 4 * H * B * (F + T)
    if name in ("eat_dw_conv_dyn_fwd", "eat_dw_conv_dyn_act_fwd"):
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        return f"dw_conv_dyn<{k},{s}>", 4 * B * C * (F * T + Fo * To + k * k), 2 * B * C * Fo * To * k * k
    if name in ("eat_dw_conv_dyn_wgrad", "eat_dw_conv_dyn_dgrad"):
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        return name.replace("eat_", "") + f"<{k},{s}>", 4 * B * C * (F * T + Fo * To + k * k), 2 * B * C * Fo * To * k * k
    if name in ("eat_mixup_fwd", "eat_kd_loss_fwd_bwd", "eat_calib_copy"):
        n = a[4] * a[5] if name == "eat_mixup_fwd" else (a[8] * a[9] if name == "eat_kd_loss_fwd_bwd" else a[2])
        return name.replace("eat_", "") + "_kernel", 12 * n, 4 * n
    # ---- round 6: bf16 activation storage of the DyMN blocks (argument order = include/eat_hip.h)
    if name in ("eat_dyn_heads_fwd", "eat_dyn_heads_bwd"):
        B, n_att, K, cexp = (a[1:5] if name.endswith("fwd") else a[5:9])
        n = B * (n_att * K + 4 * cexp)
        return name.replace("eat_", "") + "_kernel", 12 * n, 8 * n
    if name == "eat_dyn_pw_pack_b16":
        B, K, Co, Ci = a[3:7]
        return "dyn_pw_pack_bf16_kernel<plain>", 4 * K * Co * Ci + 2 * B * Co * Ci, 2 * B * K * Co * Ci
    if name == "eat_pw_conv_dyn_b16_fwd":
        x, x16, wp, bias, res, y, y16, part, B, Ci, Co, S, act = a[:13]
        mt = (Co + 15) // 16
        chunks = (mt + 7) // 8
        mtw = (mt + chunks - 1) // chunks
        nbytes = B * S * ((2 if x16 else 4) * Ci + (2 if y16 else 4) * Co + (4 * Co if res else 0)) + 2 * B * Co * Ci
        return f"pw_conv_bf16_kernel<{mtw},1,*,dyn,{'bf16' if x16 else 'float'},{'bf16' if y16 else 'float'}>", nbytes, 2 * B * S * Ci * Co
    if name == "eat_dw_conv_dyn_fwd_stats_b16":
        x, x16, ia, ib, iact, w, y, part, cap, hin, B, C, F, T, Fo, To, k, s = a[:18]
        return f"dw_conv_fwd_stats<{k},{s},dyn,bf16>", B * C * ((2 if x16 else 4) * F * T + 2 * Fo * To + 4 * k * k), 2 * B * C * Fo * To * k * k
    if name == "eat_dw_conv_dyn_bwd_bn_g_b16":
        B, C, F, T, Fo, To, k, s = a[-9:-1]
        xb = 2 if a[10] else 4
        has_res = a[15] is not None
        nbytes = B * C * (2 * 2 * Fo * To + (2 * xb + (4 if has_res else 0)) * F * T + 4 * k * k)
        return f"dw_bwd_tile_kernel<{k},{s},*,true,*,true,bf16>", nbytes, 4 * B * C * F * T * k * k // (s * s) + 2 * B * C * Fo * To * k * k
    if name == "eat_dyrelu_ca_fwd2_b16":
        B, C, Fo, To = a[-5:-1]
        return "dyrelu_ca_fwd_kernel<bf16>", B * C * (2 * 2 * Fo * To + 4 * (Fo + To + 4)), 10 * B * C * Fo * To
    if name == "eat_dyrelu_ca_bwd2_b16":
        B, C, Fo, To = a[-5:-1]
        return "dyrelu_ca_bwd_kernel<bf16>", B * C * (3 * 2 * Fo * To + 4 * (2 * (Fo + To) + 10)), 24 * B * C * Fo * To
    if name == "eat_bn_bwd_apply_b16":
        B, C, S, act = a[8:12]
        return f"bn_act_bwd_apply_kernel<{act},bf16,bf16>", 6 * B * C * S, 10 * B * C * S
    if name == "eat_pw_conv_dyn_wgrad_b16":
        dz, d16, x, x16, dW, nsl, B, Co, Ci, S = a[:10]
        return (f"pw_wgrad_wide_kernel<1,{'true' if x16 else 'false'},*,per-sample>",
                B * S * ((2 if d16 else 4) * Co + (2 if x16 else 4) * Ci) + 4 * B * Co * Ci, 2 * B * S * Co * Ci)
    if name == "eat_adam_multi":          # reads p, g, m, v, writes p, m, v (the element count: `_adam` below)
        n = _ADAM_ELEMS["n"] or a[1] * 4096
        return "adam_multi_kernel", 28 * n, 12 * n
    return name, 0, 0


_ADAM_ELEMS = {"n": 0}


# access width of a kernel's global loads -> which calibration copy corrects its FETCH_SIZE reading
_ACCESS_CLASS = {"pw_conv_kernel": "lds16", "pw_conv_bf16_kernel": "lds16", "pw_wgrad_x3_kernel": "lds16", "pw_wgrad_wide_kernel": "b16",
                 "dw_plane_kernel": "b8", "dw_tile_kernel": "b8", "dw_bwd_tile_kernel": "b8", "dw_conv_fwd_stats": "b8",
                 "dw_conv_kernel": "b4", "irb_kernel": "b4", "stem_conv_kernel": "b4", "dyrelu_ca_fwd_kernel": "b4",
                 "dyrelu_ca_bwd_kernel": "b4", "ctx_pool_kernel": "b4", "col_sum_kernel": "b4"}
# byte-model labels (an entry point runs one of several kernels) -> kernel family in the PMC file
PMC_FILES = ("pmc_traffic_r6.json", "pmc_traffic_r5.json", "pmc_traffic_r4.json")


def pmc_step_bytes(phase):
    """Calibrated FETCH + WRITE bytes of one whole step of the committed PMC pass (every dispatch, torch's included), or None."""
    for tfile in PMC_FILES:
        tpath = os.path.join(ROOT, "profiles", tfile)
        if os.path.exists(tpath):
            doc = json.load(open(tpath))
            step, cal = doc.get(phase + "_step"), doc.get("calibration")
            if step and cal:
                ff, fw = cal["fetch_factor"]["b16"], cal["write_factor"]["b16"]
                return {"bytes": int((ff * step["fetch_kib_total"] + fw * step["write_kib_total"]) * 1024),
                        "source": f"profiles/{tfile}: sum over the {step['dispatches_per_step'][0]} dispatches of one step, "
                                  f"FETCH_SIZE x {ff:.2f} + WRITE_SIZE x {fw:.2f}"}
            return None
    return None


_PMC_FAMILY = {"dw_conv_fwd_stats": "dw_tile_kernel", "se_mlp_bwd_kernels": "se_mlp_bwd_kernel", "pw_wgrad_dyn": "pw_wgrad_x3_kernel"}

_HAS_MFMA = ("pw_conv_kernel", "pw_conv_bf16_kernel", "pw_expand_kernel", "pw_kstream_kernel", "expand_dw_kernel", "irb_kernel",
             "pw_wgrad_kernel", "pw_wgrad_x3_kernel", "pw_wgrad_x3_narrow_kernel", "pw_wgrad_wide_kernel", "linear_kernel")


def roofline_of(name, d, args):
    """The `roofline` object of one kernel symbol from its event-profile row `d` (launches, total_ms, bytes, flops)."""
    import fnmatch
    per_launch_bytes = d["bytes"] / d["launches"]
    per_launch_flops = d["flops"] / d["launches"]
    per_launch_s = d["total_ms"] * 1e-3 / d["launches"]
    traffic, tsrc = None, None
    for tfile in PMC_FILES:
        tpath = os.path.join(ROOT, "profiles", tfile)
        if not os.path.exists(tpath):
            continue
        doc = json.load(open(tpath))
        step = doc.get(args.get("phase", "train") + "_step")
        cal = doc.get("calibration")
        if step and cal:
            # round 5: counter sums over the launches of this family INSIDE one step of the PMC pass - the same layers and
            # grids whose algorithmic bytes `alg_bytes_per_launch` averages (tools/pmc_traffic.py)
            pat = name.replace(" ", "")
            hit = {kk: v for kk, v in step["kernels"].items() if fnmatch.fnmatchcase(kk, pat)}
            if not hit:
                fam = _PMC_FAMILY.get(name.split("<")[0], name.split("<")[0])
                hit = {kk: v for kk, v in step["kernels"].items() if kk.split("<")[0] == fam}
            if hit:
                n = sum(v["launches"] for v in hit.values())
                fetch = sum(v["fetch_kib"] for v in hit.values()) * 1024
                write = sum(v["write_kib"] for v in hit.values()) * 1024
                cls = _ACCESS_CLASS.get(name.split("<")[0], "b16")
                ff, fw = cal["fetch_factor"][cls], cal["write_factor"][cls]
                traffic = int((ff * fetch + fw * write) / max(1, n))
                tsrc = (f"profiles/{tfile}: rocprofv3 (FETCH_SIZE {fetch / 1e6:.1f} MB x {ff:.2f} + WRITE_SIZE {write / 1e6:.1f} MB x "
                        f"{fw:.2f}) / {n} = the {n} launches of this family in ONE step of the PMC pass"
                        + ("" if n == d["launches"] else f" (the timed step has {d['launches']}: the two launch sets differ)")
                        + "; factors = known bytes / counter of eat_calib_copy in the same passes")
                break
        ks = doc["kernels"]
        # `*` in our symbol stands for template arguments chosen inside the library (tile rows, stages)
        hit = [v for kk, v in ks.items() if fnmatch.fnmatchcase(kk, name.replace(" ", ""))]
        if not hit:                                         # byte-model labels that are not kernel symbols: the family
            hit = [v for kk, v in ks.items() if kk.split("<")[0] == _PMC_FAMILY.get(name.split("<")[0], name.split("<")[0])]
        if hit:
            n = sum(h["launches_sampled"] for h in hit)
            fetch = sum(h["fetch_kib"] * h["launches_sampled"] for h in hit) / n * 1024
            write = sum(h["write_kib"] * h["launches_sampled"] for h in hit) / n * 1024
            cal = doc.get("calibration")
            if cal:
                # counter readings corrected by the copies of a KNOWN byte count that ran in the same rocprofv3 passes, per
                # access width of the kernel's loads (gfx950 reports wide coalesced reads at half size)
                cls = _ACCESS_CLASS.get(name.split("<")[0], "b16")
                ff, fw = cal["fetch_factor"][cls], cal["write_factor"][cls]
                traffic = int(ff * fetch + fw * write)
                tsrc = (f"profiles/{tfile}: rocprofv3 FETCH_SIZE {fetch / 1e6:.1f} MB x {ff:.2f} + WRITE_SIZE {write / 1e6:.1f} MB x "
                        f"{fw:.2f} per launch (largest-grid launches, separate passes); factors = known bytes / counter of "
                        f"eat_calib_copy ({cls} loads) in the same passes")
            else:
                rd = d.get("read_bytes", 0) / d["launches"]
                x2 = rd > 0 and fetch < 0.75 * rd
                traffic = int((2 if x2 else 1) * fetch + write)
                tsrc = (f"profiles/{tfile} (no calibration pass): FETCH_SIZE {fetch / 1e6:.1f} MB raw "
                        + ("x2" if x2 else "x1") + f" + WRITE_SIZE {write / 1e6:.1f} MB")
            break
    base = name.split("<")[0]
    mfma_peak = MFMA_F32_PEAK
    if base in ("pw_conv_bf16_kernel", "pw_expand_kernel", "pw_kstream_kernel"):
        mfma_peak = MFMA_BF16_PEAK / (3 if ",3," in name else 1)
    if base in ("expand_dw_kernel", "pw_wgrad_x3_narrow_kernel") or name in ("pw_wgrad_x3_kernel<3>", "pw_wgrad_wide_kernel<3,*>"):
        mfma_peak = MFMA_BF16_PEAK / 3            # every useful product costs three bf16 MFMAs
    if name in ("pw_wgrad_x3_kernel<1>", "pw_wgrad_wide_kernel<1,*>"):
        mfma_peak = MFMA_BF16_PEAK
    # which roof binds: a kernel WITHOUT matrix instructions is priced against HBM only (its fp32 VALU work is reported
    # as `valu_tflops` for information); an MFMA kernel against the larger of its HBM time and its MFMA time
    has_mfma = base in _HAS_MFMA
    mfma_bound = has_mfma and per_launch_flops / mfma_peak > per_launch_bytes / HBM_PEAK
    common = {"kernel": name, "traffic": traffic, "launches_per_step": d["launches"],
              "avg_launch_us": round(per_launch_s * 1e6, 2), "alg_bytes_per_launch": int(per_launch_bytes),
              "alg_flops_per_launch": int(per_launch_flops), "hbm_gbps": round(per_launch_bytes / per_launch_s / 1e9, 1),
              "traffic_source": tsrc, "share_of_step": round(d["total_ms"] / args["step_ms"], 3)}
    if has_mfma:
        common["mfma_tflops"] = round(per_launch_flops / per_launch_s / 1e12, 2)
    else:
        common["valu_tflops"] = round(per_launch_flops / per_launch_s / 1e12, 2)
    if mfma_bound:
        ach = per_launch_flops / per_launch_s
        return {"bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": round(mfma_peak / 1e12, 1),
                "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4), **common}
    ach = per_launch_bytes / per_launch_s
    return {"bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK, 4), **common}


def kernel_profile(step, iters=3):
    """Run `step` eagerly with a HIP event pair around every C-ABI launch (same stream)."""
    from efficientat_amd import _lib
    real_call, real_call_rc = _lib.call, _lib.call_rc
    rec = []

    def traced(name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_call(name, *args)
        e1.record()
        rec.append((name, args, e0, e1))

    def traced_rc(name, *args):
        # the entry points that may answer 1 = "nothing launched" (the 1x1 convs with statistics epilogues; until round 6
        # these 19 launches of the mn10 step went untimed and were read as inter-kernel gaps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = real_call_rc(name, *args)
        e1.record()
        if rc == 0:
            rec.append((name, args, e0, e1))
        return rc

    _lib.call, _lib.call_rc = traced, traced_rc
    try:
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
    finally:
        _lib.call, _lib.call_rc = real_call, real_call_rc
    # per launch: the MINIMUM over the iterations (the launch sequences are identical) - an event pair also spans whatever idle
    # time a host-side stall (allocator, a first-use hipMalloc) puts between its two records, and one such stall of tens of
    # milliseconds in one iteration would otherwise be booked on whichever kernel it hit
    n = len(rec) // iters
    seqs = [rec[i * n:(i + 1) * n] for i in range(iters)]
    same = len(rec) == n * iters and all([r[0] for r in sq] == [r[0] for r in seqs[0]] for sq in seqs)
    if not same:                                              # (never seen; fall back to the last iteration alone)
        seqs = [rec[-n:]]
    times = [min(sq[j][2].elapsed_time(sq[j][3]) for sq in seqs) for j in range(n)]
    last = seqs[-1]
    agg = {}
    if os.environ.get("EAT_BENCH_LAUNCHES"):      # debug: one line per launch
        for (name, args, _, _), ms in zip(last, times):
            sym, nbytes, _ = _alg_bytes(name, args)
            us = ms * 1e3
            ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < 10 ** 7]
            print(f"[launch] {sym:24s} {us:9.1f} us {nbytes / us / 1e3:8.1f} GB/s  {ints}", file=sys.stderr)
    for (name, args, _, _), ms in zip(last, times):
        sym, nbytes, flops = _alg_bytes(name, args)
        d = agg.setdefault(sym, [0, 0.0, 0, 0, 0])
        d[0] += 1
        d[1] += ms * 1e-3
        d[2] += nbytes
        d[3] += flops
        d[4] += _read_bytes(name, args, nbytes)
    return {k: dict(launches=v[0], total_ms=v[1] * 1e3, bytes=v[2], flops=v[3], read_bytes=v[4],
                    gbps=(v[2] / v[1] / 1e9) if v[1] > 0 else 0.0) for k, v in agg.items()}


def _read_bytes(name, a, nbytes):
    """Compulsory READ bytes of one launch (calibrates the FETCH_SIZE correction); 0 = unknown."""
    if name == "eat_mel_fwd":
        return 4 * a[1] * a[2]
    if name == "eat_pw_conv_wgrad":
        return nbytes - 4 * a[5] * a[6]
    if name == "eat_pw_conv_wgrad_ws":
        return nbytes - 4 * a[7] * a[8]
    if name == "eat_pw_conv_wgrad_tf":
        return nbytes - 4 * a[10] * a[11]
    if name in ("eat_bn_act_bwd_reduce", "eat_bn_stats", "eat_bn_stats_partial", "eat_plane_dot"):
        return nbytes
    if name == "eat_bn_act_bwd_apply":
        return nbytes * 2 // 3
    if name in ("eat_pw_conv_fwd", "eat_pw_conv_bf16_fwd"):
        B, Ci, Co, S = a[7:11]
        return 4 * B * S * (Ci + (Co if a[4] else 0))
    if name == "eat_dw_conv_dgrad_g":
        B, C, F, T, Fo, To = a[10:16]
        return 4 * B * C * (Fo * To + F * T)
    return 0


# ----------------------------------------------------------------------------- CPU baseline
def cpu_baseline(budget_s=7.0, batch=32):
    """The CPU oracle (a port of the reference's torch-CPU path) on this host's cores, SURVEY 8(d): batch 32, fp32,
    torch.set_num_threads(all cores), separate figures for the log-mel front-end, the mn10 eval forward and the full
    training step (mel + forward with batch-stat BN + BCE + backward + Adam; ex_audioset.py:139-199).  Each leg is a
    bounded sample: one warm-up pass, then whole passes until `budget_s` seconds are spent."""
    import torch.nn.functional as F
    from oracle import eat_oracle as O
    from oracle import synth
    cores = torch.get_num_threads()              # torch's default: one thread per physical core of the host
    sd = synth.synth_state(synth.mn_shapes(1.0), seed=0)
    g = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(batch, CLIP_SAMPLES, generator=g)).clamp_(-1, 1)
    y = (torch.rand(batch, 527, generator=g) < 2.7 / 527).float()
    with torch.no_grad():
        m = O.mel_forward(x).unsqueeze(1)
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var"))]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    opt = torch.optim.Adam(list(params.values()), lr=8e-4)

    def mel_leg():
        with torch.no_grad():
            O.mel_forward(x)

    def fwd_leg():
        with torch.no_grad():
            O.mn_forward(sd, m)

    def train_leg():
        sdt = dict(sd)
        sdt.update(params)
        opt.zero_grad(set_to_none=True)
        logits, _ = O.mn_forward(sdt, O.mel_forward(x).unsqueeze(1), train=True, stats={})
        F.binary_cross_entropy_with_logits(logits, y).backward()
        opt.step()

    def rate(fn, max_iters=50, min_iters=3):
        fn()
        t0 = time.perf_counter()
        n = 0
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if (dt > budget_s and n >= min_iters) or n >= max_iters:
                return batch * n / dt, n

    mel_r, mel_n = rate(mel_leg)
    fwd_r, fwd_n = rate(fwd_leg)
    trn_r, trn_n = rate(train_leg)
    both = 1.0 / (1.0 / mel_r + 1.0 / fwd_r)
    return {"value": round(trn_r, 2), "unit": "clips/s", "cores": cores, "kind": "port",
            "mel_clips_s": round(mel_r, 2), "fwd_clips_s": round(fwd_r, 2), "mel_plus_fwd_clips_s": round(both, 2),
            "train_step_clips_s": round(trn_r, 2),
            "sample": f"batch {batch}, fp32, torch CPU with {cores} threads ({os.cpu_count()} logical CPUs): mel {mel_n} / mn10 fwd {fwd_n} / train step "
                      f"{trn_n} passes (~{budget_s:.0f} s each); value = the training step (mel + fwd + BCE + bwd + Adam), the bench workload"}


def cpu_baseline_reference(ref_root, budget_s=7.0, batch=32):
    """The REFERENCE'S OWN modules (models/preprocess.py AugmentMelSTFT, models/mn/model.py get_model) timed on this host's
    cores - `kind: "reference"` - when a reference checkout is staged at EAT_REFERENCE_ROOT (it does not exist on the
    driver's GPU box: `cpu_baseline()` then times the oracle port).  Imported in a CHILD process with the torchaudio /
    torchvision stand-ins of oracle/ref_shims on its path (the reference's third-party imports are not installed here), so
    nothing of the reference enters this process.  Same legs and budget as the port."""
    code = r"""
import contextlib, io, json, os, sys, time
import torch, torch.nn.functional as F
ref, shims, batch, budget = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
sys.path[:0] = [shims, ref]
os.chdir(ref)
with contextlib.redirect_stdout(io.StringIO()):
    from models.preprocess import AugmentMelSTFT
    from models.mn.model import get_model
    mel = AugmentMelSTFT(freqm=0, timem=0)
    model = get_model(width_mult=1.0)
g = torch.Generator().manual_seed(1234)
x = (0.1 * torch.randn(batch, 320000, generator=g)).clamp_(-1, 1)
y = (torch.rand(batch, 527, generator=g) < 2.7 / 527).float()
opt = torch.optim.Adam(model.parameters(), lr=8e-4)
def mel_leg():
    mel.eval()
    with torch.no_grad():
        return mel(x)
m = mel_leg().unsqueeze(1)
def fwd_leg():
    model.eval()
    with torch.no_grad():
        model(m)
def train_leg():
    model.train(); mel.train()
    opt.zero_grad(set_to_none=True)
    logits, _ = model(mel(x).unsqueeze(1))
    F.binary_cross_entropy_with_logits(logits, y).backward()
    opt.step()
def rate(fn, max_iters=50, min_iters=3):
    fn()
    t0 = time.perf_counter(); n = 0
    while True:
        fn(); n += 1
        dt = time.perf_counter() - t0
        if (dt > budget and n >= min_iters) or n >= max_iters:
            return batch * n / dt, n
r = [rate(mel_leg), rate(fwd_leg), rate(train_leg)]
print("REFCPU " + json.dumps({"mel": r[0], "fwd": r[1], "train": r[2], "threads": torch.get_num_threads()}))
"""
    shims = os.path.join(ROOT, "oracle", "ref_shims")
    r = subprocess.run([sys.executable, "-c", code, ref_root, shims, str(batch), str(budget_s)], capture_output=True, text=True,
                       timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("REFCPU ")]
    if r.returncode != 0 or not line:
        raise RuntimeError(f"reference CPU baseline failed: {r.stderr[-800:]}")
    d = json.loads(line[0][7:])
    (mel_r, mel_n), (fwd_r, fwd_n), (trn_r, trn_n) = d["mel"], d["fwd"], d["train"]
    both = 1.0 / (1.0 / mel_r + 1.0 / fwd_r)
    return {"value": round(trn_r, 2), "unit": "clips/s", "cores": d["threads"], "kind": "reference",
            "mel_clips_s": round(mel_r, 2), "fwd_clips_s": round(fwd_r, 2), "mel_plus_fwd_clips_s": round(both, 2),
            "train_step_clips_s": round(trn_r, 2),
            "sample": f"the reference's own models/preprocess.py + models/mn/model.py (staged at EAT_REFERENCE_ROOT, torchaudio / "
                      f"torchvision stand-ins of oracle/ref_shims), batch {batch}, fp32, torch CPU with {d['threads']} threads: mel {mel_n} / "
                      f"mn10 fwd {fwd_n} / train step {trn_n} passes (~{budget_s:.0f} s each); value = the training step (mel + fwd + "
                      "BCE + bwd + Adam), the bench workload"}


def parity_grad_probe(dev):
    """Gradient parity of the training step against torch-CPU autograd over the oracle (5 structured clips x 2 s, calibrated
    synthetic weights, Dropout replaced by its expectation on both sides), for the step's default arithmetic ("auto") and
    for exact fp32: per-tensor relative L2 error - median, maximum, and the number of tensors above SURVEY 8(c)'s 1e-2."""
    import torch.nn.functional as F
    from oracle import eat_oracle as O
    from oracle import synth
    from efficientat_amd.mn import get_model
    wave = synth.parity_clips(64000, seed=3)
    x_ref = O.mel_forward(wave).unsqueeze(1)
    sd = synth.calibrate(synth.synth_state(synth.mn_shapes(1.0), seed=0), O.mn_forward, x_ref)
    y = (torch.rand(wave.shape[0], 527, generator=torch.Generator().manual_seed(5)) < 0.01).float()
    keep = torch.ones(wave.shape[0], 1280) * 0.8
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    logits_ref, _ = O.mn_forward(sdr, x_ref, train=True, stats={}, drop_mask=keep)
    F.binary_cross_entropy_with_logits(logits_ref, y).backward()
    gmax = max(float(v.grad.norm()) for v in sdr.values() if getattr(v, "grad", None) is not None)
    out = {"vs": "torch-CPU autograd over the oracle, 5 clips x 2 s, same weights; tensors with |grad| < 1e-4 of the largest "
                 "skipped (project-BatchNorm biases: true gradient 0)"}
    for mode in ("auto", "fp32"):
        model = quiet(get_model, width_mult=1.0)
        model.load_state_dict(sd)
        model.to(dev).train()
        model.train_precision = mode
        model._drop_mask_override = keep.to(dev)
        logits, _ = model(x_ref.to(dev))
        F.binary_cross_entropy_with_logits(logits, y.to(dev)).backward()
        rels = []
        for name, p in model.named_parameters():
            rg = sdr[name].grad
            if float(rg.norm()) >= 1e-4 * gmax:
                rels.append(float((p.grad.cpu().double() - rg.double()).norm() / rg.double().norm()))
        rels.sort()
        out[mode] = {"grad_rel_l2_median": float(f"{rels[len(rels) // 2]:.3e}"), "grad_rel_l2_max": float(f"{rels[-1]:.3e}"),
                     "n_above_1e-2": sum(r > 1e-2 for r in rels), "tensors": len(rels),
                     "train_logit_max_abs_err": float(f"{float((logits.detach().cpu() - logits_ref.detach()).abs().max()):.3e}")}
        del model
    return out


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


# ----------------------------------------------------------------------------- timing helpers
class Ranks:
    """torch.distributed plumbing of the bench: barrier + device sync on both sides of a timed region, max over ranks."""

    def __init__(self, dist, world, dev):
        self.dist, self.world, self.dev = dist, world, dev

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def timed(self, run, steps):
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        self.barrier()
        el = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([el], device=self.dev, dtype=torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t.item())
        return el


def forward_bench(args, mel, model, wave, ranks):
    """(clips/s over all ranks, ms per step, launch description) of the eval forward on `wave`."""
    from efficientat_amd.graphs import GraphedForward
    launch = "eager"
    run = None
    if not args.no_graph:
        try:
            gf = GraphedForward(model, mel, wave, streams=args.streams)
            run = gf.replay
            launch = "hipGraph replay" + (f", {len(gf.streams)} concurrent sub-batch streams" if gf.streams else "")
        except Exception as e:  # pragma: no cover - report, then measure eagerly
            print(f"[bench] hipGraph capture failed ({e}); timing eager launches", file=sys.stderr)
    if run is None:
        def run():
            with torch.no_grad():
                model(mel(wave).unsqueeze(1))
    for _ in range(args.warmup):
        run()
    els = sorted(ranks.timed(run, args.steps) for _ in range(max(1, getattr(args, "reps", 1))))
    el = els[len(els) // 2]                                   # median repetition (each: exactly args.steps steps)
    return ranks.world * wave.shape[0] * args.steps / el, el / args.steps * 1e3, launch


def make_train_model(name, dev, precision=None):
    torch.manual_seed(0)
    if name.startswith("dymn"):
        from efficientat_amd.dymn import get_model as gm
        model = quiet(gm, width_mult=2.0 if name.startswith("dymn20") else 1.0)
        if precision:
            model.train_precision = precision
        elif name.endswith("bf16"):
            # BASELINE configs[3] on the byte contract SURVEY 8(d) quotes for it: bf16 GEMM operands + bf16 storage of the wide
            # tensors of every dynamic block (EAT_ACT_STORAGE=fp32: fp32 activations in HBM, for A/B)
            model.train_precision = "bf16"
            model.act_storage = os.environ.get("EAT_ACT_STORAGE", "bf16")
    else:
        from efficientat_amd.mn import get_model as gm
        model = quiet(gm, width_mult=4.0 if name.startswith("mn40") else 1.0)
        model.train_precision = precision or ("bf16" if name.endswith("bf16") else os.environ.get("EAT_TRAIN_PRECISION", "auto"))
        # BASELINE configs[2] "train step bf16": bf16 GEMM operands AND bf16 storage of the wide activations / gradients
        # (EAT_ACT_STORAGE=fp32: the round-4 form, fp32 activations in HBM, for A/B)
        if name.endswith("bf16") and model.train_precision == "bf16":
            model.act_storage = os.environ.get("EAT_ACT_STORAGE", "bf16")
    _fan_in_init(model, head_scale=0.05 if name == "mn10" else None)
    return model.to(dev)


def _adam(params, capturable):
    """The optimizer of the timed steps (ex_audioset.py:86-91: Adam, lr 8e-4): the library's one-launch multi-tensor Adam
    (`efficientat_amd.optim.FusedAdam`, eat_adam_multi) - EAT_ADAM=torch selects torch.optim.Adam(fused=True) for A/B."""
    params = list(params)
    if os.environ.get("EAT_ADAM", "eat") == "torch":
        return torch.optim.Adam(params, lr=8e-4, capturable=capturable, fused=True)
    from efficientat_amd.optim import FusedAdam
    _ADAM_ELEMS["n"] = sum(p.numel() for p in params)
    return FusedAdam(params, lr=8e-4, capturable=capturable)


def train_bench(name, batch, steps, warmup, args, mel, wave, ranks, precision=None):
    """Full training step per GPU: log-mel (train mode) -> forward (batch-stat BN) -> BCE-with-logits ->
    hand-written backward -> [bucketed RCCL all-reduce of the gradient, overlapped with backward] -> fused Adam.
    Mirrors ex_audioset.py:139-199 without data loading / wandb / KD teacher.  The step (incl. the collectives when
    N > 1) is captured into one hipGraph and replayed; the log-mel stays outside (host RNG draws per step)."""
    import torch.nn.functional as F
    from efficientat_amd.dp import enable_data_parallel
    dev = ranks.dev
    model = make_train_model(name, dev, precision)
    bt = min(batch, wave.shape[0])
    w = wave[:bt]
    g = torch.Generator(device=dev).manual_seed(99)
    y = (torch.rand((bt, 527), device=dev, generator=g) < 2.7 / 527).float()
    use_dp = ranks.dist is not None
    if use_dp:
        enable_data_parallel(model, force_buckets=ranks.world == 1)
    graphed = not args.no_graph
    model.train()
    mel.train()
    launch = "eager"
    opt = None
    if graphed:
        # the step is launch-bound when issued eagerly (~35 ms of host time for ~500 launches): capture fwd + loss +
        # bwd + [all-reduce] + Adam once, replay per step.  fused=True: one multi-tensor kernel per step (the foreach
        # path spends ~350 tiny launches per step on the per-parameter bias-correction scalars)
        try:
            from efficientat_amd.graphs import GraphedTrainStep
            opt = _adam(model.parameters(), capturable=True)
            gstep = GraphedTrainStep(model, opt, F.binary_cross_entropy_with_logits, mel(w).unsqueeze(1), y)
            gstep.y.copy_(y)
            launch = "hipGraph replay (mel eager" + (", RCCL all-reduce captured)" if use_dp else ")")
        except Exception as e:  # pragma: no cover
            print(f"[bench] train-step graph capture failed for {name} ({type(e).__name__}: {e}); eager", file=sys.stderr)
            graphed = False
            model = make_train_model(name, dev, precision)
            if use_dp:
                enable_data_parallel(model, force_buckets=ranks.world == 1)
            model.train()
    if not graphed:
        opt = _adam(model.parameters(), capturable=False)

    def tstep():
        if graphed:
            # the log-mel kernel writes the captured step's input buffer directly (no 131 MB hand-over copy per step)
            return gstep(mel(w, out=gstep.x).view_as(gstep.x), gstep.y)
        opt.zero_grad(set_to_none=True)
        logits, _ = model(mel(w).unsqueeze(1))
        loss = F.binary_cross_entropy_with_logits(logits, y)
        loss.backward()
        opt.step()
        return loss

    out = {}
    for _ in range(max(2, warmup)):
        out["loss"] = tstep()
    evs = []
    if os.environ.get("EAT_BENCH_STEP_TIMES"):       # diagnostic: per-step GPU intervals (HIP events) and host submit times
        inner = tstep

        def tstep():  # noqa: F811
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((e, time.perf_counter()))
            return inner()
    # median of `reps` repetitions; every repetition times EXACTLY `steps` steps between barrier +
    # synchronize pairs (max over ranks) - the reported value / ms_per_step are those of the median repetition
    reps = max(1, getattr(args, "reps", 1))
    els = sorted(ranks.timed(lambda: out.__setitem__("loss", tstep()), steps) for _ in range(reps))
    el = els[len(els) // 2]
    if evs:
        gaps = [round(evs[i][0].elapsed_time(evs[i + 1][0]), 2) for i in range(len(evs) - 1)]
        host = [round((evs[i + 1][1] - evs[i][1]) * 1e3, 2) for i in range(len(evs) - 1)]
        print(f"[bench] {name} per-step GPU ms: {gaps}", file=sys.stderr)
        print(f"[bench] {name} per-step host submit ms: {host}", file=sys.stderr)
    cps = ranks.world * bt * steps / el
    alg = ALG_TRAIN.get(name)
    res = {"value": round(cps, 1), "unit": "clips/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
           "warmup": max(2, warmup), "batch_per_gpu": bt, "n_gpus": ranks.world, "final_loss": round(float(out["loss"]), 5),
           "launch": launch, "model": name, "repetitions": reps, "rep_ms_per_step": [round(e / steps * 1e3, 3) for e in els],
           "what": "mel + fwd(train BN) + BCE + bwd (HIP) + " + ("RCCL all-reduce + " if use_dp else "") + ("torch fused Adam; " if os.environ.get("EAT_ADAM", "eat") == "torch" else "multi-tensor Adam (eat_adam_multi); ")
                   + (f"bf16 MFMA 1x1 GEMMs, wide activations / gradients stored in {getattr(model, 'act_storage', 'fp32')}, fp32 statistics / "
                      "parameters / optimizer" if name.endswith("bf16") else "fp32 activations, 1x1 GEMMs per EAT_TRAIN_PRECISION"),
           "roofline_e2e_frac": round(cps / ranks.world * alg / HBM_PEAK, 4) if alg else None,
           "alg_bytes_per_clip": alg}
    if name in ALG_TRAIN_FP32_ACT and getattr(model, "act_storage", "fp32") == "fp32":
        res["alg_bytes_per_clip_fp32_activations"] = ALG_TRAIN_FP32_ACT[name]
        res["roofline_e2e_frac_fp32_activations"] = round(cps / ranks.world * ALG_TRAIN_FP32_ACT[name] / HBM_PEAK, 4)
    red = getattr(model, "_grad_reducer", None) or getattr(model, "_grad_hooks_reducer", None)
    if red is not None and red.last_stats:
        res["reducer"] = dict(red.last_stats)
    mel.eval()
    del model, opt
    if graphed:
        del gstep
    torch.cuda.empty_cache()
    return res


def kd_bench(args, mel_unused, wave, ranks, steps=12, warmup=3):
    """BASELINE configs[4], per-GPU shard: the reference's KD iteration (ex_audioset.py:139-199: log-mel with fmin / fmax
    jitter -> mixup -> forward -> BCE + KD loss against gathered teacher rows -> backward -> [RCCL all-reduce] -> Adam) on
    `train_loop.GraphedKDTrainer` / `KDTrainer`, B clips per GPU: (a) the captured step on a device-resident batch, (b) the
    same fed from pinned host memory through DevicePrefetcher (fp32 and 16-bit transport), (c) the eager trainer."""
    from efficientat_amd.dp import enable_data_parallel
    from efficientat_amd.input_pipeline import DevicePrefetcher, to_int16
    from efficientat_amd.preprocess import AugmentMelSTFT
    from efficientat_amd.train_loop import GraphedKDTrainer, KDTrainer
    dev = ranks.dev
    B, L = wave.shape
    g = torch.Generator().manual_seed(7)
    teacher = torch.randn(4096, 527, generator=g) * 2 - 5
    f2i = {"syn%07d" % i: i % 4096 for i in range(0, 8192, 2)}          # half of the files have a teacher row
    names = ["syn%07d" % i for i in range(B)]
    y = (torch.rand((B, 527), generator=g) < 2.7 / 527).float()
    out = {"workload": f"mn10_as KD training iteration of ex_audioset.py:139-199 (train-mode log-mel with fmin / fmax jitter, "
                       f"mixup alpha 0.3, BCE + KD loss lambda 0.1 against a 4096-row teacher table, backward, "
                       + ("bucketed RCCL all-reduce, " if ranks.dist is not None else "") + f"fused Adam), batch {B} per GPU "
                       "[BASELINE.json configs[4], per-GPU shard]", "batch_per_gpu": B, "n_gpus": ranks.world, "steps": steps}

    def make(graphed):
        model = make_train_model("mn10", dev)
        if ranks.dist is not None:
            enable_data_parallel(model, force_buckets=ranks.world == 1)
        model.train()
        mel = quiet(AugmentMelSTFT, freqm=0, timem=0).to(dev).train()
        opt = _adam(model.parameters(), capturable=graphed)
        kw = dict(teacher_preds=teacher, fname_to_index=f2i, kd_lambda=0.1, mixup_alpha=0.3)
        return (GraphedKDTrainer(model, mel, opt, B, L, **kw) if graphed else KDTrainer(model, mel, opt, **kw))

    def timed(run, n):
        for _ in range(warmup):
            run()
        el = ranks.timed(run, n)
        return round(ranks.world * B * n / el, 1), round(el / n * 1e3, 3)

    yd = y.to(dev)
    xd = wave.view(B, 1, L)
    torch.manual_seed(1234 + int(os.environ.get("RANK", 0)))
    np.random.seed(1234 + int(os.environ.get("RANK", 0)))
    tr = make(True)
    v, ms = timed(lambda: tr.step(xd, names, yd), steps)
    out["graphed"] = {"value": v, "unit": "clips/s", "ms_per_step": ms,
                      "launch": "one hipGraph replay per step (mel + mixup + forward + KD loss + backward + "
                                + ("all-reduce + " if ranks.dist is not None else "") + "Adam); batch resident in HBM"}
    if ranks.world > 1:
        # N > 1: the captured KD step with the RCCL all-reduce inside is the configs[4] number; the host-fed and eager legs are
        # one-GPU diagnostics (and a rank-asymmetric failure in a diagnostic leg must not be able to stall the headline line)
        out["final_loss"] = round(float(tr.loss), 5)
        red = getattr(tr.model, "_grad_reducer", None)
        if red is not None and red.last_stats:
            out["reducer"] = dict(red.last_stats)
        del tr
        torch.cuda.empty_cache()
        return out
    # fed from pinned host memory: 3 page-locked batches cycled through the prefetcher (copy stream, depth 2)
    for transport in ("fp32", "int16"):
        host = []
        for i in range(3):
            hw = wave.cpu().view(B, 1, L).roll(i, 0)
            hw = to_int16(hw) if transport == "int16" else hw
            host.append((hw.pin_memory(), names, y.pin_memory()))
        n_feed = steps + warmup

        class _Cycle:
            def __len__(self):
                return n_feed

            def __iter__(self):
                return (host[i % 3] for i in range(n_feed))

        def fed():
            it = iter(DevicePrefetcher(_Cycle(), dev, depth=2))
            for _ in range(warmup):
                tr.step(*next(it))
            ranks.barrier()
            t0 = time.perf_counter()
            n = 0
            for b in it:
                tr.step(*b)
                n += 1
            ranks.barrier()
            return n, time.perf_counter() - t0
        n, el = fed()
        out[f"graphed_fed_{transport}"] = {"value": round(ranks.world * B * n / el, 1), "unit": "clips/s",
                                           "ms_per_step": round(el / n * 1e3, 3),
                                           "h2d_bytes_per_step": int(host[0][0].numel() * host[0][0].element_size() + y.numel() * 4),
                                           "launch": "hipGraph replay; batches from pinned host memory through DevicePrefetcher "
                                                     f"(copy stream, depth 2), waveforms as {transport}"}
        del host
    out["final_loss"] = round(float(tr.loss), 5)
    del tr
    torch.cuda.empty_cache()
    tr = make(False)
    v, ms = timed(lambda: tr.step(xd, names, yd), max(4, steps // 2))
    out["eager"] = {"value": v, "unit": "clips/s", "ms_per_step": ms,
                    "launch": "eager KDTrainer.step (~450 launches from one host thread); batch resident in HBM"}
    del tr
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------- launch plumbing
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a torchrun environment: become the launcher of N ranks."""
    if not args.dry_run:
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {n_vis} GPU(s) visible: refusing to print a bench line")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (needed by RCCL on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    raise SystemExit(subprocess.call(cmd, env=env))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3, help="repetitions of the timed region (each exactly --steps steps); the median is reported")
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--streams", type=int, default=2, help="sub-batches issued on concurrent HIP streams per step")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-forward", action="store_true", help="skip the forward-only (configs[1]) measurement")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel event profiles (roofline objects)")
    ap.add_argument("--no-train-configs", action="store_true", help="skip the mn40_bf16 / dymn20 train steps (configs 2/3)")
    ap.add_argument("--train-model", default=None, choices=["mn10", "mn40", "mn40_bf16", "dymn10", "dymn20", "dymn20_bf16"],
                    help="only this train-step network (debug)")
    ap.add_argument("--no-fp32-exact", action="store_true", help="skip the exact-fp32 forward measurement")
    ap.add_argument("--no-kd", action="store_true", help="skip the KD training iteration (configs[4] per-GPU shard) measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel event profile to stderr")
    ap.add_argument("--calibrate-traffic", action="store_true",
                    help="run the known-byte copies (eat_calib_copy) first: under `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` they "
                         "calibrate the counters of that pass (tools/pmc_traffic.py)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rank plumbing only (gloo, CPU, no kernels): used by the CPU test of the N-rank launch path")
    return ap.parse_args(argv)


def _claim_stdout():
    """Keep fd 1 for the ONE JSON line: libraries print to stdout from C (RCCL's version banner at communicator init,
    flushed at exit, i.e. AFTER Python's own prints), so fd 1 is pointed at stderr for the run and the line is written to
    the saved descriptor at the end."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args)
    json_out = _claim_stdout()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a bench line")
    force_dist = os.environ.get("EAT_BENCH_FORCE_DIST") == "1"    # exercise the RCCL path with a single rank (debug)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")    # required to capture collectives in a graph
    if args.dry_run:
        dev = torch.device("cpu")
        if dist is not None:
            dist.init_process_group("gloo")
        ranks = Ranks(dist, world, dev)
        probe = torch.ones(1024)

        def dry_step():                      # the rank plumbing of a step: something to time + one collective per step
            time.sleep(0.001 * (rank + 1))
            if dist is not None:
                dist.all_reduce(probe)
                probe.fill_(1.0)
        el = ranks.timed(dry_step, args.steps)
        if rank == 0:
            line = {"metric": "clips/sec (10 s @ 32 kHz) mn10_as", "value": round(world * args.batch * args.steps / el, 1),
                    "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": round(el / args.steps * 1e3, 4), "dry_run": True}
            if dist is not None:
                line["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": None,
                                "buckets": 1, "bytes_per_step": 4096, "bucket_bytes": None, "forced_single_rank": False}
            print(json.dumps(line), file=json_out, flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist is not None:
        dist.init_process_group("nccl", device_id=dev)
    ranks = Ranks(dist, world, dev)

    from efficientat_amd import mn as mn_mod
    if args.calibrate_traffic and rank == 0:
        from efficientat_amd import _lib
        n = 1 << 28                                             # 1 GiB per buffer: past the 256 MiB Infinity Cache
        src = torch.rand(n, device=dev)
        dst = torch.empty(n, device=dev)
        for mode in range(4):
            for _ in range(3):
                _lib.call("eat_calib_copy", src.data_ptr(), dst.data_ptr(), n, mode, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        del src, dst
        torch.cuda.empty_cache()
    mel, model = build_model(dev)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    wave = (0.1 * torch.randn(args.batch, CLIP_SAMPLES, device=dev, generator=g)).clamp_(-1, 1)

    # ------------------------------------------------------------------ headline: the training step (fwd + bwd)
    train_name = args.train_model or "mn10"
    head = train_bench(train_name, args.batch, args.steps, args.warmup, args, mel, wave, ranks)
    arithmetic = {
        "auto": "fp32 activations and accumulation; 1x1 convs and their gradients: exact fp32 MFMA for C_in < 40, "
                "split-operand bf16x3 MFMA (x = hi + lo, 3 products, ~2^-16 rel. error) from C_in >= 40",
        "fp32": "fp32 activations, exact fp32 MFMA / VALU everywhere"}.get(os.environ.get("EAT_TRAIN_PRECISION", "auto"),
                                                                            os.environ.get("EAT_TRAIN_PRECISION", "auto"))
    alg_train = ALG_TRAIN.get(train_name)
    result = {
        "metric": "clips/sec (10 s @ 32 kHz) mn10_as fwd+bwd, 1/2/4/8 MI355X; logit max-abs-err",
        "value": head["value"], "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": head["warmup"],
        "ms_per_step": head["ms_per_step"], "repetitions": head["repetitions"], "rep_ms_per_step": head["rep_ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if train_name.endswith("bf16") else "f32", "data": "synthetic",
        "config": {"workload": f"{train_name}_as training step (log-mel + forward with batch-stat BatchNorm + BCE + backward + "
                               + ("bucketed RCCL all-reduce + " if world > 1 else "")
                               + f"fused Adam), batch {args.batch} synthetic 10 s @ 32 kHz clips per GPU, "
                               + ("bf16 GEMM operands and bf16 activation storage, fp32 statistics / parameters / optimizer "
                                  f"[BASELINE.json configs[{2 if train_name == 'mn40_bf16' else 3}]; not the headline configuration]"
                                  if train_name.endswith("bf16") else
                                  "fp32 [BASELINE.json metric; per-GPU shard of configs[4]; forward-only configs[1] in `forward`]"
                                  if train_name == "mn10" else "fp32 [BASELINE.json configs[3]; not the headline configuration]"),
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "arithmetic": arithmetic,
                   "launch": head["launch"], "train_plan": __import__("efficientat_amd.mn_train", fromlist=["x"])._TRAIN_V,
                   "parallelism": f"dp{world}" + (" (local BatchNorm statistics, gradients averaged by RCCL all-reduce "
                                                  "in ~4 MB buckets overlapped with backward)" if world > 1 else "")},
        "final_loss": head["final_loss"],
        "roofline_e2e": {"bound": "hbm", "achieved": round(head["value"] / world * alg_train / 1e9, 1) if alg_train else None,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": head["roofline_e2e_frac"],
                         "note": f"whole training step: clips/s per GPU x {(alg_train or 0) / 1e6:.1f} MB algorithmic bytes per clip "
                                 "(SURVEY 8d: 3 x forward activations + 28 B per parameter / batch + mel)"},
    }

    # ------------------------------------------------------------------ BASELINE configs[1]: forward only
    if not args.no_forward:
        clips_per_s, ms_step, launch = forward_bench(args, mel, model, wave, ranks)
        fwd = {"value": round(clips_per_s, 1), "unit": "clips/s", "ms_per_step": round(ms_step, 4), "steps": args.steps,
               "warmup": args.warmup, "batch_per_gpu": args.batch, "launch": launch,
               "workload": "mn10_as forward-only (log-mel front-end + MN eval forward), batch 256 per GPU, fp32 "
                           "[BASELINE.json configs[1]]; no collective",
               "pw_stream_mode": __import__("efficientat_amd.ops", fromlist=["x"]).pw_stream_mode(),
               "roofline_e2e_frac": round(clips_per_s / world * ALG_BYTES_PER_CLIP / HBM_PEAK, 4),
               "alg_bytes_per_clip": ALG_BYTES_PER_CLIP}
        if not args.no_fp32_exact and mn_mod._PW_MODE != "fp32":
            keep = mn_mod._PW_MODE
            mn_mod._PW_MODE = "fp32"
            model._cache.invalidate()
            try:
                v, ms, _ = forward_bench(args, mel, model, wave, ranks)
                fwd["fp32_exact"] = {"value": round(v, 1), "unit": "clips/s", "ms_per_step": round(ms, 4),
                                     "roofline_e2e_frac": round(v / world * ALG_BYTES_PER_CLIP / HBM_PEAK, 4),
                                     "what": "the same forward with every 1x1 conv on v_mfma_f32_16x16x4_f32 (exact fp32 products)"}
            finally:
                mn_mod._PW_MODE = keep
                model._cache.invalidate()
        result["forward"] = fwd

    if not args.no_fp32_exact and os.environ.get("EAT_TRAIN_PRECISION", "auto") != "fp32":
        # the same training step with every GEMM on the exact fp32 MFMA (the reference CPU path's arithmetic)
        try:
            ex = train_bench(train_name, args.batch, max(5, args.steps // 3), 2, args, mel, wave, ranks, precision="fp32")
            alg = ALG_TRAIN.get(train_name)
            result["train_step_fp32_exact"] = {
                "value": ex["value"], "unit": "clips/s", "ms_per_step": ex["ms_per_step"], "steps": ex["steps"],
                "roofline_e2e_frac": ex["roofline_e2e_frac"], "alg_bytes_per_clip": alg,
                "what": "the headline step with train_precision = 'fp32': every 1x1 conv, data gradient and weight gradient on "
                        "v_mfma_f32_16x16x4_f32 (exact fp32 products)"}
        except Exception as e:  # pragma: no cover
            result["train_step_fp32_exact"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    if not args.no_kd and args.train_model is None:
        try:
            result["kd_train_step"] = kd_bench(args, mel, wave, ranks)
        except Exception as e:  # pragma: no cover - one failing leg must not lose the line
            result["kd_train_step"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()

    if world == 1 and not args.no_train_configs and args.train_model is None:
        for key, name, bt, st, wu in [("train_step_mn40_bf16", "mn40_bf16", 128, 10, 2), ("train_step_dymn20", "dymn20", 128, 10, 2),
                                      ("train_step_dymn20_bf16", "dymn20_bf16", 128, 10, 2)]:
            try:
                result[key] = train_bench(name, bt, st, wu, args, mel, wave, ranks)
            except Exception as e:  # pragma: no cover - one failing leg must not lose the line
                result[key] = {"error": f"{type(e).__name__}: {e}", "model": name}
                torch.cuda.empty_cache()
        if not args.no_forward:
            # the eval forward of the other BASELINE widths (configs[2] / [3] models), batch 128: hipGraph replay, one stream
            for key, name in [("forward_mn40", "mn40"), ("forward_dymn20", "dymn20")]:
                try:
                    fm = make_train_model(name, dev).eval()
                    fargs = argparse.Namespace(**{**vars(args), "streams": 1, "steps": 10, "warmup": 2})
                    v, ms, launch = forward_bench(fargs, mel, fm, wave[:128], ranks)
                    result[key] = {"value": round(v, 1), "unit": "clips/s", "ms_per_step": round(ms, 4), "batch_per_gpu": 128,
                                   "steps": 10, "warmup": 2, "launch": launch,
                                   "workload": f"{name}_as forward-only (log-mel + eval forward), batch 128, fp32 activations"}
                    del fm
                except Exception as e:  # pragma: no cover
                    result[key] = {"error": f"{type(e).__name__}: {e}", "model": name}
                torch.cuda.empty_cache()
    model.eval()
    mel.eval()

    if rank == 0 and not args.no_profile:
        # ---- dominant kernel of the timed step: one eager training step with a HIP event pair around every launch
        import torch.nn.functional as F
        tm = make_train_model(train_name, dev)
        tm.train()
        mel.train()
        bt = min(args.batch, wave.shape[0])
        gy = torch.Generator(device=dev).manual_seed(99)
        y = (torch.rand((bt, 527), device=dev, generator=gy) < 2.7 / 527).float()
        opt = _adam(tm.parameters(), capturable=False)

        def tstep():
            opt.zero_grad(set_to_none=True)
            logits, _ = tm(mel(wave[:bt]).unsqueeze(1))
            F.binary_cross_entropy_with_logits(logits, y).backward()
            opt.step()
        tstep()
        prof = kernel_profile(tstep, iters=2)
        mel.eval()
        del tm, opt
        torch.cuda.empty_cache()
        step_ms = sum(v["total_ms"] for v in prof.values())
        # every launch carries a byte model (a launch priced at 0 bytes would drop out of `moved_bytes` and of the
        # dominant-kernel pick): listed - and reported - if one ever does not
        unmodelled = sorted(k for k, v in prof.items() if v["bytes"] <= 0)
        if unmodelled:
            print(f"[bench] ERROR: launches without a byte model: {unmodelled}", file=sys.stderr)
        result["roofline_e2e"]["launches_without_byte_model"] = unmodelled
        name, d = max(prof.items(), key=lambda kv: kv[1]["total_ms"])        # largest time share over ALL launches
        result["roofline"] = roofline_of(name, d, {"step_ms": step_ms, "phase": "train"})
        result["roofline"]["step"] = "training step, eager launches with a HIP event pair each (one stream)"
        moved = sum(v["bytes"] for v in prof.values())
        result["roofline_e2e"]["moved_bytes_per_step"] = int(moved)
        result["roofline_e2e"]["frac_moved"] = round(moved / (head["ms_per_step"] * 1e-3) / HBM_PEAK, 4)
        result["roofline_e2e"]["note_moved"] = ("moved = sum over the step's launches of each kernel's own input + output "
                                                "bytes (what the plan reads and writes), against the same step time")
        pm = pmc_step_bytes("train") if train_name == "mn10" else None
        if pm:
            result["roofline_e2e"]["pmc_bytes_per_step"] = pm["bytes"]
            result["roofline_e2e"]["pmc_source"] = pm["source"]
        if args.kernel_table:
            print(f"[bench] training step, {sum(v['launches'] for v in prof.values())} library launches, "
                  f"{step_ms:.2f} ms of kernels", file=sys.stderr)
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
                print(f"[bench-train] {k:36s} launches {v['launches']:3d}  {v['total_ms']:8.3f} ms  "
                      f"{v['bytes'] / 1e9:7.3f} GB  {v['gbps']:8.1f} GB/s", file=sys.stderr)
        if not args.no_forward:
            def profile_step():      # one stream: concurrent sub-batch streams would time overlapping kernels
                with torch.no_grad():
                    model(mel(wave).unsqueeze(1))
            fprof = kernel_profile(profile_step)
            fmoved = sum(v["bytes"] for v in fprof.values())
            fms = sum(v["total_ms"] for v in fprof.values())
            result["forward"]["moved_bytes_per_step"] = int(fmoved)
            result["forward"]["roofline_moved_frac"] = round(fmoved / (result["forward"]["ms_per_step"] * 1e-3) / HBM_PEAK, 4)
            result["forward"]["single_stream_kernel_ms"] = round(fms, 3)
            fname, fd = max(fprof.items(), key=lambda kv: kv[1]["total_ms"])
            result["forward"]["roofline"] = roofline_of(fname, fd, {"step_ms": fms, "phase": "forward"})
            pmf = pmc_step_bytes("forward")
            if pmf:
                result["forward"]["pmc_bytes_per_step"] = pmf["bytes"]
            if args.kernel_table:
                for k, v in sorted(fprof.items(), key=lambda kv: -kv[1]["total_ms"]):
                    print(f"[bench] {k:28s} launches {v['launches']:3d}  {v['total_ms']:8.3f} ms  "
                          f"{v['bytes'] / 1e9:7.3f} GB  {v['gbps']:8.1f} GB/s", file=sys.stderr)
    if dist is not None:
        # what a SCALE record needs to prove RCCL saw N ranks: the process group as the library reports it + the step's buckets
        red = head.get("reducer") or {}
        nv = None
        try:
            nv = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # pragma: no cover
            pass
        result["rccl"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": nv,
                          "buckets": red.get("buckets"), "bytes_per_step": red.get("bytes"),
                          "bucket_bytes": red.get("bucket_bytes"), "forced_single_rank": bool(force_dist and world == 1)}
    if rank == 0:
        err, scale = parity_probe(mel, model, dev)
        result["parity"] = {"logit_max_abs_err": err, "logit_abs_max": scale, "vs": "CPU oracle, 4 clips, same weights (eval forward)"}
        try:
            result["parity"]["train_grads"] = parity_grad_probe(dev)
        except Exception as e:  # pragma: no cover
            result["parity"]["train_grads"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            ref_root = os.environ.get("EAT_REFERENCE_ROOT")
            if ref_root and os.path.isfile(os.path.join(ref_root, "models", "mn", "model.py")):
                try:
                    result["cpu_baseline"] = cpu_baseline_reference(ref_root)
                except Exception as e:  # pragma: no cover - fall back to the port, say why
                    print(f"[bench] {e}; timing the oracle port instead", file=sys.stderr)
            if "cpu_baseline" not in result:
                result["cpu_baseline"] = cpu_baseline()
        print(json.dumps(result), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
