"""Train-mode forward / backward of the MN trunk on the HIP kernels.

The reference trains through autograd over ~210 per-op launches (`ex_audioset.py:147-199`,
`models/mn/model.py:212-231`).  Here ONE `torch.autograd.Function` owns the whole network: its
forward runs the train-mode launch plan (raw conv -> batch statistics -> fused BN-affine +
activation), its backward the hand-derived reverse plan and returns the
gradient of every parameter, so `loss.backward(); optimizer.step()` in the caller are unchanged.

Round-1 structure: each pass is a separate streaming kernel (stats, apply, reduce, ...); the
fusions (producer-epilogue statistics, consumer-prologue normalisation) come next.
Tiny (B, C)-shaped glue of the SE gate and the head (ReLU/sigmoid/hardswish derivatives, bias
sums, transposes) uses torch elementwise ops; every GEMM / conv / reduction over activations runs
in libeat_hip.so.
"""
import torch
import torch.nn.functional as F

from . import _lib, ops
from .dp import GradReducer

NONE, RELU, HSWISH, SIGMOID = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_HSWISH, ops.ACT_SIGMOID


class _Zeros:
    """Cached zero bias (the conv entry points take a bias pointer; train-mode convs have none)."""

    def __init__(self):
        self.buf = None

    def get(self, n, device):
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf = torch.zeros((max(n, 4096),), device=device, dtype=torch.float32)
        return self.buf[:n]


_zeros = _Zeros()

def _prepack_plan(model):
    """All weight packs of the trunk's 1x1 convs (forward and data-gradient forms) from one launch per step
    (ops.PrepackPlan); None while a hipGraph is being captured before the plan exists, or for geometries it does not take."""
    plan = getattr(model, "_eat_prepack_plan", None)
    if plan is not None and not plan.stale():
        return plan
    if not _PREPACK_PLAN or torch.cuda.is_current_stream_capturing():
        return None
    entries = []
    for bi, blk in enumerate(model.features[1:-1]):
        if blk.i_expand is not None:
            entries.append((("e", bi), blk.block[blk.i_expand][0].weight, False))
        pw = blk.block[blk.i_proj][0].weight
        entries += [(("p", bi), pw, False), (("pt", bi), pw, True)]
    lw = model.features[-1][0].weight
    entries += [(("l",), lw, False), (("lt",), lw, True)]
    try:
        plan = ops.PrepackPlan(entries)
    except _lib.EatHipError:                              # odd channel counts: per-matrix packs (which check for themselves)
        plan = None
    model._eat_prepack_plan = plan
    return plan


def _pk(plan, key, w, trans=False):
    return plan.get(key) if plan is not None else ops.pw_prepack(w.flatten(1), trans=trans)


class _Ones(_Zeros):
    def get(self, n, device):
        if self.buf is None or self.buf.numel() < n or self.buf.device != device:
            self.buf = torch.ones((max(n, 4096),), device=device, dtype=torch.float32)
        return self.buf[:n]


_ones = _Ones()


class _GradSink:
    """`sink[name] = grad` forwards to GradReducer.push; `finish()` returns the (averaged) gradients."""

    def __init__(self, reducer):
        self.reducer = reducer

    def __setitem__(self, name, grad):
        self.reducer.push(name, grad)

    def alloc(self, name, like):
        """Memory for the gradient of parameter `like` inside the reducer's flat bucket buffer (zero-filled, shape of the 2-D
        weight matrix), or None when the pass is not bucketed - the producer then allocates for itself (dp.GradReducer)."""
        return self.reducer.alloc(name, (like.shape[0], like.numel() // like.shape[0]), like.device)

    def finish(self):
        return self.reducer.finish()


def _conv_bn_stats(z, bn):
    return ops.bn_train_state(z, bn)


def _pw_conv_bn(x, wp, Co, bn, dev, tf=None, in_scale=None):
    """z = W x [of act(tf_a x + tf_b) * in_scale] and the state of the BatchNorm that follows: the batch statistics leave
    the conv's epilogue (ops.pw_conv_stats: no pass over z) where that exists and the layer is in training mode."""
    if _EPI_STATS and bn.training:
        z, parts = ops.pw_conv_stats(x, wp, Co, tf=tf, in_scale=in_scale)
        if z is not None:
            return z, ops.bn_state_from_partials(parts, bn, z.numel() // Co)
    if tf is not None:
        z = ops.pw_conv_tf(x, tf, wp, _zeros.get(Co, dev), Co, NONE, in_scale=in_scale)
    else:
        z = ops.pw_conv(x, wp, _zeros.get(Co, dev), Co, NONE, in_scale=in_scale)
    return z, _conv_bn_stats(z, bn)


# (the round-2 plan - separate statistics / apply passes per BatchNorm, behind EAT_TRAIN_V=1 - was removed in round 4)
def _backward_impl(ctx, model, sv, dlogits, dfeat, n_lead):
    """Backward of MNTrainFunction2 (one pass, reverse layer order)."""
    # every `g[name] = grad` hands the gradient to the data-parallel reducer, which all-reduces full
    # buckets on RCCL's stream while the remaining layers' backward kernels run (dp.py)
    # a model that was never handed to dp.enable_data_parallel keeps its gradients local (no hidden collective)
    g = _GradSink(getattr(model, "_grad_reducer", None) or GradReducer(local=True))
    g.reducer.begin_pass()               # (an aborted earlier backward must not leak its slots into this one: ADVICE r5)
    dev = dlogits.device
    dlogits = dlogits.contiguous().float()
    B = dlogits.shape[0]
    blocks = list(model.features[1:-1])
    nb = len(blocks)

    # the fp64 channel sums of every BatchNorm backward of the pass live in ONE zeroed buffer: dgamma / dbeta of all
    # layers are converted to fp32 by one launch at the end (31 conversions before) and handed to the sink together
    bn_total = 2 * sum(m.num_features for m in model.features.modules() if isinstance(m, torch.nn.BatchNorm2d))
    bn_buf = ops.zero_arena.zeros((bn_total,), torch.float64, dev)
    bn_reg = []

    def bn_sums(C, wname, bname):
        if bn_buf is None:
            return None
        off = sum(2 * r[2] for r in bn_reg)
        bn_reg.append((wname, bname, C, off))
        return bn_buf[off:off + 2 * C]

    def bn_unreserve(wname):
        if bn_reg and bn_reg[-1][0] == wname:
            bn_reg.pop()

    def bn_grads(dgam, dbet, wname, bname):
        if dgam is not None:
            g[wname], g[bname] = dgam, dbet

    x_l, z_l, st_l, S_l = sv["last"]
    last = model.features[-1]
    nm = f"features.{nb + 1}"
    if sv["head"] is None:
        # trunk mode: the head ran outside (torch autograd); dlogits is the gradient w.r.t. the last feature map
        dz, dgam, dbet = ops.bn_act_bwd(dlogits.view_as(z_l), z_l, *st_l, HSWISH,
                                        sums=bn_sums(z_l.shape[1], nm + ".1.weight", nm + ".1.bias"))
    else:
        # ---- head (mn/model.py:186-194)
        feat, u, h2, drop_mask = sv["head"]
        fc1, fc2 = model.classifier[2], model.classifier[5]
        # two launches of the library's stride-addressed tile GEMMs (csrc/se_train.hip: eat_mlp_head_bwd) - round 4 ran
        # four rocBLAS GEMMs + five torch ops here
        o2, o1 = g.alloc("classifier.5.weight", fc2.weight), g.alloc("classifier.2.weight", fc1.weight)   # (production order)
        dW1, db1, dW2, db2, dft = ops.mlp_head_bwd(dlogits, h2.contiguous(), u.contiguous(),
                                                  None if drop_mask is None else drop_mask.contiguous(), feat.contiguous(),
                                                  fc1.weight, fc2.weight, dW1_out=o1, dW2_out=o2)
        g["classifier.5.weight"], g["classifier.5.bias"] = dW2, db2
        g["classifier.2.weight"], g["classifier.2.bias"] = dW1, db1
        if dfeat is not None:
            dft = dft + dfeat
        # ---- last 1x1 conv + BN + hardswish, pooled (the pool's gradient is a per-plane constant)
        zeros_bc = torch.zeros((B, z_l.shape[1]), device=dev)
        dz, dgam, dbet = ops.bn_act_bwd(z_l, z_l, *st_l, HSWISH, gscale=zeros_bc, gadd=dft * (1.0 / S_l),
                                        sums=bn_sums(z_l.shape[1], nm + ".1.weight", nm + ".1.bias"))
    bn_grads(dgam, dbet, nm + ".1.weight", nm + ".1.bias")
    g[nm + ".0.weight"] = ops.pw_conv_wgrad(dz, x_l, out=g.alloc(nm + ".0.weight", last[0].weight)).view_as(last[0].weight)
    plan = sv.get("plan")
    if plan is not None and plan.runs != sv.get("plan_run"):
        # another forward of the model re-packed the plan's views since this pass's forward (two forwards before one
        # backward): the views hold the CURRENT weights - pack this backward's operands matrix by matrix instead
        plan = None
    wpt = _pk(plan, ("lt",), last[0].weight, trans=True)
    dout = ops.pw_conv(dz, wpt, _zeros.get(x_l.shape[1], dev), x_l.shape[1], NONE)
    del dz, z_l

    # ---- inverted residual blocks, last to first (mn/block_types.py:177-181)
    dout2 = None                   # second summand of the gradient w.r.t. the stem output (see res_ok below)
    for i in range(nb - 1, -1, -1):
        blk, rec = blocks[i], sv["blocks"][i]
        cnf = blk.cnf
        act = HSWISH if cnf.use_hs else RELU
        pre = f"features.{i + 1}.block"
        inp = rec["inp"]
        res_grad = dout if blk.use_res_connect else None
        # project conv + BN (no activation); the residual branch passes dout through unchanged
        cna = blk.block[blk.i_proj]
        nw, nbias = f"{pre}.{blk.i_proj}.1.weight", f"{pre}.{blk.i_proj}.1.bias"
        b16 = rec.get("b16", False)                    # the block's wide tensors are bf16 in HBM (forward: rec["b16"])
        want16 = b16 and rec["z_p"].dtype == torch.bfloat16 and _cast_narrow(cnf)
        dz_p, dgam, dbet = ops.bn_act_bwd(dout, rec["z_p"], *rec["st_p"], NONE, sums=bn_sums(cnf.out_channels, nw, nbias),
                                          copy16=want16)
        dz_p, dz16 = dz_p if want16 else (dz_p, None)
        bn_grads(dgam, dbet, nw, nbias)
        scale = rec.get("scale")
        wgrad = ops.pw_conv_wgrad_b16 if b16 else ops.pw_conv_wgrad
        o_p = g.alloc(f"{pre}.{blk.i_proj}.0.weight", cna[0].weight)
        if rec["y_d"] is None:     # y_d = act(BN(z_d)) was evaluated on load in the forward: the same here
            st_d = rec["st_d"]
            dWp = wgrad(dz_p, rec["z_d"], x_scale=scale, tf=(st_d[0], st_d[1], act), out=o_p)
        else:
            dWp = wgrad(dz_p, rec["y_d"], x_scale=scale, out=o_p)
        g[f"{pre}.{blk.i_proj}.0.weight"] = dWp.view_as(cna[0].weight)
        wpt = _pk(plan, ("pt", i), cna[0].weight, trans=True)
        # (what the depthwise stage below will do with dxs, decided here: without an SE gate between the two, the channel sums
        # of the depthwise BatchNorm's backward can leave the data-gradient GEMM's epilogue - ops.pw_conv_gstats)
        k = cnf.kernel
        y_e = rec["y_e"]
        no_expand = blk.i_expand is None
        src_shape = tuple((y_e if y_e is not None else rec["z_e"]).shape)
        # a block without expand conv hands its residual-branch gradient to the stem's backward kernel, which adds the
        # two on load (the merged kernel has no residual input)
        res_ok = not no_expand or res_grad is None or (i == 0 and sv["stem"][1] is None)
        use_merged = (_MERGED_DW_BWD and _DW_BN_ON_LOAD and (y_e is None or no_expand) and res_ok
                      and ops.dw_bwd_merged_ok(rec["z_d"].shape, src_shape, k, cnf.stride))
        sums_d = None
        if b16:
            if dz16 is None:
                dz16 = ops.cast_b16(dz_p) if _cast_narrow(cnf, dz_p) else dz_p
            gst = None
            if (_GSTATS_EPILOGUE and use_merged and scale is None and rec["z_d"].numel() >= _GSTATS_MIN_ELEMS
                    and dz16.dtype == torch.bfloat16 and rec["z_d"].dtype == torch.bfloat16):
                nw_d, nb_d = f"{pre}.{blk.i_dw}.1.weight", f"{pre}.{blk.i_dw}.1.bias"
                gst = (rec["z_d"], rec["st_d"], act, bn_sums(cnf.expanded_channels, nw_d, nb_d))
            dxs = ops.pw_conv_b16(dz16, wpt, _zeros.get(cnf.expanded_channels, dev), cnf.expanded_channels, NONE, out_b16=True,
                                  gstat=gst)
            if gst is not None:
                dxs, sums_d = dxs
            del dz16
        else:
            dxs = None
            if _GSTATS_EPILOGUE and use_merged and scale is None and rec["z_d"].numel() >= _GSTATS_MIN_ELEMS:
                nw_d, nb_d = f"{pre}.{blk.i_dw}.1.weight", f"{pre}.{blk.i_dw}.1.bias"
                slot = bn_sums(cnf.expanded_channels, nw_d, nb_d)
                dxs, sums_d = ops.pw_conv_gstats(dz_p, wpt, cnf.expanded_channels, rec["z_d"], rec["st_d"], act, sums=slot)
                if dxs is None:
                    bn_unreserve(nw_d)
            if dxs is None:
                dxs = ops.pw_conv(dz_p, wpt, _zeros.get(cnf.expanded_channels, dev), cnf.expanded_channels, NONE)
        del dz_p
        gscale = gadd = se_P = None
        if scale is not None:     # squeeze-excitation gate (mn/block_types.py:72-83)
            se = blk.block[blk.i_se].conc_se_layers[0]
            sp = f"{pre}.{blk.i_se}.conc_se_layers.0"
            h, pool, S_d = rec["h"], rec["pool"], rec["S_d"]
            # one pass over (dxs, z_d) for the gate gradient AND the BatchNorm-backward plane sums
            st_d = rec["st_d"]
            se_P = ops.se_bn_bwd_partials(dxs, rec["z_d"], st_d[0], st_d[1], st_d[2], act)
            ds = se_P[0]
            # the gate MLP's backward as two launches (csrc/se_train.hip)
            o2, o1 = g.alloc(sp + ".fc2.weight", se.fc2.weight), g.alloc(sp + ".fc1.weight", se.fc1.weight)
            dW1, db1, dW2, db2, gadd = ops.se_mlp_bwd(ds, scale, h, pool, se.fc1.weight, se.fc2.weight, S_d, dW1_out=o1, dW2_out=o2)
            g[sp + ".fc2.weight"], g[sp + ".fc2.bias"] = dW2, db2
            g[sp + ".fc1.weight"], g[sp + ".fc1.bias"] = dW1, db1
            gscale = scale
        # depthwise conv + BN + act
        cna = blk.block[blk.i_dw]
        merged = None
        if b16 and not (_MERGED_DW_BWD and _DW_BN_ON_LOAD):
            raise _lib.EatHipError("act_storage='bf16' runs the merged depthwise backward only (EAT_MERGED_DW_BWD / EAT_DW_BN_ON_LOAD = 1)")
        if use_merged:
            # large planes: dz_d is never written - the merged backward kernel evaluates the BatchNorm + activation
            # backward of the depthwise output on load from (dxs, z_d) and the channel sums of the reduce pass
            st_d = rec["st_d"]
            nw, nbias = f"{pre}.{blk.i_dw}.1.weight", f"{pre}.{blk.i_dw}.1.bias"
            if sums_d is not None:                     # (they left the data-gradient GEMM's epilogue above)
                sums, dgam, dbet = sums_d, None, None
            else:
                sums, dgam, dbet = ops.bn_act_bwd_sums(dxs, rec["z_d"], *st_d, act, gscale=gscale, gadd=gadd, se_P=se_P,
                                                       sums=bn_sums(cnf.expanded_channels, nw, nbias))
            bn_grads(dgam, dbet, nw, nbias)
            w_d = cna[0].weight.reshape(-1, k * k)
            if no_expand:
                C_d = cnf.expanded_channels
                dout, _, dw_d = ops.dw_conv_bwd_bn_g(dxs, rec["z_d"], st_d, act, sums, w_d, y_e, _ones.get(C_d, dev),
                                                     _zeros.get(C_d, dev), NONE, k, cnf.stride, gscale=gscale,
                                                     gadd=gadd, want_gsum=False)
                dout2 = res_grad
                g[f"{pre}.{blk.i_dw}.0.weight"] = dw_d.view_as(cna[0].weight)
                del dxs
                sv["blocks"][i] = None
                continue
            st_e = rec["st_e"]
            merged = ops.dw_conv_bwd_bn_g(dxs, rec["z_d"], st_d, act, sums, w_d, rec["z_e"], st_e[0], st_e[1], act, k,
                                          cnf.stride, gscale=gscale, gadd=gadd)
            g[f"{pre}.{blk.i_dw}.0.weight"] = merged[2].view_as(cna[0].weight)
            in_shape = src_shape
            dz_d = None
            del dxs
        else:
            nw, nbias = f"{pre}.{blk.i_dw}.1.weight", f"{pre}.{blk.i_dw}.1.bias"
            if se_P is not None:
                dz_d, dgam, dbet = ops.bn_act_bwd_se(dxs, rec["z_d"], *rec["st_d"], act, se_P, gscale, gadd,
                                                     sums=bn_sums(cnf.expanded_channels, nw, nbias))
            else:
                dz_d, dgam, dbet = ops.bn_act_bwd(dxs, rec["z_d"], *rec["st_d"], act, gscale=gscale, gadd=gadd,
                                                  sums=bn_sums(cnf.expanded_channels, nw, nbias))
            del dxs
            bn_grads(dgam, dbet, nw, nbias)
        if merged is not None:
            pass
        elif y_e is None and _MERGED_DW_BWD:
            # weight gradient, data gradient and the activation-derivative epilogue from ONE pass over dz_d and z_e
            st_e = rec["st_e"]
            merged = ops.dw_conv_bwd_g(dz_d, cna[0].weight.reshape(-1, k * k), rec["z_e"], st_e[0], st_e[1], act, k, cnf.stride)
            g[f"{pre}.{blk.i_dw}.0.weight"] = merged[2].view_as(cna[0].weight)
            in_shape = tuple(rec["z_e"].shape)
        elif y_e is None:           # fused expand-BN: the depthwise input is act(a z_e + b), evaluated on load
            st_e = rec["st_e"]
            g[f"{pre}.{blk.i_dw}.0.weight"] = ops.dw_conv_wgrad_tf(dz_d, rec["z_e"], st_e[0], st_e[1], act, k,
                                                                   cnf.stride).view_as(cna[0].weight)
            in_shape = tuple(rec["z_e"].shape)
        else:
            g[f"{pre}.{blk.i_dw}.0.weight"] = ops.dw_conv_wgrad(dz_d, y_e, k, cnf.stride).view_as(cna[0].weight)
            in_shape = tuple(y_e.shape)
        if not no_expand:
            # expand conv + BN + act without dz_e (csrc/train_fuse.hip): g = dy_e * act'(.) and sum g leave the
            # depthwise data-gradient kernel; dW / dx follow from Gx = sum g x^T and the forward's Gram products
            st_e = rec["st_e"]
            cna_e = blk.block[blk.i_expand]
            W = cna_e[0].weight.flatten(1)
            if merged is not None:
                g_e, gparts = merged[0], merged[1]
            else:
                g_e, gparts = ops.dw_conv_dgrad_g(dz_d, cna[0].weight.reshape(-1, k * k), in_shape, k, cnf.stride,
                                                  rec["z_e"], st_e[0], st_e[1], act)
            del dz_d
            frozen = getattr(st_e[2], "_eat_frozen", False)
            n_e = inp.shape[0] * inp.shape[2] * inp.shape[3]
            Gx = wgrad(g_e, inp)
            Tm, sx = (Gx, st_e[2]) if frozen else (rec["Tm"], rec["sx"])      # frozen: not read (m1 = m2 = 0)
            S_e = inp.shape[2] * inp.shape[3]
            o_e = g.alloc(f"{pre}.{blk.i_expand}.0.weight", cna_e[0].weight)
            cat_b16 = b16 and not frozen and _CAT_DGRAD and cnf.expanded_channels % 32 == 0
            cat_f32 = not b16 and not frozen and _CAT_DGRAD and S_e % 4 == 0
            if cat_b16 or cat_f32:
                # dx = [WaT | M] [g ; x] + c0 (+ residual-branch gradient): one GEMM over both tensors, its weight pack and
                # bias from ONE launch after the coefficient kernel (round 6: was transposes + pack + M GEMM + cat + pack)
                dW, dgam, dbet, wcat, c0 = ops.expand_bwd_coef_cat(W, Gx, Tm, sx, gparts, st_e[0], st_e[2], st_e[3], n_e,
                                                                   centered=True, dW_out=o_e)
            else:
                dW, dgam, dbet, WaT, M, c0 = ops.expand_bwd_coef(W, Gx, Tm, sx, gparts, st_e[0], st_e[2], st_e[3], n_e,
                                                                 frozen=frozen, centered=not frozen, dW_out=o_e)
            g[f"{pre}.{blk.i_expand}.1.weight"], g[f"{pre}.{blk.i_expand}.1.bias"] = dgam, dbet
            g[f"{pre}.{blk.i_expand}.0.weight"] = dW.view_as(cna_e[0].weight)
            if cat_b16:
                # the same GEMM with g read from its bf16 storage (g's channels fill whole k-chunks)
                dout = ops.pw_conv_b16(g_e, wcat, c0, cnf.input_channels, NONE, x2=inp, res=res_grad)
            elif b16:
                t = res_grad
                if not frozen:
                    t = ops.pw_conv(inp, ops.pw_prepack(M), c0, cnf.input_channels, NONE, res=res_grad)
                dout = ops.pw_conv_b16(g_e, ops.pw_prepack(WaT), _zeros.get(cnf.input_channels, dev), cnf.input_channels,
                                       NONE, res=t)
            elif cat_f32:
                dout = ops.pw_conv_cat(g_e, inp, wcat, c0, cnf.input_channels, NONE, res=res_grad)
            else:
                t = res_grad
                if not frozen:                                                  # M x + c0 (+ residual-branch gradient)
                    t = ops.pw_conv(inp, ops.pw_prepack(M), c0, cnf.input_channels, NONE, res=res_grad)
                dout = ops.pw_conv(g_e, ops.pw_prepack(WaT), _zeros.get(cnf.input_channels, dev), cnf.input_channels,
                                   NONE, res=t)
            del g_e
            sv["blocks"][i] = None
            continue
        dy_e = ops.dw_conv_dgrad(dz_d, cna[0].weight.reshape(-1, k * k), in_shape, k, cnf.stride, res=res_grad)
        del dz_d
        dout = dy_e                    # (only a block without expand conv gets here)
        sv["blocks"][i] = None

    # ---- stem
    x, z0, st0 = sv["stem"][:3]
    stem = model.features[0]
    if z0 is None:
        # one pass over the block-0 gradient: g = dout * hswish'(.) from the log-mel window, Gx = sum g p^T, sum g
        Tm0, sp0 = sv["stem"][3:]
        W0 = stem[0].weight.reshape(-1, 9)
        Gx, gparts = ops.stem_bwd(dout, x, W0, st0[0], st0[1], HSWISH, dy2=dout2)
        frozen = getattr(st0[2], "_eat_frozen", False)
        if frozen:
            Tm0, sp0 = Gx, _zeros.get(9, dev)                                         # not read (m1 = m2 = 0)
        dW, dgam, dbet = ops.expand_bwd_coef(W0, Gx, Tm0, sp0, gparts, st0[0], st0[2], st0[3], dout.numel() // W0.shape[0],
                                             frozen=frozen, need_dx=False)[:3]
        g["features.0.1.weight"], g["features.0.1.bias"] = dgam, dbet
        g["features.0.0.weight"] = dW.view_as(stem[0].weight)
    else:
        dz0, dgam, dbet = ops.bn_act_bwd(dout, z0, *st0, HSWISH)
        g["features.0.1.weight"], g["features.0.1.bias"] = dgam, dbet
        g["features.0.0.weight"] = ops.dw_conv_wgrad(dz0, x, 3, 2).view_as(stem[0].weight)
    if bn_reg:
        sf = bn_buf.float()
        for wname, bname, C, off in bn_reg:
            g[wname], g[bname] = sf[off + C:off + 2 * C], sf[off:off + C]
    grads = g.finish()
    return (None,) * n_lead + tuple(grads.get(n) for n in ctx.names)


# ---------------------------------------------------------------------------------------------------------------------
# The training plan (round 3).  Same math as one statistics pass + one apply pass per BatchNorm, fewer passes over the expanded tensors
# (csrc/train_fuse.hip has the algebra):
#   * expand conv: BatchNorm statistics from the Gram matrix of the block input (no pass over z_e); backward through
#     BN + activation WITHOUT dz_e: the depthwise data-gradient kernel's epilogue writes g = dy_e * act'(.) and sum g,
#     the weight gradient / data gradient are wgrad(g, x) and two 1x1 convs with corrected weights;
#   * depthwise conv: sum / sum of squares of its output leave the conv kernel's epilogue as per-wave partials.
# Expanded-resolution passes per block: forward 3 -> 2, backward 9 -> 5.
_TRAIN_V = 2                         # (bench.py reports it; the round-2 plan behind EAT_TRAIN_V=1 was removed in round 4)
# The pieces of the plan.  Each was an environment switch while it was being measured against the form it replaced
# (rounds 3 / 4: DESIGN 3.12, 3.14); round 5 fixed them - the other branch of every test below is still LIVE code, taken
# for the geometries / layer states the fused form does not cover (planes that are not a multiple of 4, frozen BatchNorm,
# blocks without expand conv, SE blocks on small planes, a hipGraph capture before the pack plan exists).
_FUSE_DW_BN = True        # BatchNorm + activation (+ SE scale) of the depthwise output on load in the project conv / its wgrad
_MERGED_DW_BWD = True     # depthwise weight + data gradient (+ derivative epilogue) as one kernel (csrc/dw_plane.hip)
_DW_BN_ON_LOAD = True     # ... with the depthwise BatchNorm's own backward evaluated on load
_EPI_STATS = True         # project / last conv: BatchNorm statistics in the 1x1 epilogue
_PREPACK_PLAN = True      # all weight packs of the step from one launch
_FUSE_STEM = True         # stem without its pre-activation tensor (csrc/stem_train.hip)
_CAST_NARROW_MIN_CEXP = 200    # bf16-storage plan: bf16 copy of the narrow operand of the expand / data-gradient conv from this width
_GSTATS_EPILOGUE = True  # depthwise BatchNorm backward sums from the project data-gradient GEMM's epilogue (blocks without SE gate)
# (same-box A/B of the mn10 step: 24.29 / 24.17 ms without, 23.97 / 23.96 ms with; mn40 bf16 storage: 42.19 vs 42.17 ms - there
# the reduce pass reads 2-byte tensors and the epilogue saves little)
# ... from this size of the depthwise output on (same-box microbenchmarks at B = 256, conv + reduce pass -> conv with the
# epilogue + finish: 16 -> 16 at 64 x 500 387 -> 299 us, 24 -> 64 at 32 x 250 352 -> 280, 24 -> 72 378 -> 332; the 8 x 63 layers
# 80 -> 240 / 200 / 184: 89 -> 110, 76 -> 87, 67 -> 65 us - their reduce pass is cheaper than the epilogue's extra phase)
_GSTATS_MIN_ELEMS = 1 << 26
_CAT_DGRAD = True         # expand data gradient + BatchNorm correction as one two-source GEMM
# (round 6: the pack of [WaT | M] and c0 come from ONE launch after the coefficient kernel - eat_expand_bwd_wcat; same-box
#  A/B against the five-launch form: mn10 24.23 -> 24.06 ms, mn40 bf16 42.59 -> 41.89 ms)

def _w_times_g(W, G):
    """T = W G (C_out, C_in) for a symmetric G (C_in, C_in), C_in % 4 == 0."""
    Co, Ci = W.shape
    if Ci % 4:
        return ops.linear(W, G, None, NONE)
    wp3 = ops.pw_prepack_bf16(W, None, split=True)
    return ops.pw_conv_bf16(G.view(1, Ci, Ci, 1), wp3, _zeros.get(Co, W.device), Co, NONE, split=True).view(Co, Ci)


def _cast_narrow(cnf, x=None):
    """bf16-storage plan: hand the expand / data-gradient 1x1 conv of the widest blocks a bf16 COPY of its narrow fp32 operand
    (bit-identical results - the kernel rounds the operand the same way; measured on MI355X at B = 128: 320 -> 1920 at S = 504
    209 -> 144 us, 448 -> 2688 324 -> 238 us, 640 -> 3840 at S = 128 123 -> 91 us, 160 -> 960 at S = 2000 275 -> 220 us,
    64 -> 256 at S = 32000 667 -> 528 us; same-box A/B of the mn40 step with copies from expanded width 960 / 700 / 400 / 200:
    42.71 / 42.72 / 42.60 / 42.43 ms - every block with an expand conv).  The copy leaves the pass that produces the fp32 tensor
    (project BatchNorm forward / backward apply: `copy16`); `eat_cast_b16` (18 / 24 / 10 us on the shapes above) is the
    fallback when that pass ran on the fp32 kernels (frozen BatchNorm)."""
    return cnf.expanded_channels >= _CAST_NARROW_MIN_CEXP and (x is None or x.numel() % 8 == 0)


def _act_storage_bf16(model):
    """True when the step stores its wide activations in bf16 (model.act_storage, mn.py); checks the combination."""
    st = getattr(model, "act_storage", "fp32")
    if st == "fp32":
        return False
    if st != "bf16":
        raise _lib.EatHipError(f"act_storage must be 'fp32' or 'bf16' (got {st!r})")
    if ops.precision.mode != "bf16":
        raise _lib.EatHipError("act_storage='bf16' needs train_precision='bf16' (plain bf16 GEMM operands): the split-operand "
                               f"kernels read fp32 activations (train_precision={ops.precision.mode!r})")
    return True


class MNTrainFunction2(torch.autograd.Function):
    """mode 0: the whole network incl. the mlp head -> (logits, features).  mode 1 ("trunk"): up to the last feature map
    -> (y_last, fmap_0 ... fmap_15): the caller runs the head on y_last (non-default heads, `return_fmaps`); the block
    outputs are handed out as non-differentiable tensors."""

    @staticmethod
    def forward(ctx, model, x, drop_mask, mode, *params):
        keep = torch.is_grad_enabled() or any(p.requires_grad for p in params)
        dev = x.device
        x = x.contiguous().float()
        B = x.shape[0]
        saved = {}
        blocks = list(model.features[1:-1])
        exact = ops.precision.mode == "fp32"
        store16 = _act_storage_bf16(model)
        plan = _prepack_plan(model)
        if plan is not None:
            plan.run()
        saved["plan"], saved["plan_run"] = plan, (plan.runs if plan is not None else 0)

        # stem
        stem = model.features[0]
        C0 = stem[0].out_channels
        W0 = stem[0].weight.reshape(C0, 9)
        need_sx = bool(blocks) and blocks[0].i_expand is not None
        if _FUSE_STEM and not need_sx:
            # the pre-activation stem tensor never exists: statistics from the Gram matrix of the log-mel's 3x3 patches,
            # BatchNorm + Hardswish in the conv's epilogue (csrc/stem_train.hip)
            Fo, To = (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1
            if stem[1].training:
                Tm0, sp0 = ops.stem_gram(x, W0)
                st0 = ops.gram_bn_state(Tm0, W0, sp0, stem[1], B * Fo * To)
            else:
                Tm0, sp0, st0 = None, None, ops.bn_frozen_state(stem[1])
            cur = ops.stem_conv(x, W0 * st0[0].unsqueeze(1), st0[1], HSWISH)
            sx = None
            saved["stem"] = (x, None, st0, Tm0, sp0)
        else:
            z0 = ops.stem_conv(x, W0, _zeros.get(C0, dev), NONE)
            st0 = (ops.bn_state_from_partials(ops.bn_stats_partial(z0), stem[1], z0.numel() // C0) if stem[1].training
                   else ops.bn_frozen_state(stem[1]))
            pool_c = torch.empty((B, C0), device=dev) if need_sx else None
            cur = ops.bn_act_fwd(z0, st0[0], st0[1], HSWISH, pool=pool_c)
            sx = ops.col_sum(pool_c) if need_sx else None
            saved["stem"] = (x, z0, st0)
        fmaps = [cur]

        blk_saved = []
        cur16 = None                                        # bf16 copy of `cur` (bf16-storage plan, _cast_narrow)
        for bi, blk in enumerate(blocks):
            cnf = blk.cnf
            act = HSWISH if cnf.use_hs else RELU
            inp, inp16 = cur, cur16
            rec = {"inp": inp}
            k = cnf.kernel
            cna_d = blk.block[blk.i_dw]
            w_d = cna_d[0].weight.reshape(-1, k * k)
            tf = None
            # bf16 storage of this block's wide tensors (z_e, z_d, y_d; backward: dxs, g) where the kernels cover its geometry
            # (a block without expand conv: its depthwise conv reads the fp32 block input and its backward hands an fp32
            # gradient to the layer below - z_d, y_d and dxs are the bf16 tensors there)
            b16 = (store16 and cna_d[1].training
                   and ops.b16_block_ok(B, cnf.expanded_channels, inp.shape[2], inp.shape[3], k, cnf.stride))
            if b16 and blk.i_expand is None and blk.use_res_connect and not (bi == 0 and saved["stem"][1] is None):
                # the backward's merged depthwise kernel has no residual input: a residual block without expand conv that is
                # not the stem's consumer (custom inverted_residual_setting) takes the fp32 passes there, which read fp32
                # tensors - keep this block's storage fp32 (the predicate of _backward_impl's `res_ok`)
                b16 = False
            rec["b16"] = b16
            if blk.i_expand is not None:
                cna = blk.block[blk.i_expand]
                W = cna[0].weight.flatten(1)
                n_e = B * inp.shape[2] * inp.shape[3]
                if cna[1].training:
                    # centred Gram matrix of the block input (reproducible; x - mean on load: the variance is not a
                    # difference of two (mean / std)^2-times larger sums)
                    G = ops.gram(inp, exact=exact, sx=sx, plain_bf16=store16)
                    if W.shape[1] <= 192:
                        Tm, st_e = ops.gram_bn_state_g(G, W, sx, cna[1], n_e, centered=True)   # T = W Gc and the BatchNorm state, one launch
                    else:
                        # wide inputs (mn40: up to 640 channels): every block of the one-launch form would stream the whole
                        # G from L2 (C_out x C_in^2 floats: 0.79 ms at 3840 x 640) - W G on the matrix cores instead: as the
                        # 1x1 conv of the "image" G (1, C_in, C_in, 1) with W on the split-operand bf16x3 kernel (fp32-class
                        # products whatever the step's precision: T feeds the BatchNorm variance; round 4 ran the K-split
                        # `linear` kernel here, 15 TFLOP/s on 3840 x 640 x 640)
                        Tm = ops.linear(W, G, None, NONE) if exact else _w_times_g(W, G)
                        st_e = ops.gram_bn_state(Tm, W, sx, cna[1], n_e, centered=True)
                else:
                    Tm, st_e = None, ops.bn_frozen_state(cna[1])
                wp = _pk(plan, ("e", bi), cna[0].weight)
                if b16:
                    x16 = inp16 if inp16 is not None else (ops.cast_b16(inp) if _cast_narrow(cnf, inp) else inp)
                    z_e = ops.pw_conv_b16(x16, wp, _zeros.get(cnf.expanded_channels, dev), cnf.expanded_channels, NONE,
                                          out_b16=True)
                    del x16
                else:
                    z_e = ops.pw_conv(inp, wp, _zeros.get(cnf.expanded_channels, dev), cnf.expanded_channels, NONE)
                rec.update(z_e=z_e, st_e=st_e, Tm=Tm, sx=sx)
                tf = (st_e[0], st_e[1], act)
            src = z_e if blk.i_expand is not None else inp
            if cna_d[1].training:
                z_d, parts = ops.dw_conv_stats(src, w_d, k, cnf.stride, tf=tf, out_b16=b16)
                st_d = ops.bn_state_from_partials(parts, cna_d[1], z_d.numel() // cnf.expanded_channels)
            else:
                if tf is not None:
                    z_d = ops.dw_conv_tf(src, tf[0], tf[1], act, w_d, _zeros.get(cnf.expanded_channels, dev), k, cnf.stride)
                else:
                    z_d = ops.dw_conv(src, w_d, _zeros.get(cnf.expanded_channels, dev), k, cnf.stride, NONE)
                st_d = ops.bn_frozen_state(cna_d[1])
            S_d = z_d.shape[2] * z_d.shape[3]
            pool = torch.empty((B, cnf.expanded_channels), device=dev) if blk.i_se is not None else None
            # on-load BatchNorm + activation in the project conv: saves the write AND the read of y_d in blocks without
            # SE, only the write in SE blocks (the squeeze needs a pass anyway) - there it pays on the large planes only
            # (measured: on the 8x63 / 4x32 SE blocks the transformed GEMM and weight gradient cost more than the write)
            on_load = _FUSE_DW_BN and ops.pw_tf_eligible(cnf.expanded_channels, S_d) and (blk.i_se is None or S_d >= 2000)
            y_d = ops.bn_act_fwd(z_d, st_d[0], st_d[1], act, pool=pool, write=not on_load) if (pool is not None or not on_load) else None
            rec.update(y_e=None if blk.i_expand is not None else inp, z_d=z_d, st_d=st_d, y_d=y_d)
            scale = None
            if blk.i_se is not None:
                se = blk.block[blk.i_se].conc_se_layers[0]
                h = ops.linear(pool, se.fc1.weight, se.fc1.bias, RELU, 1.0 / S_d)
                scale = ops.linear(h, se.fc2.weight, se.fc2.bias, SIGMOID)
                rec.update(pool=pool, h=h, scale=scale, S_d=S_d)
            cna = blk.block[blk.i_proj]
            wp = _pk(plan, ("p", bi), cna[0].weight)
            if b16:
                # project conv from the bf16-stored z_d (BatchNorm + activation on load) or y_d, statistics in its epilogue
                src_p, tf_p = (z_d, (st_d[0], st_d[1], act)) if on_load else (y_d, None)
                if cna[1].training:
                    # z_p too is stored in bf16 (statistics of the stored values); the block output below is fp32
                    z_p, parts = ops.pw_conv_b16(src_p, wp, _zeros.get(cnf.out_channels, dev), cnf.out_channels, NONE, tf=tf_p,
                                                 in_scale=scale, stats=True, out_b16=True)
                    st_p = ops.bn_state_from_partials(parts, cna[1], z_p.numel() // cnf.out_channels)
                else:
                    z_p = ops.pw_conv_b16(src_p, wp, _zeros.get(cnf.out_channels, dev), cnf.out_channels, NONE, tf=tf_p, in_scale=scale)
                    st_p = ops.bn_frozen_state(cna[1])
            elif on_load:
                z_p, st_p = _pw_conv_bn(z_d, wp, cnf.out_channels, cna[1], dev, tf=(st_d[0], st_d[1], act), in_scale=scale)
            else:
                z_p, st_p = _pw_conv_bn(y_d, wp, cnf.out_channels, cna[1], dev, in_scale=scale)
            need_sx = bi + 1 < len(blocks) and blocks[bi + 1].i_expand is not None
            pool_c = torch.empty((B, cnf.out_channels), device=dev) if need_sx else None
            nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
            want16 = (b16 and z_p.dtype == torch.bfloat16 and nxt is not None and nxt.i_expand is not None
                      and _cast_narrow(nxt.cnf))
            cur = ops.bn_act_fwd(z_p, st_p[0], st_p[1], NONE, res=inp if blk.use_res_connect else None, pool=pool_c, y_f32=True,
                                 copy16=want16)
            cur, cur16 = cur if want16 else (cur, None)
            sx = ops.col_sum(pool_c) if need_sx else None
            rec.update(z_p=z_p, st_p=st_p)
            blk_saved.append(rec)
            fmaps.append(cur)

        last = model.features[-1]
        c_feat = last.out_channels
        wp = _pk(plan, ("l",), last[0].weight)
        z_l, st_l = _pw_conv_bn(cur, wp, c_feat, last[1], dev)
        S_l = z_l.shape[2] * z_l.shape[3]
        ctx.mode = mode
        if mode == 1:
            y_l = ops.bn_act_fwd(z_l, st_l[0], st_l[1], HSWISH)
            if keep:
                saved.update(blocks=blk_saved, last=(cur, z_l, st_l, S_l), head=None)
                ctx.saved, ctx.model = saved, model
                ctx.names = [n for n, _ in model.named_parameters()]
            ctx.mark_non_differentiable(*fmaps)
            return (y_l,) + tuple(fmaps)
        pooled = torch.empty((B, c_feat), device=dev)
        ops.bn_act_fwd(z_l, st_l[0], st_l[1], HSWISH, pool=pooled, write=False)
        feat = pooled * (1.0 / S_l)
        fc1, fc2 = model.classifier[2], model.classifier[5]
        u = ops.linear(feat, fc1.weight, fc1.bias, NONE)
        h2 = F.hardswish(u)
        if drop_mask is not None:
            h2 = h2 * drop_mask
        logits = ops.linear(h2, fc2.weight, fc2.bias, NONE)
        if keep:
            saved.update(blocks=blk_saved, last=(cur, z_l, st_l, S_l), head=(feat, u, h2, drop_mask))
            ctx.saved, ctx.model = saved, model
            ctx.names = [n for n, _ in model.named_parameters()]
        return logits, feat

    @staticmethod
    def backward(ctx, dout0, *rest):
        model, sv = ctx.model, ctx.saved
        ctx.saved = None
        dfeat = rest[0] if ctx.mode == 0 else None
        with ops.precision(getattr(model, "train_precision", "fp32")), ops.zero_arena.scope("mn_bwd"):
            return _backward_impl(ctx, model, sv, dout0, dfeat, n_lead=4)


def forward_train(model, x, return_fmaps=False):
    """Train-mode `(logits, features)` - or `(logits, fmaps)` - with autograd support (mn/model.py:212-231 in `.train()`).

    The default network (mlp head) is one Function incl. the head.  Non-default heads (`fully_convolutional`,
    `multihead_attention_pooling`; mn/model.py:170-185) and `return_fmaps=True` run the trunk Function up to the last
    feature map and the head module on top of it under torch autograd (both heads act on the 960 x 4 x 32 map only).  The 17 feature maps of
    `return_fmaps` are values only (no gradient flows back through them; the logits are differentiable as usual)."""
    drop = model.classifier[4] if model.head_type == "mlp" else None
    mask = None
    if drop is not None and drop.p > 0 and drop.training:
        n_hidden = model.classifier[2].out_features
        mask = torch.empty((x.shape[0], n_hidden), device=x.device).bernoulli_(1.0 - drop.p) / (1.0 - drop.p)
    override = getattr(model, "_drop_mask_override", None)       # tests replay the reference's mask
    if override is not None and drop is not None and drop.training:
        mask = override.to(x.device).float() / (1.0 - drop.p)
    params = [p for _, p in model.named_parameters()]
    trunk = return_fmaps or model.head_type != "mlp"
    model._eat_trunk_active = trunk       # dp.py: the head's gradients come from torch autograd then (their own reducer hooks)
    with ops.precision(getattr(model, "train_precision", "fp32")), ops.bn_counters, ops.zero_arena.scope("mn_fwd"):
        if not trunk:
            return MNTrainFunction2.apply(model, x, mask, 0, *params)
        outs = MNTrainFunction2.apply(model, x, None, 1, *params)
        y_l, fmaps = outs[0], list(outs[1:])
        feat = y_l.mean(dim=(2, 3))
        if model.head_type == "mlp":
            fc1, fc2 = model.classifier[2], model.classifier[5]
            h = F.hardswish(F.linear(feat, fc1.weight, fc1.bias))
            if mask is not None:
                h = h * mask
            logits = F.linear(h, fc2.weight, fc2.bias)
        else:
            # fully-convolutional head (1x1 conv to the classes + BatchNorm + mean: the class count, 527, is not a
            # multiple of 4, which the library's data-gradient GEMM needs) and the attention-pooling module: a few MB of
            # torch ops on the 4 x 32 map under torch autograd
            logits = model.classifier(y_l).reshape(x.shape[0], -1)
    return (logits, fmaps + [y_l]) if return_fmaps else (logits, feat)


# ---------------------------------------------------------------------------------------------------------------------
# Modular train path for the variants the monolithic plan does not cover: dilated tails (`dilated=True`,
# models/mn/model.py:244-269) and squeeze-excitation over the time axis / several axes at once (`se_dims` beyond 'c',
# models/mn/block_types.py:10-83).  Neither has a released checkpoint or a measured configuration: every conv / BatchNorm is
# the library's kernel behind a per-layer autograd Function (the ones DyMN's static blocks use, dymn_train.py), the
# concurrent SE block - a mean over two axes, two Linears on a vector of <= 960 entries, a broadcast product and the
# max / avg / add / min of the gated copies - is torch ops under torch autograd.
def _concurrent_se_train(se_block, y):
    """ConcurrentSEBlock.forward (block_types.py:36-42) over SqueezeExcitation.forward (:72-83), differentiable."""
    outs = []
    for se in se_block.conc_se_layers:
        d = se.gate_dim                                            # the axis that keeps its extent (1 = c, 3 = t)
        m = y.mean(dim=[k for k in (1, 2, 3) if k != d])           # (B, extent)
        gate = torch.sigmoid(F.linear(F.relu(F.linear(m, se.fc1.weight, se.fc1.bias)), se.fc2.weight, se.fc2.bias))
        shape = [y.shape[0], 1, 1, 1]
        shape[d] = gate.shape[1]
        outs.append(gate.view(shape) * y)
    if len(outs) == 1:
        return outs[0]
    st = torch.stack(outs, dim=0)
    agg = se_block.se_agg
    return (st.max(dim=0)[0] if agg == "max" else st.mean(dim=0) if agg == "avg" else st.sum(dim=0) if agg == "add"
            else st.min(dim=0)[0])


def forward_train_modular(model, x, return_fmaps=False):
    """Train-mode `(logits, features)` / `(logits, fmaps)` of an MN with dilated blocks or SE beyond the channel axis."""
    from .dymn_train import BnAct, DwConv, Linear, PwConv, StemConv
    drop = model.classifier[4] if model.head_type == "mlp" else None
    with ops.precision(getattr(model, "train_precision", "fp32")), ops.bn_counters, ops.zero_arena.scope("mn_fwd"):
        x = x.contiguous().float()
        stem = model.features[0]
        cur = BnAct.apply(StemConv.apply(x, stem[0].weight), stem[1].weight, stem[1].bias, stem[1], HSWISH)
        fmaps = [cur]
        for blk in model.features[1:-1]:
            cnf = blk.cnf
            act = HSWISH if cnf.use_hs else RELU
            inp = cur
            if blk.i_expand is not None:
                conv, bn = blk.block[blk.i_expand][0], blk.block[blk.i_expand][1]
                cur = BnAct.apply(PwConv.apply(cur, conv.weight), bn.weight, bn.bias, bn, act)
            conv, bn = blk.block[blk.i_dw][0], blk.block[blk.i_dw][1]
            cur = BnAct.apply(DwConv.apply(cur, conv.weight, cnf.kernel, blk.dw_stride, cnf.dilation), bn.weight, bn.bias, bn, act)
            if blk.i_se is not None:
                cur = _concurrent_se_train(blk.block[blk.i_se], cur)
            conv, bn = blk.block[blk.i_proj][0], blk.block[blk.i_proj][1]
            cur = BnAct.apply(PwConv.apply(cur, conv.weight), bn.weight, bn.bias, bn, NONE)
            if blk.use_res_connect:
                cur = cur + inp
            fmaps.append(cur)
        last = model.features[-1]
        y_l = BnAct.apply(PwConv.apply(cur, last[0].weight), last[1].weight, last[1].bias, last[1], HSWISH)
        feat = y_l.mean(dim=(2, 3))
        if model.head_type == "mlp":
            fc1, fc2 = model.classifier[2], model.classifier[5]
            h = F.hardswish(Linear.apply(feat, fc1.weight, fc1.bias))
            override = getattr(model, "_drop_mask_override", None)
            if drop is not None and drop.training:
                if override is not None:
                    h = h * (override.to(h.device).float() / (1.0 - drop.p))
                elif drop.p > 0:
                    h = F.dropout(h, drop.p, True)
            logits = Linear.apply(h, fc2.weight, fc2.bias)
        else:
            logits = model.classifier(y_l).reshape(x.shape[0], -1)
    return (logits, fmaps + [y_l]) if return_fmaps else (logits, feat)
