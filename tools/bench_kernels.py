"""Per-kernel timings of the train step's candidates at B = 256 (GPU diagnostic, not product):
   python tools/bench_kernels.py [stem] [dwbn]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientat_amd import ops

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["stem", "dwbn", "se"]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 256
if "stem" in what:
    x = torch.randn(B, 1, 128, 1000, device=dev)
    W = torch.randn(16, 9, device=dev) * 0.3
    a, b = torch.rand(16, device=dev) + 0.5, torch.randn(16, device=dev) * 0.3
    dy = torch.randn(B, 16, 64, 500, device=dev)
    with ops.zero_arena.scope("k"):
        print("stem_gram   %8.1f us" % timeit(lambda: ops.stem_gram(x, W)))
        print("stem_conv   %8.1f us" % timeit(lambda: ops.stem_conv(x, W, b, ops.ACT_HSWISH)))
        print("stem_bwd    %8.1f us" % timeit(lambda: ops.stem_bwd(dy, x, W, a, b, ops.ACT_HSWISH)))
        print("stem wgrad (old) %8.1f us" % timeit(lambda: ops.dw_conv_wgrad(dy, x, 3, 2)))
if "dwbn" in what:
    for (C, F, T, k, s, act, noexp) in [(16, 64, 500, 3, 1, 1, True), (64, 64, 500, 3, 2, 1, False), (72, 32, 250, 3, 1, 1, False),
                                        (72, 32, 250, 5, 2, 1, False), (120, 16, 125, 5, 1, 1, False), (240, 16, 125, 3, 2, 2, False),
                                        (200, 8, 63, 3, 1, 2, False), (672, 8, 63, 3, 1, 2, False), (672, 8, 63, 5, 2, 2, False),
                                        (960, 4, 32, 5, 1, 2, False)]:
        p = (k - 1) // 2
        Fo, To = (F + 2 * p - k) // s + 1, (T + 2 * p - k) // s + 1
        x = torch.randn(B, C, F, T, device=dev)
        z = torch.randn(B, C, Fo, To, device=dev)
        dy = torch.randn(B, C, Fo, To, device=dev)
        w = torch.randn(C, k * k, device=dev) * 0.3
        st = (torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1,
              torch.rand(C, device=dev) + 0.5)
        ia, ib = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if noexp else (st[0], st[1])
        in_act = 0 if noexp else act
        res = torch.randn(B, C, F, T, device=dev) if noexp else None
        with ops.zero_arena.scope("k"):
            sums, _, _ = ops.bn_act_bwd_sums(dy, z, *st, act)
            t_new = timeit(lambda: ops.dw_conv_bwd_bn_g(dy, z, st, act, sums, w, x, ia, ib, in_act, k, s, want_gsum=not noexp))
            t_red = timeit(lambda: ops.bn_act_bwd_sums(dy, z, *st, act))
            t_app = timeit(lambda: ops.bn_act_bwd(dy, z, *st, act)) - t_red
            dz = torch.randn_like(z)
            if noexp:
                t_old = timeit(lambda: ops.dw_conv_wgrad(dz, x, k, s)) + timeit(lambda: ops.dw_conv_dgrad(dz, w, tuple(x.shape), k, s, res=res))
            else:
                t_old = timeit(lambda: ops.dw_conv_bwd_g(dz, w, x, ia, ib, in_act, k, s))
        print(f"C={C} {F}x{T} k{k} s{s}: on-load {t_new:8.1f} us | apply {t_app:8.1f} + old backward {t_old:8.1f} = {t_app + t_old:8.1f} us (reduce {t_red:.1f})")

if "se" in what:
    for (C, Cr, S) in [(72, 24, 2000), (120, 32, 2000), (480, 120, 504), (672, 168, 504), (960, 240, 128)]:
        ds, sc = torch.randn(B, C, device=dev), torch.rand(B, C, device=dev)
        h, pool = torch.relu(torch.randn(B, Cr, device=dev)), torch.randn(B, C, device=dev) * S
        W1, W2 = torch.randn(Cr, C, device=dev) * 0.1, torch.randn(C, Cr, device=dev) * 0.1
        print(f"se_mlp_bwd C={C} Cr={Cr}: %8.1f us" % timeit(lambda: ops.se_mlp_bwd(ds, sc, h, pool, W1, W2, S)))
if "wgrad" in what:
    # 1x1 weight gradients at the mn10 late-layer shapes (B = 256) on the library's shipped dispatch (the round-4 A/B switch
    # EAT_WGRAD_WIDE is a constant since round 5; compare builds with EAT_LIB instead)
    shapes = [(960, 160, 128), (160, 960, 128), (160, 672, 128), (672, 112, 504), (112, 672, 504), (112, 480, 504), (480, 80, 504),
              (184, 80, 504), (200, 80, 504), (240, 40, 2000), (120, 40, 2000), (72, 24, 8000)]
    tfs = [(80, 184, 504), (80, 240, 504), (40, 120, 2000), (40, 72, 2000), (24, 72, 8000)]
    print("library:", os.environ.get("EAT_LIB", "(in-tree build)"))
    with ops.precision("auto"):
        for (Co, Ci, S) in shapes:
            dz, x = torch.randn(B, Co, S, 1, device=dev), torch.randn(B, Ci, S, 1, device=dev)
            with ops.zero_arena.scope("k"):
                t = timeit(lambda: ops.pw_conv_wgrad(dz, x), n=20)
            mb = (Co + Ci) * S * B * 4 / 1e6
            print("wgrad %4d x %4d @ %5d  %7.1f us  %6.0f MB alg  %5.2f TB/s" % (Co, Ci, S, t, mb, mb / t))
        for (Co, Ci, S) in [(160, 960, 128), (160, 672, 128), (112, 672, 504), (112, 480, 504), (40, 120, 2000), (40, 72, 2000)]:
            dz, x = torch.randn(B, Co, S, 1, device=dev), torch.randn(B, Ci, S, 1, device=dev)
            sc = torch.rand(B, Ci, device=dev)
            with ops.zero_arena.scope("k"):
                t = timeit(lambda: ops.pw_conv_wgrad(dz, x, x_scale=sc), n=20)
            mb = (Co + Ci) * S * B * 4 / 1e6
            print("wgrad_sc %4d x %4d @ %5d  %7.1f us  %6.0f MB alg  %5.2f TB/s" % (Co, Ci, S, t, mb, mb / t))
        for (Co, Ci, S) in tfs:
            dz, x = torch.randn(B, Co, S, 1, device=dev), torch.randn(B, Ci, S, 1, device=dev)
            a, b = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
            sc = torch.rand(B, Ci, device=dev)
            with ops.zero_arena.scope("k"):
                t = timeit(lambda: ops.pw_conv_wgrad(dz, x, x_scale=sc, tf=(a, b, ops.ACT_HSWISH)), n=20)
            mb = (Co + Ci) * S * B * 4 / 1e6
            print("wgrad_tf %4d x %4d @ %5d  %7.1f us  %6.0f MB alg  %5.2f TB/s" % (Co, Ci, S, t, mb, mb / t))
