"""Debug aid for the x-resident 1x1 kernel: error map of one configuration (tools, not product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from efficientat_amd import ops
DEV = "cuda"


def run(B, Ci, Co, Fq, T, act, split, verbose=True):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, Fq, T, generator=g)
    w = torch.randn(Co, Ci, generator=g) * Ci ** -0.5
    bias = torch.randn(Co, generator=g) * 0.1
    wp = ops.pw_prepack_bf16(w.to(DEV), None, split)
    ops.pw_stream_mode(0)
    old = ops.pw_conv_bf16(x.to(DEV), wp, bias.to(DEV), Co, act, split).cpu()
    ops.pw_stream_mode(1)
    outs = [ops.pw_conv_bf16(x.to(DEV), wp, bias.to(DEV), Co, act, split).cpu() for _ in range(3)]
    ops.pw_stream_mode(0)
    new = outs[0]
    S = Fq * T
    d = (new - old).abs().view(B, Co, S)
    bad = d > 1e-3 * float(old.abs().max())
    print(f"B={B} Ci={Ci} Co={Co} S={S} act={act} split={split}: max diff {float(d.max()):.3e}, bad {int(bad.sum())} of {bad.numel()}"
          f", run-to-run identical: {bool((outs[0] == outs[1]).all() and (outs[1] == outs[2]).all())}", flush=True)
    if verbose and bad.any():
        MT = (Co + 15) // 16
        rows = bad.any(dim=2).any(dim=0)          # (Co,)
        print("  bad m-tiles:", [mt for mt in range(MT) if rows[mt * 16:(mt + 1) * 16].any()])
        print("  bad rows within m-tile (count per row%16):", [int(bad[:, r::16].sum()) for r in range(16)])
        cols = bad.any(dim=1).reshape(-1)          # flattened (b, s) columns
        N = cols.numel()
        print("  bad column tiles (of", (N + 255) // 256, "):", [t for t in range((N + 255) // 256) if cols[t * 256:(t + 1) * 256].any()])
        print("  bad per wave (col%256//64):", [int(cols[wv * 64::256].sum() + sum(int(cols[wv * 64 + i::256].sum()) for i in range(1, 64))) for wv in range(4)])
        print("  bad per lane&15 ((col%64)//4):", [int(sum(int(cols[q * 4 + i::64].sum()) for i in range(4))) for q in range(16)])
        # is a bad value the correct value of ANOTHER m-tile (stale / early fragments)?
        b0, m0, s0 = [int(v[0]) for v in torch.nonzero(bad, as_tuple=True)]
        print(f"  first bad element (b={b0}, m={m0}, s={s0}): new {float(new.view(B, Co, S)[b0, m0, s0]):.5f} old {float(old.view(B, Co, S)[b0, m0, s0]):.5f}")
        cand = old.view(B, Co, S)[b0, m0 % 16::16, s0]
        print("   old values of the same row-in-tile in every m-tile:", [round(float(v), 4) for v in cand])


if __name__ == "__main__":
    print("EAT_PW_STREAM_DBG =", os.environ.get("EAT_PW_STREAM_DBG"))
    for cfg in [(3, 80, 184, 8, 63, 0, False), (3, 80, 184, 8, 63, 0, True), (3, 80, 192, 8, 63, 0, False),
                (3, 80, 184, 8, 63, 1, False), (3, 96, 184, 8, 63, 0, False), (3, 64, 184, 8, 63, 0, False),
                (3, 128, 384, 8, 63, 0, False), (16, 80, 184, 8, 63, 0, False), (1, 80, 184, 8, 63, 0, False),
                (3, 80, 96, 8, 63, 0, False), (3, 32, 184, 8, 63, 0, False), (64, 112, 672, 8, 63, 2, False),
                (64, 112, 672, 8, 63, 2, True), (64, 40, 240, 16, 125, 2, True)]:
        run(*cfg)
