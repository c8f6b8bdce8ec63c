"""`efficientat_amd.optim.FusedAdam` (one launch of eat_adam_multi over every parameter) against torch.optim.Adam / AdamW - the
optimizer of the reference's training loop (ex_audioset.py:86-91, 197-199)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("no GPU", allow_module_level=True)

from efficientat_amd.optim import FusedAdam  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(1,), (3,), (527,), (16, 1, 3, 3), (960, 160, 1, 1), (1280, 960), (4097,), (4096,), (2, 4095)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in SHAPES]


@pytest.mark.parametrize("decoupled,wd", [(False, 0.0), (False, 1e-2), (True, 1e-2)])
@pytest.mark.parametrize("capturable", [False, True])
def test_fused_adam_matches_torch(decoupled, wd, capturable):
    pa, pb = _params(0), _params(0)
    ref_cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    ref = ref_cls(pa, lr=8e-4, weight_decay=wd, fused=True)       # the kernel whose expression types eat_adam_multi mirrors
    lr = torch.tensor(8e-4, device=DEV) if capturable else 8e-4
    opt = FusedAdam(pb, lr=lr, weight_decay=wd, decoupled=decoupled, capturable=capturable)
    g = torch.Generator().manual_seed(1)
    for it in range(7):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** (it - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, a.shape, float((a - b).abs().max()))
    for a, b in zip(pa, pb):
        sa, sb = ref.state[a], opt.state[b]
        # (the first moment is a difference of terms up to 1e3 times its own size here: round-off relative to the TERMS)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=2e-6 * float(sa["exp_avg"].abs().max()))
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-20)
        assert float(sb["step"]) == 7.0


def test_fused_adam_in_a_captured_graph_follows_the_eager_optimizer():
    pa, pb = _params(3), _params(3)
    ref = torch.optim.Adam(pa, lr=1e-3, fused=True)
    lr = torch.tensor(1e-3, device=DEV)
    opt = FusedAdam(pb, lr=lr, capturable=True)
    grads = [torch.zeros_like(p) for p in pb]
    for p, gr in zip(pb, grads):
        p.grad = gr
    opt.step()                                              # builds state + table outside the capture (zero gradients: no-op on p)
    for p in pa:
        p.grad = torch.zeros_like(p)
    ref.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            opt.step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.Generator().manual_seed(4)
    for it in range(5):
        if it == 3:
            lr.fill_(2e-4)                                  # a scheduler writing the tensor learning rate between replays
            for grp in ref.param_groups:
                grp["lr"] = 2e-4
        for a, gr in zip(pa, grads):
            v = torch.randn(a.shape, generator=g).to(DEV)
            a.grad.copy_(v)
            gr.copy_(v)
        ref.step()
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
    assert float(opt.state[pb[0]]["step"]) == 6.0
